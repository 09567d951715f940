"""Device-resident affine-invariant ensemble sampler.

Plays the role of ``emcee.EnsembleSampler`` in the reference's fit driver
(reference mbb_emcee/mbb_fit.py:80-81, :525-550; results.py:154-155) with the
whole stretch-move step on the MI355X: for each half of the ensemble ONE kernel
draws z and a partner for every walker (counter-based Philox RNG), forms the
proposal, evaluates the fused likelihood and accepts or rejects in place.  A run
of N steps is 2N dependent launches enqueued back to back; the host only sees
the chain at the end.  emcee itself is not part of the reference tree, so parity
with it is statistical.
"""
import ctypes as C

import numpy as np

from . import _native
from .ensemble import integrated_time

__all__ = ["DeviceEnsembleSampler"]


class DeviceEnsembleSampler(object):
    """emcee-2.x-shaped API: run_mcmc, chain [nw, nsteps, 5], lnprobability
    [nw, nsteps], flatchain, acceptance_fraction, acor, reset.

    lnpostfn must be this package's ``likelihood`` (it owns the device context
    and the constant block the kernel reads)."""

    def __init__(self, nwalkers, dim, lnpostfn, a=2.0, threads=1, seed=None, **unused):
        if dim != 5:
            raise ValueError("the modified blackbody model has 5 parameters")
        if nwalkers % 2 != 0:
            raise ValueError("The number of walkers must be even.")
        if nwalkers < 2 * dim:
            raise ValueError("The number of walkers needs to be more than twice the "
                             "dimension of your parameter space.")
        if not hasattr(lnpostfn, "_sync_device"):
            raise TypeError("DeviceEnsembleSampler needs a mbb_emcee_amd.likelihood")
        self.k, self.dim, self.a = int(nwalkers), 5, float(a)
        self.lnprobfn = lnpostfn
        self.seed = int(np.random.SeedSequence(seed).generate_state(1, dtype=np.uint64)[0]) \
            if seed is None else int(seed) & 0xFFFFFFFFFFFFFFFF
        self._h = None
        self._ctx = None
        self._ns = 1
        self.reset()

    @property
    def nsources(self):
        return getattr(self.lnprobfn, "nsources", 1)

    def _handle(self):
        ctx = self.lnprobfn._sync_device()
        if self._h is not None and self._ctx is ctx and self._ns != self.nsources:
            ctx.lib.mbb_sampler_destroy(ctx.h, self._h)
            self._h = None
        if self._h is None or self._ctx is not ctx:
            self._ns = self.nsources
            h = C.c_void_p()
            _native._check(ctx.lib.mbb_sampler_create(ctx.h, self.k, self.seed, C.byref(h)))
            self._h, self._ctx = h, ctx
        return ctx, self._h

    def reset(self):
        ns = self.nsources
        lead = (ns, self.k) if ns > 1 else (self.k,)
        self.naccepted = np.zeros(lead)
        self.iterations = 0
        self._chain = np.empty(lead + (0, self.dim))
        self._lnprob = np.empty(lead + (0,))
        self._last = None
        if self._h is not None:
            _native._check(self._ctx.lib.mbb_sampler_reset(self._ctx.h, self._h))

    @property
    def chain(self):
        return self._chain

    @property
    def flatchain(self):
        s = self._chain.shape
        if len(s) == 4:                       # [nsources, nw, nsteps, 5] -> [nsources, nw*nsteps, 5]
            return self._chain.reshape(s[0], s[1] * s[2], s[3])
        return self._chain.reshape(s[0] * s[1], s[2])

    @property
    def lnprobability(self):
        return self._lnprob

    @property
    def flatlnprobability(self):
        return self._lnprob.flatten()

    @property
    def acceptance_fraction(self):
        return self.naccepted / max(self.iterations, 1)

    @property
    def acor(self):
        return self.get_autocorr_time()

    def get_autocorr_time(self, c=5.0):
        ch = self._chain if self._chain.ndim == 3 else self._chain[0]
        mean_chain = ch.mean(axis=0)
        return np.array([integrated_time(mean_chain[:, i], c=c) for i in range(self.dim)])

    @property
    def random_state(self):
        return self.seed

    def run_mcmc(self, pos0, N, rstate0=None, lnprob0=None, storechain=True, **unused):
        """N stretch-move steps from pos0 [nw, 5]; returns (pos, lnprob, rstate)."""
        ctx, h = self._handle()
        # sharded with the one-hop exchange (parallel.ipc_exchange_setup): the ranks are
        # held together around set_state, because a peer's kernel writes into this rank's
        # copy of the ensemble
        barrier = getattr(ctx, "xchg_barrier", None)
        if pos0 is None:
            if self._last is None:
                raise ValueError("Cannot have pos0=None if run_mcmc has never been called.")
        else:
            if barrier:
                ctx.sync(); barrier()
            p0 = np.ascontiguousarray(pos0, dtype=np.float64)
            want = (self.nsources, self.k, 5) if self.nsources > 1 else (self.k, 5)
            if p0.shape != want:
                raise ValueError("p0 must have shape {}".format(want))
            if np.any(np.isinf(p0)):
                raise ValueError("At least one parameter value was infinite.")
            if np.any(np.isnan(p0)):
                raise ValueError("At least one parameter value was NaN.")
            l0 = None if lnprob0 is None else np.ascontiguousarray(lnprob0, dtype=np.float64)
            try:
                _native._check(ctx.lib.mbb_sampler_set_state(
                    ctx.h, h, _native._d(p0), _native._d(l0) if l0 is not None else None))
            except _native.NativeError as e:
                raise ValueError(str(e))
            if barrier:
                barrier()
        N = int(N)
        lead = (self.nsources, self.k) if self.nsources > 1 else (self.k,)
        # (zeros where a run is sharded over ranks -- the one-hop exchange or a communicator, whether or not a
        # barrier was handed in: only this rank's walkers are filled in there)
        alloc = np.zeros if (barrier or ctx.info("nranks") > 1) else np.empty
        chain = alloc(lead + (N, 5)) if storechain else None
        lnp = alloc(lead + (N,)) if storechain else None
        pos = np.empty(lead + (5,))
        lnprob = np.empty(lead)
        nacc = np.zeros(lead)
        fallbacks = ctx.info("flow_fallbacks")
        rc = ctx.lib.mbb_sampler_run(ctx.h, h, N, self.a,
                                     _native._d(chain) if storechain else None,
                                     _native._d(lnp) if storechain else None,
                                     _native._d(pos), _native._d(lnprob), _native._d(nacc))
        if rc == -2:
            raise ValueError(ctx.lib.mbb_last_error().decode())
        _native._check(rc)
        if ctx.info("flow_fallbacks") > fallbacks:
            import warnings
            warnings.warn("the one-launch form of the device sampler gave up waiting (a workgroup was not "
                          "resident: another process on the GPU?) and the run was redone as a train of "
                          "launches, about 2.5x slower; same chain.  %d such run(s) on this context so far%s"
                          % (ctx.info("flow_fallbacks"),
                             "; the one-launch form now rests for %d runs" % ctx.info("flow_resting")
                             if ctx.info("flow_resting") else ""), RuntimeWarning, stacklevel=2)
        self.iterations += N
        self.naccepted = nacc
        if storechain:
            ax = len(lead)
            if self._chain.shape[ax] == 0:            # the usual case (after reset): no second copy of the chain
                self._chain, self._lnprob = chain, lnp
            else:
                self._chain = np.concatenate((self._chain, chain), axis=ax)
                self._lnprob = np.concatenate((self._lnprob, lnp), axis=ax)
        self._last = (pos, lnprob)
        return pos, lnprob, self.seed

    def sample(self, p0, lnprob0=None, rstate0=None, iterations=1, storechain=True, chunk=64):
        """emcee's generator form (``for pos, lnprob, rstate in sampler.sample(p0, iterations=N)``): the ensemble after
        every step.  The steps are made ``chunk`` at a time in one launch on the device -- a launch has a fixed cost of
        ~20 us beside ~6 us per step -- and handed out one by one.  The chain is run_mcmc(p0, iterations)'s whatever the
        chunk: a step's draws are keyed by its number in the sampler's life, not by its place in a launch.  While step j is
        out, ``chain``, ``lnprobability``, ``iterations`` and ``naccepted`` are those of the steps handed out so far -- as
        with emcee, where a step exists only once it has been yielded -- although the device is up to ``chunk`` - 1 steps
        ahead; breaking out of the loop early leaves the sampler at the end of the chunk it was in (``run_mcmc(None, n)``
        goes on from there, and the attributes then show that whole chunk).  ``rstate0`` is accepted for emcee's call
        convention and ignored: the random stream is the sampler's ``seed`` (Philox, counted by step and row)."""
        iterations, chunk = int(iterations), max(1, int(chunk))
        ax = 1 if self.nsources == 1 else 2
        kept_c, kept_l = self._chain, self._lnprob
        n_prev = kept_c.shape[ax]
        it_prev = self.iterations
        if storechain:
            # (room for the whole run once, filled chunk by chunk: no chain is copied more than that)
            big_c = np.empty(kept_c.shape[:ax] + (n_prev + iterations, self.dim))
            big_l = np.empty(kept_l.shape[:ax] + (n_prev + iterations,))
            big_c[..., :n_prev, :] = kept_c
            big_l[..., :n_prev] = kept_l
        pos0, l0 = p0, lnprob0
        done = made = 0                       # steps handed out / made on the device by this call
        acc_made = self.naccepted
        cur = None if p0 is None else np.asarray(p0, dtype=np.float64)
        if cur is None and self._last is not None:
            cur = self._last[0]
        try:
            while done < iterations:
                k = min(chunk, iterations - done)
                acc_before = np.array(self.naccepted, dtype=np.float64, copy=True)
                self._chain, self._lnprob = kept_c[..., :0, :], kept_l[..., :0]
                self.run_mcmc(pos0, k, lnprob0=l0, storechain=True)
                steps, lnps = self._chain, self._lnprob          # (this chunk's own arrays)
                made, acc_made = made + k, self.naccepted
                pos0 = l0 = None
                if storechain:
                    big_c[..., n_prev + done:n_prev + done + k, :] = steps
                    big_l[..., n_prev + done:n_prev + done + k] = lnps
                # a walker's accepted moves step by step: a stretch move that is accepted changes the walker's position
                prev = np.concatenate((cur[..., None, :], steps[..., :-1, :]), axis=ax)
                moved = np.cumsum(np.any(steps != prev, axis=-1), axis=-1)
                cur = steps[..., k - 1, :]
                for j in range(k):
                    done += 1
                    if storechain:
                        self._chain, self._lnprob = big_c[..., :n_prev + done, :], big_l[..., :n_prev + done]
                    else:
                        self._chain, self._lnprob = kept_c, kept_l
                    self.iterations = it_prev + done
                    self.naccepted = acc_made if j == k - 1 else acc_before + moved[..., j]
                    yield steps[..., j, :], lnps[..., j], self.seed
        finally:
            # (left early, inside a chunk: the device made the whole chunk and run_mcmc(None, n) goes on from its end --
            # the attributes then show all of it)
            self.iterations, self.naccepted = it_prev + made, acc_made
            if storechain:
                self._chain, self._lnprob = big_c[..., :n_prev + made, :], big_l[..., :n_prev + made]
            else:
                self._chain, self._lnprob = kept_c, kept_l

    def advance_async(self, N):
        """Enqueue N steps without storing or synchronising (benchmarks)."""
        ctx, h = self._handle()
        _native._check(ctx.lib.mbb_sampler_advance_async(ctx.h, h, int(N), self.a))

    def advance_timed(self, N):
        """N steps as advance_async enqueues them, timed inside one native call: (wall seconds from an
        idle stream to an idle stream, stream milliseconds between two events).  Benchmarks."""
        ctx, h = self._handle()
        wall, ms = C.c_double(), C.c_float()
        _native._check(ctx.lib.mbb_sampler_advance_timed(ctx.h, h, int(N), self.a, C.byref(wall), C.byref(ms)))
        return wall.value, ms.value

    def __del__(self):
        try:
            if self._h is not None and self._ctx is not None and self._ctx.h:
                self._ctx.lib.mbb_sampler_destroy(self._ctx.h, self._h)
        except Exception:
            pass
