"""Affine-invariant ensemble sampler (Goodman & Weare stretch move) with the
slice of the emcee 1.x/2.x ``EnsembleSampler`` API the reference uses
(reference mbb_emcee/mbb_fit.py:80-81, :525-550; results.py:154-155).

emcee is a third-party dependency of the reference that is not vendored and is
not installed in this image, so the role it plays -- propose a half-ensemble,
evaluate lnprob on every proposed row, accept/reject -- is provided here.  The
only thing that matters for the MI355X path is how lnprob is called: each
half-step hands all proposed rows to the callable at once (``(n, 5)`` array ->
``float64[n]``), i.e. one fused kernel launch per half-step instead of n Python
calls.  Parity with emcee is statistical only.
"""
import numpy as np

__all__ = ["EnsembleSampler", "integrated_time"]


def integrated_time(x, c=5.0):
    """Integrated autocorrelation time of a 1-D series (Sokal's automatic
    windowing); NaN when the series is too short to tell."""
    x = np.asarray(x, dtype=np.float64)
    n = x.size
    if n < 8:
        return np.nan
    x = x - x.mean()
    nfft = 1 << (2 * n - 1).bit_length()
    f = np.fft.rfft(x, nfft)
    acf = np.fft.irfft(f * np.conjugate(f))[:n]
    if acf[0] <= 0:
        return np.nan
    acf /= acf[0]
    taus = 2.0 * np.cumsum(acf) - 1.0
    window = np.arange(n) >= c * taus
    m = np.argmax(window) if window.any() else n - 1
    return float(taus[m])


class EnsembleSampler(object):
    """Stretch-move ensemble sampler.

    Parameters
    ----------
    nwalkers : even int, at least 2*dim
    dim : int
    lnpostfn : callable.  With ``vectorize=True`` (default when the callable has
        a ``context`` attribute, i.e. is this package's likelihood) it receives
        an ``(n, dim)`` array and returns n values; otherwise it is called row
        by row, or through ``pool.map`` when a pool is given.
    a : stretch scale (emcee default 2.0)
    threads : accepted for signature compatibility with emcee 2.x, unused.
    """

    def __init__(self, nwalkers, dim, lnpostfn, a=2.0, args=(), kwargs=None,
                 threads=1, pool=None, vectorize=None, seed=None):
        if nwalkers % 2 != 0:
            raise ValueError("The number of walkers must be even.")
        if nwalkers < 2 * dim:
            raise ValueError("The number of walkers needs to be more than twice the "
                             "dimension of your parameter space.")
        self.k = int(nwalkers)
        self.dim = int(dim)
        self.a = float(a)
        self.lnprobfn = lnpostfn
        self.args = tuple(args)
        self.kwargs = dict(kwargs or {})
        self.pool = pool
        self.threads = threads
        if vectorize is None:
            vectorize = hasattr(lnpostfn, "context")
        self.vectorize = bool(vectorize)
        self._random = np.random.RandomState(seed)
        self.reset()

    # ---- state ---------------------------------------------------------------
    def reset(self):
        self.naccepted = np.zeros(self.k)
        self.iterations = 0
        self._chain = np.empty((self.k, 0, self.dim))
        self._lnprob = np.empty((self.k, 0))
        self._last_run_mcmc_result = None

    @property
    def random_state(self):
        return self._random.get_state()

    @random_state.setter
    def random_state(self, state):
        try:
            self._random.set_state(state)
        except Exception:
            pass

    @property
    def chain(self):
        """[nwalkers, nsteps, dim]"""
        return self._chain

    @property
    def flatchain(self):
        s = self._chain.shape
        return self._chain.reshape(s[0] * s[1], s[2])

    @property
    def lnprobability(self):
        """[nwalkers, nsteps]"""
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
        mean_chain = self._chain.mean(axis=0)          # [nsteps, dim]
        return np.array([integrated_time(mean_chain[:, i], c=c) for i in range(self.dim)])

    # ---- lnprob ----------------------------------------------------------------
    def _get_lnprob(self, p):
        p = np.ascontiguousarray(p, dtype=np.float64)
        if not np.isfinite(p).all():                       # one pass in the usual case
            if np.any(np.isinf(p)):
                raise ValueError("At least one parameter value was infinite.")
            raise ValueError("At least one parameter value was NaN.")
        if self.vectorize:
            lnp = np.asarray(self.lnprobfn(p, *self.args, **self.kwargs), dtype=np.float64)
        elif self.pool is not None:
            lnp = np.array(list(self.pool.map(self.lnprobfn, list(p))), dtype=np.float64)
        else:
            lnp = np.array([self.lnprobfn(row, *self.args, **self.kwargs) for row in p],
                           dtype=np.float64)
        if np.isnan(lnp).any():
            bad = p[np.isnan(lnp)]
            raise ValueError("lnprob returned NaN for parameters {}".format(bad[0]))
        return lnp

    # ---- sampling ----------------------------------------------------------------
    def _begin(self, p0, lnprob0, rstate0, iterations, storechain):
        if rstate0 is not None:
            self.random_state = rstate0
        p = np.array(p0, dtype=np.float64)
        if p.shape != (self.k, self.dim):
            raise ValueError("p0 must have shape (nwalkers, dim)")
        lnprob = lnprob0 if lnprob0 is not None else self._get_lnprob(p)
        lnprob = np.array(lnprob, dtype=np.float64)
        if np.any(np.isnan(lnprob)):
            raise ValueError("The initial lnprob was NaN.")
        i0 = self._chain.shape[1]
        if storechain:
            self._chain = np.concatenate(
                (self._chain, np.zeros((self.k, iterations, self.dim))), axis=1)
            self._lnprob = np.concatenate((self._lnprob, np.zeros((self.k, iterations))), axis=1)
        return p, lnprob, i0

    def _advance(self, p, lnprob, i0, its, storechain):
        """The stretch move proper, `its` = range of iteration numbers; p and lnprob are
        updated in place.  One likelihood launch per half-step."""
        halfk = self.k // 2
        halves = ((slice(halfk), slice(halfk, self.k), 0), (slice(halfk, self.k), slice(halfk), halfk))
        rand, randint, a, dm1 = self._random.rand, self._random.randint, self.a, self.dim - 1.0
        for it in its:
            self.iterations += 1
            for S0, S1, off in halves:
                s, cset = p[S0], p[S1]
                ns, nc = s.shape[0], cset.shape[0]
                zz = ((a - 1.0) * rand(ns) + 1.0) ** 2 / a
                partner = cset[randint(nc, size=ns)]
                q = partner - zz[:, None] * (partner - s)
                newlnprob = self._get_lnprob(q)
                lnpdiff = dm1 * np.log(zz) + newlnprob - lnprob[S0]
                idx = np.flatnonzero(lnpdiff > np.log(rand(ns)))
                p[idx + off] = q[idx]
                lnprob[idx + off] = newlnprob[idx]
                self.naccepted[idx + off] += 1
            if storechain:
                self._chain[:, i0 + it, :] = p
                self._lnprob[:, i0 + it] = lnprob

    def sample(self, p0, lnprob0=None, rstate0=None, iterations=1, storechain=True):
        p, lnprob, i0 = self._begin(p0, lnprob0, rstate0, int(iterations), storechain)
        for it in range(int(iterations)):
            self._advance(p, lnprob, i0, (it,), storechain)
            yield p, lnprob, self.random_state

    def run_mcmc(self, pos0, N, rstate0=None, lnprob0=None, storechain=True):
        """Run N steps from pos0; returns (pos, lnprob, rstate) like emcee 2.x.  (Same moves
        as iterating sample(), without copying the generator's state out at every step.)"""
        if pos0 is None:
            if self._last_run_mcmc_result is None:
                raise ValueError("Cannot have pos0=None if run_mcmc has never been called.")
            pos0, lnprob0, rstate0 = self._last_run_mcmc_result
        N = int(N)
        p, lnprob, i0 = self._begin(pos0, lnprob0, rstate0, N, storechain)
        if N <= 0:
            return None
        self._advance(p, lnprob, i0, range(N), storechain)
        results = (p, lnprob, self.random_state)
        self._last_run_mcmc_result = (p.copy(), lnprob.copy(), results[2])
        return results
