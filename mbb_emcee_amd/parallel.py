"""Multi-GPU evaluation of an ensemble: walkers partitioned across ranks, one
all-gather of log-probabilities per call (i.e. per emcee half-step).

The reference's only parallelism is emcee's multiprocessing pool over walkers
(reference mbb_emcee/mbb_fit.py:80-81).  Each walker's lnL depends only on its
own five parameters (likelihood.py:790-834), so the ensemble shards trivially:
one process per GPU, rank r evaluates the contiguous block
``[r*per, (r+1)*per)`` of the rows it is given and the blocks are exchanged with
a single collective.  Every rank runs the same sampler with the same random
stream, so positions never have to be scattered -- the all-gather of lnprob is
the only data-path communication.

Communicators:
  RcclComm   -- ncclAllGather on device buffers through the C-ABI (the MI355X path)
  any object with ``rank``, ``world`` and ``allgather_host(x) -> concatenation over
  ranks`` -- a host-side gather; the CPU tests of the sharding logic bring one built on
  torch.distributed/gloo (tests/_dist_worker.py).  Nothing in this package imports torch.
"""
import numpy as np

__all__ = ["block_bounds", "RcclComm", "ShardedLikelihood", "ipc_exchange_setup"]


def block_bounds(n, world):
    """Rows per rank (ceil) and the [lo, hi) block of every rank."""
    per = (n + world - 1) // world
    return per, [(min(n, r * per), min(n, (r + 1) * per)) for r in range(world)]


class RcclComm(object):
    """ncclAllGather over xGMI on the likelihood context's stream (C-ABI)."""

    def __init__(self, ctx, rank, world, unique_id):
        self.ctx, self.rank, self.world = ctx, int(rank), int(world)
        if self.world > 1:
            ctx.comm_init(self.world, self.rank, unique_id)
        self._cap = 0
        self._bufs = None

    def buffers(self, per):
        """Device buffers for callers that keep the rows resident (mbb_lnlike_allgather_device)."""
        if per > self._cap:
            if self._bufs:
                for b in self._bufs:
                    b.free()
            cap = max(256, 2 * per)
            self._bufs = (self.ctx.alloc(cap * 40), self.ctx.alloc(cap * 8),
                          self.ctx.alloc(cap * 4), self.ctx.alloc(self.world * cap * 8))
            self._cap = cap
        return self._bufs

    def close(self):
        if self.world > 1:
            self.ctx.comm_destroy()


class ShardedLikelihood(object):
    """lnprob callable for an ensemble partitioned across ranks.

    ``like`` is this package's likelihood when ``comm`` is an RcclComm (the
    shard is evaluated by the fused kernel and gathered device-to-device), or
    any callable ``(m, 5) -> float64[m]`` with a host-side communicator.  Calling it with the
    same ``(n, 5)`` array on every rank returns the same ``float64[n]`` everywhere.
    """

    def __init__(self, like, comm):
        self.like = like
        self.comm = comm

    @property
    def context(self):             # lets EnsembleSampler default to vectorize=True
        return getattr(self.like, "context", None)

    def __call__(self, pars):
        p = np.ascontiguousarray(np.atleast_2d(pars), dtype=np.float64)
        n = p.shape[0]
        world, rank = self.comm.world, self.comm.rank
        per, bounds = block_bounds(n, world)
        lo, hi = bounds[rank]
        local = np.empty((per, 5))
        local[:hi - lo] = p[lo:hi]
        if hi - lo < per:                       # ragged tail: pad with a valid row
            local[hi - lo:] = p[0]
        if isinstance(self.comm, RcclComm):
            # one native call: rows through the BAR / pinned memory, the fused kernel, ONE ncclAllGather of
            # `per` doubles per rank, the gathered vector into a pinned landing buffer, one stream wait
            # (round 3 made three blocking copies around an asynchronous call here)
            ctx = self.like._sync_device()
            full, st = ctx.lnlike_allgather(local, world)
            from . import _native
            _native.raise_for_status(st[:hi - lo])
        else:
            full = self.comm.allgather_host(np.asarray(self.like(local), dtype=np.float64))
        out = np.empty(n)
        for r, (a, b) in enumerate(bounds):
            out[a:b] = full[r * per:r * per + (b - a)]
        return out


def ipc_exchange_setup(ctx, rank, world, side_channel, max_rows=4096):
    """Bring up the one-hop exchange of the device-resident sampler (include/mbb_hip.h,
    mbb_xchg_*) on this rank's context: every rank keeps the whole ensemble in a buffer
    its peers map through hipIpc and the kernel itself stores a moved walker into every
    copy -- no collective library, one xGMI hop per moved row.

    side_channel: how the 64-byte handles travel and how the ranks are held together
    between host-side steps.  Either a torch.distributed-like module (``all_gather_object``
    + ``barrier``; the launcher's gloo group) or any object with
    ``allgather_bytes(b) -> [b_0, ..., b_{world-1}]`` and ``barrier()``.
    Afterwards ``DeviceEnsembleSampler(nwalkers <= max_rows, ...)`` on this context runs
    sharded; its ``barrier`` hook is set from the side channel."""
    handle = ctx.xchg_open(world, rank, max_rows)
    if hasattr(side_channel, "allgather_bytes"):
        handles = side_channel.allgather_bytes(handle)
    else:
        handles = [None] * world
        side_channel.all_gather_object(handles, handle)
    side_channel.barrier()                 # every rank has its buffer before anyone maps it
    ctx.xchg_connect([bytes(h) for h in handles])
    side_channel.barrier()
    ctx.xchg_barrier = side_channel.barrier
    return ctx
