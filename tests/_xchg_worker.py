"""Worker of the two-process test of the one-hop exchange (tests/test_gpu_parity.py::
test_one_hop_exchange_two_processes_one_gpu): both processes run on the SAME GPU (IPC
mappings work within a device), each moves its share of every half-ensemble and stores
the moved rows into the other's copy; chain, final state and acceptance counts must be
bitwise those of the unsharded device sampler."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from _filecomm import FileComm
    dist = FileComm()                                  # ranks started by the test itself: files, no port, no torch
    rank, world = dist.rank, dist.world
    import mbb_emcee_amd as mbb
    from mbb_emcee_amd import parallel
    g = np.load(os.path.join(ROOT, "tests", "golden", "lnlike.npz"))
    bands = [str(b) for b in g["cfg2/bands"]]

    def make():
        like = mbb.likelihood(response=True, device=0)
        like.set_phot(bands, g["cfg2/thick_walpha/flux"], g["cfg2/thick_walpha/unc"])
        return like
    nw, nsteps = int(os.environ.get("MBB_XCHG_TEST_WALKERS", "48")), 12
    rng = np.random.RandomState(3)                     # the same on every rank
    p0 = np.array([12.0, 1.8, 600.0, 3.0, 40.0]) * (1.0 + 0.02 * rng.normal(size=(nw, 5)))
    like = make()
    ctx = like._sync_device()
    parallel.ipc_exchange_setup(ctx, rank, world, dist, max_rows=256)
    assert ctx.info("nranks") == world and ctx.info("rank") == rank
    # the sharded run in both its forms -- one launch per run, decisions / rows / words stored into
    # every rank's copy as they are made (k_lnlike SMODE 6), and one launch per half-step with the
    # moved rows exchanged after it (SMODE 2) -- against the unsharded run on a context of its own
    launches_before = 0
    for one_launch in (1, 0):
        ctx.set_option("sharded_flow_sampler", one_launch)
        ref_like = make()
        # (the reference is the plain launch train: every rank runs one at the same time on this one
        # GPU, and four one-launch runs of the whole ensemble would not all be resident)
        ref_like.context.set_option("lookahead_sampler", 0)
        ref = mbb.DeviceEnsembleSampler(nw, 5, ref_like, seed=21)
        rpos, rlnp, _ = ref.run_mcmc(p0, nsteps)
        rchain, rlnpc, racc = ref.chain.copy(), ref.lnprobability.copy(), ref.naccepted.copy()
        rpos2, rlnp2, _ = ref.run_mcmc(None, 5)
        smp = mbb.DeviceEnsembleSampler(nw, 5, like, seed=21)
        pos, lnp, _ = smp.run_mcmc(p0, nsteps)
        assert ctx.info("last_kernel_form") == (6 if one_launch else 2), ctx.info("last_kernel_form")
        half, per = nw // 2, nw // 2 // world
        mine = np.r_[rank * per:(rank + 1) * per, half + rank * per:half + (rank + 1) * per]
        assert np.array_equal(pos, rpos) and np.array_equal(lnp, rlnp), "final state differs on rank %d (form %d)" % (rank, one_launch)
        assert np.array_equal(smp.chain[mine], rchain[mine]), "chain differs on rank %d" % rank
        assert np.array_equal(smp.lnprobability[mine], rlnpc[mine])
        assert np.array_equal(smp.naccepted[mine], racc[mine])
        others = np.setdiff1d(np.arange(nw), mine)
        assert not smp.chain[others].any()                 # left untouched for the caller to gather
        if not one_launch:
            assert ctx.info("xchg_launches") - launches_before == 2 * nsteps
        # continuing the chain, and the asynchronous form the benchmark uses
        pos2, lnp2, _ = smp.run_mcmc(None, 5)
        assert np.array_equal(pos2, rpos2) and np.array_equal(lnp2, rlnp2)
        dist.barrier()
        smp.advance_async(40)
        ctx.sync()
        ref.advance_async(40)
        ref_like.context.sync()
        a, la, _ = smp.run_mcmc(None, 0)
        b, lb, _ = ref.run_mcmc(None, 0)
        assert np.array_equal(a, b) and np.array_equal(la, lb)
        # the whole chain, gathered by the caller over the side channel
        full = smp.chain.copy()
        parts = [np.frombuffer(b, dtype=full.dtype).reshape(full.shape) for b in dist.allgather_bytes(full.tobytes())]
        full = np.sum(parts, axis=0)                       # disjoint rows, zeros elsewhere: the sum is the gather
        assert np.array_equal(full[:, :nsteps], rchain)
        dist.barrier()
        launches_before = ctx.info("xchg_launches")
        if one_launch:
            del smp
            import gc
            gc.collect()
    ctx.set_option("sharded_flow_sampler", 1)
    ctx.set_option("flow_spin_log2", 13)
    # a peer that stops taking part must surface as an error on the others, not as a hang:
    # rank 1 sits this run out, rank 0's second launch waits for rank 1's flag, gives up
    # after xchg_spin_max polls and the run reports it
    ctx.set_option("xchg_spin_max", 20000)
    if os.environ.get("MBB_XCHG_TEST_SKIP_LOST_PEER"):
        del smp
        import gc
        gc.collect()
        ctx.xchg_close()
        if rank == 0:
            print("XCHG_OK")
        return
    if rank == 0:
        smp2 = mbb.DeviceEnsembleSampler(nw, 5, like, seed=3)
        smp2.barrier = None
        saved, ctx.xchg_barrier = ctx.xchg_barrier, None      # rank 1 will not be at the barriers either
        try:
            smp2.run_mcmc(p0, 3)
            raise AssertionError("a run without its peer must not succeed")
        except Exception as e:
            assert "timed out" in str(e), str(e)
        ctx.xchg_barrier = saved
    dist.barrier()
    try:
        ctx.xchg_close()
        raise AssertionError("closing the exchange under a live sampler must be refused")
    except Exception as e:
        assert "sampler still lives" in str(e), str(e)
    del smp
    if rank == 0:
        del smp2
    import gc
    gc.collect()
    ctx.xchg_close()
    if rank == 0:
        print("XCHG_OK")


if __name__ == "__main__":
    try:
        main()
    except BaseException:
        try:                                           # the other ranks stop waiting for this one
            open(os.path.join(os.environ["MBB_TEST_RDZV_DIR"], "abort"), "w").close()
        except Exception:
            pass
        raise
