"""Worker of test_rccl_paths_with_several_ranks_on_one_gpu: the library's N > 1 communicator code -- mbb_comm_init,
mbb_allgather_f64, mbb_lnlike_allgather_device, mbb_lnlike_allgather (rank-major offsets, the in-place send pointer,
the ragged last block through ShardedLikelihood) and the device sampler's in-place gathers of moved rows / chain /
counts -- with `WORLD_SIZE` ranks as processes on ONE GPU.  Real RCCL wants a device per rank; the ranks here bind
tests/rccl_standin/ through MBB_RCCL_LIB (an all-gather through hipIpc-mapped staging buffers): test
infrastructure, never a product fall-back.  Everything is held bitwise to the unsharded evaluation on a context of
its own.  Role replaced: emcee's pool, reference mbb_fit.py:80-81."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from _filecomm import FileComm
    side = FileComm()
    rank, world = side.rank, side.world
    assert os.environ.get("MBB_RCCL_LIB", "").endswith("librccl_standin.so")
    import mbb_emcee_amd as mbb
    from mbb_emcee_amd.parallel import RcclComm, ShardedLikelihood, block_bounds
    g = np.load(os.path.join(ROOT, "tests", "golden", "lnlike.npz"))
    bands = [str(b) for b in g["cfg2/bands"]]

    def make():
        like = mbb.likelihood(response=True, device=0)
        like.set_phot(bands, g["cfg2/thick_walpha/flux"], g["cfg2/thick_walpha/unc"])
        return like
    like, plain = make(), make()
    ctx = like._sync_device()
    allp = np.ascontiguousarray(np.tile(g["cfg2/thick_walpha/pars"], (3, 1)))
    allp[5, 0] = 0.5                                              # a row below the lower limit: -inf
    want = plain(allp)
    assert np.isneginf(want[5]) and np.isfinite(want[6])

    uid = side.allgather_bytes(ctx.comm_unique_id() if rank == 0 else b"")[0]
    comm = RcclComm(ctx, rank, world, uid)
    assert ctx.info("nranks") == world and ctx.info("rank") == rank

    # mbb_allgather_f64, out of place and in place (send = recv + rank * count: what the sampler does with its rows)
    for count in (1, 7, 125, 4099):
        mine = np.arange(count, dtype=np.float64) + 1000.0 * (rank + 1)
        a, b = ctx.alloc(mine.nbytes), ctx.alloc(world * mine.nbytes)
        a.upload(mine)
        ctx.allgather_f64(a, b, count)
        ctx.sync()
        got = b.download(np.float64, world * count).reshape(world, count)
        for r in range(world):
            assert np.array_equal(got[r], np.arange(count) + 1000.0 * (r + 1)), (count, r)
        buf = np.zeros((world, count))
        buf[rank] = mine
        b.upload(buf)
        ctx.allgather_f64(b.ptr.value + rank * mine.nbytes, b, count)
        ctx.sync()
        assert np.array_equal(b.download(np.float64, world * count).reshape(world, count), got), count
        a.free(); b.free()

    # north_star's split: likelihood.__call__ sharded, ONE all-gather of lnprob per call (mbb_lnlike_allgather) --
    # sizes that divide, sizes with a ragged last block, fewer rows than ranks, with and without zero-copy
    sharded = ShardedLikelihood(like, comm)
    for zc in (1, 0):
        ctx.set_option("zero_copy", zc)
        for n in (1, 2, world, world + 1, 125, 250, 251, 256, allp.shape[0]):
            got = sharded(allp[:n])
            assert np.array_equal(got, want[:n], equal_nan=True), (rank, zc, n, np.flatnonzero(got != want[:n])[:5])
            per, bounds = block_bounds(n, world)
            lo, hi = bounds[rank]
            local = np.ascontiguousarray(np.vstack([allp[lo:hi], np.tile(allp[0], (per - (hi - lo), 1))]))
            full, st = ctx.lnlike_allgather(local, world)         # the native call itself: rank-major blocks of `per`
            assert full.shape == (world * per,) and st.shape == (per,)
            for r, (a_, b_) in enumerate(bounds):
                assert np.array_equal(full[r * per:r * per + (b_ - a_)], want[a_:b_], equal_nan=True), (rank, n, r)
    ctx.set_option("zero_copy", 1)

    # rows that stay on the device (mbb_lnlike_allgather_device): d_lnl is this rank's slice of d_all (in place)
    per = 125
    lo = rank * per
    dp, ds, da = ctx.alloc(per * 40), ctx.alloc(per * 4), ctx.alloc(world * per * 8)
    dp.upload(np.ascontiguousarray(allp[lo:lo + per]))
    import ctypes as C
    from mbb_emcee_amd import _native
    _native._check(ctx.lib.mbb_lnlike_allgather_device(ctx.h, dp.ptr, per, C.c_void_p(da.ptr.value + rank * per * 8), ds.ptr,
                                                       da.ptr))
    ctx.sync()
    assert np.array_equal(da.download(np.float64, world * per), want[:world * per], equal_nan=True)
    for x in (dp, ds, da):
        x.free()

    # the device sampler over the communicator: a launch per half-step, the moved rows gathered in place after each,
    # chain and counts gathered at the end -- every rank ends with the WHOLE chain, bitwise the unsharded run's
    nw = 24 * world
    rng = np.random.RandomState(3)                                # the same on every rank
    p0 = np.array([12.0, 1.8, 600.0, 3.0, 40.0]) * (1.0 + 0.02 * rng.normal(size=(nw, 5)))
    plain.context.set_option("lookahead_sampler", 0)              # (the reference: the plain launch train)
    ref = mbb.DeviceEnsembleSampler(nw, 5, plain, seed=21)
    rpos, rlnp, _ = ref.run_mcmc(p0, 9)
    rpos2, rlnp2, _ = ref.run_mcmc(None, 4)
    smp = mbb.DeviceEnsembleSampler(nw, 5, like, seed=21)
    pos, lnp, _ = smp.run_mcmc(p0, 9)
    assert ctx.info("last_kernel_form") == 1
    assert np.array_equal(pos, rpos) and np.array_equal(lnp, rlnp), "state differs on rank %d" % rank
    pos2, lnp2, _ = smp.run_mcmc(None, 4)
    assert np.array_equal(pos2, rpos2) and np.array_equal(lnp2, rlnp2)
    assert np.array_equal(smp.chain, ref.chain) and np.array_equal(smp.lnprobability, ref.lnprobability)
    assert np.array_equal(smp.naccepted, ref.naccepted)
    smp.advance_async(10); ctx.sync()
    ref.advance_async(10); plain.context.sync()
    a, la, _ = smp.run_mcmc(None, 0)
    b, lb, _ = ref.run_mcmc(None, 0)
    assert np.array_equal(a, b) and np.array_equal(la, lb)
    # a half-ensemble that does not divide by the ranks is refused, on every rank alike
    if world == 3:
        try:
            mbb.DeviceEnsembleSampler(40, 5, like, seed=1).run_mcmc(p0[:40], 1)
            raise AssertionError("20 walkers per half over 3 ranks must be refused")
        except AssertionError:
            raise
        except Exception as e:
            assert "multiple of the number of ranks" in str(e), str(e)
    del smp
    import gc
    gc.collect()
    side.barrier()
    comm.close()
    assert ctx.info("nranks") == 1
    assert np.array_equal(like(allp[:7]), want[:7], equal_nan=True)     # (and the context is an ordinary one again)
    side.barrier()
    print("RCCL_RANKS_OK %d" % rank)


if __name__ == "__main__":
    try:
        main()
    except BaseException:
        try:
            open(os.path.join(os.environ["MBB_TEST_RDZV_DIR"], "abort"), "w").close()
        except Exception:
            pass
        raise
