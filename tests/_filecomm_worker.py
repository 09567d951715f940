"""Worker of test_sharded_three_ranks_file_side_channel (CPU): the sharding logic of mbb_emcee_amd.parallel -- partition,
pad, all-gather, reassemble, ragged last block, fewer rows than ranks -- with three ranks that meet through files
(tests/_filecomm.py: the side channel the multi-process GPU tests use), the per-shard lnprob being the CPU oracle (test
infrastructure standing in for the GPU kernel, as in tests/_dist_worker.py).  No torch, no port."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    from _filecomm import FileComm
    comm = FileComm()
    rank, world = comm.rank, comm.world
    from mbb_emcee_amd.parallel import ShardedLikelihood, block_bounds
    from mbb_emcee_amd.ensemble import EnsembleSampler
    from mbb_emcee_amd.response import response_set
    from oracle import oracle as O
    names = ["PACS_160um", "SPIRE_250um", "SPIRE_350um", "SPIRE_500um"]
    wheel = response_set()
    bands = [(wheel[n].wavelength, wheel[n]._sedmult, wheel[n]._normfac) for n in names]
    flux = np.array([95.0, 90.0, 60.0, 30.0])
    orc = O.OracleLikelihood(flux, 0.1 * flux, bands=bands, opthin=False, noalpha=False)
    sharded = ShardedLikelihood(lambda p: orc(p), comm)
    rng = np.random.RandomState(5)             # the same stream on every rank
    for n in (1, 2, 3, 4, 10, 49, 50):
        pars = np.column_stack([rng.normal(20, 2, n), rng.normal(1.8, 0.2, n), rng.normal(600.0, 30.0, n),
                                rng.normal(3.0, 0.2, n), rng.normal(30, 3, n)])
        if n > 3:
            pars[n - 1, 0] = 0.5                # a -inf row in the ragged last block
        got, ref = sharded(pars), orc(pars)
        assert np.array_equal(got, ref), (rank, n)
        per, bounds = block_bounds(n, world)
        assert sum(b - a for a, b in bounds) == n and all(b - a <= per for a, b in bounds)
    # objects and bytes through the channel, a barrier, and a whole sharded sampler run: the same chain on every rank
    assert comm.allgather_object({"rank": rank}) == [{"rank": r} for r in range(world)]
    p0 = np.column_stack([rng.normal(20, 1, 30), rng.normal(1.8, 0.1, 30), rng.normal(600.0, 20.0, 30),
                          rng.normal(3.0, 0.1, 30), rng.normal(30, 2, 30)])
    s = EnsembleSampler(30, 5, sharded, vectorize=True, seed=11)
    s.run_mcmc(p0, 10)
    chains = comm.allgather_bytes(np.ascontiguousarray(s.chain).tobytes())
    assert all(c == chains[0] for c in chains)
    s1 = EnsembleSampler(30, 5, lambda p: orc(p), vectorize=True, seed=11)
    s1.run_mcmc(p0, 10)
    assert np.array_equal(s1.chain, s.chain)
    comm.barrier()
    print("FILECOMM_OK %d" % rank)


if __name__ == "__main__":
    try:
        main()
    except BaseException:
        try:
            open(os.path.join(os.environ["MBB_TEST_RDZV_DIR"], "abort"), "w").close()
        except Exception:
            pass
        raise
