"""Worker for the world_size-2 gloo test of the sharding logic (CPU).
The per-shard lnprob is the CPU oracle -- test infrastructure standing in for the
GPU kernel so that the partition / pad / all-gather / reassemble path and a full
sharded sampler run can be exercised without a GPU."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class TorchComm(object):
    """Host-side all-gather through torch.distributed (gloo): the communicator protocol
    of mbb_emcee_amd.parallel.ShardedLikelihood, kept with the tests so that the product
    package has no torch dependency."""

    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)

    def allgather_host(self, x):
        import torch
        x = np.ascontiguousarray(x, dtype=np.float64)
        out = torch.empty(self.world * x.size, dtype=torch.float64)
        self.dist.all_gather_into_tensor(out, torch.from_numpy(x), group=self.group)
        return out.numpy()


def main():
    import torch.distributed as dist
    # ranks started by the test itself; they meet through a FILE the parent named: no port is picked by anybody but
    # gloo's own listeners (round 5: a port chosen, closed and handed to a launcher was taken when it went to listen)
    dist.init_process_group(backend="gloo", init_method="file://" + os.environ["MBB_TEST_RDZV_FILE"],
                            rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    rank, world = dist.get_rank(), dist.get_world_size()
    from mbb_emcee_amd.parallel import ShardedLikelihood, block_bounds
    from mbb_emcee_amd.ensemble import EnsembleSampler
    from mbb_emcee_amd.response import response_set
    from oracle import oracle as O
    names = ["PACS_100um", "PACS_160um", "SPIRE_250um", "SPIRE_350um", "SPIRE_500um"]
    wheel = response_set()
    bands = [(wheel[n].wavelength, wheel[n]._sedmult, wheel[n]._normfac) for n in names]
    flux = np.array([60.0, 95.0, 90.0, 60.0, 30.0])
    orc = O.OracleLikelihood(flux, 0.1 * flux, bands=bands, opthin=True, noalpha=True,
                             has_uplim=[0, 1, 1, 1, 0, 0], uplim=[np.inf, 20, 1500.0, 20, np.inf, np.inf])
    sharded = ShardedLikelihood(lambda p: orc(p), TorchComm())
    rng = np.random.RandomState(7)            # same stream on every rank
    for n in (1, 2, 7, 50, 51):
        pars = np.column_stack([rng.normal(20, 2, n), rng.normal(1.8, 0.2, n), np.full(n, 600.0),
                                np.full(n, 3.0), rng.normal(30, 3, n)])
        pars[0, 0] = 0.5 if n > 2 else pars[0, 0]            # a -inf row
        got = sharded(pars)
        ref = orc(pars)
        assert np.array_equal(got, ref), (rank, n)
        per, bounds = block_bounds(n, world)
        assert sum(b - a for a, b in bounds) == n
    # a whole sharded sampler run: identical chains on every rank
    p0 = np.column_stack([rng.normal(20, 1, 32), rng.normal(1.8, 0.1, 32), np.full(32, 600.0),
                          np.full(32, 3.0), rng.normal(30, 2, 32)])
    s = EnsembleSampler(32, 5, sharded, vectorize=True, seed=11)
    s.run_mcmc(p0, 20)
    chain = np.ascontiguousarray(s.chain)
    import torch
    gathered = [torch.empty(chain.shape, dtype=torch.float64) for _ in range(world)]
    dist.all_gather(gathered, torch.from_numpy(chain))
    for g in gathered:
        assert np.array_equal(g.numpy(), chain)
    # and equal to the unsharded run
    s1 = EnsembleSampler(32, 5, lambda p: orc(p), vectorize=True, seed=11)
    s1.run_mcmc(p0, 20)
    assert np.array_equal(s1.chain, chain)
    dist.barrier()
    dist.destroy_process_group()
    if rank == 0:
        print("DIST_OK")


if __name__ == "__main__":
    main()
