#!/usr/bin/env python3
"""Sharded device sampler, ranks as processes on ONE GPU (a rehearsal: inside a device the "peer" copies
are a system-scope hop away, not an xGMI link): one launch per run across the ranks (k_lnlike SMODE 6)
against one launch per half-step with the rows exchanged after it (SMODE 2).

    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29600 tools/probe_xflow.py [walkers]
"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch.distributed as dist
dist.init_process_group(backend="gloo")
rank, world = dist.get_rank(), dist.get_world_size()
import mbb_emcee_amd as mbb
from mbb_emcee_amd import parallel
from bench import make_likelihood, walkers

nw = int(sys.argv[1]) if len(sys.argv) > 1 else 100
like, flux = make_likelihood(0)
ctx = like._sync_device()
parallel.ipc_exchange_setup(ctx, rank, world, dist, max_rows=1024)
p0 = walkers(1)[:nw]
for form in (1, 0, 1):
    ctx.set_option("sharded_flow_sampler", form)
    smp = mbb.DeviceEnsembleSampler(nw, 5, like, seed=11)
    smp.run_mcmc(p0, 40, storechain=False)
    smp.advance_async(200); ctx.sync(); dist.barrier()
    best = 1e9
    for _ in range(3):
        e0, e1 = ctx.event(), ctx.event()
        ctx.record(e0); smp.advance_async(1500); ctx.record(e1); ctx.sync(); dist.barrier()
        best = min(best, ctx.elapsed_ms(e0, e1) * 1e3 / 1500)
    pos, lnp, _ = smp.run_mcmc(None, 0)
    if rank == 0:
        print("%d ranks x %d movers, %s: %.2f us per step (kernel form %d)" %
              (world, nw // 2 // world, "one launch per run" if form else "one launch per half-step", best, ctx.info("last_kernel_form")))
    del smp
    import gc; gc.collect()
ctx.xchg_close()
dist.destroy_process_group()
