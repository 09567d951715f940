#!/usr/bin/env python3
"""Sampler form 7: stream time per step against options of the band layout / workgroup width."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mbb_emcee_amd as mbb
from bench import BANDS, TRUTH, walkers, NW_PER_GPU


def rate(opts, nsteps=2000):
    like = mbb.likelihood(response=True)
    for k, v in opts.items():
        if k != "debug":
            like.context.set_option(k, v)
    like.set_phot(BANDS, np.ones(8), np.ones(8))
    flux = like.model_flux(TRUTH)[0]
    like.set_phot(BANDS, flux, 0.1 * flux + 1.0)
    ctx = like._sync_device()
    if "debug" in opts:
        ctx.set_option("debug", opts["debug"])
    smp = mbb.DeviceEnsembleSampler(NW_PER_GPU, 5, like, seed=11)
    smp.run_mcmc(walkers(1)[:NW_PER_GPU], 20, storechain=False)
    smp.advance_async(200); ctx.sync()
    best = 1e9
    for _ in range(5):
        e0, e1 = ctx.event(), ctx.event()
        ctx.record(e0); smp.advance_async(nsteps); ctx.record(e1); ctx.sync()
        best = min(best, ctx.elapsed_ms(e0, e1) * 1e3 / nsteps)
    return best, int(ctx.info("nunit")), int(ctx.info("last_threads")), int(ctx.info("last_kernel_form"))


for opts in ({}, {"seg_chunks": 3}, {"seg_chunks": 6}, {"block_threads": 768}, {"stage_tables": 0}, {}):
    print(opts, "-> %.3f us per step, %d units per walker, %d threads, form %d" % rate(opts), flush=True)
