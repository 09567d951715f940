#!/usr/bin/env python3
"""Device sampler: plain launch train vs the look-ahead form (option lookahead_sampler):
chains must be bitwise equal; stream time per step of each (HIP events)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mbb_emcee_amd as mbb
from bench import make_likelihood, walkers, NW_PER_GPU

like, flux = make_likelihood(0)
ctx = like._sync_device()
p0 = walkers(1)[:NW_PER_GPU]


def run(look, rows=1, nsteps=40, runs=2, flow=0):
    ctx.set_option("flow_sampler", flow)
    ctx.set_option("lookahead_sampler", look)
    ctx.set_option("lookahead_rows", rows)
    smp = mbb.DeviceEnsembleSampler(NW_PER_GPU, 5, like, seed=11)
    pos, lnp, _ = smp.run_mcmc(p0, nsteps)
    for _ in range(runs - 1):
        pos, lnp, _ = smp.run_mcmc(None, nsteps)
    return smp.chain.copy(), smp.lnprobability.copy(), smp.naccepted.copy(), pos, lnp


def rate(look, rows=1, nsteps=2000, flow=0):
    ctx.set_option("flow_sampler", flow)
    ctx.set_option("lookahead_sampler", look)
    ctx.set_option("lookahead_rows", rows)
    smp = mbb.DeviceEnsembleSampler(NW_PER_GPU, 5, like, seed=11)
    smp.run_mcmc(p0, 20, storechain=False)
    smp.advance_async(200); ctx.sync()
    best = 1e9
    for _ in range(5):
        e0, e1 = ctx.event(), ctx.event()
        ctx.record(e0); smp.advance_async(nsteps); ctx.record(e1); ctx.sync()
        best = min(best, ctx.elapsed_ms(e0, e1) * 1e3 / nsteps)
    return best


a = run(0)
for rows in (0, 1, 4):
    b = run(1, rows)
    same = [bool(np.array_equal(x, y)) for x, y in zip(a, b)]
    print("rows per wave %d: chain, lnprob, naccepted, pos, lnp bitwise equal:" % rows, same,
          "acceptance %.3f" % (a[2].mean() / 80))
    assert all(same)
if "flow" in sys.argv:
    b = run(1, 0, flow=1)
    same = [bool(np.array_equal(x, y)) for x, y in zip(a, b)]
    print("one launch per run (SMODE 5): chain, lnprob, naccepted, pos, lnp bitwise equal:", same)
    if not all(same):
        d = np.argwhere(a[0] != b[0])
        print("first differences (walker, step, dim):", d[:5].tolist())
    for waves, rows in ((0, 0), (4, 1), (2, 1), (8, 1), (4, 2)):
        ctx.set_option("lookahead_waves", waves)
        print("one launch per run, %d waves x %d rows ahead per workgroup: %.3f us per step" % (waves, rows, rate(1, rows, flow=1)))
    ctx.set_option("lookahead_waves", 0)
print("plain      : %.3f us per step" % rate(0))
for rows, waves in ((0, 0), (1, 2), (1, 4), (1, 8), (2, 4), (4, 4), (4, 16)):
    ctx.set_option("lookahead_waves", waves)
    print("look-ahead (%d rows/wave, %d waves; 0 = the host's plan): %.3f us per step" % (rows, waves, rate(1, rows)))
ctx.set_option("lookahead_waves", 0)
print("plain      : %.3f us per step" % rate(0))
