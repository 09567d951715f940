import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mbb_emcee_amd as mbb
from bench import make_likelihood, walkers, NW_PER_GPU
like, flux = make_likelihood(0)
ctx = like._sync_device()
smp = mbb.DeviceEnsembleSampler(NW_PER_GPU, 5, like, seed=11)
smp.run_mcmc(walkers(1)[:NW_PER_GPU], 50, storechain=False)
for n in (250, 1000, 4000):
    smp.advance_async(n); ctx.sync()
    best = 1e9
    for _ in range(5):
        e0, e1 = ctx.event(), ctx.event()
        ctx.record(e0); smp.advance_async(n); ctx.record(e1); ctx.sync()
        best = min(best, ctx.elapsed_ms(e0, e1) * 1e3)
    print("%5d steps per call: %.1f us in all, %.3f us per step" % (n, best, best / n))
