#!/usr/bin/env python3
"""Form 7: stream time of short runs (one launch each) against their number of steps -- the fixed cost of a launch
(dispatch, table staging, filling of the pipeline) and the steady rate."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mbb_emcee_amd as mbb
from bench import make_likelihood, walkers, NW_PER_GPU
like, flux = make_likelihood(0)
ctx = like._sync_device()
smp = mbb.DeviceEnsembleSampler(NW_PER_GPU, 5, like, seed=11)
smp.run_mcmc(walkers(1)[:NW_PER_GPU], 50, storechain=False)
ks, ts = [], []
for k in (1, 2, 3, 5, 10, 20, 50, 200):
    best = 1e9
    for _ in range(15):
        e0, e1 = ctx.event(), ctx.event()
        ctx.record(e0); smp.advance_async(k); ctx.record(e1); ctx.sync()
        best = min(best, ctx.elapsed_ms(e0, e1) * 1e3)
    ks.append(k); ts.append(best)
    print("%4d steps: %8.2f us on the stream, %.2f per step" % (k, best, best / k), flush=True)
b, a = np.polyfit(ks[3:], ts[3:], 1)
print("fit over 5..200 steps: %.2f us fixed + %.3f us per step (form %d)" % (a, b, ctx.info("last_kernel_form")))
