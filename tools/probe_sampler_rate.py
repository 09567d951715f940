"""Stream time per step of the device-resident sampler (250 walkers, cfg2)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mbb_emcee_amd as mbb
from bench import make_likelihood, walkers
like, flux = make_likelihood(0)
ctx = like._sync_device()
s = mbb.DeviceEnsembleSampler(250, 5, like, seed=5)
s.run_mcmc(walkers(1)[:250], 50, storechain=False)
for rep in range(3):
    e0, e1 = ctx.event(), ctx.event()
    ctx.record(e0); s.advance_async(2000); ctx.record(e1); ctx.sync()
    print("device sampler: %.2f us per step (stream)" % (ctx.elapsed_ms(e0, e1) * 1e3 / 2000), flush=True)
