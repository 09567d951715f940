import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import mbb_emcee_amd as mbb
from bench import make_likelihood, walkers, NW_PER_GPU
like, flux = make_likelihood(0)
ctx = like._sync_device()
for shift in (0, 1, 2, 4, 8, 9, 16, 24, 32, 48, 64, 96, 128, 160, 192, 255):     # units of 64 bytes
    ctx.set_option("dbg_done_shift", shift)
    smp = mbb.DeviceEnsembleSampler(NW_PER_GPU, 5, like, seed=11)
    smp.run_mcmc(walkers(1)[:NW_PER_GPU], 60, storechain=False)
    out = [smp.advance_timed(1000)[1] * 1e3 for _ in range(8)]
    print("shift %3d x 64 B: even %.1f odd %.1f us per 1000 steps   %s" % (shift, np.median(out[0::2]), np.median(out[1::2]), " ".join("%.0f" % o for o in out)), flush=True)
    del smp
