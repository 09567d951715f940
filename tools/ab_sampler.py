#!/usr/bin/env python3
"""us per MCMC step of the device sampler for a few ensemble sizes (stream time, median of 12 launches of 1000 steps) -- run
through tools/ab_lib.py to compare two builds:  python tools/ab_lib.py <lib.so> tools/ab_sampler.py [nwalkers ...]"""
import os, sys, zlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mbb_emcee_amd as mbb
from bench import make_likelihood, walkers
like, flux = make_likelihood(0)
ctx = like._sync_device()
out = []
for nw in [int(a) for a in sys.argv[1:]] or [250, 64, 512, 2000]:
    smp = mbb.DeviceEnsembleSampler(nw, 5, like, seed=11)
    pos, lnp, _ = smp.run_mcmc(walkers(8)[:nw], 60, storechain=False)
    k = 1000 if nw <= 512 else 300
    t = np.median([smp.advance_timed(k)[1] for _ in range(12)]) * 1e3 / k
    out.append("%d: %.3f (form %d, crc %08x)" % (nw, t, ctx.info("last_kernel_form"), zlib.crc32(pos.tobytes() + lnp.tobytes())))
    del smp
print("   ".join(out), flush=True)
