#!/usr/bin/env python3
"""What the host adds to a short timed region (bench.py's on one GPU: clock, event, ONE launch of K steps, event,
stream wait, clock -- mbb_sampler_advance_timed): wall clock against the stream time between the two events.

    python tools/probe_timed_region.py [K ...]

(Round 4 tried polling hipStreamQuery for up to 200 us before blocking in hipStreamSynchronize: no difference, 12.5 us of
host either way -- profiles/r04/timed_region.txt; the option was not kept.)
"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mbb_emcee_amd as mbb
from bench import make_likelihood, walkers, NW_PER_GPU

like, flux = make_likelihood(0)
ctx = like._sync_device()
smp = mbb.DeviceEnsembleSampler(NW_PER_GPU, 5, like, seed=11)
smp.run_mcmc(walkers(1)[:NW_PER_GPU], 50, storechain=False)
for k in [int(a) for a in sys.argv[1:]] or [20, 5, 100, 2000]:
    res = []
    for rnd in range(100):
        smp.advance_async(5); ctx.sync()
        wall, ms = smp.advance_timed(k)
        res.append((wall * 1e6, ms * 1e3))
    v = np.median(np.array(res), axis=0)
    print("K = %4d: wall %8.2f us = stream %8.2f + host %5.2f   -> %.3f us per step, %.3g evals/s"
          % (k, v[0], v[1], v[0] - v[1], v[0] / k, NW_PER_GPU * k / (v[0] * 1e-6)), flush=True)
