#!/usr/bin/env python3
"""The device sampler under the launch-geometry options (block_threads, walkers_per_group) at a few ensemble sizes: whatever
form a run takes, its chain against the plain launch train's, bit for bit.   python tools/soak_sampler_options.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mbb_emcee_amd as mbb
from bench import make_likelihood, TRUTH
bad = 0
for thr in (64, 128, 256, 512, 1024, 0):
    for wpb in (0, 1, 4):
        for nw in (18, 250, 300, 1000):
            like, flux = make_likelihood(0)
            ctx = like._sync_device()
            ctx.set_option("block_threads", thr); ctx.set_option("walkers_per_group", wpb)
            p0 = np.asarray(TRUTH) * (1.0 + 0.02 * np.random.RandomState(nw).normal(size=(nw, 5)))
            res = []
            for look in (1, 0):
                ctx.set_option("lookahead_sampler", look)
                s = mbb.DeviceEnsembleSampler(nw, 5, like, seed=5)
                pos, lnp, _ = s.run_mcmc(p0, 40)
                res.append((pos.copy(), lnp.copy(), s.chain.copy(), ctx.info("last_kernel_form"), ctx.info("flow_fallbacks")))
                del s
            same = all(np.array_equal(res[0][i], res[1][i]) for i in range(3))
            if not same or res[0][4]:
                bad += 1
            print("threads %4d wpb %d walkers %4d: form %d vs %d %s fallbacks %d" % (thr, wpb, nw, res[0][3], res[1][3], "equal" if same else "DIFFERENT", res[0][4]), flush=True)
print("wrong:", bad)
sys.exit(1 if bad else 0)
