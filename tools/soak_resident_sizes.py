#!/usr/bin/env python3
"""The resident sampler form (9: k_flowa) at many ensemble sizes against the plain launch train, long runs:
final state, acceptance counts and the run's error flag, bit for bit; whether any run had to fall back.
    python tools/soak_resident_sizes.py [steps] [sizes ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mbb_emcee_amd as mbb
from bench import make_likelihood, TRUTH

nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
sizes = [int(a) for a in sys.argv[2:]] or [258, 300, 510, 514, 770, 1000, 1026, 1500, 1538, 2000, 2050, 2560, 3000, 3600, 4096]
bad = 0
for nw in sizes:
    like, flux = make_likelihood(0)
    ctx = like._sync_device()
    p0 = np.asarray(TRUTH) * (1.0 + 0.02 * np.random.RandomState(nw).normal(size=(nw, 5)))
    res = []
    for name, opts in (("resident", {"lookahead_sampler": 1}), ("train", {"lookahead_sampler": 0})):
        for k, v in opts.items():
            ctx.set_option(k, v)
        s = mbb.DeviceEnsembleSampler(nw, 5, like, seed=7)
        t0 = time.time()
        pos, lnp, _ = s.run_mcmc(p0, nsteps, storechain=False)
        pos, lnp, _ = s.run_mcmc(None, 37, storechain=False)
        res.append((pos.copy(), lnp.copy(), s.naccepted.copy(), ctx.info("last_kernel_form"), ctx.info("last_wpb"), ctx.info("flow_fallbacks"), time.time() - t0))
        del s
    same = all(np.array_equal(res[0][i], res[1][i]) for i in range(3))
    bad += 0 if same and res[0][5] == 0 else 1
    print("%5d walkers: form %d (W %d) against form %d: %s, fall-backs %d  (%.1f s / %.1f s)" % (nw, res[0][3], res[0][4], res[1][3], "equal" if same else "DIFFERENT", res[0][5], res[0][6], res[1][6]), flush=True)
print("soak: %d size(s) wrong" % bad)
sys.exit(1 if bad else 0)
