#!/usr/bin/env python3
"""Soak of sampler form 5 (k_lnlike SMODE 5, what 258-454 walkers take) against the plain launch train after
round 3's role-local argument views: final state, counts and stored chains bit for bit at several ensemble sizes."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mbb_emcee_amd as mbb
from bench import make_likelihood, TRUTH

nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
for nw in (258, 320, 400, 450, 250):
    like, flux = make_likelihood(0)
    ctx = like._sync_device()
    p0 = np.asarray(TRUTH) * (1.0 + 0.02 * np.random.RandomState(nw).normal(size=(nw, 5)))
    res = []
    for name, opts in (("form 5", {"lookahead_sampler": 1, "flow_sampler": 1, "merged_flow_sampler": 0}), ("plain train", {"lookahead_sampler": 0})):
        for o, v in opts.items():
            ctx.set_option(o, v)
        s = mbb.DeviceEnsembleSampler(nw, 5, like, seed=nw)
        t0 = time.perf_counter()
        s.run_mcmc(p0, 300)
        s.run_mcmc(None, nsteps, storechain=False)
        pos, lnp, _ = s.run_mcmc(None, 300)
        dt = time.perf_counter() - t0
        res.append((pos, lnp, s.naccepted.copy(), s.chain.copy(), s.lnprobability.copy()))
        print("%4d walkers  %-12s %6d steps %.2f us per step incl. host, kernel form %d, fall-backs %d"
              % (nw, name, s.iterations, dt / s.iterations * 1e6, ctx.info("last_kernel_form"), ctx.info("flow_fallbacks")), flush=True)
    same = [bool(np.array_equal(a, b)) for a, b in zip(res[0], res[1])]
    print("      bitwise equal (positions, lnprob, counts, chain, stored lnprob):", same, flush=True)
    assert all(same)
