#!/usr/bin/env python3
"""Soak of the one-launch look-ahead sampler: a long run against the plain launch train, final
state / acceptance counts / a thinned chain compared bit for bit.  python tools/soak_flow.py [steps]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mbb_emcee_amd as mbb
from bench import make_likelihood, walkers, NW_PER_GPU

nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
like, flux = make_likelihood(0)
ctx = like._sync_device()
p0 = walkers(1)[:NW_PER_GPU]
res = []
for name, opts in (("one launch per run", {"lookahead_sampler": 1, "flow_sampler": 1}), ("plain train", {"lookahead_sampler": 0})):
    for o, v in opts.items():
        ctx.set_option(o, v)
    s = mbb.DeviceEnsembleSampler(NW_PER_GPU, 5, like, seed=2026)
    t0 = time.perf_counter()
    pos, lnp, _ = s.run_mcmc(p0, 1000)                 # stored
    for _ in range((nsteps - 2000) // 50000):
        pos, lnp, _ = s.run_mcmc(None, 50000, storechain=False)
    pos, lnp, _ = s.run_mcmc(None, 1000)               # stored again
    dt = time.perf_counter() - t0
    res.append((pos, lnp, s.naccepted.copy(), s.chain.copy(), s.lnprobability.copy()))
    print("%-20s %d steps in %.2f s (%.2f us per step incl. host), acceptance %.4f, kernel form %d"
          % (name, s.iterations, dt, dt / s.iterations * 1e6, s.acceptance_fraction.mean(), ctx.info("last_kernel_form")))
same = [bool(np.array_equal(a, b)) for a, b in zip(res[0], res[1])]
print("final positions, lnprob, acceptance counts, stored chain, stored lnprob bitwise equal:", same)
print("half-step hand-overs checked: %d walkers x %d half-steps" % (NW_PER_GPU // 2, 2 * res and 2 * (1000 + ((nsteps - 2000) // 50000) * 50000 + 1000)))
assert all(same)
