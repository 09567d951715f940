#!/usr/bin/env python3
"""Sampler form 7 at other ensemble sizes (up to one workgroup on every CU) against the plain launch train:
final state, acceptance counts and a stored chain bit for bit, and whether any run had to fall back."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mbb_emcee_amd as mbb
from bench import make_likelihood, TRUTH

nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
# (further arguments: ensemble sizes; form 7 ends at 256 walkers, beyond it the option is ignored and form 9 runs)
sizes = [int(a) for a in sys.argv[2:]] or [256, 254, 130, 64, 18, -250]
for nw in sizes:
    like, flux = make_likelihood(0)
    if nw < 0:
        # a covariance matrix: the two accept-test waves of a workgroup each need a scratch row of their own
        nw = -nw
        unc = 0.1 * flux + 1.0
        A = np.random.RandomState(3).normal(size=(8, 8))
        like.set_cov(np.diag(unc ** 2) + 0.02 * np.median(unc) ** 2 * A.dot(A.T))
        print("with a covariance matrix:")
    ctx = like._sync_device()
    p0 = np.asarray(TRUTH) * (1.0 + 0.02 * np.random.RandomState(nw).normal(size=(nw, 5)))
    res = []
    for name, opts in (("form 7", {"lookahead_sampler": 1, "flow_sampler": 1, "merged_flow_sampler": 1}), ("plain train", {"lookahead_sampler": 0})):
        for o, v in opts.items():
            ctx.set_option(o, v)
        s = mbb.DeviceEnsembleSampler(nw, 5, like, seed=nw)
        t0 = time.perf_counter()
        s.run_mcmc(p0, 500)
        s.run_mcmc(None, nsteps, storechain=False)
        pos, lnp, _ = s.run_mcmc(None, 500)
        dt = time.perf_counter() - t0
        res.append((pos, lnp, s.naccepted.copy(), s.chain.copy(), s.lnprobability.copy()))
        print("%4d walkers  %-12s %6d steps %.2f us per step incl. host, kernel form %d, fall-backs %d"
              % (nw, name, s.iterations, dt / s.iterations * 1e6, ctx.info("last_kernel_form"), ctx.info("flow_fallbacks")), flush=True)
    same = [bool(np.array_equal(a, b)) for a, b in zip(res[0], res[1])]
    print("      bitwise equal (positions, lnprob, counts, chain, stored lnprob):", same, flush=True)
    assert all(same)
