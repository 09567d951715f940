#!/usr/bin/env python3
"""Debug: where does a sampler that gave up (and was redone) part from one that did not?"""
import os, sys, warnings
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mbb_emcee_amd as mbb
from bench import make_likelihood
warnings.simplefilter("ignore")

for merged in (1, 0):
    like, flux = make_likelihood(0)
    ctx = like._sync_device()
    p0 = np.array([12.0, 1.8, 600.0, 3.0, 40.0]) * (1.0 + 0.02 * np.random.RandomState(8).normal(size=(100, 5)))
    ctx.set_option("lookahead_sampler", 0)
    s0 = mbb.DeviceEnsembleSampler(100, 5, like, seed=4)
    s0.run_mcmc(p0, 6)
    ctx.set_option("lookahead_sampler", 1); ctx.set_option("merged_flow_sampler", merged)
    s1 = mbb.DeviceEnsembleSampler(100, 5, like, seed=4)
    s1.run_mcmc(p0, 6)

    def step(n, spin, tag):
        ctx.set_option("flow_spin_log2", spin)
        b = s1.run_mcmc(None, n)
        f1 = ctx.info("last_kernel_form")
        ctx.set_option("flow_spin_log2", 0)
        ctx.set_option("lookahead_sampler", 0)
        a = s0.run_mcmc(None, n)
        ctx.set_option("lookahead_sampler", 1)
        print("merged %d %-28s n=%d form %d fallbacks %d resting %d  equal pos %s lnp %s" % (
            merged, tag, n, f1, ctx.info("flow_fallbacks"), ctx.info("flow_resting"),
            np.array_equal(a[0], b[0]), np.array_equal(a[1], b[1])), flush=True)
    step(5, 1, "give-up 1")
    step(4, 0, "one launch")
    for k in range(3):
        step(2, 1, "give-up in a row %d" % k)
    for k in range(ctx.info("flow_resting")):
        step(2, 0, "resting %d" % k)
    step(2, 0, "back")
    step(7, 0, "again")
