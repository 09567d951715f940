#!/usr/bin/env python3
"""Form 7 with two pairs of walkers per workgroup against the plain launch train: chains bit for bit, and timing."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from mbb_emcee_amd import _native
if os.environ.get("MBB_LIB"):
    _native.LIB_PATH = os.path.abspath(os.environ["MBB_LIB"])
import mbb_emcee_amd as mbb
from bench import make_likelihood, TRUTH

like, flux = make_likelihood(0)
ctx = like._sync_device()
for nw, pairs in ((100, 2), (102, 2), (250, 2), (384, 2), (510, 2), (512, 2), (250, 0)):
    p0 = TRUTH * (1.0 + 0.02 * np.random.RandomState(nw).normal(size=(nw, 5)))
    res = []
    for name, opts in (("form 7", {"lookahead_sampler": 1, "merged_flow_sampler": 1, "merged_flow_pairs": pairs}), ("train", {"lookahead_sampler": 0})):
        for o, v in opts.items():
            ctx.set_option(o, v)
        s = mbb.DeviceEnsembleSampler(nw, 5, like, seed=nw)
        a = s.run_mcmc(p0, 40)
        b = s.run_mcmc(None, 300, storechain=False)
        s.advance_async(200); ctx.sync()
        e0, e1 = ctx.event(), ctx.event()
        ctx.record(e0); s.advance_async(1000); ctx.record(e1); ctx.sync()
        us = ctx.elapsed_ms(e0, e1)
        c = s.run_mcmc(None, 10)
        res.append((a[0], a[1], b[0], c[0], c[1], s.chain.copy(), s.naccepted.copy()))
        print("%4d walkers pairs=%d %-7s form %d wpb %d grid %3d  %.2f us/step  fallbacks %d" % (
            nw, pairs, name, ctx.info("last_kernel_form"), ctx.info("last_wpb"), ctx.info("last_grid"), us, ctx.info("flow_fallbacks")), flush=True)
    print("      equal:", [bool(np.array_equal(x, y)) for x, y in zip(res[0], res[1])], flush=True)
ctx.set_option("lookahead_sampler", 1); ctx.set_option("merged_flow_pairs", 0)
