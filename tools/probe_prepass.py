#!/usr/bin/env python3
"""cfg5's 250 000-row launch with the constructors on a row of 16 lanes per walker inside k_lnlike (option prepass 0) and by
k_walker_pre, a lane per walker, ahead of it (1); interleaved, HIP events, and a sweep of the launch geometry with the pre-pass.
    python tools/probe_prepass.py [--sweep]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.bench_cfg5 import setup

like, truths, p0 = setup(1000, 250)
ctx = like._sync_device()
n = 250000
flat = np.ascontiguousarray(p0.reshape(-1, 5))
d_pars = ctx.alloc(flat.nbytes); d_pars.upload(flat)
d_lnl = ctx.alloc(n * 8); d_st = ctx.alloc(n * 4)


def t(reps=10):
    ctx.lnlike_repeat_device(d_pars, n, d_lnl, d_st, 3); ctx.sync()
    e0, e1 = ctx.event(), ctx.event()
    ctx.record(e0); ctx.lnlike_repeat_device(d_pars, n, d_lnl, d_st, reps); ctx.record(e1); ctx.sync()
    return ctx.elapsed_ms(e0, e1) / reps


ref = None
ctx.lnlike_repeat_device(d_pars, n, d_lnl, d_st, 20); ctx.sync()       # (clocks)
for rnd in range(3):
    for pp in (0, 1):
        ctx.set_option("prepass", pp)
        ms = t()
        got = d_lnl.download(np.float64, n)
        if ref is None:
            ref = got
        assert np.array_equal(got, ref, equal_nan=True)
        print("prepass %d: %.4f ms  (%.3g evals/s)   wpb %d threads %d" % (pp, ms, n / ms * 1e3, ctx.info("last_wpb"), ctx.info("last_threads")), flush=True)
if "--sweep" in sys.argv:
    ctx.set_option("prepass", 1)
    for wpb, thr, stage in [(32, 512, -1), (16, 256, -1), (16, 512, -1), (64, 1024, -1), (64, 512, -1), (32, 256, -1), (48, 768, -1), (32, 1024, 1), (64, 1024, 1), (24, 384, -1), (8, 128, -1), (16, 128, -1), (32, 128, -1)]:
        ctx.set_option("walkers_per_group", wpb); ctx.set_option("block_threads", thr); ctx.set_option("stage_tables", stage)
        ms = t(5)
        assert np.array_equal(d_lnl.download(np.float64, n), ref, equal_nan=True)
        print("  prepass 1, wpb %2d threads %4d stage %2d (%d): %.4f ms" % (ctx.info("last_wpb"), ctx.info("last_threads"), stage, ctx.info("last_stage"), ms), flush=True)
