"""Interleaved A/B of the L2-table and LDS-staged variants (one process, rounds)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mbb_emcee_amd as mbb
from tools.bench_cfg5 import setup
from bench import make_likelihood, walkers

def timeit(ctx, d_pars, n, d_lnl, d_st, reps=300):
    """us per launch of the fused kernel, HIP events around `reps` back-to-back launches"""
    ctx.lnlike_repeat_device(d_pars, n, d_lnl, d_st, 20); ctx.sync()
    e0, e1 = ctx.event(), ctx.event()
    ctx.record(e0); ctx.lnlike_repeat_device(d_pars, n, d_lnl, d_st, reps); ctx.record(e1); ctx.sync()
    return ctx.elapsed_ms(e0, e1) * 1e3 / reps


def run(ctx, d_pars, n, d_lnl, d_st, cfgs, reps, rounds=7):
    res = {c: [] for c in cfgs}
    for r in range(rounds):
        for c in cfgs:
            ctx.set_option("walkers_per_group", c[0]); ctx.set_option("block_threads", c[1]); ctx.set_option("stage_tables", c[2])
            res[c].append(timeit(ctx, d_pars, n, d_lnl, d_st, reps))
    for c in cfgs:
        v = np.array(res[c]); print("  wpb=%d thr=%d stage=%d: median %.2f us  min %.2f" % (c[0], c[1], c[2], np.median(v), v.min()), flush=True)
    ctx.set_option("walkers_per_group", 0); ctx.set_option("block_threads", 0); ctx.set_option("stage_tables", -1)

like, truths, p0 = setup(1000, 250)
ctx = like._sync_device()
flat = np.ascontiguousarray(p0.reshape(-1, 5)); n = flat.shape[0]
d_pars = ctx.alloc(flat.nbytes); d_pars.upload(flat); d_lnl = ctx.alloc(n * 8); d_st = ctx.alloc(n * 4)
print("cfg5 n=250000")
run(ctx, d_pars, n, d_lnl, d_st, [(16, 256, 0), (32, 512, 1), (16, 512, 1), (64, 512, 1), (32, 512, 0)], 5)
like1, flux = make_likelihood(0)
c1 = like1._sync_device()
for n in (125, 2000, 16000):
    p = np.tile(walkers(1), (8, 1))[:n]
    dp = c1.alloc(p.nbytes); dp.upload(p); dl = c1.alloc(n * 8); ds = c1.alloc(n * 4)
    print("cfg2 n=%d" % n)
    if n == 125:
        run(c1, dp, n, dl, ds, [(1, 768, 0), (1, 768, 1), (1, 1024, 1)], 200)
    elif n == 2000:
        run(c1, dp, n, dl, ds, [(1, 256, 0), (2, 256, 0), (2, 512, 1), (4, 512, 1), (8, 512, 1)], 100)
    else:
        run(c1, dp, n, dl, ds, [(8, 256, 0), (16, 256, 0), (16, 512, 1), (32, 512, 1)], 20)
