#!/usr/bin/env python3
"""Times the fused lnL kernel for model variants / batch sizes / launch geometries
(HIP events around back-to-back launches).  Development tool, GPU box only."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mbb_emcee_amd as mbb
from bench import BANDS, TRUTH, walkers

def timeit(ctx, d_pars, n, d_lnl, d_st, reps=300):
    ctx.lnlike_repeat_device(d_pars, n, d_lnl, d_st, 20); ctx.sync()
    e0, e1 = ctx.event(), ctx.event()
    ctx.record(e0); ctx.lnlike_repeat_device(d_pars, n, d_lnl, d_st, reps); ctx.record(e1); ctx.sync()
    return ctx.elapsed_ms(e0, e1) * 1e3 / reps

def main():
    allw = np.tile(walkers(1), (200, 1))
    for name, opthin, noalpha in [("thick_walpha", 0, 0), ("thick_noalpha", 0, 1), ("thin_walpha", 1, 0), ("thin_noalpha", 1, 1)]:
        like = mbb.likelihood(response=True, opthin=bool(opthin), noalpha=bool(noalpha))
        like.set_phot(BANDS, np.ones(8), np.ones(8))
        flux = like.model_flux(TRUTH)[0]
        like.set_phot(BANDS, flux, 0.1 * flux + 1)
        ctx = like._sync_device()
        for n in (1, 125, 250, 2000, 32768, 250000):
            p = allw[:n]
            d_pars = ctx.alloc(p.nbytes); d_pars.upload(p)
            d_lnl = ctx.alloc(n * 8); d_st = ctx.alloc(n * 4)
            res = []
            geos = [(0, 0)] if n > 4000 else [(0, 0), (1, 256), (1, 512), (1, 1024), (2, 256), (2, 512), (4, 512), (8, 512)]
            if n > 4000: geos += [(16, 256), (16, 512), (32, 512), (32, 1024), (64, 1024)]
            for wpb, thr in geos:
                ctx.set_option("walkers_per_group", wpb); ctx.set_option("block_threads", thr)
                us = timeit(ctx, d_pars, n, d_lnl, d_st, reps=200 if n < 4000 else 5)
                res.append("wpb=%d thr=%d: %.1f us" % (ctx.info("last_wpb"), ctx.info("last_threads"), us))
            ctx.set_option("walkers_per_group", 0); ctx.set_option("block_threads", 0)
            print(name, "n=%d" % n, " | ".join(res), flush=True)
            d_pars.free(); d_lnl.free(); d_st.free()

if __name__ == "__main__":
    main()
