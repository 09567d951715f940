"""Interleaved A/B of the tail-chunk packing (option pack_tails) on one box."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mbb_emcee_amd as mbb
from bench import BANDS, TRUTH, walkers
from tools.probe_kernel import timeit

like = mbb.likelihood(response=True)
like.set_phot(BANDS, np.ones(8), np.ones(8))
flux = like.model_flux(TRUTH)[0]
like.set_phot(BANDS, flux, 0.1 * flux + 1)
allw = np.tile(walkers(1), (1000, 1))
for n in (125, 250000):
    res = {0: [], 1: []}
    for rep in range(4):
        for pk in (0, 1):
            like.context.set_option("pack_tails", pk)
            like._dirty = True
            ctx = like._sync_device()
            p = allw[:n]
            d_pars = ctx.alloc(p.nbytes); d_pars.upload(p)
            d_lnl = ctx.alloc(n * 8); d_st = ctx.alloc(n * 4)
            res[pk].append(timeit(ctx, d_pars, n, d_lnl, d_st, reps=400 if n < 4000 else 10))
            d_pars.free(); d_lnl.free(); d_st.free()
    print("n=%6d  chunk per leftover: %s us | shared tail chunks: %s us" %
          (n, " ".join("%.2f" % v for v in res[0]), " ".join("%.2f" % v for v in res[1])), flush=True)
