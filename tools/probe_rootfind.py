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
ctx = like._sync_device()
allw = walkers(1)
for label, p in [("125 distinct", allw[:125]), ("125 x walker0", np.tile(allw[:1], (125, 1))), ("1 walker0", allw[:1]),
                 ("8 distinct", allw[:8]), ("32 distinct", allw[:32]), ("64 distinct", allw[:64])]:
    n = p.shape[0]
    d_pars = ctx.alloc(p.nbytes); d_pars.upload(p)
    d_lnl = ctx.alloc(n * 8); d_st = ctx.alloc(n * 4)
    us = timeit(ctx, d_pars, n, d_lnl, d_st, 200)
    ctx.set_option("debug", 1)
    ctx.lnlike_repeat_device(d_pars, n, d_lnl, d_st, 1); ctx.sync()
    st = d_st.download(np.int32, n)
    ctx.set_option("debug", 0)
    print(label, "%.1f us" % us, "iters:", np.bincount(st >> 8), flush=True)
