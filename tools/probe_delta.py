"""Delta-function photometry (the reference's default, response_integrate = False):
kernel time per launch for a few band counts and batch sizes."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mbb_emcee_amd as mbb
from tools.probe_kernel import timeit

rng = np.random.RandomState(3)
for nb in (5, 12, 40):
    wave = np.exp(np.linspace(np.log(70.0), np.log(2000.0), nb))
    for opthin, noalpha in ((True, True), (False, False)):
        like = mbb.likelihood(opthin=opthin, noalpha=noalpha)
        like.set_phot(wave, np.ones(nb), np.ones(nb))
        truth = np.array([15.0, 1.8, 200.0, 3.0, 40.0])
        flux = like.model_flux(truth)[0]
        like.set_phot(wave, flux, 0.1 * flux + 0.5)
        ctx = like._sync_device()
        for n in (50, 125, 250000):
            p = truth * (1.0 + 0.05 * rng.normal(size=(n, 5)))
            d_pars = ctx.alloc(p.nbytes); d_pars.upload(p)
            d_lnl = ctx.alloc(n * 8); d_st = ctx.alloc(n * 4)
            us = timeit(ctx, d_pars, n, d_lnl, d_st, reps=200 if n < 4000 else 10)
            print("delta bands %2d  %s  n=%6d: %8.1f us  (wpb %d, %d threads, nseg %d)" %
                  (nb, "thin,noalpha" if opthin else "thick,alpha ", n, us, ctx.info("last_wpb"),
                   ctx.info("last_threads"), ctx.info("nseg")), flush=True)
            d_pars.free(); d_lnl.free(); d_st.free()
