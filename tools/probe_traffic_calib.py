"""Calibration run for the FETCH_SIZE counter on this kernel's access pattern:
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d <dir> -- python3 tools/probe_traffic_calib.py
launches the staged 125-walker kernel shape with 1, 8, 32 and 125 workgroups.  One
workgroup must fetch the passband tables exactly once (known byte count); the ratio of
the reported to the known bytes is the correction for the other rows."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mbb_emcee_amd as mbb
from bench import make_likelihood, walkers
like, flux = make_likelihood(0)
ctx = like._sync_device()
allw = walkers(1)
print("table bytes per copy:", ctx.info("nchunk") * 64 * 3 * 8, "nchunk", ctx.info("nchunk"))
for n in (1, 8, 32, 125):
    p = allw[:n]
    d_pars = ctx.alloc(p.nbytes); d_pars.upload(p)
    d_lnl = ctx.alloc(n * 8); d_st = ctx.alloc(n * 4)
    ctx.lnlike_repeat_device(d_pars, n, d_lnl, d_st, 40); ctx.sync()
    print("n =", n, "grid", ctx.info("last_grid"), "threads", ctx.info("last_threads"), "staged", ctx.info("last_stage"))

# the same with every band twice: the table doubles, nothing else changes
from bench import BANDS, TRUTH
like2 = mbb.likelihood(response=True)
like2.set_phot(BANDS + BANDS, np.ones(16), np.ones(16))
f2 = like2.model_flux(TRUTH)[0]
like2.set_phot(BANDS + BANDS, f2, 0.1 * f2 + 1)
c2 = like2._sync_device()
print("doubled table bytes per copy:", c2.info("nchunk") * 64 * 3 * 8, "nchunk", c2.info("nchunk"))
for n in (2, 9):
    p = allw[:n]
    d_pars = c2.alloc(p.nbytes); d_pars.upload(p)
    d_lnl = c2.alloc(n * 8); d_st = c2.alloc(n * 4)
    c2.lnlike_repeat_device(d_pars, n, d_lnl, d_st, 40); c2.sync()
    print("doubled: n =", n, "grid", c2.info("last_grid"), "threads", c2.info("last_threads"), "staged", c2.info("last_stage"))
