"""Diagnostic build only (-DMBB_STAMPS): where a 125-walker launch spends its cycles."""
import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mbb_emcee_amd import _native
_native.LIB_PATH = os.path.join(ROOT, "tools", "libmbb_hip_stamps.so")
import mbb_emcee_amd as mbb
from bench import BANDS, TRUTH, walkers
for name, opthin, noalpha in [("thick_walpha", 0, 0), ("thin_noalpha", 1, 1)]:
    like = mbb.likelihood(response=True, opthin=bool(opthin), noalpha=bool(noalpha))
    like.set_phot(BANDS, np.ones(8), np.ones(8))
    flux = like.model_flux(TRUTH)[0]
    like.set_phot(BANDS, flux, 0.1 * flux + 1)
    ctx = like._sync_device()
    lib = ctx.lib
    lib.mbb_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    lib.mbb_stamps(ctx.h, None, 0)
    n = 125
    p = walkers(1)[:n]
    d_pars = ctx.alloc(p.nbytes); d_pars.upload(p)
    d_lnl = ctx.alloc(n * 8); d_st = ctx.alloc(n * 4)
    ctx.lnlike_repeat_device(d_pars, n, d_lnl, d_st, 50); ctx.sync()
    st = np.zeros((n, 8), dtype=np.uint64)
    lib.mbb_stamps(ctx.h, st.ctypes.data_as(C.c_void_p), n)
    d = (st[:, 1:7].astype(np.int64) - st[:, 0:6].astype(np.int64))
    lab = ["warm+prologue", "barrier1", "phase2", "barrier2", "bandsum+barrier3", "lnL"]
    print(name, "cycles (median over blocks):", {l: int(np.median(d[:, i])) for i, l in enumerate(lab)},
          "total", int(np.median(st[:, 6].astype(np.int64) - st[:, 0].astype(np.int64))),
          "span over blocks", int(st[:, 6].max() - st[:, 0].min()))
