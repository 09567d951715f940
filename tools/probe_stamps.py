"""Diagnostic build only (-DMBB_STAMPS): where a 125-walker launch spends its cycles."""
import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mbb_emcee_amd import _native
_native.LIB_PATH = os.path.join(ROOT, "tools", "libmbb_hip_stamps.so"); _native.SIGNATURES["mbb_stamps"] = (C.c_int, [C.c_void_p, C.c_void_p, C.c_int])
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
    st = np.zeros((n, 32), dtype=np.uint64)
    lib.mbb_stamps(ctx.h, st.ctypes.data_as(C.c_void_p), n)
    si = st.astype(np.int64)
    fine = {"entry->kernarg": si[:, 0] - si[:, 7], "kernarg->row loaded": si[:, 8] - si[:, 0],
            "sed_prologue": si[:, 9] - si[:, 8], "penalties": si[:, 10] - si[:, 9],
            "LDS write": si[:, 1] - si[:, 10]}
    if name == "thick_walpha":
        inner = {"log+fp32 bracket": si[:, 11] - si[:, 8], "exp round 1": si[:, 12] - si[:, 11],
                 "exp round 2": si[:, 13] - si[:, 12], "Newton step": si[:, 14] - si[:, 13],
                 "kappa exps": si[:, 15] - si[:, 14], "quotients": si[:, 9] - si[:, 15]}
        print(name, "sed_prologue detail:", {k: int(np.median(v)) for k, v in inner.items()})
    print(name, "phase 1 detail:", {k: int(np.median(v)) for k, v in fine.items()})
    d = (st[:, 1:7].astype(np.int64) - st[:, 0:6].astype(np.int64))
    print(name, "phase 3 detail:", {"band sums": int(np.median(si[:, 5] - si[:, 4])),
          "reduction": int(np.median(si[:, 16] - si[:, 5])), "lnL + stores": int(np.median(si[:, 6] - si[:, 16]))})
    lab = ["warm+prologue", "barrier1", "phase2", "barrier2", "band sums", "chi2+lnL"]
    print(name, "cycles (median over blocks):", {l: int(np.median(d[:, i])) for i, l in enumerate(lab)},
          "total", int(np.median(st[:, 6].astype(np.int64) - st[:, 0].astype(np.int64))),
          "span over blocks", int(st[:, 6].max() - st[:, 0].min()))
