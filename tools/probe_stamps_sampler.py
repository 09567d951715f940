"""Diagnostic build only (-DMBB_STAMPS): phase stamps of the sampler variant of the fused
kernel (device-resident stretch move), next to the plain variant."""
import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mbb_emcee_amd import _native
_native.LIB_PATH = os.path.join(ROOT, "tools", "libmbb_hip_stamps.so"); _native.SIGNATURES["mbb_stamps"] = (C.c_int, [C.c_void_p, C.c_void_p, C.c_int])
import mbb_emcee_amd as mbb
from bench import BANDS, TRUTH, walkers

like = mbb.likelihood(response=True)
like.set_phot(BANDS, np.ones(8), np.ones(8))
flux = like.model_flux(TRUTH)[0]
like.set_phot(BANDS, flux, 0.1 * flux + 1)
ctx = like._sync_device()
lib = ctx.lib
lib.mbb_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
lib.mbb_stamps(ctx.h, None, 0)
n = 125
lab = ["warm+prologue", "barrier1", "phase2", "barrier2", "band sums", "chi2+lnL(+accept)"]

def report(name):
    st = np.zeros((n, 32), dtype=np.uint64)
    lib.mbb_stamps(ctx.h, st.ctypes.data_as(C.c_void_p), n)
    si = st.astype(np.int64)
    d = si[:, 1:7] - si[:, 0:6]
    print(name, {l: int(np.median(d[:, i])) for i, l in enumerate(lab)},
          "entry->kernarg", int(np.median(si[:, 0] - si[:, 7])), "total", int(np.median(si[:, 6] - si[:, 7])))

p = walkers(1)[:n]
d_pars = ctx.alloc(p.nbytes); d_pars.upload(p)
d_lnl = ctx.alloc(n * 8); d_st = ctx.alloc(n * 4)
ctx.lnlike_repeat_device(d_pars, n, d_lnl, d_st, 50); ctx.sync()
report("plain  ")
s = mbb.DeviceEnsembleSampler(250, 5, like, seed=3)
s.run_mcmc(walkers(1)[:250], 30, storechain=False)
ctx.sync()
report("sampler")
