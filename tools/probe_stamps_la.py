"""Diagnostic build only (-DMBB_STAMPS): phases of the look-ahead sampler's launch, movers and workers-ahead."""
import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mbb_emcee_amd import _native
_native.LIB_PATH = os.path.join(ROOT, "tools", "libmbb_hip_stamps.so"); _native.SIGNATURES["mbb_stamps"] = (C.c_int, [C.c_void_p, C.c_void_p, C.c_int])
import mbb_emcee_amd as mbb
from bench import make_likelihood, walkers, NW_PER_GPU
like, flux = make_likelihood(0)
ctx = like._sync_device()
lib = ctx.lib
lib.mbb_stamps(ctx.h, None, 0)
if len(sys.argv) > 1: ctx.set_option("lookahead_rows", int(sys.argv[1]))
if len(sys.argv) > 2: ctx.set_option("lookahead_waves", int(sys.argv[2]))
for look in (0, 1):
    ctx.set_option("lookahead_sampler", look & 1)
    smp = mbb.DeviceEnsembleSampler(NW_PER_GPU, 5, like, seed=11)
    smp.run_mcmc(walkers(1)[:NW_PER_GPU], 30, storechain=False)
    nb = int(ctx.info("last_grid"))
    st = np.zeros((nb, 32), dtype=np.uint64)
    lib.mbb_stamps(ctx.h, st.ctypes.data_as(C.c_void_p), nb)
    si = st.astype(np.int64)
    t0 = si[:, 7].min()
    nm = 125
    m = si[nb - nm:]
    med = lambda v: int(np.median(v))
    print(["plain", "look-ahead", "", "", "", "look-ahead, passband tables not staged (timing only)", "", "", "", "look-ahead, polynomial tables not staged (timing only)", "", "", "", "look-ahead, neither staged (timing only)"][look], "grid", nb)
    print("  movers: entry (after the launch's first entry) median %d max %d; entry->kernarg %d; ->records/prologue done %d; ->barrier1 passed %d; quadrature %d; barrier2 %d; epilogue %d; total %d; last exit %d"
          % (med(m[:, 7] - t0), (m[:, 7] - t0).max(), med(m[:, 0] - m[:, 7]), med(m[:, 10] - m[:, 0]), med(m[:, 2] - m[:, 10]),
             med(m[:, 3] - m[:, 2]), med(m[:, 4] - m[:, 3]), med(m[:, 6] - m[:, 4]), med(m[:, 6] - m[:, 7]), (m[:, 6] - t0).max()))
    if look:
        w = si[:nb - nm]
        print("  workers ahead (first row of each workgroup): entry median %d max %d; ->proposal formed %d; constructor %d; penalties+record %d; total %d; last exit %d"
              % (med(w[:, 7] - t0), (w[:, 7] - t0).max(), med(w[:, 8] - w[:, 7]), med(w[:, 9] - w[:, 8]), med(w[:, 10] - w[:, 9]),
                 med(w[:, 6] - w[:, 7]), (w[:, 6] - t0).max()))
    d = m[:, 17:32] - m[:, 7:8]
    print("  movers: arrival of waves 1..15 at the first barrier, after wave 0's entry (median over workgroups):", [int(x) for x in np.median(d, axis=0)])
