"""Diagnostic build only (-DMBB_STAMPS): phases of a half-step inside the one-launch look-ahead run (SMODE 5),
movers and workers-ahead; values of the run's last half-step, cycles (s_memtime), median over workgroups."""
import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mbb_emcee_amd import _native
_native.LIB_PATH = os.environ.get("MBB_STAMPS_LIB", os.path.join(ROOT, "tools", "libmbb_hip_stamps.so")); _native.SIGNATURES["mbb_stamps"] = (C.c_int, [C.c_void_p, C.c_void_p, C.c_int])
import mbb_emcee_amd as mbb
from bench import make_likelihood, walkers, NW_PER_GPU
like, flux = make_likelihood(0)
ctx = like._sync_device()
lib = ctx.lib
lib.mbb_stamps(ctx.h, None, 0)
ctx.set_option("flow_sampler", 1)
smp = mbb.DeviceEnsembleSampler(NW_PER_GPU, 5, like, seed=11)
smp.run_mcmc(walkers(1)[:NW_PER_GPU], 200, storechain=False)
nb = int(ctx.info("last_grid"))
st = np.zeros((nb, 32), dtype=np.uint64)
lib.mbb_stamps(ctx.h, st.ctypes.data_as(C.c_void_p), nb)
si = st.astype(np.int64)
nm = 125
m, w = si[nb - nm:], si[:nb - nm]
med = lambda v: int(np.median(v))
print("movers: wait %d; loads+select %d; barrier1 %d; quadrature %d; barrier2 %d; epilogue %d; half-step %d"
      % (med(m[:, 12] - m[:, 11]), med(m[:, 10] - m[:, 12]), med(m[:, 2] - m[:, 10]), med(m[:, 3] - m[:, 2]),
         med(m[:, 4] - m[:, 3]), med(m[:, 6] - m[:, 4]), med(m[:, 6] - m[:, 11])))
print("workers ahead (first row of each workgroup): wait %d; loads+proposal %d; constructor %d; penalties+record %d; half-step %d"
      % (med(w[:, 12] - w[:, 11]), med(w[:, 8] - w[:, 12]), med(w[:, 9] - w[:, 8]), med(w[:, 10] - w[:, 9]), med(w[:, 10] - w[:, 11])))
d = m[:, 17:32] - m[:, 2:3]
print("movers: quadrature time of waves 1..15 (after the first barrier; wave 0: %d):" % med(m[:, 3] - m[:, 2]), [int(x) for x in np.median(d, axis=0)])
ut = ctx.info("nunit")
print("units per walker:", ut)
big = np.zeros((256 * 8, 32), dtype=np.uint64)
lib.mbb_stamps(ctx.h, big.ctypes.data_as(C.c_void_p), 256 * 8)
bi = big.astype(np.int64)
mv = slice(nb - nm, nb)
pg = lambda p: bi[256 * p:256 * p + nb][mv][:, 16:32]
t1, t2, t3, t4 = pg(1), pg(2), pg(3), pg(4)
row = lambda x: [int(v) for v in np.median(x, axis=0)]
print("per wave 0..15, medians over the movers: barrier-1 exit after wave 0's:", row(t1 - t1[:, :1]))
print("  walker constants read:", row(t2 - t1))
print("  sample arithmetic:", row(t3 - t2))
print("  reduction + store:", row(t4 - t3))
