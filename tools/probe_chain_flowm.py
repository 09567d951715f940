"""Diagnostic build only (-DMBB_STAMPS): the hand-over chain of sampler form 7 from event times of the launch's
last 64 half-steps (s_memrealtime, 10 ns ticks; all CUs share that clock)."""
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
ctx.set_option("flow_sampler", 1); ctx.set_option("merged_flow_sampler", 1)
smp = mbb.DeviceEnsembleSampler(NW_PER_GPU, 5, like, seed=11)
smp.run_mcmc(walkers(1)[:NW_PER_GPU], 100, storechain=False)
NS = 500
e0, e1 = ctx.event(), ctx.event()
ctx.record(e0); smp.advance_async(NS); ctx.record(e1); ctx.sync()
print("%.3f us per step" % (ctx.elapsed_ms(e0, e1) * 1e3 / NS))
nb = int(ctx.info("last_grid")); half = nb // 2
big = np.zeros((32768 + 256 + nb * 16 + 8, 32), dtype=np.uint64)
lib.mbb_stamps(ctx.h, big.ctypes.data_as(C.c_void_p), big.shape[0])
ev = big.reshape(-1)[(1 << 20) + 8192:(1 << 20) + 8192 + nb * 64 * 8].reshape(nb, 64, 8).astype(np.int64)
niter = 2 * NS
js = np.arange(niter - 56, niter - 2)                     # half-steps with complete records
T = lambda g, j, e: ev[g, j & 63, e] * 10.0               # ns
# decision time of row (half h, pair w) at half-step j: whichever sibling published
def D(w, j):
    return max(T(2 * w, j, 6), T(2 * w + 1, j, 6))
lag1, lag2, lag3, cons, w2w, qd, fin, per = [], [], [], [], [], [], [], []
for j in js:
    for g in range(nb):
        w = g >> 1
        prow = int(ev[g, j & 63, 7]); pw = prow % half
        d3, d2, d1 = D(pw, j - 3), D(w, j - 2), D(pw, j - 1)
        lag1.append(T(g, j, 0) - d3)                      # W1 satisfied after the partner's decision of j - 3
        cons.append(T(g, j, 1) - T(g, j, 0))
        lag2.append(T(g, j, 2) - d2)                      # W2 satisfied after the walker's own decision of j - 2
        w2w.append(T(g, j, 2) - T(g, j, 1))
        qd.append(T(g, j, 4) - T(g, j, 3))                # record handed to Q -> E starts the sums
        lag3.append(T(g, j, 5) - d1)                      # E's words satisfied after the partner's decision of j - 1
        if T(g, j, 6) > 0:
            fin.append(T(g, j, 6) - max(T(g, j, 4), T(g, j, 5)))
            per.append(T(g, j, 6) - D(w, j - 2))
pc = lambda v: "median %5.0f  p10 %5.0f  p90 %5.0f ns" % (np.median(v), np.percentile(v, 10), np.percentile(v, 90))
print("W1 satisfied after the partner's decision of j - 3 :", pc(lag1))
print("constructor (W1 -> end)                            :", pc(cons))
print("W2 wait (constructor end -> satisfied)             :", pc(w2w))
print("W2 satisfied after the walker's decision of j - 2  :", pc(lag2))
print("record handed over -> E starts the sums            :", pc(qd))
print("E's words satisfied after partner's decision j - 1 :", pc(lag3))
print("later of (sums start, words) -> decision out       :", pc(fin))
print("a walker's decision after its previous one (2 half-steps):", pc(per))
