"""Diagnostic build only (-DMBB_STAMPS): where the waves of sampler form 7 (k_flowm) spend their cycles, summed
over a launch: per role (Q quadrature waves, E, the two C waves) the average cycles per half-step in each part
of the loop, median over workgroups."""
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
NS = 1000
e0, e1 = ctx.event(), ctx.event()
ctx.record(e0); smp.advance_async(NS); ctx.record(e1); ctx.sync()
us = ctx.elapsed_ms(e0, e1) * 1e3 / NS
nb = int(ctx.info("last_grid")); nwave = int(ctx.info("last_threads")) // 64
print("form", ctx.info("last_kernel_form"), "grid", nb, "waves", nwave, "%.3f us per step" % us)
st = np.zeros((nb * 4, 32), dtype=np.uint64)
lib.mbb_stamps(ctx.h, st.ctypes.data_as(C.c_void_p), nb * 4)
t = st.reshape(-1)[: nb * 16 * 8].reshape(nb, 16, 8).astype(np.float64) / (2 * NS)    # cycles per half-step
nq = nwave - 5
spread = nwave == 16
QW = [wv for wv in range(nwave) if (wv & 3) != 3 and wv != 14] if spread else list(range(nq))
CW = (3, 7, 11) if spread else (nq, nq + 1, nq + 2)
EWS = (15, 14) if spread else (nq + 3, nq + 4)
EW = EWS[0]
raw = st.reshape(-1)[: nb * 16 * 8].reshape(nb, 16, 8).astype(np.float64)
med = lambda v: [int(x) for x in np.median(v, axis=0)]
tot = t.sum(axis=2)
print("cycles per half-step of the run (sum of a wave's parts, median over workgroups and waves): %d" % np.median(tot[:, :nwave]))
print("Q waves 0..%d  [wait for record, units + reduce]:" % (nq - 1))
for wv in QW:
    print("   wave %2d" % wv, med(t[:, wv, :2]))
for e in EWS:
    print("E wave %d (per half-step; it works every other one) [words (+ the sums, when Q is through first), wait for Q, band sums after the words, decide + publish]:" % e, med(t[:, e, :4]))
for cb in range(3):
    print("C wave %d (per half-step; it works every third one) [draws + wait 1, staging, constructor, wait 2, wait E, hand-over]:" % cb,
          med(t[:, CW[cb], :6]), " constructor of one proposal: min %d max %d cycles (medians over workgroups)"
          % (np.median(raw[:, CW[cb], 7]), np.median(raw[:, CW[cb], 6])))
for c in range(2):
    sel = t[c::2]
    print("candidate %d workgroups: E" % c, med(sel[:, EW, :4]), " C0", med(sel[:, CW[0], :6]))

big = np.zeros((32768 + nb, 32), dtype=np.uint64)
lib.mbb_stamps(ctx.h, big.ctypes.data_as(C.c_void_p), 32768 + nb)
hw = big.reshape(-1)[(1 << 20):(1 << 20) + nb * 16].reshape(nb, 16)
simd = ((hw >> np.uint64(4)) & np.uint64(3)).astype(int)
cu = ((hw >> np.uint64(8)) & np.uint64(15)).astype(int) + 16 * ((hw >> np.uint64(13)) & np.uint64(7)).astype(int) + 128 * ((hw >> np.uint64(32)) & np.uint64(7)).astype(int)
print("SIMD of waves 0..15, first workgroups:")
for g in range(4):
    print("  ", simd[g].tolist())
print("workgroups per (XCC, SE, CU): max %d; distinct CUs %d" % (np.bincount(cu[:, 0]).max(), len(set(cu[:, 0].tolist()))))
same = sum(1 for g in range(nb) if len(set(simd[g][[3, 7, 11]].tolist())) == 1)
print("workgroups whose waves 3, 7, 11 share one SIMD: %d of %d" % (same, nb))
