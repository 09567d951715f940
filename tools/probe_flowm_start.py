#!/usr/bin/env python3
"""Diagnostic build only (-DMBB_STAMPS): the first half-steps of a short form-7 launch on the clock all CUs share
(s_memrealtime, 10 ns ticks) -- where the ~14 us a launch costs beside its steps go.

    MBB_STAMPS_LIB=tools/libmbb_hip_stamps.so python tools/probe_flowm_start.py [steps]

Times are relative to the first workgroup getting past its set-up; medians (and the slowest) over workgroups."""
import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mbb_emcee_amd import _native
_native.LIB_PATH = os.environ.get("MBB_STAMPS_LIB", os.path.join(ROOT, "tools", "libmbb_hip_stamps.so")); _native.SIGNATURES["mbb_stamps"] = (C.c_int, [C.c_void_p, C.c_void_p, C.c_int])
import mbb_emcee_amd as mbb
from bench import make_likelihood, walkers, NW_PER_GPU
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 20
like, flux = make_likelihood(0)
ctx = like._sync_device()
lib = ctx.lib
lib.mbb_stamps(ctx.h, None, 0)
smp = mbb.DeviceEnsembleSampler(NW_PER_GPU, 5, like, seed=11)
smp.run_mcmc(walkers(1)[:NW_PER_GPU], 100, storechain=False)
smp.advance_async(5); ctx.sync()
e0, e1 = ctx.event(), ctx.event()
ctx.record(e0); smp.advance_async(NS); ctx.record(e1); ctx.sync()
print("%d steps: %.2f us on the stream, form %d" % (NS, ctx.elapsed_ms(e0, e1) * 1e3, ctx.info("last_kernel_form")))
nb = int(ctx.info("last_grid"))
big = np.zeros((32768 + 256 + nb * 32 + 8, 32), dtype=np.uint64)
lib.mbb_stamps(ctx.h, big.ctypes.data_as(C.c_void_p), big.shape[0])
flat = big.reshape(-1)
start = flat[(1 << 20) + 4096:(1 << 20) + 4096 + nb].astype(np.int64) * 10.0
ev = flat[(1 << 20) + 8192:(1 << 20) + 8192 + nb * 64 * 16].reshape(nb, 64, 16).astype(np.int64) * 10.0
t0 = start.min()
print("workgroups past their set-up: first 0, median %.0f ns, last %.0f ns" % (np.median(start) - t0, start.max() - t0))
names = {13: "C: draws made, asks for rows / proposals / decisions", 0: "C: everything gathered", 1: "C: constructor through",
         2: "C: decisions of j-2 known", 3: "C: record handed to Q", 14: "Q (wave 1): record seen", 15: "Q: last wave through",
         9: "E: turns to this half-step", 4: "E: sums begin", 5: "E: words there", 6: "E: decision published"}
niter = 2 * NS
for j in range(min(6, niter)):
    print("half-step %d" % j)
    for e in (13, 0, 1, 2, 3, 14, 15, 9, 4, 5, 6):
        t = ev[:, j & 63, e]
        ok = t > 0
        if e == 6:                     # published by the sibling whose candidate it was
            pair = np.maximum(ev[0::2, j & 63, 6], ev[1::2, j & 63, 6])
            t, ok = pair, pair > 0
        if ok.any():
            print("   %-52s median %6.0f ns   last %6.0f ns" % (names[e], np.median(t[ok]) - t0, t[ok].max() - t0))
last = np.maximum(ev[0::2, (niter - 1) & 63, 6], ev[1::2, (niter - 1) & 63, 6])
print("last decision of the launch (half-step %d): median %.0f ns, last %.0f ns after the first workgroup's set-up" % (niter - 1, np.median(last) - t0, last.max() - t0))
