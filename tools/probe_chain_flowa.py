#!/usr/bin/env python3
"""Diagnostic build only (-DMBB_STAMPS): where a half-step of sampler form 9 (k_flowa) goes, from stamps on the clock all CUs
share (100 MHz) over the last 64 half-steps of a launch.   python tools/probe_chain_flowa.py [nwalkers ...]"""
import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mbb_emcee_amd import _native
_native.LIB_PATH = os.environ.get("MBB_STAMPS_LIB", os.path.join(ROOT, "tools", "libmbb_hip_stamps.so")); _native.SIGNATURES["mbb_stamps"] = (C.c_int, [C.c_void_p, C.c_void_p, C.c_int])
import mbb_emcee_amd as mbb
from bench import make_likelihood, walkers
like, flux = make_likelihood(0)
ctx = like._sync_device()
lib = ctx.lib
lib.mbb_stamps(ctx.h, None, 0)
NS = 200
for nw in [int(a) for a in sys.argv[1:]] or [2000, 1000, 512]:
    smp = mbb.DeviceEnsembleSampler(nw, 5, like, seed=11)
    smp.run_mcmc(walkers(8)[:nw], 60, storechain=False)
    wall, ms = smp.advance_timed(NS)
    nb = int(ctx.info("last_grid"))
    st = np.zeros((nb * 32, 32), dtype=np.uint64)
    lib.mbb_stamps(ctx.h, st.ctypes.data_as(C.c_void_p), nb * 32)
    ev = st.reshape(-1)[: nb * 64 * 16].reshape(nb, 64, 16).astype(np.float64) * 10.0     # ns
    niter = 2 * NS
    # half-step numbers of the slots: the last 64
    js = np.arange(niter - 64, niter)
    slot = js & 63
    E = ev[:, slot, :]                                  # [wg, k, event] in half-step order
    t0 = E[:, :, 1].min()                               # first record ready
    print("nw %d: form %d, %d workgroups x %d walkers, %.2f us per step on the stream" % (nw, ctx.info("last_kernel_form"), nb, ctx.info("last_wpb"), ms * 1e3 / NS))
    def D(a, b, what):
        x = E[:, 4:60, a] - E[:, 4:60, b]
        print("   %-62s median %6.0f ns   p90 %6.0f   max %6.0f" % (what, np.median(x), np.percentile(x, 90), x.max()))
    per = np.median(np.diff(E[:, 4:60, 1], axis=1))
    print("   record ready -> record ready of the next half-step: median %.0f ns" % per)
    D(1, 0, "Q wave 0: waits for the first record (loop top -> a walker ready)")
    D(2, 1, "Q wave 0: first record -> through with its units of every walker")
    D(3, 1, "first record -> every Q wave through (wave 0 sees it)")
    D(4, 3, "wave 0: band sums, accept test, decision + row published")
    D(9, 8, "C wave 0: gathers the rows / proposals it needs (incl. lag guard)")
    D(10, 9, "C wave 0: constructor")
    D(11, 10, "C wave 0: waits for the partner's decision of j - 1")
    D(12, 11, "C wave 0: select + hand-over")
    # decision published (event 4 of half-step j, any workgroup) -> C select done (event 11 of j+1)
    x = E[:, 5:60, 11] - np.max(E[:, 4:59, 4], axis=0)[None, :]
    print("   last decision of half-step j published anywhere -> C wave 0 has its partner's decision (j+1): median %.0f ns" % np.median(x))
    x = E[:, 5:60, 1] - np.max(E[:, 4:59, 4], axis=0)[None, :]
    print("   last decision of half-step j published anywhere -> record of j+1 ready: median %.0f ns" % np.median(x))
    x = np.max(E[:, 4:60, 4], axis=0) - np.min(E[:, 4:60, 4], axis=0)
    print("   spread of the workgroups' decision times within a half-step: median %.0f ns" % np.median(x))
    del smp
