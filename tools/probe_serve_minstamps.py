#!/usr/bin/env python3
"""Diagnostic build with two stamps only (-DMBB_STAMPS -DMBB_STAMPS_MIN): 'request seen' -> 'result stored' inside k_serve,
next to the host's clock around the same calls.   MBB_STAMPS_LIB=tools/libmbb_hip_stamps_min.so python tools/probe_serve_minstamps.py"""
import os, sys, time, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mbb_emcee_amd import _native
_native.LIB_PATH = os.environ.get("MBB_STAMPS_LIB", os.path.join(ROOT, "tools", "libmbb_hip_stamps_min.so")); _native.SIGNATURES["mbb_stamps"] = (C.c_int, [C.c_void_p, C.c_void_p, C.c_int])
from bench import make_likelihood, walkers
like, flux = make_likelihood(0)
ctx = like._sync_device()
lib = ctx.lib
lib.mbb_stamps(ctx.h, None, 0)
for n in (125, 1):
    for ovl in (0, 1):
        ctx.set_option("serve", 1); ctx.set_option("serve_overlap", ovl)
        p = np.ascontiguousarray(walkers(1)[:n])
        for _ in range(200):
            like(p)
        ts = np.empty(1000)
        for i in range(1000):
            t0 = time.perf_counter(); like(p); ts[i] = time.perf_counter() - t0
        assert ctx.info("serving") == 1
        ctx.set_option("serve", 1)
        st = np.zeros((256, 32), dtype=np.uint64)
        lib.mbb_stamps(ctx.h, st.ctypes.data_as(C.c_void_p), 256)
        ev = st.reshape(-1)[: 256 * 16].reshape(256, 16)[:n].astype(np.float64) * 10.0
        d = ev[:, 6] - ev[:, 0]
        print("rows %3d  overlap %d: host p50 %.2f us | in the kernel, request seen -> result stored: median %.0f ns, max %.0f; first seen -> last stored %.0f ns"
              % (n, ovl, np.median(ts) * 1e6, np.median(d), d.max(), ev[:, 6].max() - ev[:, 0].min()))
