#!/usr/bin/env python3
"""Diagnostic build only (-DMBB_STAMPS): where a served request's time goes inside k_serve, from stamps on the clock all CUs
share (100 MHz), medians over the workgroups of the last of many requests.   python tools/probe_serve_stamps.py [rows]"""
import os, sys, ctypes as C
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mbb_emcee_amd import _native
_native.LIB_PATH = os.environ.get("MBB_STAMPS_LIB", os.path.join(ROOT, "tools", "libmbb_hip_stamps.so")); _native.SIGNATURES["mbb_stamps"] = (C.c_int, [C.c_void_p, C.c_void_p, C.c_int])
from bench import make_likelihood, walkers
like, flux = make_likelihood(0)
ctx = like._sync_device()
lib = ctx.lib
lib.mbb_stamps(ctx.h, None, 0)
for n in [int(a) for a in sys.argv[1:]] or [125, 1]:
    for ovl in (0, 1):
        ctx.set_option("serve", 1); ctx.set_option("serve_overlap", ovl)
        p = np.ascontiguousarray(walkers(1)[:n])
        for _ in range(200):
            like(p)
        assert ctx.info("serving") == 1
        ctx.set_option("serve", 1)            # (any option: the server leaves, the stream is idle, the stamps are final)
        st = np.zeros((256, 32), dtype=np.uint64)
        lib.mbb_stamps(ctx.h, st.ctypes.data_as(C.c_void_p), 256)
        ev = st.reshape(-1)[: 256 * 16].reshape(256, 16)[:n].astype(np.float64) * 10.0          # ns
        names = ["request seen -> row in registers", "-> scalars out, quadrature may start", "-> constructor through",
                 "-> record replaced, units begin", "-> units summed", "-> result stored"]
        d = [ev[:, i + 1] - ev[:, i] for i in range(6)]
        print("rows %d, quadrature %s the constructor:" % (n, "beside" if ovl else "after"))
        for nm, x in zip(names, d):
            print("   %-44s median %6.0f ns   max %6.0f" % (nm, np.median(x), x.max()))
        print("   %-44s median %6.0f ns   max %6.0f" % ("request seen -> result stored", np.median(ev[:, 6] - ev[:, 0]), (ev[:, 6] - ev[:, 0]).max()))
        if n == 1:
            raw = st.reshape(-1).astype(np.int64)
            print("   constructor, core cycles (one row): sed_prologue %d, the rest (record, penalties) %d" % (raw[9] - raw[8], raw[10] - raw[9]))
        if ovl:
            # per wave, workgroup 0: ahead pass start / end, barrier passed, unit summed -- ns after the workgroup's 'request seen'
            wv = st.reshape(-1)[4096:4096 + 64].reshape(16, 4).astype(np.float64) * 10.0 - ev[0, 0]
            print("   per wave of workgroup 0 (ns after 'request seen'):  ahead begins / ends | behind the barrier / unit stored")
            for w_ in range(1, 16):
                print("      wave %2d  %6.0f %6.0f | %6.0f %6.0f" % (w_, wv[w_, 0], wv[w_, 1], wv[w_, 2], wv[w_, 3]))
        print("   spread of 'request seen' over the workgroups: %.0f ns; of 'result stored': %.0f ns" % (ev[:, 0].max() - ev[:, 0].min(), ev[:, 6].max() - ev[:, 6].min()))
