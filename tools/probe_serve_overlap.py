#!/usr/bin/env python3
"""The served boundary with a row's quadrature started beside its constructor (option serve_overlap 1, the default) against
the three phases one after the other (0) and against a launch per call, interleaved; results bit for bit the launch's.
    python tools/probe_serve_overlap.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_likelihood, walkers

like, flux = make_likelihood(0)
ctx = like._sync_device()
for n in (125, 250, 1, 8):
    p = np.ascontiguousarray(walkers(1)[:n])
    arg = p if n > 1 else p[0].copy()
    ctx.set_option("serve", 0)
    want = like(arg)
    res = {}
    for rnd in range(3):
        for mode in ("launch", "served, phases in turn", "served, quadrature beside the constructor"):
            ctx.set_option("serve", 0 if mode == "launch" else 1)
            ctx.set_option("serve_overlap", 1 if "beside" in mode else 0)
            for _ in range(30):
                got = like(arg)
            assert np.array_equal(got, want), (n, mode)
            ts = np.empty(1500)
            for i in range(1500):
                t0 = time.perf_counter(); like(arg); ts[i] = time.perf_counter() - t0
            res.setdefault(mode, []).append((np.median(ts) * 1e6, np.percentile(ts, 90) * 1e6, ctx.info("serve_fallbacks")))
    for mode, v in res.items():
        m = np.median(np.array(v), axis=0)
        print("rows %3d  %-44s p50 %6.2f us  p90 %6.2f   fallbacks %d   rounds %s" % (n, mode, m[0], m[1], m[2], " ".join("%.2f" % r[0] for r in v)), flush=True)
ctx.set_option("serve", 1); ctx.set_option("serve_overlap", 1)
