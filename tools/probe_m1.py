#!/usr/bin/env python3
"""M1 by the host's clock: likelihood.__call__ in a loop, served and with a launch per call, 125 / 250 / 1 rows; medians of
3 rounds of 1500 calls.  With an alternative library: python tools/ab_lib.py <lib.so> tools/probe_m1.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_likelihood, walkers

like, flux = make_likelihood(0)
ctx = like._sync_device()
for n in (125, 250, 1):
    p = np.ascontiguousarray(walkers(1)[:n])
    arg = p if n > 1 else p[0].copy()
    ctx.set_option("serve", 0)
    want = like(arg)
    res = {}
    for rnd in range(3):
        for mode in ("launch", "served"):
            ctx.set_option("serve", 0 if mode == "launch" else 1)
            for _ in range(30):
                got = like(arg)
            assert np.array_equal(got, want), (n, mode)
            ts = np.empty(1500)
            for i in range(1500):
                t0 = time.perf_counter(); like(arg); ts[i] = time.perf_counter() - t0
            res.setdefault(mode, []).append((np.median(ts) * 1e6, np.percentile(ts, 90) * 1e6, ctx.info("serve_fallbacks")))
    for mode, v in res.items():
        m = np.median(np.array(v), axis=0)
        print("rows %3d  %-8s p50 %6.2f us  p90 %6.2f   fallbacks %d   rounds %s" % (n, mode, m[0], m[1], m[2], " ".join("%.2f" % r[0] for r in v)), flush=True)
