#!/usr/bin/env python3
"""M1 with a launch per call against the served boundary (k_serve: a kernel that stays resident between the calls and is
rung through the BAR), interleaved:   python tools/probe_served_boundary.py"""
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
    res = {0: [], 1: []}
    for rnd in range(3):
        for serve in (0, 1):
            ctx.set_option("serve", serve)
            for _ in range(30):
                got = like(arg)
            assert np.array_equal(got, want), (n, serve)
            ts = np.empty(1500)
            for i in range(1500):
                t0 = time.perf_counter(); like(arg); ts[i] = time.perf_counter() - t0
            res[serve].append((np.median(ts) * 1e6, np.percentile(ts, 90) * 1e6, ctx.info("serving"), ctx.info("serve_fallbacks"),
                               ctx.info("last_prep_ns") / 1e3, ctx.info("last_launch_ns") / 1e3, ctx.info("last_wait_ns") / 1e3))
    for serve in (0, 1):
        v = np.median(np.array(res[serve]), axis=0)
        print("rows %3d  %-22s p50 %6.2f us  p90 %6.2f   (serving %d, fallbacks %d; in C: prep %.2f, ring/launch %.2f, wait %.2f)  rounds %s"
              % (n, "served" if serve else "a launch per call", v[0], v[1], v[2], v[3], v[4], v[5], v[6],
                 " ".join("%.2f" % r[0] for r in res[serve])), flush=True)
ctx.set_option("serve", 1)
