#!/usr/bin/env python3
"""Where a served boundary call's time goes on the HOST (mbb_lnlike_call's own clock: sentinels, request out, waiting for the
records) next to the whole call by Python's clock.   python tools/probe_host_phases.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_likelihood, walkers
like, flux = make_likelihood(0)
ctx = like._sync_device()
for serve, n in [(1, 125), (1, 64), (1, 16), (1, 1), (0, 125), (0, 1)]:
    p = np.ascontiguousarray(walkers(1)[:n])
    ctx.set_option("serve", serve)
    for _ in range(100):
        like(p)
    rows = []
    for i in range(2000):
        t0 = time.perf_counter(); like(p); t = time.perf_counter() - t0
        rows.append((t * 1e9, ctx.info("last_prep_ns"), ctx.info("last_launch_ns"), ctx.info("last_wait_ns")))
    r = np.median(np.array(rows, dtype=np.float64), axis=0)
    print(("served  " if serve else "one-shot") + " rows %3d: whole call %6.0f ns | in mbb_lnlike_call: sentinels + fence %5.0f, doorbell %5.0f, waiting for the records %6.0f | the rest (Python, rows through the BAR, result copy) %5.0f"
          % (n, r[0], r[1], r[2], r[3], r[0] - r[1] - r[2] - r[3]))
