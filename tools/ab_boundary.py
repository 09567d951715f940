#!/usr/bin/env python3
"""M1 by the host's clock, served (option serve_overlap 0 and 1) -- run through tools/ab_lib.py to compare two builds."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_likelihood, walkers
like, flux = make_likelihood(0)
ctx = like._sync_device()
out = []
for n in (125, 1):
    p = np.ascontiguousarray(walkers(1)[:n]); arg = p if n > 1 else p[0].copy()
    for ovl in (0, 1, 0, 1):
        ctx.set_option("serve_overlap", ovl)
        for _ in range(50): like(arg)
        ts = np.empty(2000)
        for i in range(2000):
            t0 = time.perf_counter(); like(arg); ts[i] = time.perf_counter() - t0
        out.append("rows %d ovl %d: %.2f" % (n, ovl, np.median(ts) * 1e6))
print("   ".join(out), flush=True)
