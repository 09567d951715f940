"""Interleaved A/B of the host boundary with the parameter rows pushed through the BAR
into device memory (bar_params = 1) or left in pinned host memory (0)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_likelihood, walkers
like, flux = make_likelihood(0)
ctx = like._sync_device()
allw = np.tile(walkers(1), (16, 1))
for n in (125, 250, 1000):
    p = np.ascontiguousarray(allw[:n])
    res = {0: [], 1: []}
    for rep in range(6):
        for bar in (0, 1):
            ctx.set_option("bar_params", bar)
            for _ in range(300): like(p)
            t0 = time.perf_counter()
            for _ in range(3000): like(p)
            res[bar].append((time.perf_counter() - t0) / 3000 * 1e6)
    print("n=%5d  pinned host rows: %s | rows pushed through the BAR: %s  (us per likelihood.__call__)" %
          (n, " ".join("%.2f" % v for v in res[0]), " ".join("%.2f" % v for v in res[1])), flush=True)
