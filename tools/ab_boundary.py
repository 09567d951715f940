import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
from bench import make_likelihood, walkers
like, flux = make_likelihood(0)
ctx = like._sync_device()
out = []
for n in (125, 1):
    p = np.ascontiguousarray(walkers(1)[:n]); arg = p if n > 1 else p[0].copy()
    for _ in range(50): like(arg)
    ts = np.empty(3000)
    for i in range(3000):
        t0 = time.perf_counter(); like(arg); ts[i] = time.perf_counter() - t0
    out.append("rows %d: p50 %.2f p90 %.2f (serving %d)" % (n, np.median(ts) * 1e6, np.percentile(ts, 90) * 1e6, ctx.info("serving")))
print("   ".join(out), flush=True)
