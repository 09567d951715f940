#!/usr/bin/env python3
"""M1 served with a row's quadrature beside (serve_overlap 1) or after (0) its constructor, for cfg1 and cfg4 (half an ensemble
per call), by the host's clock:   python tools/ab_boundary_configs.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_configs as bc
for cfg in ("cfg1", "cfg4"):
    like, p, desc, okw = bc.setup(cfg)
    ctx = like._sync_device()
    n = len(p) // 2
    arg = np.ascontiguousarray(p[:n])
    out = []
    for ovl in (0, 1, 0, 1):
        ctx.set_option("serve", 2); ctx.set_option("serve_overlap", ovl)
        for _ in range(50): like(arg)
        ts = np.empty(2000)
        for i in range(2000):
            t0 = time.perf_counter(); like(arg); ts[i] = time.perf_counter() - t0
        out.append("overlap %d: %.2f us" % (ovl, np.median(ts) * 1e6))
    print("%s, %d rows, %d chunks of samples:  %s" % (cfg, n, ctx.info("nchunk"), "   ".join(out)), flush=True)
