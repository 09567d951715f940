#!/usr/bin/env python3
"""What a resident server (k_serve) means for OTHER streams of its process: the time a small torch kernel takes from its launch to
its completion (torch's own streams) while a sampler-like loop of boundary calls is being served, against the same with
nothing resident.  (Measured once with the library's stream made at high and at low priority as well: the same, and M1 the
same: profiles/r05/served_boundary.txt 16.)      python tools/probe_other_streams.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from bench import make_likelihood, walkers

p = np.ascontiguousarray(walkers(1)[:125])
x = torch.zeros(1024, device="cuda")
streams = [torch.cuda.Stream() for _ in range(6)]


def torch_latency(like, nrep=300):
    """(median, p99, max) us of x.add_(1) + synchronize on each of torch's streams, a boundary call before each"""
    ts = []
    for i in range(nrep):
        if like is not None:
            like(p)
        s = streams[i % len(streams)]
        t0 = time.perf_counter()
        with torch.cuda.stream(s):
            x.add_(1.0)
        s.synchronize()
        ts.append(time.perf_counter() - t0)
    ts = np.array(ts) * 1e6
    return np.median(ts), np.percentile(ts, 99), ts.max()


print("a torch kernel on one of six torch streams, launch -> done (us: median, p99, max)")
print("  nothing of the library resident:          %.1f  %.1f  %.1f" % torch_latency(None))
for name in ("default",):
    like = make_likelihood(0)[0]
    ctx = like._sync_device()
    for serve in (0, 1):
        ctx.set_option("serve", serve)
        for _ in range(10):
            like(p)
        r = torch_latency(like)
        print("  library stream %-7s serve %d (%s): %.1f  %.1f  %.1f   (requests served %d, fell back %d)" % (
            name, serve, "resident" if ctx.info("serving") else "launches", r[0], r[1], r[2], ctx.info("serve_requests"), ctx.info("serve_fallbacks")), flush=True)
    ctx.set_option("serve", 0)
    del like
