#!/usr/bin/env python3
"""Interleaved sweep of ONE context option's values on the served boundary (M1 by the host's clock, bench.py's workload):
    python tools/ab_option.py <option> <v0> <v1> ... [--rows 125,250,1] [--rounds 6]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_likelihood, walkers


def main():
    argv = sys.argv[1:]
    args = [a for i, a in enumerate(argv) if not a.startswith("--") and not (i and argv[i - 1] in ("--rows", "--rounds"))]
    opt, vals = args[0], [int(v) for v in args[1:]]
    rows = [125, 250, 1]
    rounds = 6
    for i, a in enumerate(sys.argv):
        if a == "--rows": rows = [int(x) for x in sys.argv[i + 1].split(",")]
        if a == "--rounds": rounds = int(sys.argv[i + 1])
    like, _ = make_likelihood(0)
    ctx = like._sync_device()
    res = {}
    for rnd in range(rounds):
        for v in vals:
            ctx.set_option(opt, v)
            if opt in ("seg_chunks", "pack_tails"):       # (options of the band layout: the tables are made anew)
                like._dirty = True
                like._sync_device()
            for n in rows:
                p = np.ascontiguousarray(walkers(1)[:n])
                arg = p if n > 1 else p[0].copy()
                for _ in range(30):
                    like(arg)
                ts = np.empty(1000)
                for i in range(1000):
                    t0 = time.perf_counter(); like(arg); ts[i] = time.perf_counter() - t0
                res.setdefault((v, n), []).append(np.median(ts) * 1e6)
    print("served boundary by %s (medians of %d interleaved rounds of 1000 calls, us; serve_requests %d, fall-backs %d)" % (
        opt, rounds, ctx.info("serve_requests"), ctx.info("serve_fallbacks")))
    for n in rows:
        print("  rows %3d   " % n + "   ".join("%s=%d: %.2f" % (opt, v, np.median(res[(v, n)])) for v in vals))


if __name__ == "__main__":
    main()
