#!/usr/bin/env python3
"""Plain likelihood launch (k_lnlike SMODE 0) of n rows against the launch geometry, cfg2 bands:

    python tools/sweep_geometry.py [n ...]

us per launch by HIP events for walkers-per-workgroup x threads combinations, with and without the passband
tables staged in LDS; the host's own choice (pick_geometry) first.  For the mid-size launches of a large
single-GPU ensemble (500-4096 rows per half-step)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_likelihood, TRUTH


WPBS = (1, 2, 3, 4, 6, 8, 16, 24, 32, 48, 64)


def main():
    ns = [int(a) for a in sys.argv[1:]] or [256, 257, 288, 320, 384, 500, 1000, 2048, 4096]
    like, flux = make_likelihood(0)
    ctx = like._sync_device()
    for n in ns:
        p = np.ascontiguousarray(TRUTH * (1.0 + 0.03 * np.random.RandomState(2).normal(size=(n, 5))))
        dp = ctx.alloc(p.nbytes); dp.upload(p)
        dl, ds = ctx.alloc(n * 8), ctx.alloc(n * 4)
        best = None
        for wpb, thr, stage in [(0, 0, -1)] + [(w, t, s) for w in WPBS for t in (256, 512, 768, 1024) for s in (0, 1)]:
            ctx.set_option("walkers_per_group", wpb); ctx.set_option("block_threads", thr); ctx.set_option("stage_tables", stage)
            try:
                ctx.lnlike_repeat_device(dp, n, dl, ds, 30); ctx.sync()
            except Exception as e:
                continue
            e0, e1 = ctx.event(), ctx.event()
            ctx.record(e0); ctx.lnlike_repeat_device(dp, n, dl, ds, 200); ctx.record(e1); ctx.sync()
            us = ctx.elapsed_ms(e0, e1) * 1e3 / 200
            tag = "host's choice" if wpb == 0 else ""
            row = (us, ctx.info("last_wpb"), ctx.info("last_threads"), ctx.info("last_stage"), ctx.info("last_grid"))
            if wpb == 0:
                print("n=%5d  %-14s wpb %2d thr %4d stage %d grid %4d : %8.2f us  %.3g evals/s" % (n, tag, row[1], row[2], row[3], row[4], us, n / us * 1e6), flush=True)
            if best is None or us < best[0]:
                best = row
            if os.environ.get("SWEEP_ALL") and wpb:
                print("        wpb %2d thr %4d stage %d grid %5d : %8.2f us" % (row[1], row[2], row[3], row[4], us), flush=True)
        print("n=%5d  %-14s wpb %2d thr %4d stage %d grid %4d : %8.2f us  %.3g evals/s" % (n, "best of sweep", best[1], best[2], best[3], best[4], best[0], n / best[0] * 1e6), flush=True)
        ctx.set_option("walkers_per_group", 0); ctx.set_option("block_threads", 0); ctx.set_option("stage_tables", -1)


if __name__ == "__main__":
    main()
