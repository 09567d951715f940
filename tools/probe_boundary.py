#!/usr/bin/env python3
"""Boundary rate: host float64[n,5] in -> host float64[n] out, one call per
emcee half-step (pinned copy + launch + sync).  PCIe-inclusive, never `value`."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_likelihood, walkers

def main():
    like, flux = make_likelihood(0)
    ctx = like._sync_device()
    allw = np.tile(walkers(1), (16, 1))
    for zc, spin, bar in ((0, 0, 0), (1, 0, 0), (1, 2, 0), (1, 2, 1), (1, 2, 0), (1, 2, 1)):
        ctx.set_option("zero_copy", zc); ctx.set_option("spin_wait", spin); ctx.set_option("bar_params", bar)
        for n in (125, 250, 2000):
            p = allw[:n]
            for _ in range(50): like(p)
            t0 = time.perf_counter(); reps = 500 if n <= 2000 else 50
            for _ in range(reps): like(p)
            dt = (time.perf_counter() - t0) / reps
            t0 = time.perf_counter()
            for _ in range(reps): ctx.lnlike_batch(p)
            dt2 = (time.perf_counter() - t0) / reps
            ph = np.array([[ctx.info("last_prep_ns"), ctx.info("last_launch_ns"), ctx.info("last_wait_ns")]
                           for _ in range(200) if ctx.lnlike_batch(p) is not None])
            print("   C-ABI phases (median ns): copy-in+sentinels %d, hipLaunchKernel %d, wait %d" % tuple(np.median(ph, axis=0)))
            print("spin=%d bar=%d " % (spin, bar), end=""); print("zero_copy=%d n=%6d  likelihood.__call__ %.1f us/call %.3g evals/s | C-ABI only %.1f us/call %.3g evals/s"
                  % (zc, n, dt * 1e6, n / dt, dt2 * 1e6, n / dt2), flush=True)

if __name__ == "__main__":
    main()
