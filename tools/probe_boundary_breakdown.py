#!/usr/bin/env python3
"""Where the time of one boundary call goes (SURVEY.md 8d metric M1: host float64[n,5] in -> host
float64[n] out through likelihood.__call__, what an external sampler such as emcee pays per half-step).

    python tools/probe_boundary_breakdown.py

Layers, each timed over many calls (medians, us): likelihood.__call__ (Python: asarray, shape checks,
_sync_device, raise_for_status) > Context.lnlike_batch (ctypes marshalling, numpy allocations) >
mbb_lnlike_batch (C: parameter rows into device memory through the BAR + sentinels | hipLaunchKernel |
watching the result slots in pinned memory) > the kernel itself (HIP events around back-to-back launches on
device-resident rows)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_likelihood, walkers


def med_us(fn, reps=2000):
    for _ in range(100):
        fn()
    ts = np.empty(reps)
    for i in range(reps):
        t0 = time.perf_counter(); fn(); ts[i] = time.perf_counter() - t0
    return float(np.median(ts) * 1e6), float(np.percentile(ts, 90) * 1e6)


def main():
    like, flux = make_likelihood(0)
    ctx = like._sync_device()
    for n in (125, 250):
        p = np.ascontiguousarray(walkers(1)[:n])
        call, call90 = med_us(lambda: like(p))
        ctypes_, _ = med_us(lambda: ctx.lnlike_batch(p))
        ph = []
        for _ in range(2000):
            ctx.lnlike_batch(p)
            ph.append((ctx.info("last_prep_ns"), ctx.info("last_launch_ns"), ctx.info("last_wait_ns")))
        prep, launch, wait = (np.median(np.array(ph), axis=0) / 1e3)
        dp = ctx.alloc(p.nbytes); dp.upload(p)
        dl, ds = ctx.alloc(n * 8), ctx.alloc(n * 4)
        ctx.lnlike_repeat_device(dp, n, dl, ds, 200); ctx.sync()
        e0, e1 = ctx.event(), ctx.event()
        ctx.record(e0); ctx.lnlike_repeat_device(dp, n, dl, ds, 1000); ctx.record(e1); ctx.sync()
        kern = ctx.elapsed_ms(e0, e1)
        c_total = prep + launch + wait
        print("rows %d" % n)
        print("  likelihood.__call__            %6.2f us  (p90 %.2f)" % (call, call90))
        print("    Python above the C-ABI call  %6.2f us  (likelihood.__call__ - Context.lnlike_batch)" % (call - ctypes_))
        print("    ctypes + numpy in the binding %5.2f us  (Context.lnlike_batch - time inside mbb_lnlike_batch)" % (ctypes_ - c_total))
        print("    inside mbb_lnlike_batch      %6.2f us" % c_total)
        print("      rows -> device (BAR) + sentinels   %5.2f us" % prep)
        print("      hipLaunchKernel                    %5.2f us" % launch)
        print("      launch latency + kernel + poll     %5.2f us   of which the kernel alone %.2f us (events, back to back)" % (wait, kern))
        print("  => %.3g evals/s at the boundary; the kernel is %.0f %% of a call" % (n / (call * 1e-6), 100.0 * kern / call), flush=True)


if __name__ == "__main__":
    main()
