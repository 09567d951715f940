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
    for n in (125, 250, 1):
        p = np.ascontiguousarray(walkers(1)[:n])
        row = p[0].copy()
        res = {}
        # interleaved A/B of how the launch is handed to the runtime (option launch_api), three rounds each
        for rnd in range(3):
            for api in (0, 1, 2):                     # 2: the served boundary (no launch per call)
                ctx.set_option("launch_api", 1 if api == 2 else api)
                ctx.set_option("serve", 1 if api == 2 else 0)
                call, call90 = med_us((lambda: like(p)) if n > 1 else (lambda: like(row)), reps=1500)
                ph = []
                for _ in range(1500):
                    like(p) if n > 1 else like(row)
                    ph.append((ctx.info("last_prep_ns"), ctx.info("last_launch_ns"), ctx.info("last_wait_ns")))
                prep, launch, wait = (np.median(np.array(ph), axis=0) / 1e3)
                res.setdefault(api, []).append((call, call90, prep, launch, wait))
        ctx.set_option("launch_api", 1); ctx.set_option("serve", 0)
        old, _ = med_us(lambda: ctx.lnlike_batch(p))            # round 3's binding: two allocations, three addresses, memcpy in and out
        dp = ctx.alloc(p.nbytes); dp.upload(p)
        dl, ds = ctx.alloc(n * 8), ctx.alloc(n * 4)
        ctx.lnlike_repeat_device(dp, n, dl, ds, 200); ctx.sync()
        e0, e1 = ctx.event(), ctx.event()
        ctx.record(e0); ctx.lnlike_repeat_device(dp, n, dl, ds, 1000); ctx.record(e1); ctx.sync()
        kern = ctx.elapsed_ms(e0, e1)
        print("rows %d%s" % (n, "  (one row: what emcee calls per walker without vectorize)" if n == 1 else ""))
        for api in (0, 1, 2):
            call, call90, prep, launch, wait = np.median(np.array(res[api]), axis=0)
            c_total = prep + launch + wait
            print("  %s" % ("launch per call, hipLaunchKernel" if api == 0 else "launch per call, hipModuleLaunchKernel with a packed argument buffer"
                            if api == 1 else "served: a kernel resident between the calls, rung through the BAR (the default in a loop of calls)"))
            print("    likelihood.__call__            %6.2f us  (p90 %.2f)   rounds: %s" % (call, call90, " ".join("%.2f" % r[0] for r in res[api])))
            print("      Python + ctypes around the C call %5.2f us  (checks, rows written into the BAR block, the call, the copy out)" % (call - c_total))
            print("      inside mbb_lnlike_call       %6.2f us" % c_total)
            print("        fence + sentinels                  %5.2f us" % prep)
            print("        %s %5.2f us" % ("the launch call                   " if api < 2 else "the request word through the BAR   ", launch))
            print("        %s %5.2f us   (the launched kernel alone: %.2f us, events, back to back)"
                  % ("launch latency + kernel + poll    " if api < 2 else "doorbell + evaluation + poll       ", wait, kern))
        print("  round 3's binding (Context.lnlike_batch: copies in and out, status array) %6.2f us" % old)
        best = min(np.median(np.array(res[a]), axis=0)[0] for a in (0, 1, 2))
        print("  => %.3g evals/s at the boundary" % (n / (best * 1e-6)), flush=True)
    ctx.set_option("serve", 1)


if __name__ == "__main__":
    main()
