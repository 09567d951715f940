#!/usr/bin/env python3
"""Interleaved A/B of two builds of libmbb_hip.so in ONE process on one box:

    python tools/ab_two_libs.py <libA.so> <libB.so> [rounds]

Both libraries are loaded side by side (each context keeps the library it was made
with); each round times, for A then B: the 125-walker plain launch, the dependent
device-sampler step (250 walkers) and the 250 000-walker cfg5 launch.  Medians over
the rounds, so that clock and thermal drift hit both builds alike.
"""
import os, sys, json
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mbb_emcee_amd import _native
import mbb_emcee_amd as mbb


_ALL_SIGNATURES = dict(_native.SIGNATURES)


def build_side(libpath):
    _native._lib = None
    _native.LIB_PATH = os.path.abspath(libpath)
    # an older build may lack entry points added since: bind what it has
    import ctypes
    probe = ctypes.CDLL(_native.LIB_PATH)
    _native.SIGNATURES.clear()
    _native.SIGNATURES.update({k: v for k, v in _ALL_SIGNATURES.items() if hasattr(probe, k)})
    from bench import make_likelihood, walkers, NW_PER_GPU
    from tools.bench_cfg5 import setup
    like, flux = make_likelihood(0)
    ctx = like._sync_device()
    p = walkers(1)[:125]
    dp = ctx.alloc(p.nbytes); dp.upload(p)
    dl, ds = ctx.alloc(125 * 8), ctx.alloc(125 * 4)
    smp = mbb.DeviceEnsembleSampler(NW_PER_GPU, 5, like, seed=7)
    smp.run_mcmc(walkers(1)[:NW_PER_GPU], 20, storechain=False)
    like5, _, p5 = setup(1000, 250)
    c5 = like5._sync_device()
    flat = np.ascontiguousarray(p5.reshape(-1, 5))
    d5 = c5.alloc(flat.nbytes); d5.upload(flat)
    l5, s5 = c5.alloc(flat.shape[0] * 8), c5.alloc(flat.shape[0] * 4)
    return dict(like=like, ctx=ctx, dp=dp, dl=dl, ds=ds, smp=smp, like5=like5, c5=c5, d5=d5, l5=l5, s5=s5,
                n5=flat.shape[0])


def ev_time(ctx, fn):
    e0, e1 = ctx.event(), ctx.event()
    ctx.record(e0); fn(); ctx.record(e1); ctx.sync()
    return ctx.elapsed_ms(e0, e1)


def measure(s):
    ctx, c5 = s["ctx"], s["c5"]
    ctx.lnlike_repeat_device(s["dp"], 125, s["dl"], s["ds"], 50); ctx.sync()
    k125 = ev_time(ctx, lambda: ctx.lnlike_repeat_device(s["dp"], 125, s["dl"], s["ds"], 500)) * 1e3 / 500
    s["smp"].advance_async(50); ctx.sync()
    step = ev_time(ctx, lambda: s["smp"].advance_async(500)) * 1e3 / 500
    c5.lnlike_repeat_device(s["d5"], s["n5"], s["l5"], s["s5"], 6); c5.sync()
    k5 = ev_time(c5, lambda: c5.lnlike_repeat_device(s["d5"], s["n5"], s["l5"], s["s5"], 6)) * 1e3 / 6
    # the same chain in the resident form with the constructor ahead (form 9: what ensembles beyond 256 walkers take)
    ctx.set_option("merged_flow_sampler", 0)
    s["smp"].advance_async(50); ctx.sync()
    step5 = ev_time(ctx, lambda: s["smp"].advance_async(500)) * 1e3 / 500
    ctx.set_option("merged_flow_sampler", 1)
    return k125, step, k5, step5


def main():
    a, b = sys.argv[1], sys.argv[2]
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 7
    A, B = build_side(a), build_side(b)
    res = {"A": [], "B": []}
    for r in range(rounds):
        res["A"].append(measure(A))
        res["B"].append(measure(B))
    out = {}
    for k, lib in (("A", a), ("B", b)):
        v = np.array(res[k])
        out[k] = {"lib": lib, "kernel_125_us": float(np.median(v[:, 0])), "sampler_step_us": float(np.median(v[:, 1])),
                  "cfg5_launch_us": float(np.median(v[:, 2])), "sampler_step_form9_us": float(np.median(v[:, 3])),
                  "min": [float(x) for x in v.min(axis=0)]}
        print("%s %-40s 125-walker launch %.3f us   sampler step %.3f us (form 9: %.3f)   cfg5 launch %.1f us" %
              (k, os.path.basename(lib), out[k]["kernel_125_us"], out[k]["sampler_step_us"], out[k]["sampler_step_form9_us"],
               out[k]["cfg5_launch_us"]))
    print("B/A: %.4f  %.4f (form 9: %.4f)  %.4f" % tuple(out["B"][x] / out["A"][x] for x in
                                                      ("kernel_125_us", "sampler_step_us", "sampler_step_form9_us", "cfg5_launch_us")))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
