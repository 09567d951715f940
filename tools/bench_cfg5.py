#!/usr/bin/env python3
"""cfg5 (BASELINE.json configs[4]): 1000 independent synthetic SEDs x 250 walkers,
8 passbands (NQ=2209), thick+alpha, fp64.  Times the full-ensemble likelihood
launch (250 000 evaluations) and the multi-ensemble device sampler, sweeping the
launch geometry."""
import os, sys, time, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mbb_emcee_amd as mbb
from bench import BANDS

def setup(ns=1000, nw=250, seed=5):
    rng = np.random.RandomState(seed)
    truths = np.column_stack([rng.uniform(8, 20, ns), rng.uniform(1.2, 2.4, ns), rng.uniform(300, 900, ns),
                              rng.uniform(2, 4.5, ns), rng.uniform(10, 80, ns)])
    one = mbb.likelihood(response=True)
    one.set_phot(BANDS, np.ones(8), np.ones(8))
    flux = one.model_flux(truths)
    like = mbb.likelihood(response=True)
    like.set_phot_multi(BANDS, flux, 0.1 * flux + 1.0)
    p0 = truths[:, None, :] * (1.0 + 0.03 * rng.normal(size=(ns, nw, 5)))
    return like, truths, p0

def main():
    ns, nw = 1000, 250
    like, truths, p0 = setup(ns, nw)
    ctx = like._sync_device()
    n = ns * nw
    flat = np.ascontiguousarray(p0.reshape(-1, 5))
    d_pars = ctx.alloc(flat.nbytes); d_pars.upload(flat)
    d_lnl = ctx.alloc(n * 8); d_st = ctx.alloc(n * 4)
    nq = ctx.info("nq")
    if "--prepass" in sys.argv:
        ctx.set_option("prepass", int(sys.argv[sys.argv.index("--prepass") + 1]))
    best = None
    sweep = [(0, 0)] if "--quick" in sys.argv else [(0, 0), (16, 256), (8, 512), (16, 512), (24, 512), (32, 512), (24, 384), (32, 1024), (48, 768), (48, 1024), (64, 1024)]
    if "--stage" in sys.argv:
        sweep = [(16, 256, 0), (16, 256, 1), (16, 512, 1), (32, 512, 1), (32, 1024, 1), (64, 1024, 1), (64, 512, 1), (32, 256, 1), (8, 256, 1)]
    for item in sweep:
        wpb, thr = item[0], item[1]
        ctx.set_option("stage_tables", item[2] if len(item) > 2 else -1)
        ctx.set_option("walkers_per_group", wpb); ctx.set_option("block_threads", thr)
        ctx.lnlike_repeat_device(d_pars, n, d_lnl, d_st, 2); ctx.sync()
        e0, e1 = ctx.event(), ctx.event()
        ctx.record(e0); ctx.lnlike_repeat_device(d_pars, n, d_lnl, d_st, 5); ctx.record(e1); ctx.sync()
        ms = ctx.elapsed_ms(e0, e1) / 5
        print("stage=%d " % ctx.info("last_stage"), end="")
        print("lnlike n=%d wpb=%d thr=%d: %.3f ms  %.3g evals/s  %.3g samples/s" %
              (n, ctx.info("last_wpb"), ctx.info("last_threads"), ms, n / ms * 1e3, n * nq / ms * 1e3), flush=True)
        if best is None or ms < best[0]:
            best = (ms, ctx.info("last_wpb"), ctx.info("last_threads"))
    ctx.set_option("walkers_per_group", 0); ctx.set_option("block_threads", 0); ctx.set_option("stage_tables", -1)
    s = mbb.DeviceEnsembleSampler(nw, 5, like, seed=3)
    s.run_mcmc(p0, 5, storechain=False)
    e0, e1 = ctx.event(), ctx.event()
    t0 = time.perf_counter()
    ctx.record(e0); s.advance_async(20); ctx.record(e1); ctx.sync()
    dt = time.perf_counter() - t0
    print(json.dumps({"cfg5": {"sources": ns, "walkers_per_source": nw, "best_lnlike_ms": best[0], "best_geometry": best[1:],
                               "lnlike_evals_per_s": n / best[0] * 1e3, "samples_per_s": n * nq / best[0] * 1e3,
                               "sampler_ms_per_step": dt / 20 * 1e3, "sampler_evals_per_s": n * 20 / dt}}))

if __name__ == "__main__":
    main()
