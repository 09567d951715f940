#!/usr/bin/env python3
"""Sampler form 7 (k_flowm: quadrature of both candidates ahead) against the plain launch train and form 9:
bitwise equality of a short run, then stream time per step.  MBB_PROBE_SPIN: log2 of the polls before a wait gives up."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mbb_emcee_amd as mbb
from bench import BANDS, TRUTH, walkers, NW_PER_GPU

SPIN = int(os.environ.get("MBB_PROBE_SPIN", "16"))
NW = int(os.environ.get("MBB_PROBE_NW", str(NW_PER_GPU)))


def make(opts):
    like = mbb.likelihood(response=True)
    for k, v in opts.items():
        like.context.set_option(k, v)
    like.set_phot(BANDS, np.ones(8), np.ones(8))
    flux = like.model_flux(TRUTH)[0]
    like.set_phot(BANDS, flux, 0.1 * flux + 1.0)
    return like


FORMS = {"plain": {"lookahead_sampler": 0, "flow_sampler": 0},
         "form9": {"lookahead_sampler": 1, "flow_sampler": 1, "resident_sampler": 2, "resident_ahead": 1},
         "form7": {"lookahead_sampler": 1, "flow_sampler": 1, "merged_flow_sampler": 1, "flow_spin_log2": SPIN}}
res = {}
for name, opts in FORMS.items():
    like = make(opts)
    s = mbb.DeviceEnsembleSampler(NW, 5, like, seed=11)
    t0 = time.time()
    pos, lnp, _ = s.run_mcmc(walkers(1)[:NW], 30)
    b = s.run_mcmc(None, 11, storechain=False)
    res[name] = (pos, lnp, b[0], b[1], s.chain.copy(), s.lnprobability.copy(), s.naccepted.copy())
    print(name, "form", like.context.info("last_kernel_form"), "fallbacks", like.context.info("flow_fallbacks"),
          "threads", like.context.info("last_threads"), "grid", like.context.info("last_grid"),
          "%.2f s" % (time.time() - t0), flush=True)
for name in ("form9", "form7"):
    same = all(np.array_equal(x, y) for x, y in zip(res["plain"], res[name]))
    print(name, "bitwise equal to the plain train:", same, flush=True)
    if not same:
        print("  equal parts (pos, lnp, pos2, lnp2, chain, lnprob, nacc):", [bool(np.array_equal(x, y)) for x, y in zip(res["plain"], res[name])])
        for i in (0, 1, 2, 3, 5, 6):
            d = np.argwhere(res["plain"][i] != res[name][i])
            print("   part", i, "differs at", len(d), "places, first", d[:6].tolist())
        c0, c1 = res["plain"][4], res[name][4]
        bad = np.argwhere((c0 != c1).any(axis=2))
        print("  first differing (walker, step):", bad[np.argsort(bad[:, 1])][:8].tolist())
if not all(np.array_equal(x, y) for x, y in zip(res["plain"], res["form7"])):
    sys.exit(1)
for name in ("form9", "form7", "form9", "form7"):
    like = make(FORMS[name])
    ctx = like._sync_device()
    smp = mbb.DeviceEnsembleSampler(NW, 5, like, seed=11)
    smp.run_mcmc(walkers(1)[:NW], 20, storechain=False)
    smp.advance_async(200); ctx.sync()
    best = 1e9
    for _ in range(5):
        e0, e1 = ctx.event(), ctx.event()
        ctx.record(e0); smp.advance_async(2000); ctx.record(e1); ctx.sync()
        best = min(best, ctx.elapsed_ms(e0, e1) * 1e3 / 2000)
    print(name, "%.3f us per step" % best, "form", ctx.info("last_kernel_form"), "fallbacks", ctx.info("flow_fallbacks"), flush=True)
