#!/usr/bin/env python3
"""Device-sampler cost against ensemble size (cfg2 bands: 8 passbands, NQ 2209, thick+alpha):

    python tools/sweep_walkers.py [sizes ...]

For every ensemble size: the form the host chooses (mbb_hip.hip, sampler_enqueue), and -- where they are
eligible -- each of the forms (7, 8, 9, the launch train) forced through the options: us per MCMC step by HIP events on the stream,
evals/s = walkers / that.  The cut-overs between the forms (form 7 up to 2 walkers per CU, form 9 up to four
walkers per CU and half, form 8 up to eight, the launch train beyond) are then read off the curve.
250 000 walkers = cfg5's count run as ONE ensemble of that size (the multi-source launch is bench.py's cfg5).
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mbb_emcee_amd as mbb
from bench import make_likelihood, TRUTH

SIZES = [18, 64, 128, 250, 256, 258, 300, 340, 384, 450, 512, 514, 768, 1000, 1500, 2000, 3000, 4096, 16384]
FORMS = (("chosen", {}),
         ("form7", {"lookahead_sampler": 1, "flow_sampler": 1, "merged_flow_sampler": 1, "resident_sampler": 0}),
         ("form8", {"lookahead_sampler": 1, "flow_sampler": 1, "resident_sampler": 2, "resident_ahead": 0}),
         ("form9", {"lookahead_sampler": 1, "flow_sampler": 1, "resident_sampler": 2, "resident_ahead": 1}),
         ("train", {"lookahead_sampler": 0}))


def main():
    sizes = [int(a) for a in sys.argv[1:]] or SIZES
    like, flux = make_likelihood(0)
    ctx = like._sync_device()
    rows = []
    print("%8s %-7s %5s %6s %12s %12s" % ("walkers", "asked", "form", "wgs", "us/step", "evals/s"), flush=True)
    for nw in sizes:
        rng = np.random.RandomState(1)
        p0 = TRUTH * (1.0 + 0.03 * rng.normal(size=(nw, 5)))
        steps = 2000 if nw <= 512 else (400 if nw <= 4096 else 20)
        done = {}
        for name, opts in FORMS:
            ctx.set_option("lookahead_sampler", 1); ctx.set_option("flow_sampler", 1); ctx.set_option("merged_flow_sampler", 1)
            ctx.set_option("resident_sampler", 1); ctx.set_option("resident_ahead", 1)
            for o, v in opts.items():
                ctx.set_option(o, v)
            s = mbb.DeviceEnsembleSampler(nw, 5, like, seed=5)
            s.run_mcmc(p0, max(2, steps // 20), storechain=False)
            form = ctx.info("last_kernel_form")
            if name != "chosen" and form in done.values() and name != "train":
                del s
                continue                      # not eligible at this size: it fell to a form already timed
            s.advance_async(max(2, steps // 10)); ctx.sync()
            e0, e1 = ctx.event(), ctx.event()
            ctx.record(e0); s.advance_async(steps); ctx.record(e1); ctx.sync()
            us = ctx.elapsed_ms(e0, e1) * 1e3 / steps
            form = ctx.info("last_kernel_form")
            pos, lnp, _ = s.run_mcmc(None, 0, storechain=False)
            assert np.all(np.isfinite(lnp))
            if name != "chosen":
                done[name] = form
            rows.append({"walkers": nw, "asked": name, "form": form, "workgroups": ctx.info("last_grid"),
                         "us_per_step": us, "evals_per_s": nw / (us * 1e-6), "steps": steps})
            print("%8d %-7s %5d %6d %12.2f %12.4g" % (nw, name, form, ctx.info("last_grid"), us, nw / (us * 1e-6)), flush=True)
            del s
    ctx.set_option("lookahead_sampler", 1); ctx.set_option("flow_sampler", 1); ctx.set_option("merged_flow_sampler", 1)
    ctx.set_option("resident_sampler", 1)
    print(json.dumps(rows))


if __name__ == "__main__":
    main()
