#!/usr/bin/env python3
"""The served boundary under the table / layout options and a few band sets, against a launch per call, bit for bit.
    python tools/soak_served_options.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import mbb_emcee_amd as mbb
import bench_configs as bc
from bench import BANDS, TRUTH


def cfg2(opthin=False, noalpha=False, cov=False):
    like = mbb.likelihood(response=True, opthin=opthin, noalpha=noalpha)
    like.set_phot(BANDS, np.ones(8), np.ones(8))
    flux = like.model_flux(TRUTH)[0]
    unc = 0.1 * flux + 1.0
    like.set_phot(BANDS, flux, unc)
    if cov:
        A = np.random.RandomState(3).normal(size=(8, 8))
        like.set_cov(np.diag(unc ** 2) + 0.02 * np.median(unc) ** 2 * A.dot(A.T))
    return like


def mixed():
    names = ["SPIRE_250um", "Y_delta_1300um", "PACS_100um", "Y_delta_2000um", "Y_delta_70um", "SCUBA2_850um"]
    like = mbb.likelihood(response=True)
    like.set_phot(names, np.ones(6), np.ones(6))
    flux = like.model_flux(np.array([22.0, 1.6, 180.0, 2.8, 50.0]))[0]
    like.set_phot(names, flux, 0.1 * flux + 0.1)
    return like


sets = {"cfg2": cfg2, "cfg2 thin": lambda: cfg2(opthin=True), "cfg2 noalpha": lambda: cfg2(noalpha=True), "cfg2 cov": lambda: cfg2(cov=True),
        "cfg1": lambda: bc.setup("cfg1")[0], "cfg4": lambda: bc.setup("cfg4")[0], "mixed delta + passbands": mixed}
rng = np.random.RandomState(5)
bad = n_cases = 0
for name, make in sets.items():
    for opts in ({}, {"stage_tables": 0}, {"pack_tails": 0}, {"seg_chunks": 1}, {"seg_chunks": 2, "stage_tables": 0}, {"block_threads": 128},
                 {"serve_overlap": 0}, {"serve_overlap": 2}):
        like = make()
        ctx = like.context
        for k, v in opts.items():
            ctx.set_option(k, v)
        p = np.asarray(TRUTH if "mixed" not in name else [22.0, 1.6, 180.0, 2.8, 50.0]) * (1.0 + 0.08 * rng.normal(size=(200, 5)))
        p[7, 1] = -0.3; p[11, 3] = 0.01                                 # (rows the gate sends back)
        ctx.set_option("serve", 0)
        want = {n: like(p[:n]).copy() for n in (1, 7, 125, 200)}
        ctx.set_option("serve", 1)
        ok = True
        for rep in range(4):
            for n in (125, 125, 125, 125, 7, 1, 200):
                got = like(p[:n])
                ok = ok and np.array_equal(got, want[n], equal_nan=True)
        served = ctx.info("serving") == 1 or ctx.info("serve_requests") > 0
        n_cases += 1
        if not ok or not served or ctx.info("serve_fallbacks"):
            bad += 1
        print("%-24s %-40s %s  (requests %d, fall-backs %d, chunks %d)" % (name, opts, "equal" if ok else "DIFFERENT", ctx.info("serve_requests"),
              ctx.info("serve_fallbacks"), ctx.info("nchunk")), flush=True)
        del like
print("served boundary: %d of %d cases wrong" % (bad, n_cases))
sys.exit(1 if bad else 0)
