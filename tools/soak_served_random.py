#!/usr/bin/env python3
"""The served boundary on random configurations (band subsets from one delta band to 13 bands, model variant, priors incl.
the peak wavelength's, limits, covariance, workgroup width, staging, batch sizes up to a row per CU) against a launch per call,
bit for bit.      python tools/soak_served_random.py [first seed] [count]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mbb_emcee_amd as mbb

POOL = ["PACS_70um", "PACS_100um", "PACS_160um", "SPIRE_250um", "SPIRE_350um", "SPIRE_500um", "SCUBA2_450um", "SCUBA2_850um",
        "Bolocam_1.1mm", "MAMBO2_1.2mm", "GISMO_2mm", "LABOCA_870um", "MIPS_160um", "X_box_850um_60", "ALMA_alma_343",
        "ALMA_alma_230", "S_gauss_345_10", "D_dsb_300_20_6", "Y_delta_1300um", "Y_delta_450um"]
first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 500
bad, t0, served_cases = [], time.time(), 0
for seed in range(first, first + count):
    rng = np.random.RandomState(90000 + seed)
    nb = int(rng.choice([1, 2, 3, 5, 8, 13]))
    names = [str(x) for x in rng.choice(POOL, nb, replace=(nb > 8))]
    like = mbb.likelihood(opthin=bool(rng.randint(2)), noalpha=bool(rng.randint(2)), response=True)
    ctx = like.context
    for o, v in (("block_threads", int(rng.choice([0, 0, 64, 256, 768]))), ("seg_chunks", int(rng.choice([0, 0, 1, 2, 6]))),
                 ("stage_tables", int(rng.choice([-1, -1, 0]))), ("pack_tails", int(rng.choice([1, 1, 0]))),
                 ("serve_overlap", int(rng.choice([1, 1, 0, 2])))):
        if (o, v) not in (("block_threads", 0), ("seg_chunks", 0), ("stage_tables", -1), ("pack_tails", 1), ("serve_overlap", 1)):
            ctx.set_option(o, v)
    truth = np.array([rng.uniform(10, 30), rng.uniform(1.2, 2.4), rng.uniform(100, 700), rng.uniform(2.0, 4.5), rng.uniform(10, 80)])
    like.set_phot(names, np.ones(nb), np.ones(nb))
    flux = like.model_flux(truth)[0]
    unc = 0.08 * flux + 0.3
    like.set_phot(names, flux, unc)
    if nb > 1 and rng.rand() < 0.5:
        A = rng.normal(0, 1, (nb, nb))
        like.set_cov(np.diag(unc ** 2) + 0.02 * np.median(unc) ** 2 * A.dot(A.T))
    for i in range(6):
        if rng.rand() < 0.3:
            centre = truth[i] if i < 5 else 200.0
            like.set_gaussian_prior(i, centre * rng.uniform(0.95, 1.05), abs(centre) * rng.uniform(0.05, 0.3))
    if rng.rand() < 0.4:
        like.set_uplim("T", truth[0] * 1.02)
    if rng.rand() < 0.3:
        like.set_uplim("lambda_peak", 220.0)
    if rng.rand() < 0.3:
        like.set_lowlim("beta", truth[1] * 0.97)
    n = int(rng.choice([1, 3, 64, 125, 250, 256]))
    p = truth * (1.0 + 0.06 * rng.normal(size=(n, 5)))
    if n > 8:
        p[5, 1] = -0.2; p[2, 0] = np.nan                      # (a row the gate sends back, a row that is not a number)
    arg = p if n > 1 else p[0].copy()
    ctx = like._sync_device()
    ctx.set_option("serve", 0)
    try:
        want = np.atleast_1d(like(arg)).copy()
    except ValueError:
        continue                                             # (a row the reference raises for: not this probe's business)
    ctx.set_option("serve", 1)
    ok = True
    for _ in range(6):
        ok = ok and np.array_equal(np.atleast_1d(like(arg)), want, equal_nan=True)
    if ctx.info("serve_requests") > 0:
        served_cases += 1
    if not ok or ctx.info("serve_fallbacks"):
        bad.append(seed)
        print("seed %d: %s rows %d %s" % (seed, names, n, "DIFFERENT" if not ok else "fell back"), flush=True)
    del like
    if (seed - first) % 100 == 99:
        print("... %d configurations, %d served, %d wrong, %.0f s" % (seed - first + 1, served_cases, len(bad), time.time() - t0), flush=True)
print("served boundary, random configurations: %d of %d wrong %s (%d of them served)" % (len(bad), count, bad, served_cases))
sys.exit(1 if bad else 0)
