#!/usr/bin/env python3
"""What a NEW fitter costs beside its 2.3 ms of sampling: context, tables, first launches, sampler set-up.

    python tools/probe_first_fit.py

A catalogue loop makes one mbb_fitter per source; this times, for the 2nd..6th fitter of a process, every stage of
`fit = mbb_fitter(...); fit.set_data(...); p0 = ...; fit.run(50, 250, p0)`."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mbb_emcee_amd as mbb
from bench import BANDS, TRUTH, make_likelihood

like0, flux = make_likelihood(0)
unc = 0.1 * flux + 1.0
rows = []
for i in range(6):
    t = [time.perf_counter()]
    fit = mbb.mbb_fitter(nwalkers=250, response=True, seed=3); t.append(time.perf_counter())
    fit.set_data(BANDS, flux, unc); t.append(time.perf_counter())
    p0 = fit.generate_initial_values(np.array([10.0, 2.0, 600.0, 4.0, 40.0]), np.array([2.0, 0.2, 100.0, 0.3, 5.0])); t.append(time.perf_counter())
    ctx = fit.like.context; t.append(time.perf_counter())                 # mbb_ctx_create
    fit.like._sync_device(); t.append(time.perf_counter())               # band layout + uploads
    fit.like(p0[:2]); t.append(time.perf_counter())                       # first launch of this context
    fit.run(50, 250, p0); t.append(time.perf_counter())
    fit.run(50, 250, p0); t.append(time.perf_counter())
    rows.append(np.diff(t) * 1e3)
names = ["mbb_fitter()", "set_data (passbands, host)", "generate_initial_values", "context (mbb_ctx_create)", "_sync_device (layout, uploads)",
         "first likelihood launch", "first run(50, 250)", "second run(50, 250)"]
med = np.median(np.array(rows[1:]), axis=0)
for n, v, f in zip(names, med, rows[0]):
    print("%-34s %7.3f ms   (first fitter of the process: %.3f)" % (n, v, f))
print("%-34s %7.3f ms" % ("a new fitter, all told", med[:7].sum()))
