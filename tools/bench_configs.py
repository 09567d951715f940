#!/usr/bin/env python3
"""BASELINE.json configs[0] and configs[3] (SURVEY.md 8d "cfg1", "cfg4") as measurable workloads.

    python tools/bench_configs.py cfg1|cfg4 [--profile]

cfg1  5-band delta-function photometry (100, 160, 250, 350, 500 um), optically thin + alpha, 50 walkers
      (what run_mbb_emcee.py does without --response; reference likelihood.py:817)
cfg4  12 bands incl. SCUBA2 / ALMA boxcar and double-sideband passbands (NQ 3246), optically thick + alpha,
      covariance-matrix likelihood (likelihood.py:823), 250 walkers

`measure()` is what bench.py puts on its line under `configs`: M1 (the boundary, host arrays in and out),
M2 (the device-resident sampler in its default form), the plain launch of a half-ensemble by HIP events,
the fp64 roofline of that launch from the committed PMC pass of THIS script (profiles/rNN/pmc_valu_<cfg>.json),
and the CPU oracle on the same rows.  With --profile the script only enqueues the launches the counters
are taken from (`rocprofv3 --pmc ... -- python3 tools/bench_configs.py cfg1 --profile`).
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

CFG4_BANDS = ["PACS_70um", "PACS_100um", "PACS_160um", "SPIRE_250um", "SPIRE_350um", "SPIRE_500um",
              "SCUBA2_450um", "SCUBA2_850um", "SCUBA2_box_850um_85", "ALMA_alma_343", "ALMA_alma_230",
              "ALMA_box_145_7.5"]
CFG1_WAVE = [100.0, 160.0, 250.0, 350.0, 500.0]
TRUTH = np.array([12.0, 1.8, 600.0, 3.0, 40.0])
PROFILE_STEPS, PROFILE_WARM, PROFILE_LAUNCHES = 300, 50, 400


def walkers(n):
    """RandomState(0), N(12,1), N(1.8,.2), N(600,50), N(3,.3), N(40,3)."""
    rng = np.random.RandomState(0)
    return np.column_stack([rng.normal(12, 1, n), rng.normal(1.8, 0.2, n), rng.normal(600, 50, n),
                            rng.normal(3, 0.3, n), rng.normal(40, 3, n)])


def setup(name, device=0):
    """(likelihood, walkers [nw, 5], description, keyword arguments of the matching oracle)"""
    import mbb_emcee_amd as mbb
    if name == "cfg1":
        like = mbb.likelihood(opthin=True, device=device)
        like.set_phot(CFG1_WAVE, np.ones(5), np.ones(5))
        flux = like.model_flux(TRUTH)[0]
        unc = 0.1 * flux + 1.0
        like.set_phot(CFG1_WAVE, flux, unc)
        okw = dict(flux=flux, unc=unc, wave=np.asarray(CFG1_WAVE), opthin=True,
                   has_uplim=[int(b) for b in like.has_uplims], uplim=like.uplims)
        desc = ("cfg1: 5 delta-function bands (100-500 um), optically thin + alpha, 50 walkers "
                "(BASELINE.json configs[0]; reference likelihood.py:817)")
        return like, walkers(50), desc, okw
    if name == "cfg4":
        like = mbb.likelihood(response=True, device=device)
        like.set_phot(CFG4_BANDS, np.ones(12), np.ones(12))
        flux = like.model_flux(TRUTH)[0]
        unc = 0.1 * flux + 1.0
        like.set_phot(CFG4_BANDS, flux, unc)
        A = np.random.RandomState(4).normal(0, 1, (12, 12))
        cov = np.diag(unc ** 2) + 0.01 * A.dot(A.T)
        like.set_cov(cov)
        okw = dict(flux=flux, unc=unc, cov=cov,
                   bands=[(r.wavelength, r._sedmult, r._normfac) for r in like._responses],
                   has_uplim=[int(b) for b in like.has_uplims], uplim=like.uplims)
        desc = ("cfg4: 12 bands incl. SCUBA2/ALMA boxcar and double-sideband passbands (NQ %d), optically "
                "thick + alpha, covariance-matrix likelihood, 250 walkers (BASELINE.json configs[3]; "
                "reference likelihood.py:823)" % like._sync_device().info("nq"))
        return like, walkers(250), desc, okw
    raise ValueError(name)


def ev_us(ctx, fn, reps):
    e0, e1 = ctx.event(), ctx.event()
    ctx.record(e0); fn(); ctx.record(e1); ctx.sync()
    return ctx.elapsed_ms(e0, e1) * 1e3 / reps


def profile_run(name):
    """What the counter passes see: PROFILE_LAUNCHES plain launches of a half-ensemble, and the device
    sampler for PROFILE_WARM + PROFILE_STEPS steps in its default form."""
    import mbb_emcee_amd as mbb
    like, pos, desc, _ = setup(name)
    ctx = like._sync_device()
    half = pos.shape[0] // 2
    p = np.ascontiguousarray(pos[:half])
    dp = ctx.alloc(p.nbytes); dp.upload(p)
    dl, ds = ctx.alloc(half * 8), ctx.alloc(half * 4)
    ctx.lnlike_repeat_device(dp, half, dl, ds, PROFILE_LAUNCHES); ctx.sync()
    smp = mbb.DeviceEnsembleSampler(pos.shape[0], 5, like, seed=11)
    smp.run_mcmc(pos, PROFILE_WARM, storechain=False)
    smp.advance_async(PROFILE_STEPS); ctx.sync()
    print(json.dumps({"config": name, "plain_launches": PROFILE_LAUNCHES, "half": half,
                      "sampler_half_steps": 2 * (PROFILE_WARM + PROFILE_STEPS), "form": ctx.info("last_kernel_form")}))


def measure(name, roofline_fn=None, cpu=True, device=0):
    """The object bench.py prints under configs[name]."""
    import mbb_emcee_amd as mbb
    like, pos, desc, okw = setup(name, device)
    ctx = like._sync_device()
    nw = pos.shape[0]
    half = nw // 2
    out = {"workload": desc, "walkers": nw, "bands": ctx.info("nb"), "nq": ctx.info("nq")}

    # M1: the boundary, synchronous likelihood.__call__ on host arrays (what emcee calls per half-step), in a loop of
    # calls.  (A fit has ONE likelihood, and its loop of calls is served by the resident kernel; beside bench.py's own
    # context that has to be asked for: "serve" 2.)
    ctx.set_option("serve", 2)
    bnd = {}
    for n in (half, nw):
        p = np.ascontiguousarray(pos[:n])
        for _ in range(50):
            like(p)
        ts = []
        for _ in range(300):
            t0 = time.perf_counter(); like(p); ts.append(time.perf_counter() - t0)
        med = float(np.median(ts))
        bnd["rows_%d" % n] = {"median_us": med * 1e6, "evals_per_s": n / med, "served_by_resident_kernel": bool(ctx.info("serving"))}
    ctx.set_option("serve", 1)
    out["boundary_M1"] = bnd

    # the plain launch of a half-ensemble, device-resident rows, by HIP events
    p = np.ascontiguousarray(pos[:half])
    dp = ctx.alloc(p.nbytes); dp.upload(p)
    dl, ds = ctx.alloc(half * 8), ctx.alloc(half * 4)
    ctx.lnlike_repeat_device(dp, half, dl, ds, 100); ctx.sync()
    k_us = ev_us(ctx, lambda: ctx.lnlike_repeat_device(dp, half, dl, ds, 500), 500)
    out["plain_launch"] = {"rows": half, "kernel_us": k_us, "evals_per_s": half / (k_us * 1e-6),
                           "geometry": {"walkers_per_workgroup": ctx.info("last_wpb"), "threads": ctx.info("last_threads"),
                                        "workgroups": ctx.info("last_grid")}}

    # M2: the device-resident sampler in its default form
    smp = mbb.DeviceEnsembleSampler(nw, 5, like, seed=11)
    smp.run_mcmc(pos, 60, storechain=False)
    smp.advance_async(200); ctx.sync()
    steps = 2000
    t0 = time.perf_counter()
    s_us = ev_us(ctx, lambda: smp.advance_async(steps), steps)
    wall = time.perf_counter() - t0
    form = ctx.info("last_kernel_form")
    pos_end, lnp_end, _ = smp.run_mcmc(None, 0, storechain=False)
    assert np.all(np.isfinite(lnp_end))
    out["sampler_M2"] = {"stream_us_per_step": s_us, "evals_per_s": nw / (s_us * 1e-6), "mcmc_steps_per_s": 1e6 / s_us,
                         "wall_us_per_step": wall * 1e6 / steps, "steps": steps, "kernel_form": form,
                         "workgroups": ctx.info("last_grid"), "half_step_us": s_us / 2,
                         "acceptance_fraction": float(np.mean(smp.naccepted)) / (60 + 200 + steps)}
    # the same chain as a train of plain launches
    ctx.set_option("lookahead_sampler", 0)
    s2 = mbb.DeviceEnsembleSampler(nw, 5, like, seed=11)
    s2.run_mcmc(pos, 60, storechain=False)
    s2.advance_async(100); ctx.sync()
    t_us = ev_us(ctx, lambda: s2.advance_async(500), 500)
    ctx.set_option("lookahead_sampler", 1)
    out["sampler_M2"]["launch_train_us_per_step"] = t_us

    if roofline_fn is not None:
        out["roofline"] = roofline_fn(name, k_us, s_us / 2, form, half, ctx)

    if cpu:
        from oracle import oracle as O
        orc = O.OracleLikelihood(**okw)
        cores = max(1, min(len(os.sched_getaffinity(0)), O.num_threads(), int(os.environ.get("MBB_CPU_THREADS", "16"))))
        rows = np.tile(pos, (max(1, 2000 // nw), 1))
        orc(rows[:nw], nthreads=1)
        t0 = time.perf_counter(); ref = orc(rows, nthreads=1); t1 = time.perf_counter() - t0
        rate1 = rows.shape[0] / t1
        # about 3 seconds of wall time with every core busy
        nbig = max(nw * cores, int(3.0 * rate1 * cores * 0.7))
        big = np.tile(pos, (nbig // nw + 1, 1))[:nbig]
        orc(big[:cores * nw], nthreads=cores)
        t0 = time.perf_counter(); orc(big, nthreads=cores); tm = time.perf_counter() - t0
        out["cpu_baseline"] = {"value": big.shape[0] / tm, "unit": "walker-likelihood evals/s", "cores": cores, "kind": "port",
                               "sample": "%d evals of this config's walkers (tiled), OpenMP over rows; single thread %.0f evals/s"
                                         % (big.shape[0], rate1), "single_thread_value": rate1}
        got = like(pos)
        err = np.abs(got - ref[:nw]) / np.maximum(1.0, np.abs(ref[:nw]))
        out["parity_max_err_vs_oracle"] = float(err.max())
        assert err.max() < 1e-10
    return out


if __name__ == "__main__":
    cfg = sys.argv[1]
    if "--profile" in sys.argv:
        profile_run(cfg)
    else:
        print(json.dumps(measure(cfg), indent=1))
