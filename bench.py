#!/usr/bin/env python3
"""Benchmark of the per-walker likelihood hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

Workload (BASELINE.json configs[1] / configs[2], SURVEY.md 8d "cfg2/cfg3"):
8 passbands (PACS 70/100/160, SPIRE 250/350/500, SCUBA2 850, Bolocam 1.1mm;
2209 quadrature samples), optically thick + alpha model, 250 walkers per GPU.

One *step* is one MCMC step of the whole ensemble (metric M2): the
device-resident stretch-move sampler advances 250 N walkers, two DEPENDENT half-steps
per step (emcee's, mbb_fit.py:80-81 / :533), each evaluating the fused likelihood of
its half.  On one GPU a run is ONE launch per 4096 steps (k_flowm, form 7): a workgroup
per (pair of walkers, candidate), the quadrature of both proposals a walker can end up
making and the SED constructor for every outcome still open running ahead of the
decisions they depend on -- the same chain, bit for bit, as one launch per half-step
(tests/test_gpu_parity.py).  With N > 1 ranks the ensemble is sharded (125 moving
walkers per GPU per half-step); the one-launch run goes across the ranks (SMODE 6:
decisions, rows and progress words stored into every rank's copy as they are made) or,
if that does not come up, one launch per half-step with the moved state rows exchanged
after every launch -- by the one-hop peer-write exchange (mbb_xchg_*: the accepting lane
stores the row into every rank's copy through hipIpc mappings), or by an in-place
ncclAllGather over RCCL (--exchange rccl; also the automatic fall-back).  Positions live
in HBM for the whole run: there is no host round trip inside the timed region.  `value`
= walker-likelihood evaluations per second of that real chain = 250 N K / t.

Also measured (rank 0, N = 1, outside the timed region; in the side file, the starred ones on the line too):
  boundary *    SURVEY.md 8d metric M1: synchronous likelihood.__call__ on host arrays
                (PCIe inclusive), median of >= 200 calls, for 125 and 250 rows
  pipelined     independent launches on pre-computed proposals enqueued back to back
                (an upper bound: no dependence between launches)
  roofline *    the binding roof of the dominant kernel: fp64 vector arithmetic
                (flop and VALU counts from the committed rocprofv3 PMC pass of this very
                launch), the empirical sample-arithmetic roof measured in this run, and
                the HBM figures north_star asks for
  cfg5          1000 sources x 250 walkers in one launch (BASELINE.json configs[4])
  cpu_baseline * the CPU oracle timed on this box's cores

Prints ONE short JSON line on rank 0 (< 4 KB: the contract's keys, roofline, roofline_hbm,
boundary_M1, cpu_baseline, cfg5's three numbers) and writes everything measured -- the
legs listed here in full -- to gpurun_out/bench_full.json.  torch is used only as launcher plumbing
(torch.distributed gloo rendezvous + barrier); all device work goes through the C-ABI
of libmbb_hip.so.  A collective that cannot be set up or does not return is a failure:
the line says so and the exit status is non-zero.
"""
import argparse
import glob
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BANDS = ["PACS_70um", "PACS_100um", "PACS_160um", "SPIRE_250um",
         "SPIRE_350um", "SPIRE_500um", "SCUBA2_850um", "Bolocam_1.1mm"]
TRUTH = np.array([12.0, 1.8, 600.0, 3.0, 40.0])
NW_PER_GPU = 250
# (rehearsals of the N > 1 machinery on one GPU only: a smaller ensemble per rank lets the ranks' one-launch
# kernels all be resident on the shared device; the line is then marked invalid)
if os.environ.get("MBB_BENCH_WALKERS_PER_GPU"):
    NW_PER_GPU = int(os.environ["MBB_BENCH_WALKERS_PER_GPU"])
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
FP64_VALU_PEAK_TFLOPS = 78.6   # vendor fp64 vector peak
N_SIMD = 1024                  # 256 CUs x 4
ONE_LAUNCH_FORMS = (7, 9)   # sampler forms whose launches cover many half-steps: counters are taken per half-step
# N > 1: what the supervisor allows the whole run, what a rank allows any one step that may wedge (communicator
# set-up, a rehearsal, a timed run), and the age of the run beyond which nothing optional is started any more
SUPERVISOR_DEADLINE_S, GUARD_S, OPTIONAL_UNTIL_S = 480.0, 45.0, 240.0
PRECONDITION_S = 0.05      # one GPU: a scratch ensemble is stepped this long right before the W warm-up steps (GPU clocks)
PRECONDITION_STEPS = 2000  # N > 1: that many untimed steps of the sharded sampler itself instead
CLOCK_HZ = 2.4e9
# tools/issue_cost.hip, profiles/r02/issue_cost_v1.txt: cycles one wave64 VALU
# instruction holds its SIMD, four waves per SIMD
CYC_FP64, CYC_OTHER = 4.3, 4.0        # (32-bit ops alone: 2.5; beside fp64 work: ~3.7-4)


def walkers(nranks):
    """RandomState(0), N(12,1), N(1.8,.2), N(600,50), N(3,.3), N(40,3)."""
    rng = np.random.RandomState(0)
    n = max(2000, NW_PER_GPU * nranks)
    return np.column_stack([rng.normal(12, 1, n), rng.normal(1.8, 0.2, n),
                            rng.normal(600, 50, n), rng.normal(3, 0.3, n),
                            rng.normal(40, 3, n)])


def proposals(pos, nsets, seed):
    """Stretch-move proposals for both half-ensembles, nsets independent draws."""
    rng = np.random.RandomState(seed)
    half = pos.shape[0] // 2
    out = []
    for _ in range(nsets):
        for S0, S1 in ((slice(0, half), slice(half, None)), (slice(half, None), slice(0, half))):
            s, c = pos[S0], pos[S1]
            zz = ((2.0 - 1.0) * rng.rand(s.shape[0]) + 1.0) ** 2 / 2.0
            partner = c[rng.randint(c.shape[0], size=s.shape[0])]
            out.append(partner - zz[:, None] * (partner - s))
    return out


def make_likelihood(device):
    import mbb_emcee_amd as mbb
    like = mbb.likelihood(response=True, device=device)
    like.set_phot(BANDS, np.ones(8), np.ones(8))
    flux = like.model_flux(TRUTH)[0]
    like.set_phot(BANDS, flux, 0.1 * flux + 1.0)
    return like, flux


def cpu_baseline(like, flux, pars):
    """The CPU oracle (a port of the reference path) timed on this box's cores."""
    from oracle import oracle as O
    orc = O.OracleLikelihood(
        flux, 0.1 * flux + 1.0,
        bands=[(r.wavelength, r._sedmult, r._normfac) for r in like._responses],
        has_uplim=[int(b) for b in like.has_uplims], uplim=like.uplims)
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else os.cpu_count()
    # a one-GPU box has a CPU share of 16 however many CPUs it shows
    cores = max(1, min(cores, O.num_threads(), int(os.environ.get("MBB_CPU_THREADS", "16"))))
    p1 = np.tile(pars, (8, 1))            # 2000 evals single thread
    orc(p1[:250], nthreads=1)
    t0 = time.perf_counter(); orc(p1, nthreads=1); t1 = time.perf_counter() - t0
    rate1 = p1.shape[0] / t1
    # about 12 CPU-seconds of work in all: wall target = 12 s / cores
    pm = np.tile(pars, (8 * cores, 1))
    orc(pm, nthreads=cores)
    t0 = time.perf_counter(); orc(pm, nthreads=cores); tb = time.perf_counter() - t0
    reps = max(1, int(round((12.0 / cores) / max(tb, 1e-4))))
    pm = np.tile(pm, (reps, 1))
    t0 = time.perf_counter(); ref = orc(pm, nthreads=cores); tm = time.perf_counter() - t0
    reps = pm.shape[0] // 250
    return {"value": pm.shape[0] / tm, "unit": "walker-likelihood evals/s", "cores": cores,
            "kind": "port",
            "sample": "%d evals of the bench workload (250 walkers x 8 bands, NQ=2209) "
                      "tiled %dx, OpenMP over walkers; single-thread rate %.0f evals/s"
                      % (pm.shape[0], reps, rate1),
            "sample_short": "%d evals of the bench workload, OpenMP over walkers, ~12 CPU-s" % pm.shape[0],
            "single_thread_value": rate1}, ref[:250]


def profile_files(pattern):
    """Committed summaries matching `pattern`, newest round (then newest name) first."""
    return sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", pattern)), reverse=True)


def newest_profile(pattern):
    files = profile_files(pattern)
    return files[0] if files else None


def lookup_kernel(pattern, kernel_substr, prefer=None):
    """(entry, source file, error): the kernel's entry in the newest committed summary matching `pattern`
    that holds it.  A newer summary that lacks the kernel is named in the error text even when an older
    one answers, so a renamed instantiation cannot hide behind stale counters."""
    files = profile_files(pattern)
    if not files:
        return None, None, "no profiles/r*/%s committed" % pattern
    missed = []
    for f in files:
        src = os.path.relpath(f, ROOT)
        try:
            d = json.load(open(f))
            ks = d.get("kernels", {d.get("kernel", ""): d})
        except Exception as e:      # noqa
            missed.append("%s: %r" % (src, e))
            continue
        k, v = find_kernel(ks, kernel_substr, prefer)
        if v is not None:
            return v, src, ("older summary used; " + "; ".join(missed)) if missed else None
        missed.append("%s holds no kernel matching %r (it has: %s)" % (src, kernel_substr, "; ".join(sorted(ks))))
    return None, os.path.relpath(files[0], ROOT), "; ".join(missed[:2])


def kernel_key(form, opthin=False, noalpha=False, staged=True, pairs=None):
    """The substring that names a kernel instantiation in rocprofv3's output, from what was launched
    (`last_kernel_form`) -- never a literal copied from an old summary: round 3 lost its headline roofline
    to a template parameter added after the literal was written.  k_flowm's trailing parameter (pairs of
    walkers per workgroup) is left open unless `pairs` is given."""
    b = ("false", "true")
    if form == 7:
        key = "k_flowm<%s, %s, %s," % (b[bool(opthin)], b[bool(noalpha)], b[bool(staged)])
        return key + (" %d>" % pairs if pairs else "")
    if form == 9:               # the resident form: k_flowa (a workgroup owns walkers, the constructor a half-step ahead)
        return "k_flowa<%s, %s, %s>" % (b[bool(opthin)], b[bool(noalpha)], b[bool(staged)])
    return "k_lnlike<%s, %s, %d, %s>" % (b[bool(opthin)], b[bool(noalpha)], form, b[bool(staged)])


def find_kernel(kernels, key, prefer=None):
    """The entry of a summary's `kernels` whose name contains `key` (`prefer`: a second substring that
    decides between several hits).  Returns (name, entry) or (None, None)."""
    hits = [(k, v) for k, v in kernels.items() if key in k]
    if prefer:
        hits = [h for h in hits if prefer in h[0]] or hits
    return hits[0] if hits else (None, None)


def measured_traffic(kernel_substr, prefer=None):
    """HBM-side bytes of the dominant kernel from the committed rocprofv3 PMC summaries (separate FETCH_SIZE /
    WRITE_SIZE passes of this same command; tools/summarize_pmc.py).  bench.py cannot run the profiler on itself."""
    return lookup_kernel("pmc_traffic*.json", kernel_substr, prefer)


def measured_valu(pattern, kernel_substr, prefer=None):
    """VALU / fp64 instruction counts per launch from the committed PMC summaries (tools/summarize_valu.py)."""
    return lookup_kernel(pattern, kernel_substr, prefer)


def valu_roofline(pm, pm_src, kernel_s, label):
    """fp64-vector roofline object from PMC counts per launch and a launch duration."""
    if not pm:
        return None
    c = pm["counters_per_launch"]
    fma, add, mul = (c.get("SQ_INSTS_VALU_" + k, 0.0) for k in ("FMA_F64", "ADD_F64", "MUL_F64"))
    valu = c.get("SQ_INSTS_VALU", 0.0)
    flops = 64.0 * (2.0 * fma + add + mul)
    f64 = fma + add + mul + c.get("SQ_INSTS_VALU_TRANS_F64", 0.0)
    # issue bound: every wave64 VALU instruction holds its SIMD 4.3 (fp64 pipe) or about 4
    # cycles (everything else once it is mixed with fp64 work): tools/issue_cost.hip
    issue_s = (f64 * CYC_FP64 + (valu - f64) * CYC_OTHER) / N_SIMD / CLOCK_HZ
    tf = flops / kernel_s / 1e12
    return {"bound": "fp64-valu", "kernel": label, "achieved": tf, "peak": FP64_VALU_PEAK_TFLOPS,
            "unit": "TFLOP/s", "frac": tf / FP64_VALU_PEAK_TFLOPS,
            "fp64_flops_per_launch": flops, "valu_wave_instructions_per_launch": valu,
            "fp64_wave_instructions_per_launch": f64,
            "fp64_share_of_valu": f64 / valu if valu else None,
            "valu_issue_bound_us": issue_s * 1e6, "valu_issue_frac": issue_s / kernel_s,
            "issue_cost_model": "fp64 %.1f / other %.1f cycles per wave instruction per SIMD, %d SIMDs at "
                                "%.1f GHz (profiles/r02/issue_cost_v2.txt)" % (CYC_FP64, CYC_OTHER, N_SIMD, CLOCK_HZ / 1e9),
            "kernel_us": kernel_s * 1e6, "counters_source": pm_src}


def dominant_kernel_roofline(form, pairs, staged, k_us, steps, kern_label, nq, nb, half):
    """`roofline` and `roofline_hbm` of the dominant kernel from the half-step time measured in this run and
    the newest committed counter summaries (profiles/rNN/pmc_valu_cfg2*, pmc_valu_plain*, pmc_traffic*).
    Touches no GPU: tests/test_host_cpu.py calls it for every sampler form.

    `achieved` / `frac` are SURVEY.md 8(d)(ii)'s ALGORITHMIC flops -- 90 flop per quadrature sample of the
    thick+alpha model (3 exp-class operations at 24, one division at 10, four FMAs at 2; fnu.pyx:82-108) x NQ
    samples (response.py:572-576) x the walkers a half-step moves -- over the measured half-step, against the
    fp64 vector peak; they need no committed file and are never null.  `counted` prices the same half-step
    with the fp64 instructions the profiler counted (fewer than the nominal weights: the polynomial tables
    removed the division and two of the three exp-class operations per sample), `executed_*` with everything
    the launch ran, work for outcomes that did not happen included.  A summary that lacks the kernel is said
    in `error`, never passed over."""
    # ---- rooflines of the dominant kernel.  The one-launch run's launches cover different
    # numbers of half-steps, so its counters are taken per half-step (all launches of the
    # profiled run / the half-steps they cover: tools/summarize_valu.py) and priced against
    # the half-step time measured above; a launch of the timed region is 2 K of those.
    kname = kernel_key(form, staged=staged, pairs=pairs if form == 7 else None)
    roof_errors = []
    pm, pm_src, err = measured_valu("pmc_valu_cfg2*.json", kname)
    if err:
        roof_errors.append(err)
    per_launch = 1.0
    if pm and form in ONE_LAUNCH_FORMS and "counters_per_half_step" in pm:
        pm = dict(pm)
        pm["counters_per_launch"] = pm["counters_per_half_step"]
        per_launch = 2.0 * min(steps, 4096)
    elif pm and form in ONE_LAUNCH_FORMS:
        roof_errors.append("%s: the entry of %s has no counters_per_half_step" % (pm_src, kname))
        pm = None
    roof = valu_roofline(pm, pm_src, k_us * 1e-6, kern_label)
    if roof and form in ONE_LAUNCH_FORMS:
        roof["unit_of_counts"] = ("one half-step (125 walkers moved; the quadrature of 250 candidates and the constructor of up to "
                                  "1000 variants run for them)" if form == 7 else
                                  "one half-step (125 walkers moved, 250 proposals prepared ahead)")
        roof["half_steps_per_launch"] = per_launch
        roof["fp64_flops_per_launch"] = roof["fp64_flops_per_launch"] * per_launch
        roof["fp64_flops_per_half_step"] = roof["fp64_flops_per_launch"] / per_launch
        roof["valu_wave_instructions_per_half_step"] = roof.pop("valu_wave_instructions_per_launch")
        roof["fp64_wave_instructions_per_half_step"] = roof.pop("fp64_wave_instructions_per_launch")
        roof["kernel_us"] = k_us * per_launch
        roof["half_step_us"] = k_us
        roof["note"] = (("the counts include the work for outcomes that did not happen -- half of the quadrature, up to three "
                         "quarters of the constructor -- and the instructions spent polling; `useful_fp64_flops_per_half_step` "
                         "is the same chain's count in the form that computes nothing twice (the plain launch train)")
                        if form == 7 else
                        ("the counts include the proposals prepared for the outcome that did not happen (half of the "
                         "constructor work) and the instructions spent polling"))
    if roof and form == 7:
        # `achieved` is algorithmic work over time: the fp64 flops of a half-step in the form that computes
        # nothing twice (the plain launch of 125 walkers, counted in its own profiler pass); what the launch
        # executes, outcomes that did not happen included, is kept beside it
        plain, plain_src, err = measured_valu("pmc_valu_plain*.json", kernel_key(1, staged=staged))
        if err:
            roof_errors.append(err)
        roof["executed_tflops"] = roof["achieved"]
        roof["executed_fp64_flops_per_half_step"] = roof["fp64_flops_per_half_step"]
        if plain:
            useful = plain["fp64_flops_per_launch"]
            roof["useful_fp64_flops_per_half_step"] = useful
            roof["achieved"] = useful / (k_us * 1e-6) / 1e12
            roof["frac"] = roof["achieved"] / FP64_VALU_PEAK_TFLOPS
            roof["useful_counters_source"] = plain_src
            # the same split for the issue bound: `valu_issue_frac` prices every instruction the launch
            # executes (speculative work and polling included); this one only those of the form that
            # computes nothing twice
            pu = valu_roofline(plain, plain_src, k_us * 1e-6, "")
            roof["valu_issue_frac_useful"] = pu["valu_issue_frac"]
            roof["useful_valu_wave_instructions_per_half_step"] = pu["valu_wave_instructions_per_launch"]
        else:
            roof["note"] += "; no summary of the plain launch found: `achieved` is the executed count"
    alg_bytes = 48.0 * half + 16.0 * nq + 16.0 * nb
    tr, traffic_src, err = measured_traffic(kname)
    if err:
        roof_errors.append(err)
    traffic = None
    if tr:
        traffic = tr.get("traffic_bytes_per_half_step" if form in ONE_LAUNCH_FORMS else "traffic_bytes_per_launch")
        if traffic is None:
            roof_errors.append("%s: the entry of %s has no per-half-step traffic" % (traffic_src, kname))
    hbm = {"bound": "hbm", "achieved": alg_bytes / (k_us * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": alg_bytes / (k_us * 1e-6) / 1e9 / HBM_PEAK_GBS, "traffic": traffic,
           "traffic_ratio": (traffic / alg_bytes) if traffic else None, "traffic_source": traffic_src,
           "algorithmic_bytes_per_launch": alg_bytes * per_launch, "algorithmic_bytes_per_half_step": alg_bytes,
           "note": "not the binding roof (SURVEY.md 8d): fp64 transcendental work on 41 KB per half-step.  "
                   + ("Per half-step; the tables are staged once per launch, what crosses the fabric every half-step "
                      "is the hand-over between workgroups (records, rows, the words they poll)" if form in ONE_LAUNCH_FORMS else
                      "The traffic above the algorithmic bytes is the passband table, the polynomial tables and the "
                      "kernel code reaching each of the 8 XCD L2s once per launch")}
    if roof is None:
        roof = {"bound": "fp64-valu", "kernel": kern_label, "achieved": None, "peak": FP64_VALU_PEAK_TFLOPS,
                "unit": "TFLOP/s", "frac": None}
    # the same half-step priced with the survey's nominal weights instead of counted
    # instructions -- per quadrature sample of the thick+alpha model 3 exp-class operations at 24 flop, one
    # division at 10, four FMAs at 2 = 90 flop, x NQ samples x 125 walkers.  The polynomial tables removed
    # every division and two of the three exp-class operations from the sample loop, so the counted figure
    # (`achieved`) is the smaller one.
    nominal = 90.0 * nq * half
    roof["survey_weights"] = {"flop_per_sample": 90.0, "flop_per_half_step": nominal,
                              "achieved": nominal / (k_us * 1e-6) / 1e12,
                              "frac": nominal / (k_us * 1e-6) / 1e12 / FP64_VALU_PEAK_TFLOPS}
    roof["traffic"] = traffic
    if roof_errors:
        # never a silent null: say which committed summary lacks which kernel
        roof["error"] = "; ".join(roof_errors)
    # the headline pair is the algorithmic one (see the docstring); the profiler-counted figure stays beside it
    roof["counted"] = {"achieved": roof.get("achieved"), "frac": roof.get("frac"), "unit": "TFLOP/s",
                       "what": "fp64 flops the profiler counted for a half-step of the form that computes nothing "
                               "twice (useful_fp64_flops_per_half_step) over the same time" if form == 7 else
                               "fp64 flops the profiler counted for this launch over the same time"}
    roof["achieved"] = roof["survey_weights"]["achieved"]
    roof["frac"] = roof["survey_weights"]["frac"]
    roof["algorithmic_flops_per_half_step"] = nominal
    roof["definition"] = "achieved = 90 flop x NQ x walkers moved per half-step / measured half-step (SURVEY.md 8d ii)"
    return roof, hbm


def user_runs(like, pos, bench_steps=None):
    import mbb_emcee_amd as mbb
    from tools.bench_configs import CFG1_WAVE
    ctx = like._sync_device()
    out = {}
    # M2 with the chain stored: enqueue + D2H + re-ordering on the host, wall clock
    smp = mbb.DeviceEnsembleSampler(NW_PER_GPU, 5, like, seed=11)
    smp.run_mcmc(pos, 60, storechain=False)
    stored = {}
    lengths = [250, 2000] + ([int(bench_steps)] if bench_steps and int(bench_steps) not in (250, 2000) else [])
    for k in lengths:
        cur, lnp = smp.run_mcmc(None, k)[:2]                   # (warm: the chain buffer of this length exists)
        ts = []
        for _ in range(5):
            smp.reset()
            t0 = time.perf_counter(); cur, lnp = smp.run_mcmc(cur, k, lnprob0=lnp)[:2]; ts.append(time.perf_counter() - t0)
        t = float(np.median(ts))
        ctx.sync()
        e0, e1 = ctx.event(), ctx.event()
        ctx.record(e0); smp.advance_async(k); ctx.record(e1); ctx.sync()
        stored["steps_%d" % k] = {"wall_us_per_step": t * 1e6 / k, "evals_per_s": NW_PER_GPU * k / t,
                                  "unstored_stream_us_per_step": ctx.elapsed_ms(e0, e1) * 1e3 / k,
                                  "chain_bytes": NW_PER_GPU * k * 48}
    stored["note"] = ("DeviceEnsembleSampler.run_mcmc(pos, K, lnprob0=lnprob) with storechain=True, as mbb_fitter.run calls it: "
                      "state to the device, launch, 48 B per walker per "
                      "step back over PCIe, re-ordering into chain[walker, step, 5] + lnprobability[walker, step]; "
                      "median of 5 by the wall clock")
    out["sampler_M2_stored_chain"] = stored
    del smp
    # whole fits
    fits = {}
    for cfg in ("cfg1", "cfg2"):
        for sampler in ("device", "native"):
            if cfg == "cfg1":
                fit = mbb.mbb_fitter(nwalkers=50, opthin=True, seed=3, sampler=sampler)
                one = mbb.likelihood(opthin=True)
                one.set_phot(CFG1_WAVE, np.ones(5), np.ones(5))
                f = one.model_flux(TRUTH)[0]
                fit.set_data(CFG1_WAVE, f, 0.1 * f + 1.0)
            else:
                fit = mbb.mbb_fitter(nwalkers=NW_PER_GPU, response=True, seed=3, sampler=sampler)
                fit.set_data(BANDS, like._flux, like._flux_unc)
            p0 = fit.generate_initial_values(np.array([10.0, 2.0, 600.0, 4.0, 40.0]), np.array([2.0, 0.2, 100.0, 0.3, 5.0]))
            t0 = time.perf_counter(); fit.run(50, 250, p0); first = time.perf_counter() - t0
            ts = []
            for _ in range(3):
                t0 = time.perf_counter(); fit.run(50, 250, p0); ts.append(time.perf_counter() - t0)
            ch = fit.sampler.chain
            fits["%s_%s" % (cfg, sampler)] = {"fit_wall_s": float(np.median(ts)), "first_fit_wall_s": first,
                                              "chain_shape": list(ch.shape),
                                              "acceptance_fraction": float(np.mean(fit.sampler.acceptance_fraction)),
                                              "median_T": float(np.median(ch[:, :, 0]))}
    fits["note"] = ("mbb_fitter(...).run(nburn=50, nsteps=250, p0) end to end: validation, burn-in, reset, main chain, chain "
                    "on the host; first = including the context, the tables' upload and the first launches.  cfg1: 50 "
                    "walkers, 5 delta bands, thin; cfg2: 250 walkers, 8 passbands, thick+alpha.  `device` is the default "
                    "sampler of mbb_fitter, `native` the host stretch move over likelihood.__call__ (one launch per "
                    "half-step: the boundary an external sampler sees)")
    out["fit"] = fits
    return out


def config_roofline(name, plain_us, half_step_us, form, half, ctx):
    """fp64 roofline objects of a config's plain launch and of its sampler half-step, from the committed
    counter passes of `tools/bench_configs.py <name> --profile` (profiles/rNN/pmc_valu_<name>.json)."""
    f = newest_profile("pmc_valu_%s.json" % name)
    if not f:
        return {"bound": "fp64-valu", "achieved": None, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": None,
                "note": "no committed PMC summary for this config"}
    d = json.load(open(f))
    src = os.path.relpath(f, ROOT)
    plain = next((v for k, v in d["kernels"].items() if "k_lnlike<" in k and ", 0, " in k), None)
    roof = valu_roofline(plain, src, plain_us * 1e-6,
                         "k_lnlike<plain> n=%d (%d workgroups x %d threads)" % (half, ctx.info("last_grid"), ctx.info("last_threads"))) \
        if plain else {"bound": "fp64-valu", "achieved": None, "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": None}
    smp = next((v for k, v in d["kernels"].items() if any(n in k for n in ("k_flowm<", "k_flowa<")) and "counters_per_half_step" in v), None)
    if smp and form in ONE_LAUNCH_FORMS:
        sm = dict(smp)
        sm["counters_per_launch"] = smp["counters_per_half_step"]
        r2 = valu_roofline(sm, src, half_step_us * 1e-6, "sampler form %d, per half-step" % form)
        roof["sampler_half_step"] = {"executed_tflops": r2["achieved"], "executed_frac": r2["frac"],
                                     "valu_issue_frac": r2["valu_issue_frac"], "half_step_us": half_step_us,
                                     "valu_wave_instructions_per_half_step": r2["valu_wave_instructions_per_launch"]}
        if plain:
            # algorithmic work of a half-step = the plain launch's count (it computes nothing twice)
            useful = plain["fp64_flops_per_launch"]
            roof["sampler_half_step"]["achieved_tflops"] = useful / (half_step_us * 1e-6) / 1e12
            roof["sampler_half_step"]["frac"] = roof["sampler_half_step"]["achieved_tflops"] / FP64_VALU_PEAK_TFLOPS
    return roof


_POOL_WORKER = r"""
import json, os, pickle, sys, time
import numpy as np
sys.path.insert(0, sys.argv[1])
d, rank, world, ncalls, serve = sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6])
import mbb_emcee_amd  # noqa
like = pickle.load(open(os.path.join(d, "like.pkl"), "rb"))
pars = np.load(os.path.join(d, "pars.npy"))
ctx = like._sync_device()
ctx.set_option("serve", 0)
first = like(pars)
ctx.set_option("serve", serve)
open(os.path.join(d, "ready.%d.%d" % (serve, rank)), "w").close()
t0 = time.time()
while not all(os.path.exists(os.path.join(d, "ready.%d.%d" % (serve, r))) for r in range(world)):
    if time.time() - t0 > 60:
        raise SystemExit("the other worker never came")
    time.sleep(0.0005)
ts = np.empty(ncalls)
ok = True
for i in range(ncalls):
    a = time.perf_counter(); r = like(pars); ts[i] = time.perf_counter() - a
    if i % 97 == 0:
        ok = ok and bool(np.array_equal(r, first, equal_nan=True))
json.dump({"p50_us": float(np.median(ts) * 1e6), "p99_us": float(np.percentile(ts, 99) * 1e6), "max_us": float(ts.max() * 1e6),
           "served": int(ctx.info("serve_requests")), "fell_back": int(ctx.info("serve_fallbacks")), "ok": ok,
           "server_workgroups": int(ctx.info("serve_grid"))}, open(os.path.join(d, "out.%d.%d.json" % (serve, rank)), "w"))
"""


def pool_leg(like, pos, world=2, ncalls=1500):
    """emcee's pool (reference mbb_fit.py:80-81 with threads > 1): the likelihood pickled into `world` worker PROCESSES, each in
    a loop of boundary calls of NW/2 rows at the same time on the one GPU; as the library does it (every worker's own
    resident kernel, as wide as its calls have rows, side by side) and with a launch per call."""
    import pickle, subprocess, tempfile, shutil
    d = tempfile.mkdtemp(prefix="mbb_pool_")
    try:
        pickle.dump(like, open(os.path.join(d, "like.pkl"), "wb"))
        np.save(os.path.join(d, "pars.npy"), np.ascontiguousarray(pos[:NW_PER_GPU // 2]))
        out = {"workers": world, "rows": NW_PER_GPU // 2, "calls": ncalls}
        for name, serve in (("served", 1), ("launch_per_call", 0)):
            procs = [subprocess.Popen([sys.executable, "-c", _POOL_WORKER, ROOT, d, str(r), str(world), str(ncalls), str(serve)],
                                      stdout=subprocess.DEVNULL, stderr=subprocess.PIPE) for r in range(world)]
            errs = []
            for pr in procs:
                try:
                    e = pr.communicate(timeout=90)[1]
                    if pr.returncode:
                        errs.append(e.decode(errors="replace")[-300:])
                except subprocess.TimeoutExpired:
                    pr.kill(); errs.append("timeout")
            if errs:
                out[name] = {"error": errs[0]}
                continue
            reps = [json.load(open(os.path.join(d, "out.%d.%d.json" % (serve, r)))) for r in range(world)]
            out[name] = {"p50_us": [r["p50_us"] for r in reps], "p99_us": [r["p99_us"] for r in reps],
                         "served_requests": [r["served"] for r in reps], "fell_back": [r["fell_back"] for r in reps],
                         "server_workgroups": [r["server_workgroups"] for r in reps], "results_right": all(r["ok"] for r in reps)}
        return out
    finally:
        shutil.rmtree(d, ignore_errors=True)


def postprocess_leg(like, pos, cpu=True):
    """SURVEY.md 8f rank 4 measured: the chain post-processing of mbb_results (results.py:570-581 peak wavelength,
    :627-674 L_IR, :746-801 dust mass, :895-944 predicted fluxes) over a stored 250 x 250 chain of the bench
    workload, every entry one row of a batched kernel call -- wall clock of the user-level calls (host arrays in
    and out), median of 5 -- with the CPU oracle's restatement of the same functions timed beside it on a bounded
    sample (one thread: the reference does these one chain entry at a time in Python) and compared to it."""
    import mbb_emcee_amd as mbb
    from mbb_emcee_amd import postprocess as pp
    nsteps = 250
    smp = mbb.DeviceEnsembleSampler(NW_PER_GPU, 5, like, seed=21)
    cur, lnp = smp.run_mcmc(pos, 50, storechain=False)[:2]
    smp.reset()
    smp.run_mcmc(cur, nsteps, lnprob0=lnp)
    chain = np.ascontiguousarray(smp.chain)                 # [walker, step, 5], emcee's layout
    del smp
    rows = chain.shape[0] * chain.shape[1]
    z, dl = 2.3, 18700.0
    bands2 = ["SPIRE_250um", "SCUBA2_450um"]                # one fitted band, one that is not
    calls = {"peak_wavelength": lambda: pp.peak_wavelength(like, chain),
             "lir": lambda: pp.lir(like, chain, z, dl),
             "predict_flux_2_bands": lambda: pp.predict_flux(like, chain, bands2),
             "predict_flux_2_wavelengths": lambda: pp.predict_flux(like, chain, [70.0, 1100.0]),
             "dustmass_host": lambda: pp.dustmass(like, chain, z, dl)}
    out = {"chain": [int(x) for x in chain.shape], "rows": rows}
    got = {}
    for name, fn in calls.items():
        got[name] = fn()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
        t = float(np.median(ts))
        out[name] = {"wall_us": t * 1e6, "rows_per_s": rows / t}
    out["note"] = ("wall clock of the user-level call (chain on the host in, result on the host out: PCIe and the Python "
                   "around the kernel included); the reference computes each of these one chain entry at a time")
    if cpu:
        from oracle import oracle as O
        flat = chain.reshape(-1, 5)
        resp = like._responsewheel
        cb = {"cores": 1, "kind": "port"}

        def timed(fn, sample):
            t0 = time.perf_counter(); r = fn(sample); dt = time.perf_counter() - t0
            return r, dt
        # bounded samples: about 2-5 s of CPU each
        sel = flat[:: max(1, rows // 2000)][:2000]
        ref, dt = timed(lambda c: O.post_peaklambda(c, like.opthin, like.noalpha, as_reference=False), sel)
        g = got["peak_wavelength"].reshape(-1)[:: max(1, rows // 2000)][:2000]
        cb["peak_wavelength"] = {"rows_per_s": len(sel) / dt, "sample_rows": len(sel),
                                 "max_rel_err_gpu_vs_oracle": float(np.max(np.abs(g - ref) / ref))}
        assert cb["peak_wavelength"]["max_rel_err_gpu_vs_oracle"] < 1e-10
        sel = flat[:: max(1, rows // 300)][:300]
        ref, dt = timed(lambda c: O.post_lir(c, z, dl, like.opthin, like.noalpha), sel)
        g = got["lir"].reshape(-1)[:: max(1, rows // 300)][:300]
        cb["lir"] = {"rows_per_s": len(sel) / dt, "sample_rows": len(sel),
                     "max_rel_err_gpu_vs_oracle": float(np.max(np.abs(g - ref) / ref)),
                     "note": "the oracle integrates with scipy's quad as the reference does: its own error is up to 4e-7 "
                             "(tests/test_gpu_parity.py::test_postprocess_vs_reference_results)"}
        assert cb["lir"]["max_rel_err_gpu_vs_oracle"] < 1e-6
        sel = flat[:: max(1, rows // 2000)][:2000]
        r = resp[bands2[0]]
        ref, dt = timed(lambda c: O.post_predict_flux(c, (r.wavelength, r._sedmult, r._normfac), like.opthin, like.noalpha), sel)
        g = got["predict_flux_2_bands"][..., 0].reshape(-1)[:: max(1, rows // 2000)][:2000]
        cb["predict_flux_per_band"] = {"rows_per_s": len(sel) / dt, "sample_rows": len(sel),
                                       "max_rel_err_gpu_vs_oracle": float(np.max(np.abs(g - ref) / np.abs(ref)))}
        assert cb["predict_flux_per_band"]["max_rel_err_gpu_vs_oracle"] < 1e-12
        out["cpu_baseline"] = cb
    return out


def ensemble_crc(pos, lnp):
    import zlib
    return zlib.crc32(np.ascontiguousarray(pos).tobytes()) ^ zlib.crc32(np.ascontiguousarray(lnp).tobytes())


LINE_LIMIT = 4096           # bytes of the ONE line on stdout (round 4's 20.7 KB line was not parsed by the driver)
FULL_PATH = os.environ.get("MBB_BENCH_FULL") or os.path.join(ROOT, "gpurun_out", "bench_full.json")


def _num(x, sig=6):
    """A number as it goes on the short line: `sig` significant digits, never NaN/Infinity (strict JSON)."""
    if isinstance(x, (bool, np.bool_)):
        return bool(x)
    if isinstance(x, (int, np.integer)):
        return int(x)
    if isinstance(x, (float, np.floating)):
        x = float(x)
        if not np.isfinite(x):
            return None
        return float("%.*g" % (sig, x))
    if isinstance(x, str):
        return _txt(x, 48)          # (units, kinds, bounds: short words; the long texts are cut where they are picked)
    if x is None:
        return None
    return _txt(x, 48)


def _txt(s, n):
    s = str(s)
    return s if len(s) <= n else s[:n - 3] + "..."


def _pick(d, keys, sig=6):
    """{k: d[k]} for the keys `d` has, numbers rounded."""
    return {k: _num(d[k], sig) for k in keys if isinstance(d, dict) and k in d}


def short_line(full):
    """The ONE line the driver parses, from everything that was measured (`full`): the contract's keys, `roofline`,
    `cpu_baseline`, the boundary figure M1 and the path of the side file that holds all the rest -- nothing else,
    strings cut, numbers rounded to six digits, strict JSON (no NaN), below LINE_LIMIT bytes whatever `full` holds
    (tests/test_host_cpu.py builds it from a fully populated line of 1 and of 8 ranks)."""
    out = {}
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data"):
        out[k] = _num(full.get(k), 7)
    out["metric"] = _txt(full.get("metric", ""), 120)           # (BASELINE.json's metric, whole)
    cfg = full.get("config") or {}
    c = _pick(cfg, ("walkers_per_gpu", "walkers", "bands", "nq"))
    c = dict({"workload": _txt(cfg.get("workload_short") or cfg.get("workload", ""), 200)}, **c)
    for k, n in (("sampler_form", 140), ("preconditioning", 140), ("collective", 60), ("rehearsal", 90), ("note", 120)):
        if (cfg.get(k + "_short") or cfg.get(k)) and cfg.get(k) != "none":
            c[k] = _txt(cfg.get(k + "_short") or cfg[k], n)
    if cfg.get("collective_fallback_from"):
        c["collective_fallback_from"] = [_txt(t, 80) for t in cfg["collective_fallback_from"][:3]]
    out["config"] = c
    for k in ("value_median_of_9", "us_per_step_median_of_9", "stored_chain_us_per_step", "stored_chain_us_per_step_at_K",
              "half_step_us", "kernel_avg_us", "acceptance_fraction"):
        if k in full:
            out[k] = _num(full[k])
    if isinstance(full.get("us_per_step_min_max_of_9"), (list, tuple)):
        out["us_per_step_min_max_of_9"] = [_num(float(v), 5) for v in full["us_per_step_min_max_of_9"][:2]]
    r = full.get("roofline")
    if isinstance(r, dict):
        o = _pick(r, ("bound",))
        o["kernel"] = _txt(r.get("kernel_short") or r.get("kernel", ""), 80)
        o.update(_pick(r, ("achieved", "peak", "unit", "frac")))
        if isinstance(r.get("counted"), dict):
            o["counted_frac"] = _num(r["counted"].get("frac"))
        o.update(_pick(r, ("valu_issue_frac", "traffic", "launch_slot_us")))
        if r.get("counters_source"):
            o["counters_source"] = _txt(r["counters_source"], 60)
        if r.get("definition"):
            o["definition"] = "90 flop x NQ x walkers / half_step_us (SURVEY 8d ii)"
        if r.get("error"):
            o["error"] = _txt(r["error"], 160)
        if r.get("note") and o.get("frac") is None:
            o["note"] = _txt(r["note"], 100)
        out["roofline"] = o
    h = full.get("roofline_hbm")
    if isinstance(h, dict):
        out["roofline_hbm"] = _pick(h, ("achieved", "peak", "unit", "frac", "traffic_ratio"))
    b = full.get("boundary_M1")
    if isinstance(b, dict):
        o = _pick(b, ("rows", "p50_us", "p90_us", "p99_us", "max_us", "evals_per_s", "calls", "slow_calls", "slow_runs",
                      "serve_requests", "serve_fallbacks", "serve_lease_yields", "serve_resizes", "serve_rests",
                      "launch_per_call_p50_us", "two_processes_p50_us"), 5)
        if isinstance(b.get("last_leg"), dict):
            o["last_leg"] = _pick(b["last_leg"], ("median_us", "p90_us", "p99_us", "max_us", "slow_calls", "slow_runs",
                                                  "serve_fallbacks", "serve_lease_yields", "serve_resizes", "serve_rests"), 5)
        out["boundary_M1"] = o
    cb = full.get("cpu_baseline")
    if isinstance(cb, dict):
        o = _pick(cb, ("value", "unit", "cores", "kind", "single_thread_value"))
        o["sample"] = _txt(cb.get("sample_short") or cb.get("sample", ""), 80)
        out["cpu_baseline"] = o
    if "parity_max_err_vs_oracle" in full:
        out["parity_max_err_vs_oracle"] = _num(full["parity_max_err_vs_oracle"], 3)
    if isinstance(full.get("cfg5"), dict):          # BASELINE.json configs[4] asks for its roofline fraction
        o = _pick(full["cfg5"], ("kernel_ms", "evals_per_s", "algorithmic_frac"))
        if isinstance(full["cfg5"].get("roofline"), dict):
            o["counted_frac"] = _num(full["cfg5"]["roofline"].get("frac"))
        out["cfg5"] = o
    # N > 1: one word per exchange (what the outcomes mean: DESIGN.md section 6), the sharded boundary's medians
    ev = full.get("exchange_validation")
    if isinstance(ev, dict):
        out["exchange_validation"] = {_txt(m, 16): ("ok" if v.get("ok") else _txt(v.get("why", "failed"), 70)) if isinstance(v, dict) else _txt(v, 70)
                                      for m, v in list(ev.items())[:4]}
    sb = full.get("boundary_sharded")
    if isinstance(sb, dict):
        o = {"ok": _num(sb.get("ok"))}
        if not sb.get("ok"):
            o["why"] = _txt(sb.get("why", ""), 90)
        o.update({_txt(k, 12): _num(v.get("median_us"), 5) for k, v in list(sb.items())[:8] if k.startswith("rows_") and isinstance(v, dict)})
        out["boundary_sharded"] = o
    for k in ("ranks_agree", "valid_for_scaling", "supervisor_timeout", "supervisor_deadline_s", "collective_hung"):
        if k in full:
            out[k] = _num(full[k])
    if "ranks_ended_badly" in full:
        out["ranks_ended_badly"] = {_txt(k, 8): _num(v) for k, v in list(full["ranks_ended_badly"].items())[:8]}
    for k, n in (("error", 300), ("extras_error", 160), ("invalid", 120), ("collective", 160), ("hang", 100)):
        if full.get(k):
            out[k] = _txt(full[k], n)
    out["full"] = _txt(os.path.relpath(FULL_PATH, ROOT), 100)
    return out


def emit(obj):
    """Rank-to-supervisor traffic (a pipe, never the driver's stdout): anything goes."""
    print(json.dumps(obj), flush=True)


def emit_final(full):
    """What the driver reads: everything measured goes to the side file gpurun_out/bench_full.json, the short
    line -- strict JSON, one line, below LINE_LIMIT bytes -- is the last thing on stdout."""
    try:
        os.makedirs(os.path.dirname(FULL_PATH), exist_ok=True)
        with open(FULL_PATH + ".tmp", "w") as f:
            json.dump(full, f, indent=1, default=str)
        os.replace(FULL_PATH + ".tmp", FULL_PATH)
    except OSError as e:
        sys.stderr.write("bench.py: cannot write %s: %r\n" % (FULL_PATH, e))
    if os.environ.get("MBB_BENCH_FULL_LINE"):
        # (the profile scripts' summarizers read legs of the line that are in the side file only: tools/run_profiles.sh)
        sys.stdout.flush()
        print(json.dumps(full, default=str), flush=True)
        return
    line = json.dumps(short_line(full), allow_nan=False, separators=(",", ":"))
    if len(line.encode()) >= LINE_LIMIT:      # (cannot happen: every field above is bounded; never print a long line)
        keep = short_line({k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step",
                                                    "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "error")})
        keep["line_cut"] = True
        line = json.dumps(keep, allow_nan=False, separators=(",", ":"))
    sys.stdout.flush()
    print(line, flush=True)


def emit_line(full):
    """The main line: to the supervisor whole (it prints the short one), to the driver short."""
    if os.environ.get("MBB_BENCH_WORKER") == "1":
        emit(full)
    else:
        emit_final(full)


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--no-cfg5", action="store_true", help="skip the 250 000-walker launch")
    ap.add_argument("--no-configs", action="store_true", help="skip the per-config objects (cfg1, cfg4)")
    ap.add_argument("--no-fit", action="store_true", help="skip the end-to-end fit timings")
    ap.add_argument("--no-extras", action="store_true", help="timed region only (profiler passes)")
    ap.add_argument("--exchange", choices=("auto", "rccl", "ipc"), default="auto",
                    help="N > 1: how the moved state rows travel after each launch (auto: the one-hop "
                         "peer-write exchange, RCCL all-gather if that cannot be brought up)")
    ap.add_argument("--exchange-order", choices=("fastest", "simplest"), default="fastest",
                    help="N > 1, --exchange auto: which exchange is tried first for the timed run -- the one "
                         "expected to be fastest (one launch per run across the ranks, then one launch per "
                         "half-step with peer writes, then RCCL) or the simplest (the reverse).  Whichever is "
                         "timed, the others are rehearsed afterwards and reported in `exchange_validation`")
    ap.add_argument("--no-validate", action="store_true",
                    help="N > 1: do not rehearse the exchanges the timed run did not use")
    ap.add_argument("--no-sharded-boundary", action="store_true",
                    help="N > 1: do not time likelihood.__call__ sharded over the ranks with one ncclAllGather per call")
    ap.add_argument("--oversubscribe", action="store_true",
                    help="allow more ranks than devices (rehearsal of the N > 1 path on one GPU; "
                         "needs --exchange ipc; the line is marked invalid for scaling)")
    return ap.parse_args(argv)


def base_line(args, world):
    """The fields of the line that do not depend on anything measured (no GPU touched)."""
    try:
        metric = json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:
        metric = "walker-likelihood evals/sec (+ MCMC steps/sec), 250 walkers x 8 bands"
    return {"metric": metric, "value": None, "unit": "evals/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "cfg2/cfg3: 8-band PACS+SPIRE+SCUBA2_850+Bolocam passband integration "
                                   "(NQ=2209), thick+alpha, one ensemble of 250 walkers per GPU advanced by "
                                   "the device-resident stretch move, two dependent half-steps per step",
                       "workload_short": "cfg2/cfg3: 8 passbands (NQ 2209) with full response integration, thick+alpha, 250 walkers "
                                         "per GPU, device-resident stretch move; 1 step = 2 dependent half-steps",
                       "walkers_per_gpu": NW_PER_GPU, "walkers": NW_PER_GPU * world, "bands": 8,
                       "nq": 2209, "half_steps_per_step": 2}}


# ---------------------------------------------------------------------------------------
# The supervisor.  `python bench.py --gpus N` with N > 1 is started by the driver either
# bare or under torch.distributed.run.  Either way the process the driver started never
# touches the GPU: it starts the rank(s) as fresh child processes (the role of emcee's
# pool, mbb_fit.py:80-81 / run_mbb_emcee.py:166-167), collects what rank 0 prints, and
# prints ONE line.  A rank that dies while the exchanges the timed run did not use are
# being rehearsed therefore cannot take the measured line with it.
# ---------------------------------------------------------------------------------------
def rendezvous_file():
    """Where the ranks this supervisor starts meet: a FILE (torch's FileStore) in a fresh directory.  Nobody picks a
    port -- round 5's harness picked one, closed it and handed the number on; it was taken when the listener came
    (GPUTEST_r05: EADDRINUSE) -- gloo's own listeners bind port 0 and publish what they got through the file."""
    import tempfile
    return os.path.join(tempfile.mkdtemp(prefix="mbb_bench_rdzv_"), "store")


def supervise(args):
    import subprocess
    launched = "WORLD_SIZE" in os.environ            # under torch.distributed.run: one rank per supervisor
    world = int(os.environ.get("WORLD_SIZE", args.gpus))
    my_rank = int(os.environ.get("RANK", "0"))
    base = base_line(args, world)
    if args.gpus != world:
        if my_rank == 0:
            emit_final(dict(base, error="--gpus %d but WORLD_SIZE=%d" % (args.gpus, world)))
        return 2
    if world < 1 or world > 64:
        emit_final(dict(base, error="--gpus %d: not a rank count" % args.gpus))
        return 2
    ranks = [my_rank] if launched else list(range(world))
    # The whole run -- ranks started, exchanges tried, line printed -- has to fit the driver's limit for one
    # bench command (600 s in round 3's record): 480 s here, every step inside a rank bounded well below that
    # (SUPERVISOR_DEADLINE_S, GUARD_S), and when the deadline comes the ranks are ended and the line is printed
    # from whatever rank 0 has handed over by then: the measured value if there is one, an error otherwise.
    t_start = time.time()
    budget = float(os.environ.get("MBB_BENCH_DEADLINE_S", SUPERVISOR_DEADLINE_S))
    # under torch.distributed.run the launcher's own store is there (MASTER_PORT names a port that IS listening: the
    # ranks are its clients); started bare, the ranks meet through a file
    rdzv = None if (launched and os.environ.get("MASTER_PORT")) else rendezvous_file()
    cmd = [sys.executable, os.path.abspath(__file__)] + sys.argv[1:]
    procs, lines = {}, []
    for r in ranks:
        env = dict(os.environ, MBB_BENCH_WORKER="1", WORLD_SIZE=str(world), RANK=str(r),
                   MASTER_ADDR=os.environ.get("MASTER_ADDR", "127.0.0.1"),
                   MBB_BENCH_T0=repr(t_start), MBB_BENCH_DEADLINE_S=repr(budget))
        if not launched:
            env["LOCAL_RANK"] = str(r)
        if rdzv:
            env["MBB_BENCH_RDZV_FILE"] = rdzv
            env.setdefault("GLOO_SOCKET_IFNAME", "lo")            # one node: the loopback interface, no hostname look-up
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        try:
            # rank 0's stdout is collected; the other ranks have nothing to say there
            procs[r] = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE if r == 0 else sys.stderr.fileno())
        except OSError as e:
            for pr in procs.values():
                pr.kill()
            emit_final(dict(base, error="cannot start rank %d: %r" % (r, e)))
            return 2

    def collect(pipe):
        for raw in iter(pipe.readline, b""):
            lines.append(raw.decode("utf-8", "replace").strip())
    th = None
    if 0 in procs:
        th = threading.Thread(target=collect, args=(procs[0].stdout,), daemon=True)
        th.start()
    deadline = t_start + budget
    timed_out, orphaned_since = False, None
    while any(pr.poll() is None for pr in procs.values()):
        now = time.time()
        if now > deadline:
            timed_out = True
            break
        # a rank that has ended -- badly, or after a wedged exchange made it leave at once -- leaves the
        # others waiting in a rendezvous or an exchange: they get half a minute to notice by themselves
        # (bounded polls, gloo's own errors) or to finish their own tear-down, then they are ended
        if any(pr.poll() is not None for pr in procs.values()):
            orphaned_since = orphaned_since or now
            if now - orphaned_since > float(os.environ.get("MBB_BENCH_GRACE_S", "30")):
                break
        time.sleep(0.05)
    ended_here = set()
    for r, pr in procs.items():
        if pr.poll() is None:
            pr.kill()                      # exactly the processes started above
            ended_here.add(r)
    for pr in procs.values():
        pr.wait()
    if th is not None:
        th.join(timeout=10.0)
    rcs = {r: pr.returncode for r, pr in procs.items()}
    worst = 0
    for rc in rcs.values():
        if rc != 0:
            worst = max(worst, 128 - rc if rc < 0 else rc)
    if my_rank != 0:
        return worst
    main_line, parts = None, {}
    for ln in lines:
        try:
            obj = json.loads(ln)
        except ValueError:
            continue
        if not isinstance(obj, dict):
            continue
        if "_part" in obj:
            parts[obj["_part"]] = obj.get("data")
        elif "metric" in obj and main_line is None:
            main_line = obj
    if main_line is None:
        main_line = dict(base, error=("the ranks were ended at the supervisor's deadline of %.0f s before rank 0 had a line"
                                      % budget) if timed_out else
                         "rank 0 ended with status %s without a line" % rcs.get(0))
        worst = worst or 1
    main_line.update(parts)
    bad = {str(r): rc for r, rc in rcs.items() if rc != 0}
    if bad and main_line.get("value") is not None:
        # the measured line stands; what died afterwards (while the other exchanges were being rehearsed)
        # is said beside it, and does not turn a valid measurement into a failed run when rank 0 itself
        # left in good order
        main_line["ranks_ended_badly"] = bad
        if rcs.get(0) == 0 or 0 in ended_here:
            # (rank 0 ended by the supervisor itself: it was wedged in something optional after the line)
            worst = 0
    if timed_out:
        main_line["supervisor_timeout"] = True
        main_line["supervisor_deadline_s"] = budget
        if main_line.get("value") is None and "error" not in main_line:
            main_line["error"] = "the ranks were ended at the supervisor's deadline of %.0f s before a run was validated" % budget
    emit_final(main_line)
    return worst


def main():
    args = parse_args()
    hook = os.environ.get("MBB_BENCH_RANK_HOOK")
    if hook and (os.environ.get("MBB_BENCH_WORKER") == "1" or (args.gpus == 1 and os.environ.get("MBB_BENCH_WORKER_FAKE_TOP"))):
        # tests only (tests/_fake_bench_rank.py): a stand-in for a rank, so that the supervisor's collecting, merging,
        # waiting and ending of ranks and the printing of the line can be tested without a GPU
        import importlib.util
        spec = importlib.util.spec_from_file_location("_bench_rank_hook", hook)
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        sys.exit(mod.run(sys.modules[__name__], args))
    if os.environ.get("MBB_BENCH_WORKER") == "1" or (args.gpus == 1 and "WORLD_SIZE" not in os.environ):
        return worker(args)
    sys.exit(supervise(args))


class Watchdog(Exception):
    pass


class Hung(Exception):
    """A step that may wedge (a collective, a launch that waits for peers) did not return within its guard."""


def guarded(fn, seconds):
    """Run fn on a thread and give it `seconds`; a stream that never drains must not hang the bench."""
    box = {}

    def run():
        try:
            box["ret"] = fn()
        except BaseException as e:          # noqa
            box["err"] = e
    th = threading.Thread(target=run, daemon=True)
    th.start()
    th.join(timeout=seconds)
    if th.is_alive():
        raise Watchdog()
    if "err" in box:
        raise box["err"]
    return box.get("ret")


EXCHANGE_TEXT = {
    "none": "none",
    "ipc": "one-hop peer writes, one launch per run: every decision, moved row and progress word is "
           "stored into every rank's copy of the run's state as it is made (hipIpc mappings, system "
           "scope) and a row's half-step starts when the rows it depends on are done, on whatever "
           "GPU; no collective library",
    "ipc-launches": "one-hop peer writes: the lane that accepts a move stores the state row (6 f64) "
                    "into every rank's copy of the ensemble (hipIpc mappings, system scope) and the "
                    "launch's last walker raises a flag in every peer; no collective library",
    "rccl": "in-place ncclAllGather of %d state rows x 6 f64 per launch (RCCL via C-ABI)" % (NW_PER_GPU // 2)}


def worker(args):
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    base = base_line(args, world)

    def fail(code, **kw):
        """A failed run still prints its line (rank 0) and leaves with a non-zero status."""
        if rank == 0:
            out = dict(base)
            out.update(kw)
            emit_line(out)
        sys.stdout.flush()
        os._exit(code)          # a wedged stream would also hang interpreter teardown

    try:
        return worker_body(args, rank, world, local_rank, base, fail)
    except SystemExit:
        raise
    except BaseException as e:              # noqa -- whatever it was, the driver gets a line
        import traceback
        traceback.print_exc()
        fail(1, error="%s: %s" % (type(e).__name__, e))


def worker_body(args, rank, world, local_rank, base, fail):
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # gloo announces its connections on stdout; the line printed at the end must be the
        # only thing there
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            import datetime
            # (a collective of the side channel that a lost peer never joins ends by itself well inside the
            # supervisor's deadline)
            rdzv = os.environ.get("MBB_BENCH_RDZV_FILE")
            dist.init_process_group(backend="gloo", rank=rank, world_size=world,
                                    timeout=datetime.timedelta(seconds=120),
                                    **({"init_method": "file://" + rdzv} if rdzv else {}))
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)

    def barrier():
        if dist is not None:
            dist.barrier()

    from mbb_emcee_amd import _native
    ndev = _native.load().mbb_device_count()
    if ndev <= 0:
        fail(2, error="no ROCm device on this box: the likelihood path has no CPU fall-back",
             collective="unavailable: no device")
    # (ranks that share a device -- a rehearsal on one GPU -- can go through the collective library too when MBB_RCCL_LIB names
    # one that lets them: tests/rccl_standin/, which is how the rccl exchange and the sharded boundary get rehearsed at all)
    rccl_named = bool(os.environ.get("MBB_RCCL_LIB"))
    if world > ndev and not (args.oversubscribe and (args.exchange in ("ipc", "auto") or rccl_named)):
        fail(2, error="%d ranks on %d device(s): RCCL needs one device per rank" % (world, ndev),
             collective="unavailable: %d ranks on %d device(s)" % (world, ndev), valid_for_scaling=False)
    like, flux = make_likelihood(local_rank % ndev)     # one GPU per rank on a real node
    ctx = like._sync_device()
    if os.environ.get("MBB_BENCH_PLAIN_TRAIN"):
        # profiler passes only (tools/run_profiles.sh): the same chain as a launch per half-step, the
        # form that computes nothing twice -- its counters are the algorithmic work of a half-step
        ctx.set_option("lookahead_sampler", 0)
    nq, nb = ctx.info("nq"), ctx.info("nb")
    base["config"]["nq"] = nq
    half = NW_PER_GPU // 2

    # ---- N > 1: the exchange of the moved state rows.  Every exchange is trusted only after a
    # guarded rehearsal of exactly what the timed loop does (60 steps, then every rank's copy of
    # the ensemble and the unsharded sampler's state must have the same CRC).  The first exchange of
    # the order that passes carries the timed run; the others are rehearsed and timed afterwards,
    # outside `value`, so that one run on a multi-GPU node says which of the three protocols survive
    # xGMI (`exchange_validation`).  A collective that never returns is a failure of the run, not
    # something to time around: the line says so and the exit status is non-zero.
    import mbb_emcee_amd as mbb
    allw = walkers(world)
    nwt = NW_PER_GPU * world
    # "ipc": the one-hop exchange as ONE launch per run on every rank (k_lnlike SMODE 6: decisions, rows
    # and words stored into every rank's copy as they are made); "ipc-launches": one launch per half-step
    # with the rows exchanged after it; "rccl": one launch per half-step and an in-place ncclAllGather
    all_modes = ["ipc", "ipc-launches", "rccl"]
    modes = {"auto": list(all_modes), "ipc": ["ipc", "ipc-launches"], "rccl": ["rccl"]}[args.exchange]
    if args.exchange_order == "simplest":
        modes.reverse()
    skipped = {}
    if world > ndev:
        # (ranks sharing a device: no RCCL -- unless a library that allows it was named --, and their one-launch kernels
        # cannot all be resident)
        if not rccl_named:
            skipped["rccl"] = {"ok": None, "why": "skipped: %d ranks share %d device(s), RCCL needs a device per rank" % (world, ndev)}
            modes = [m for m in modes if m != "rccl"]
        if not os.environ.get("MBB_BENCH_TRY_ONE_LAUNCH"):
            skipped["ipc"] = {"ok": None, "why": "skipped: ranks sharing a device cannot all keep a one-launch run resident"}
            modes = [m for m in modes if m != "ipc"]
    if world == 1:
        modes = ["none"]

    def all_ok(ok):
        if dist is None:
            return bool(ok)
        import torch
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        return int(flag[0]) == 1

    def ranks_agree_on(crc):
        """every rank must hold the same ensemble, bit for bit"""
        if dist is None:
            return True
        import torch
        lo = torch.tensor([float(crc)], dtype=torch.float64)
        hi = lo.clone()
        dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
        return bool(lo[0] == hi[0])

    ref_crc = {}

    def unsharded_crc():
        """rank 0 runs the same 60 steps of the whole ensemble by itself: a walker's result does not depend
        on the launch that evaluates it, so a sharded run must agree with it bit for bit"""
        if "crc" not in ref_crc:
            like_ref, _ = make_likelihood(local_rank % ndev)
            sref = mbb.DeviceEnsembleSampler(nwt, 5, like_ref, seed=11)
            pr, lr, _ = sref.run_mcmc(allw[:nwt], 60, storechain=False)
            ref_crc["crc"] = ensemble_crc(pr, lr)
            del sref, like_ref
        return ref_crc["crc"]

    def run_mode(mode, steps, warmup):
        """Set the exchange up, rehearse it, time `steps` dependent MCMC steps.  Returns a dict with `ok`;
        when ok the sampler is still alive in it (the caller tears it down)."""
        res = {"ok": False, "why": "", "smp": None}

        def teardown():
            res["smp"] = None                      # (the sampler lives in the exchange buffer: it goes first)
            import gc
            gc.collect()
            try:
                ctx.sync()
                ctx.xchg_close() if mode.startswith("ipc") else (ctx.comm_destroy() if mode == "rccl" else None)
            except Exception:
                pass
        res["teardown"] = teardown
        ok, why = True, ""
        try:
            if mode in ("ipc", "ipc-launches"):
                from mbb_emcee_amd import parallel
                parallel.ipc_exchange_setup(ctx, rank, world, dist, max_rows=max(4096, nwt))
                ctx.set_option("sharded_flow_sampler", 1 if mode == "ipc" else 0)
                ctx.set_option("flow_spin_log2", 20)      # (a run that cannot proceed gives up within seconds)
            elif mode == "rccl":
                uid = [ctx.comm_unique_id() if rank == 0 else None]
                dist.broadcast_object_list(uid, src=0)
                guarded(lambda: ctx.comm_init(world, rank, uid[0]), GUARD_S)
        except Watchdog:
            raise Hung("%s: ncclCommInitRank did not return within %.0f s" % (mode, GUARD_S))
        except Exception as e:
            ok, why = False, "set-up: " + repr(e)
        if not all_ok(ok):
            res["why"] = why or "set-up failed on another rank"
            teardown()
            return res
        smp = mbb.DeviceEnsembleSampler(nwt, 5, like, seed=11)
        res["smp"] = smp
        state = {"ok": False, "err": None, "crc": rank}

        def rehearse():
            try:
                pos_r, lnp_r, _ = smp.run_mcmc(allw[:nwt], 60, storechain=False)
                ctx.sync()
                state["crc"] = ensemble_crc(pos_r, lnp_r)
                state["ok"] = bool(np.all(np.isfinite(lnp_r)))
            except Exception as e:           # noqa
                state["err"] = repr(e)
        try:
            guarded(rehearse, GUARD_S)
        except Watchdog:
            raise Hung("%s: the rehearsal did not return within %.0f s" % (mode, GUARD_S))
        if not all_ok(state["ok"]):
            res["why"] = "rehearsal: %s" % (state["err"] or "failed on another rank")
            teardown()
            return res
        if not ranks_agree_on(state["crc"]):
            res["why"] = "rehearsal: the ranks' copies of the ensemble differ after 60 steps"
            teardown()
            return res
        if world > 1:
            # ... and it is the chain of the UNSHARDED sampler; an exchange that is consistently wrong
            # on every rank ends here
            ok_ref = True
            if rank == 0:
                try:
                    ok_ref = unsharded_crc() == state["crc"]
                except Exception as e:       # noqa
                    ok_ref, state["err"] = False, repr(e)
            if not all_ok(ok_ref):
                res["why"] = "rehearsal: differs from the unsharded sampler's chain after 60 steps"
                teardown()
                return res
            res["rehearsal"] = "60 steps: every rank's copy and the unsharded sampler's state have the same CRC"

        # ---- the timed region: K dependent MCMC steps ----------------------------
        def timed():
            if dist is None:
                # one GPU: clock, event, launch, event, stream wait, clock -- inside one native call
                # (mbb_sampler_advance_timed), so that the harness around a 20-step region is not four Python-to-C
                # round trips; the stream is idle before and after (the wait inside).  This function runs on a thread
                # of its own (guarded), possibly on a core that has been asleep, and the W warm-up steps go through the
                # very call that is timed next.  The GPU: a process that has run a millisecond of kernels so far (the rehearsal) finds it ~6 % slower
                # than one that has kept it busy for 20 ms (profiles/r04/timed_region.txt: 141 against 133 us on the
                # stream for the same 20 steps) -- a scratch ensemble is stepped for 50 ms first, said on the line.
                scratch = mbb.DeviceEnsembleSampler(nwt, 5, like, seed=12)
                scratch.run_mcmc(allw[:nwt], 2, storechain=False)
                ran = [60, 2]                                     # steps of every sampler run so far (rehearsal first)
                t_end = time.perf_counter() + PRECONDITION_S
                pre = min(4096, max(2, steps))                    # (launches exactly as long as the timed one: see PRECONDITION_S)
                while time.perf_counter() < t_end:
                    scratch.advance_timed(pre)
                    ran.append(pre)
                smp.advance_timed(warmup)
                first = smp.advance_timed(steps)
                # (the line's value is that ONE region; eight more of the same right behind it, reported beside it,
                # say how far the single shot is from the typical one)
                state["again"] = [smp.advance_timed(steps) for _ in range(8)]
                del scratch
                # (for the profile summaries: which of the sampler kernel's launches was the timed one, and how many
                # half-steps they cover together -- runs of fewer than two steps are not launches of that kernel)
                state["runs"] = {"steps_of_every_run": ran + [warmup, steps] + [steps] * 8, "timed_run": len(ran) + 1}
                return first
            # (N > 1: the same sharded sampler, PRECONDITION_STEPS untimed steps further first -- the GPUs as a long run leaves
            # them, as at N = 1; a fixed count, the ranks must agree on it)
            smp.advance_async(PRECONDITION_STEPS)
            ctx.sync(); barrier()
            state["pre_steps"] = PRECONDITION_STEPS
            smp.advance_async(warmup)
            ctx.sync(); barrier()
            e0, e1 = ctx.event(), ctx.event()
            t0 = time.perf_counter()
            ctx.record(e0)
            smp.advance_async(steps)
            ctx.record(e1)
            ctx.sync(); barrier()
            return time.perf_counter() - t0, ctx.elapsed_ms(e0, e1)
        try:
            elapsed, stream_ms = guarded(timed, GUARD_S)
        except Watchdog:
            raise Hung("%s: the timed run did not return within %.0f s" % (mode, GUARD_S))
        except Exception as e:               # noqa
            elapsed, stream_ms, state["err"] = None, None, repr(e)
        if not all_ok(elapsed is not None):
            res["why"] = "timed run: %s" % (state["err"] or "failed on another rank")
            teardown()
            return res
        if dist is not None:
            import torch
            t = torch.tensor([elapsed], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t[0])
        # the chain is still a valid one: every rank holds the same finite state
        try:
            pos_end, lnp_end, _ = smp.run_mcmc(None, 0, storechain=False)
            fine = bool(np.all(np.isfinite(lnp_end)))
            crc = ensemble_crc(pos_end, lnp_end)
        except Exception as e:
            fine, crc, state["err"] = False, rank, repr(e)
        if not (all_ok(fine) and ranks_agree_on(crc)):
            # a number measured on an exchange that lost or mixed up rows is not a number
            res["why"] = "timed run: %s" % ("the ranks' copies of the ensemble differ afterwards"
                                            if fine else (state["err"] or "non-finite state"))
            teardown()
            return res
        res.update(ok=True, elapsed=elapsed, stream_ms=stream_ms, again=state.get("again"), runs=state.get("runs"),
                   pre_steps=state.get("pre_steps", 0),
                   form=ctx.info("last_kernel_form"),
                   us_per_step=1e6 * elapsed / steps,
                   # which instantiation that was (form 7: last_wpb = pairs of walkers per workgroup)
                   pairs=ctx.info("last_wpb"), staged=bool(ctx.info("last_stage")))
        return res

    tried, validation, run = [], dict(skipped), None
    mode = None
    for mode in modes:
        try:
            r = run_mode(mode, args.steps, args.warmup)
        except Hung as e:
            fail(4, error=str(e), collective_hung=True, hang=EXCHANGE_TEXT[mode],
                 collective_fallback_from=tried)
        if r["ok"]:
            run = r
            break
        tried.append("%s %s" % (mode, r["why"]))
        validation[mode] = {"ok": False, "why": r["why"]}
    if run is None:
        fail(3, error="no exchange gave a valid run: " + "; ".join(tried),
             collective="unavailable: " + "; ".join(tried), exchange_validation=validation or None)
    smp, elapsed, stream_ms = run["smp"], run["elapsed"], run["stream_ms"]
    mode_used = mode
    base["config"]["collective"] = EXCHANGE_TEXT[mode_used]
    if "rehearsal" in run:
        base["config"]["rehearsal"] = run["rehearsal"]
    if tried:
        base["config"]["collective_fallback_from"] = tried
    if os.environ.get("MBB_BENCH_WALKERS_PER_GPU"):
        base["valid_for_scaling"] = False
        base["invalid"] = "MBB_BENCH_WALKERS_PER_GPU=%d: not BASELINE.json's ensemble, a rehearsal" % NW_PER_GPU
    if world > ndev:
        base["valid_for_scaling"] = False
        base["config"]["note"] = "%d ranks share %d device(s): a rehearsal of the exchange, not a scaling point" % (world, ndev)
    validation[mode_used] = {"ok": True, "us_per_step": run["us_per_step"], "steps": args.steps,
                             "kernel_form": run["form"], "why": "carried the timed run"}

    if rank == 0:
        out = dict(base)
        out.update({"value": nwt * args.steps / elapsed, "ms_per_step": 1e3 * elapsed / args.steps,
                    "mcmc_steps_per_s": args.steps / elapsed,
                    # (`value` is ONE region of K steps; the median of it and the eight regions right behind it says what
                    # a typical one is)
                    **({"value_median_of_9": nwt * args.steps / float(np.median([elapsed] + [w for w, _ in run["again"]])),
                        "us_per_step_median_of_9": 1e6 * float(np.median([elapsed] + [w for w, _ in run["again"]])) / args.steps,
                        "us_per_step_min_max_of_9": [1e6 * min([elapsed] + [w for w, _ in run["again"]]) / args.steps,
                                                     1e6 * max([elapsed] + [w for w, _ in run["again"]]) / args.steps]}
                       if run.get("again") else {}),
                    "stream_us_per_step": stream_ms * 1e3 / args.steps,
                    **({"same_region_again_us": {"wall": [round(w * 1e6, 2) for w, _ in run["again"]],
                                                 "stream": [round(m * 1e3, 2) for _, m in run["again"]],
                                                 "note": "eight more timed regions of the same K steps right behind the reported one"}}
                       if run.get("again") else {}),
                    **({"sampler_runs": run["runs"]} if run.get("runs") else {}),
                    # (sharded with the one-hop exchange: the counts of this rank's own walkers; over RCCL every rank ends up
                    # with everybody's -- found by the first rehearsal of that leg with three ranks: 1.18)
                    "acceptance_fraction": float(np.sum(smp.naccepted)) / (nwt if mode_used == "rccl" else nwt / world) / (60 + run.get("pre_steps", 0) + args.warmup + args.steps * (1 + len(run.get("again") or []))),
                    "ranks_agree": True})
        k_us = stream_ms * 1e3 / (2 * args.steps)       # one half-step of the dominant kernel
        form = run["form"]
        out["config"] = dict(out["config"])
        if world > 1:
            out["config"]["preconditioning"] = ("%d untimed steps of the same sharded sampler before the W warm-up steps: the GPUs' "
                                                "clocks" % run.get("pre_steps", 0))
            out["config"]["preconditioning_short"] = "%d untimed steps of the same sampler before warm-up (clocks)" % run.get("pre_steps", 0)
        if world == 1:
            out["config"]["preconditioning"] = ("%.0f ms of sampler steps on a scratch ensemble right before the W warm-up steps: "
                                                "the GPU's clocks, not the chain that is timed" % (PRECONDITION_S * 1e3))
            out["config"]["preconditioning_short"] = "%.0f ms of scratch-ensemble steps before the W warm-up steps (GPU clocks)" % (PRECONDITION_S * 1e3)
        if form in (6,) + ONE_LAUNCH_FORMS:
            nlaunch = (args.steps + 4095) // 4096
            if form == 7:
                kern_label = ("k_flowm<thick,alpha,staged>: %d workgroups, one per (pair of walkers, candidate); constructor, "
                              "quadrature and accept test of a half-step in waves of their own" % ctx.info("last_grid"))
                out["config"]["sampler_form"] = ("one launch per 4096 steps, the quadrature of both candidates ahead of the partner's "
                                                 "decision (k_flowm, form 7): the timed region is %d launch(es) of %d half-steps"
                                                 % (nlaunch, 2 * args.steps))
            elif form == 9:
                kern_label = ("k_flowa<thick,alpha,staged>: %d workgroups, each owning %d walker(s) of each half; the constructor a "
                              "half-step ahead for both outcomes of the partner's pending move" % (ctx.info("last_grid"), ctx.info("last_wpb")))
                out["config"]["sampler_form"] = ("one launch per 4096 steps, resident (form %d): the timed region is %d launch(es) of %d "
                                                 "half-steps" % (form, nlaunch, 2 * args.steps))
            else:
                kern_label = ("k_lnlike<thick,alpha,one-launch look-ahead run,staged>: %d workgroups move walkers, %d work "
                              "ahead" % (half, ctx.info("last_workgroups_ahead")))
                out["config"]["sampler_form"] = ("one launch per 4096 steps on every rank (k_lnlike SMODE %d): the timed region is %d "
                                                 "launch(es) of %d half-steps" % (form, nlaunch, 2 * args.steps))
            out["kernel_avg_us"] = stream_ms * 1e3 / nlaunch
        else:
            kern_label = "k_lnlike<thick,alpha,sampler,staged> n=%d" % half
            out["config"]["sampler_form"] = "one launch per half-step (k_lnlike SMODE %d)" % form
            out["kernel_avg_us"] = k_us
        out["half_step_us"] = k_us
        short_kern = ({7: "k_flowm", 9: "k_flowa"}.get(form, "k_lnlike") + "<thick,alpha,%sform %d> %d workgroups x %d threads"
                      % ("staged," if run["staged"] else "", form, ctx.info("last_grid"), ctx.info("last_threads")))
        out["config"]["sampler_form_short"] = ("form %d: %s" % (form, "one launch per <=4096 steps" if form in (6,) + ONE_LAUNCH_FORMS
                                                                  else "one launch per half-step"))
        if world == 1 and not args.no_extras:
            # (whatever goes wrong beside the timed region is said on the line; it does not take the
            # measured value with it)
            try:
                out.update(extras(args, like, flux, ctx, allw, k_us, kern_label, nq, nb, form,
                                  pairs=run["pairs"], staged=run["staged"], partial=out, kern_short=short_kern))
            except Exception as e:           # noqa
                import traceback
                traceback.print_exc()
                out["extras_error"] = "%s: %s" % (type(e).__name__, e)
        elif world > 1:
            alg_bytes = 48.0 * half + 16.0 * nq + 16.0 * nb
            out["roofline"] = {"bound": "fp64-valu", "kernel": kern_label, "kernel_short": short_kern, "achieved": None,
                               "peak": FP64_VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": None, "traffic": None,
                               "note": "launch slot = kernel + exchange at N > 1; the roofline is reported at N = 1",
                               "launch_slot_us": k_us,
                               "hbm": {"algorithmic_bytes_per_launch": alg_bytes,
                                       "achieved_GBps": alg_bytes / (k_us * 1e-6) / 1e9, "peak_GBps": HBM_PEAK_GBS}}
        emit_line(out)          # (N > 1: held by the supervisor until this process has ended)
    barrier()
    smp = None
    run["smp"] = None
    if world > 1:
        run["teardown"]()
        t0_run = float(os.environ.get("MBB_BENCH_T0", "0")) or time.time()

        def still_time():
            """Is the run young enough to start something optional?  Decided together: a collective that only
            some ranks enter is a hang of its own."""
            return all_ok(time.time() - t0_run < OPTIONAL_UNTIL_S)

        # ---- M1 of the sharded run: north_star's own split.  Every rank holds the proposed rows of the
        # moving half (every rank runs the same sampler with the same random stream), evaluates its block in
        # the fused kernel and ONE ncclAllGather of the blocks' log-probabilities gives every rank the whole
        # vector: literally emcee's pool (mbb_fit.py:80-81) replaced.  Host arrays in and out, never `value`.
        hung = None
        if (world <= ndev or rccl_named) and not args.no_sharded_boundary:
            if still_time():
                try:
                    def bcast(obj):
                        box = [obj]
                        dist.broadcast_object_list(box, src=0)
                        return box[0]

                    def reduce_max(vals):
                        import torch
                        t = torch.tensor(vals, dtype=torch.float64)
                        dist.all_reduce(t, op=dist.ReduceOp.MAX)
                        return [float(x) for x in t]
                    sb = sharded_boundary(ctx, like, rank, world, allw, nwt, barrier, all_ok, bcast, reduce_max)
                except Hung as e:
                    sb, hung = {"ok": False, "why": str(e), "hung": True}, "boundary_sharded"
                except Exception as e:           # noqa
                    sb = {"ok": False, "why": repr(e)}
            else:
                sb = {"ok": None, "why": "skipped: the run was already %.0f s old" % (time.time() - t0_run)}
            if rank == 0:
                emit({"_part": "boundary_sharded", "data": sb})
            if hung is not None:
                sys.stdout.flush()
                os._exit(0)             # the measured run was valid; a wedged stream would hang teardown
        elif rank == 0:
            emit({"_part": "boundary_sharded", "data": {"ok": None, "why": ("not asked for" if args.no_sharded_boundary else
                  "skipped: %d ranks share %d device(s), RCCL needs a device per rank" % (world, ndev))}})
        # ---- the exchanges the timed run did not use: the same rehearsal and a short timed run each,
        # outside `value`.  The measured line is already with the supervisor; whatever happens here
        # only adds to it.
        if not args.no_validate:
            vsteps = max(50, min(args.steps, 500))
            hung = None
            for m in all_modes:
                if m in validation:
                    continue
                if not still_time():
                    validation[m] = {"ok": None, "why": "skipped: the run was already more than %.0f s old" % OPTIONAL_UNTIL_S}
                    continue
                if m not in modes and args.exchange != "auto":
                    validation[m] = {"ok": None, "why": "not asked for (--exchange %s)" % args.exchange}
                    continue
                try:
                    r = run_mode(m, vsteps, 50)
                except Hung as e:
                    validation[m] = {"ok": False, "why": str(e), "hung": True}
                    hung = m
                    break
                except Exception as e:           # noqa
                    validation[m] = {"ok": False, "why": repr(e)}
                    continue
                if r["ok"]:
                    validation[m] = {"ok": True, "us_per_step": r["us_per_step"], "steps": vsteps,
                                     "kernel_form": r["form"],
                                     "why": "rehearsal and a timed run of %d steps after the measured one" % vsteps}
                    r["smp"] = None
                    r["teardown"]()
                else:
                    validation[m] = {"ok": False, "why": r["why"]}
            if rank == 0:
                emit({"_part": "exchange_validation", "data": validation})
            if hung is not None:
                sys.stdout.flush()
                os._exit(0)             # the measured run was valid; a wedged stream would hang teardown
        elif rank == 0:
            emit({"_part": "exchange_validation", "data": validation})
        barrier()
        dist.destroy_process_group()


def sharded_boundary(ctx, like, rank, world, allw, nwt, barrier, all_ok, bcast, reduce_max):
    """SURVEY.md 8d M1 at N > 1 (see the call site): parallel.ShardedLikelihood over parallel.RcclComm.  Every
    rank makes the same calls in the same order (a collective inside each); times are this rank's wall clock
    around a call, the figure reported is the slowest rank's median.  bcast(obj) -> rank 0's obj on every rank and
    reduce_max([floats]) -> their maxima over the ranks are the caller's (gloo in bench.py; nothing of torch here)."""
    from mbb_emcee_amd import parallel
    res = {"ok": False,
           "what": "likelihood.__call__ of the moving half-ensemble sharded over %d ranks: this rank's block through the "
                   "fused kernel, one in-place ncclAllGather of the blocks' lnprob (RCCL), the gathered vector back on "
                   "the host; host arrays in and out, median of 200 synchronous calls, slowest rank" % world}
    uid = bcast(ctx.comm_unique_id() if rank == 0 else None)
    comm, why = None, ""
    try:
        comm = guarded(lambda: parallel.RcclComm(ctx, rank, world, uid), GUARD_S)
    except Watchdog:
        raise Hung("boundary_sharded: ncclCommInitRank did not return within %.0f s" % GUARD_S)
    except Exception as e:           # noqa
        why = repr(e)
    if not all_ok(comm is not None):
        res["why"] = "set-up: " + (why or "failed on another rank")
        if comm is not None:
            comm.close()
        return res
    sl = parallel.ShardedLikelihood(like, comm)
    try:
        for n in (nwt // 2, nwt):
            p = np.ascontiguousarray(allw[:n])

            def loop():
                for _ in range(20):
                    out = sl(p)
                barrier()
                ts = []
                for _ in range(200):
                    t0 = time.perf_counter(); out = sl(p); ts.append(time.perf_counter() - t0)
                return ts, out
            try:
                ts, out = guarded(loop, GUARD_S)
            except Watchdog:
                raise Hung("boundary_sharded: %d sharded calls of %d rows did not return within %.0f s" % (220, n, GUARD_S))
            # the gathered vector is the unsharded evaluation, bit for bit (a walker's lnL does not depend on the
            # launch or the GPU that evaluates it), and the same on every rank
            same = bool(np.array_equal(out, like(p), equal_nan=True))
            if not all_ok(same):
                res["why"] = "the gathered lnprob of %d rows differs from the unsharded evaluation on some rank" % n
                return res
            t = reduce_max([float(np.median(ts)), float(np.percentile(ts, 90))])
            res["rows_%d" % n] = {"rows_per_rank": -(-n // world), "median_us": float(t[0]) * 1e6, "p90_us": float(t[1]) * 1e6,
                                  "evals_per_s": n / float(t[0]), "calls": len(ts),
                                  "equals_unsharded_bitwise": True}
        res["ok"] = True
    finally:
        try:
            ctx.sync()
            comm.close()
        except Exception:           # noqa
            pass
    return res


def boundary_loop(like, ctx, p, n, ncalls=400, warm=50):
    """`ncalls` synchronous boundary calls of the rows `p` in a loop, each timed by the wall clock: median, tail, which
    calls were slow and what the library's serve counters moved by over exactly these calls."""
    for _ in range(warm):
        like(p)
    names = ("serve_requests", "serve_fallbacks", "serve_lease_yields", "serve_resizes", "serve_rests")
    before = [ctx.info(k) for k in names]
    ts = np.empty(ncalls)
    clock = time.perf_counter
    for i in range(ncalls):
        t0 = clock(); like(p); ts[i] = clock() - t0
    after = [ctx.info(k) for k in names]
    med = float(np.median(ts))
    slow = np.flatnonzero(ts > 1.5 * med)
    runs = int(np.sum(np.diff(slow) > 1) + 1) if slow.size else 0
    res = {"median_us": med * 1e6, "p90_us": float(np.percentile(ts, 90)) * 1e6, "p99_us": float(np.percentile(ts, 99)) * 1e6,
           "max_us": float(ts.max()) * 1e6, "evals_per_s": n / med, "calls": ncalls,
           "slow_calls": int(slow.size), "slow_runs": runs,
           "slow_call_indices": [int(i) for i in slow[:40]], "slow_call_us": [round(float(ts[i]) * 1e6, 2) for i in slow[:40]]}
    res.update({k: int(a - b) for k, a, b in zip(names, after, before)})
    return res


def extras(args, like, flux, ctx, allw, k_us, kern_label, nq, nb, form=1, pairs=1, staged=True, partial=None, kern_short=None):
    """Everything on the line besides the timed region (rank 0, one GPU).  `partial`: the caller's line, filled in
    leg by leg, so that what was measured before a leg failed is kept."""
    import mbb_emcee_amd as mbb
    out = partial if partial is not None else {}
    half = NW_PER_GPU // 2
    pos = allw[:NW_PER_GPU]

    # ---- M1, the boundary: synchronous likelihood.__call__, host arrays in and out.  Twice: as the library does it by
    # default -- after a few calls in a row the rows go to a kernel that stays resident between the calls and is rung
    # through the BAR (k_serve) -- and with a launch per call (option "serve" 0)
    bnd = {}
    for mode, serve in (("served", 1), ("launch_per_call", 0)):
        ctx.set_option("serve", serve)
        sub = {}
        for n in (half, NW_PER_GPU, 1):
            p = np.ascontiguousarray(pos[:n]) if n > 1 else np.ascontiguousarray(pos[0])
            sub["rows_%d" % n] = boundary_loop(like, ctx, p, n)
        sub["served_by_resident_kernel"] = bool(ctx.info("serving"))
        sub["serve_fallbacks"] = ctx.info("serve_fallbacks")
        bnd[mode] = sub
    ctx.set_option("serve", 1)
    bnd["rows_%d" % half] = bnd["served"]["rows_%d" % half]
    bnd["rows_%d" % NW_PER_GPU] = bnd["served"]["rows_%d" % NW_PER_GPU]
    bnd["note"] = ("SURVEY.md 8d M1: host float64[n,5] in -> host float64[n] out through likelihood.__call__ "
                   "(what emcee calls per half-step, mbb_fit.py:80-81), PCIe inclusive, never `value`; `served`: the library's "
                   "default for a sampler's loop of calls; `launch_per_call`: every call a kernel launch")
    out["boundary"] = bnd
    # M1 where a reader of the line looks first: what an external sampler such as emcee gets per half-step of the
    # bench's ensemble (125 rows), next to `value`, which is the device-resident sampler's rate (M2).  With the tail and
    # what the library did meanwhile: a slow call is either the library's doing (a request that fell back to a launch, a
    # server sent away when its lease was up or made anew for another width: the deltas of its counters over exactly
    # these calls) or the box's (none of those, the slow calls in a few runs of neighbours: a host thread descheduled)
    b125 = bnd["rows_%d" % half]
    out["boundary_M1"] = {"rows": half, "p50_us": b125["median_us"], "p90_us": b125["p90_us"],
                          "evals_per_s": b125["evals_per_s"],
                          **{k: b125[k] for k in ("p99_us", "max_us", "calls", "slow_calls", "slow_runs", "serve_requests",
                                                  "serve_fallbacks", "serve_lease_yields", "serve_resizes", "serve_rests")},
                          "launch_per_call_p50_us": bnd["launch_per_call"]["rows_%d" % half]["median_us"],
                          "what": "synchronous likelihood.__call__(float64[125, 5]) -> float64[125], host arrays in and out, in a "
                                  "loop of calls (the rows are served by a kernel resident between the calls); slow_calls: "
                                  "those beyond 1.5 x the median, slow_runs: in how many runs of neighbours they came; serve_*: "
                                  "what the library's counters moved by over exactly these calls",
                          # (how the three host steps of a call -- rows in, native call, results out -- were made)
                          "host_glue": ("CPython extension (mbb_emcee_amd/csrc/mbb_fastcall.c)"
                                        if like._fast is not None and like._fast[7] is not None else "numpy + ctypes")}

    # ---- pipelined upper bound: independent launches on pre-computed proposals
    NSETS = 8
    props = proposals(pos, NSETS, seed=100)
    d_pars = []
    for p in props:
        b = ctx.alloc(p.nbytes); b.upload(p); d_pars.append(b)
    d_lnl = [ctx.alloc(half * 8) for _ in range(2)]
    d_status = ctx.alloc(half * 4)
    ksteps = max(200, min(args.steps, 2000))

    def pstep(i):
        for h in range(2):
            ctx.lnlike_batch_device(d_pars[(2 * i + h) % (2 * NSETS)], half, d_lnl[h], d_status)
    for i in range(100):
        pstep(i)
    ctx.sync()
    p0, p1 = ctx.event(), ctx.event()
    t0 = time.perf_counter()
    ctx.record(p0)
    for i in range(ksteps):
        pstep(i)
    ctx.record(p1); ctx.sync()
    tp = time.perf_counter() - t0
    plain_us = ctx.elapsed_ms(p0, p1) * 1e3 / (2 * ksteps)
    out["pipelined"] = {"evals_per_s": NW_PER_GPU * ksteps / tp, "steps": ksteps,
                        "kernel_avg_us": plain_us,
                        "note": "k_lnlike<thick,alpha,plain,staged> n=125 on device-resident, pre-computed "
                                "proposals, independent launches back to back: an upper bound, not an MCMC"}
    # what the GPU computed for the last of those launches equals a fresh evaluation
    got = d_lnl[1].download(np.float64, half)
    last = props[(2 * (ksteps - 1) + 1) % (2 * NSETS)]
    assert np.array_equal(like(last), got, equal_nan=True)

    # ---- the same chain in the sampler's other forms: a train of launches, one per half-step (what round 1 and the
    # first half of round 2 timed), and the resident form larger ensembles take (form 9: the constructor a half-step
    # ahead), here with one walker of each half per workgroup
    forms = {}
    for name, opts in (("one_launch_per_half_step", {"lookahead_sampler": 0}),
                       ("resident_constructor_ahead_form9", {"lookahead_sampler": 1, "flow_sampler": 1, "resident_sampler": 2})):
        for o, v in opts.items():
            ctx.set_option(o, v)
        s2 = mbb.DeviceEnsembleSampler(NW_PER_GPU, 5, like, seed=11)
        s2.run_mcmc(pos, 60, storechain=False)
        s2.advance_async(100); ctx.sync()
        f0, f1 = ctx.event(), ctx.event()
        ctx.record(f0); s2.advance_async(ksteps); ctx.record(f1); ctx.sync()
        forms[name] = {"stream_us_per_step": ctx.elapsed_ms(f0, f1) * 1e3 / ksteps,
                       "evals_per_s": NW_PER_GPU * ksteps / (ctx.elapsed_ms(f0, f1) * 1e-3),
                       "kernel_form": ctx.info("last_kernel_form")}
        del s2
    ctx.set_option("lookahead_sampler", 1); ctx.set_option("flow_sampler", 1); ctx.set_option("merged_flow_sampler", 1)
    ctx.set_option("resident_sampler", 1)
    forms["note"] = "the other forms of the device sampler, same chain bit for bit (stream time, HIP events)"
    out["other_sampler_forms"] = forms

    # ---- larger ensembles on this one GPU (BASELINE.json configs[2] is 2000 walkers over 8 GPUs; here the whole of it
    # on one): the resident forms, us per MCMC step and evals/s, with the launch train beside them
    big = {}
    for nwb in (512, 1000, 2000, 4096):
        pb = allw[:nwb] if allw.shape[0] >= nwb else np.tile(allw, (nwb // allw.shape[0] + 1, 1))[:nwb] * (1.0 + 1e-3 * np.arange(nwb)[:, None] / nwb)
        row = {}
        for name, look in (("one_launch", 1), ("launch_train", 0)):
            ctx.set_option("lookahead_sampler", look)
            sb_ = mbb.DeviceEnsembleSampler(nwb, 5, like, seed=11)
            sb_.run_mcmc(pb, 20, storechain=False)
            sb_.advance_async(50); ctx.sync()
            g0, g1 = ctx.event(), ctx.event()
            ctx.record(g0); sb_.advance_async(300); ctx.record(g1); ctx.sync()
            us = ctx.elapsed_ms(g0, g1) * 1e3 / 300
            row[name] = {"us_per_step": us, "evals_per_s": nwb / (us * 1e-6), "kernel_form": ctx.info("last_kernel_form"),
                         "workgroups": ctx.info("last_grid"), "walkers_per_workgroup_and_half": ctx.info("last_wpb"),
                         # (90 flop per quadrature sample, nwb / 2 walkers per half-step, fp64 vector peak)
                         "roofline_frac": 90.0 * nq * (nwb / 2) / (0.5 * us * 1e-6) / (FP64_VALU_PEAK_TFLOPS * 1e12)}
            assert np.all(np.isfinite(sb_.run_mcmc(None, 0, storechain=False)[1]))
            del sb_
        big["walkers_%d" % nwb] = row
    ctx.set_option("lookahead_sampler", 1)
    big["note"] = ("one ensemble of that many walkers on this GPU, 300 steps by HIP events; kernel_form 9 = resident with the "
                   "constructor a half-step ahead (k_flowa), 1 = one launch per half-step; "
                   "roofline_frac = algorithmic flops of a half-step (90 x NQ x walkers / 2) over its time, against the fp64 "
                   "vector peak")
    out["large_ensembles"] = big

    # ---- the empirical roof of the sample arithmetic, measured now
    sec, slots, roof_mhz = ctx.roof_probe(TRUTH, reps=40)
    nchunk = ctx.info("nchunk")
    slots_per_launch = half * nchunk * 64.0
    roof_rate = slots / sec
    kern_rate = slots_per_launch / (k_us * 1e-6)
    arith = {"bound": "sample-arithmetic (empirical)", "unit": "quadrature samples/s (lane slots, padding included)",
             "peak": roof_rate, "achieved": kern_rate, "frac": kern_rate / roof_rate,
             "probe": "k_roof: 512 workgroups x 512 threads walk all %d chunks 40 times with one walker's "
                      "constants -- the exp + two degree-7 polynomials per sample and nothing else" % nchunk,
             "probe_seconds": sec, "probe_shader_clock_mhz": roof_mhz, "samples_per_launch": half * nq, "lane_slots_per_launch": slots_per_launch}

    roof, hbm = dominant_kernel_roofline(form, pairs, staged, k_us, args.steps, kern_label, nq, nb, half)
    roof["kernel_avg_us"] = k_us
    roof["kernel_short"] = kern_short
    roof["sample_arithmetic"] = arith
    roof["why_far_below"] = ("a half-step of 125 walkers is a chain of latencies, not a stream: constructor (one dependent chain of "
                             "fp64 transcendentals on 16 lanes), quadrature (12 chunks of samples per SIMD, three dependent LDS "
                             "look-ups per sample), band sums and accept test, then ~1 us until the workgroups that depend on the "
                             "decision see it across XCDs.  Form 7 runs those stages ahead of the decisions they depend on, for every "
                             "outcome still possible, so a half-step is the longest of three shorter chains instead of their sum; "
                             "see cfg5 for the same arithmetic when the chip is full")
    out["roofline"] = roof
    out["roofline_hbm"] = hbm

    # ---- cfg5 (BASELINE.json configs[4]): 1000 independent SEDs x 250 walkers in
    # one launch -- the regime where the kernel is VALU-bound, not latency-bound
    if not args.no_cfg5:
        from tools.bench_cfg5 import setup as cfg5_setup
        like5, _, p5 = cfg5_setup(1000, NW_PER_GPU)
        c5 = like5._sync_device()
        n5 = p5.shape[0] * p5.shape[1]
        flat5 = np.ascontiguousarray(p5.reshape(-1, 5))
        dp5 = c5.alloc(flat5.nbytes); dp5.upload(flat5)
        dl5, ds5 = c5.alloc(n5 * 8), c5.alloc(n5 * 4)
        # (the first ~10 ms of sustained load run at a lower clock: warm up past that)
        c5.lnlike_repeat_device(dp5, n5, dl5, ds5, 20); c5.sync()
        q0, q1 = c5.event(), c5.event()
        c5.record(q0); c5.lnlike_repeat_device(dp5, n5, dl5, ds5, 20); c5.record(q1); c5.sync()
        ms5 = c5.elapsed_ms(q0, q1) / 20
        smp5 = mbb.DeviceEnsembleSampler(NW_PER_GPU, 5, like5, seed=3)
        smp5.run_mcmc(p5, 3, storechain=False)
        c5.sync(); t5 = time.perf_counter()
        smp5.advance_async(20); c5.sync()
        t5 = (time.perf_counter() - t5) / 20
        sec5, slots5, mhz5 = c5.roof_probe(TRUTH, reps=200)
        pm5, pm5_src, err5 = measured_valu("pmc_valu_cfg5*.json", kernel_key(0, staged=False))
        r5 = valu_roofline(pm5, pm5_src, ms5 * 1e-3, "k_lnlike<thick,alpha,plain> n=250000")
        if r5 and mhz5 > 0:
            # the chip does not hold 2.4 GHz under this load (tools/probe_clock.py): the same
            # bound at the clock the sample-arithmetic probe measured just now
            r5["shader_clock_mhz_under_load"] = mhz5
            r5["valu_issue_frac_at_that_clock"] = r5["valu_issue_frac"] * (CLOCK_HZ / 1e6) / mhz5
        cfg5 = {"workload": "1000 sources x 250 walkers, 8 bands, NQ=2209, thick+alpha, one launch",
                "evals_per_launch": n5, "kernel_ms": ms5, "evals_per_s": n5 / ms5 * 1e3,
                "samples_per_s": n5 * nq / ms5 * 1e3,
                "sampler_ms_per_step": t5 * 1e3, "sampler_evals_per_s": n5 / t5,
                "geometry": {"walkers_per_workgroup": c5.info("last_wpb"), "threads": c5.info("last_threads")},
                "roofline": r5,
                # SURVEY.md 8d (ii)'s algorithmic count, 90 flop per quadrature sample, against the fp64 vector peak: the
                # fraction that can be compared across rounds (the COUNTED flops fall when the same integrals take fewer
                # operations -- round 6: 3.28e10 -> 2.58e10 per launch -- so counted / time rewards executing more)
                "algorithmic_frac": 90.0 * nq * n5 / (ms5 * 1e-3) / (FP64_VALU_PEAK_TFLOPS * 1e12),
                "sample_arithmetic": {"peak": slots5 / sec5, "achieved": n5 * nchunk * 64.0 / (ms5 * 1e-3),
                                      "frac": n5 * nchunk * 64.0 / (ms5 * 1e-3) / (slots5 / sec5),
                                      "unit": "quadrature samples/s (lane slots)",
                                      "probe_shader_clock_mhz": mhz5, "probe_seconds": sec5}}
        if pm5:
            cfg5["valu_wave_instructions_per_walker"] = pm5["counters_per_launch"].get("SQ_INSTS_VALU", 0) / n5
        out["cfg5"] = cfg5

    # ---- what a user runs (outside `value`): the sampler WITH its chain stored and brought back in
    # emcee's [walker, step, dim] layout (reference mbb_fit.py:542 -> results.py:153-154), by the wall
    # clock; and whole fits, mbb_fitter.run(50, 250, p0) (run_mbb_emcee.py:73-74, :137-142 defaults), with
    # the device-resident sampler (the default) and with the host stretch move that calls
    # likelihood.__call__ once per half-step (what an external sampler such as emcee does)
    if not args.no_fit:
        out["user_runs"] = user_runs(like, pos, bench_steps=args.steps)
        st = out["user_runs"].get("sampler_M2_stored_chain", {})
        # what mbb_fitter.run / run_mcmc(storechain=True) delivers -- the chain on the host in emcee's layout (consumed at
        # results.py:154-155) -- beside `value`, which is the unstored advance: for a user's 2000 steps and for this run's K
        if "steps_2000" in st:
            out["stored_chain_us_per_step"] = st["steps_2000"]["wall_us_per_step"]
        if "steps_%d" % args.steps in st:
            out["stored_chain_us_per_step_at_K"] = st["steps_%d" % args.steps]["wall_us_per_step"]
        out["postprocess"] = postprocess_leg(like, pos, cpu=not args.no_cpu)
        try:
            ctx.set_option("serve", 1)            # (this process steps aside: nothing of it resident, it makes no calls meanwhile)
            out["pool_two_processes"] = pool_leg(like, pos)
            pw = out["pool_two_processes"].get("served", {})
            if "p50_us" in pw and isinstance(out.get("boundary_M1"), dict):
                out["boundary_M1"]["two_processes_p50_us"] = float(max(pw["p50_us"]))
        except Exception as e:      # noqa -- a side leg
            out["pool_two_processes"] = {"error": repr(e)}

    # ---- the other single-GPU configurations of BASELINE.json (configs[0], configs[3]): M1, M2, the
    # plain launch, its fp64 roofline from the committed PMC pass of that launch, the CPU oracle on
    # the same rows (tools/bench_configs.py)
    if not args.no_configs:
        from tools.bench_configs import measure as cfg_measure
        out["configs"] = {c: cfg_measure(c, roofline_fn=config_roofline, cpu=not args.no_cpu) for c in ("cfg1", "cfg4")}

    # ---- M1 once more, as the LAST measurement on the GPU of this run: is a tail in the first leg the leg's place (right
    # behind the timed region) or the box?
    try:
        ctx.set_option("serve", 1)
        again = boundary_loop(like, ctx, np.ascontiguousarray(pos[:half]), half)
        if isinstance(out.get("boundary_M1"), dict):
            out["boundary_M1"]["last_leg"] = {k: again[k] for k in ("median_us", "p90_us", "p99_us", "max_us", "slow_calls", "slow_runs",
                                                                     "serve_requests", "serve_fallbacks", "serve_lease_yields", "serve_resizes", "serve_rests")}
        out["boundary"]["last_leg_rows_%d" % half] = again
    except Exception as e:      # noqa -- a side leg
        out["boundary"]["last_leg_error"] = repr(e)

    if not args.no_cpu:
        cb, ref = cpu_baseline(like, flux, pos)
        out["cpu_baseline"] = cb
        chk = like(pos)
        err = np.abs(chk - ref) / np.maximum(1.0, np.abs(ref))
        out["parity_max_err_vs_oracle"] = float(err.max())
        assert err.max() < 1e-10
    return out


if __name__ == "__main__":
    main()
