"""CPU tests of the host logic: passband construction (against fixtures made by
the reference), filter-wheel / photometry text formats, likelihood bookkeeping,
initial positions, the sampler, the C-ABI symbol table, and the 2-rank (gloo)
sharding path.  No GPU is needed and none is used."""
import os
import pickle
import re
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT


# ---------------------------------------------------------------- response
def test_default_wheel_members():
    """reference tests/test_response.py:10-17"""
    from mbb_emcee_amd import response_set
    wheel = response_set()
    for nm in ("SCUBA2_850um", "SPIRE_250um", "SPIRE_350um", "SPIRE_500um", "Bolocam_1.1mm"):
        assert nm in wheel
    assert len(wheel) == 18 and "MAMBO2_1.2mm" in wheel


def test_spire250_kat():
    """reference tests/test_response.py:19-29"""
    from mbb_emcee_amd import response_set
    r = response_set()["SPIRE_250um"]
    assert r.data_read and r.name == "SPIRE_250um"
    np.testing.assert_allclose(r.normfac, 3.0796e-3, atol=1e-4)
    np.testing.assert_allclose(r.effective_wavelength, 247.268656, atol=1e-4)
    np.testing.assert_allclose(r(lambda x: 1), 1.011046, atol=1e-4)


def test_add_special_kat():
    """reference tests/test_response.py:31-43"""
    from mbb_emcee_amd import response_set
    wheel = response_set()
    wheel.add_special("ZSpec_box_1050um_100")
    r = wheel["ZSpec_box_1050um_100"]
    assert r.data_read and r.name == "ZSpec_box_1050um_100"
    np.testing.assert_allclose(r.effective_frequency, 286.1655, atol=1e-3)
    np.testing.assert_allclose(r(lambda x: 1), 1.0, atol=1e-4)
    del wheel["ZSpec_box_1050um_100"]
    assert "ZSpec_box_1050um_100" not in wheel


def test_builtin_wheel_is_cached_not_shared():
    """The built-in wheel is set up once per process and handed out as copies: what one wheel adds, deletes or
    sets up again never shows in another."""
    from mbb_emcee_amd import response_set
    a = response_set()
    before = (a["SPIRE_350um"].effective_wavelength, a["SPIRE_350um"]._nresp)
    a.add_special("ZSpec_box_1050um_100")
    del a["SPIRE_250um"]
    a["SPIRE_350um"].setup("box_300_20", xtype="freq", xunits="ghz", senstype="energy", normtype="flat", xnorm=300.0, normparam=0.0)
    b = response_set()
    assert "SPIRE_250um" in b and "ZSpec_box_1050um_100" not in b and len(b) == 18
    assert (b["SPIRE_350um"].effective_wavelength, b["SPIRE_350um"]._nresp) == before
    assert a["SPIRE_350um"]._nresp != b["SPIRE_350um"]._nresp
    assert b["SPIRE_500um"] is not response_set()["SPIRE_500um"]
    np.testing.assert_allclose(b["SPIRE_250um"].normfac, 3.0796e-3, atol=1e-4)
    # every wheel has arrays of its own, writable, as in the reference (response.py:252-332): the reference's pattern of
    # changing a curve in place works -- on the first wheel of the process (`a`) as on a later one -- and changes that
    # wheel only, never the cached master behind the later wheels
    keep = b["SPIRE_500um"].response.copy()
    for w in (a, b):
        w["SPIRE_500um"].response[:] *= 2.0
        w["SPIRE_500um"].wavelength[0] = 1.0
        assert np.array_equal(w["SPIRE_500um"].response, 2.0 * keep)
    fresh = response_set()
    assert np.array_equal(fresh["SPIRE_500um"].response, keep) and fresh["SPIRE_500um"].wavelength[0] != 1.0
    assert fresh["SPIRE_500um"].response.flags.writeable


def test_passband_tables_match_reference(g_pb):
    """G1: wave, freq, response, trapezoid weights, normalisation and effective
    wavelength of every wheel band and every special type, as the reference
    builds them (response.py:252-332)."""
    from mbb_emcee_amd import response_set
    wheel = response_set()
    for nm in [str(n) for n in g_pb["names"]]:
        if nm not in wheel:
            wheel.add_special(nm)
        r = wheel[nm]
        sc = g_pb[nm + "/scalars"]
        np.testing.assert_allclose(r.wavelength, g_pb[nm + "/wave"], rtol=1e-15)
        np.testing.assert_allclose(r.frequency, g_pb[nm + "/freq"], rtol=1e-15)
        np.testing.assert_allclose(r.response, g_pb[nm + "/resp"], rtol=1e-15)
        if not r.isdelta:
            np.testing.assert_allclose(r._sedmult, g_pb[nm + "/sedmult"], rtol=1e-14)
            f, w = r.quadrature()
            np.testing.assert_allclose(np.sum(w), sc[7], rtol=1e-13)   # flat-SED response
        np.testing.assert_allclose([r._normfac, r.effective_wavelength, r.effective_frequency],
                                   sc[:3], rtol=1e-14)
        assert r._nresp == int(sc[5]) and bool(sc[6]) == r.isdelta


@pytest.mark.parametrize("spec,sens,ntype,xn,npar", [
    ("SPIRE_250.txt", "counts", "power", 250.0, -1.0),
    ("MIPS_70.txt", "energy", "none", 71.44, 0.0),
    ("PACS_100.txt", "counts", "flat", 100.0, 0.0)])
def test_counts_and_none_normalisations(g_pb, spec, sens, ntype, xn, npar):
    """response.py:284-301"""
    from mbb_emcee_amd import response
    r = response("x")
    r.setup(spec, senstype=sens, normtype=ntype, xnorm=xn, normparam=npar, dir="!package-dir!")
    key = "setup:%s:%s:%s" % (spec, sens, ntype)
    np.testing.assert_allclose(r._sedmult, g_pb[key + "/sedmult"], rtol=1e-14)
    np.testing.assert_allclose([r._normfac, r.effective_wavelength, r.effective_frequency],
                               g_pb[key + "/scalars"], rtol=1e-14)


def test_wheel_and_curve_text_formats(tmp_path):
    """A user-supplied filter wheel + curve files in the reference's text format
    (mbb_filterwheel.txt columns; two-column curves, '#' comments)."""
    from mbb_emcee_amd import response_set
    from mbb_emcee_amd.response import _packaged_curve
    x, r = _packaged_curve("SPIRE_350.txt")
    with open(tmp_path / "my350.txt", "w") as fh:
        fh.write("# wavelength  transmission\n")
        for a, b in zip(x, r):
            fh.write("%.17g  %.17g\n" % (a, b))
    with open(tmp_path / "wheel.txt", "w") as fh:
        fh.write("#Name File Xtype Xunit Sens NormType XNorm NormPar\n")
        fh.write("My_350 my350.txt wave microns energy power 350.0 -1.0\n")
        fh.write("My_350A my350A.txt wave angstroms energy power 3500000.0 -1.0\n")
    with open(tmp_path / "my350A.txt", "w") as fh:
        for a, b in zip(x, r):
            fh.write("%.17g %.17g\n" % (a * 1e4, b))
    wheel = response_set("wheel.txt", dir=str(tmp_path))
    ref = response_set()["SPIRE_350um"]
    np.testing.assert_allclose(wheel["My_350"]._sedmult, ref._sedmult, rtol=1e-15)
    np.testing.assert_allclose(wheel["My_350"].normfac, ref.normfac, rtol=1e-15)
    np.testing.assert_allclose(wheel["My_350A"].normfac, ref.normfac, rtol=1e-12)
    np.testing.assert_allclose(wheel["My_350A"].effective_wavelength, ref.effective_wavelength, rtol=1e-12)


def test_special_spec_errors():
    from mbb_emcee_amd import response, response_set
    wheel = response_set()
    for bad in ("X_blob_3", "X_box", "X_box_abc_3", "X_box_100mm_3"):
        with pytest.raises(ValueError):
            wheel.add_special(bad)
    with pytest.raises(ValueError):
        wheel.add_special("ALMA_alma_200")          # between bands 4 and 6
    with pytest.raises(ValueError):
        response("x").setup("box_100")
    # dsb given in wavelength units is converted consistently
    a, b = response("a"), response("b")
    a.setup("dsb_230_16_8", xtype="freq", xunits="ghz", normtype="flat", xnorm=230.0)
    lam = 299792.458 / 230.0
    assert a._nresp == 29 and np.all(a.response[13:16] == 0)
    np.testing.assert_allclose(a.effective_frequency, 230.0, rtol=1e-3)


def test_delta_passband_in_frequency_units_is_defined_here_not_parity_checked():
    """a delta-function passband given in FREQUENCY units (`X_delta_300`, the `add_special` default) raises
    AttributeError in the reference -- `_setup_delta` sets `_normwave` only in its THz branch (response.py:352-364) --
    so there is nothing to pin this to.  Here it is defined as lambda = c / nu and behaves like any delta band."""
    from mbb_emcee_amd import response, response_set
    r = response("d")
    r.setup("delta_300", xtype="freq", xunits="ghz")
    np.testing.assert_allclose(r.effective_wavelength, 299792.458 / 300)
    wheel = response_set()
    wheel.add_special("X_delta_300")
    assert wheel["X_delta_300"].isdelta
    np.testing.assert_allclose(wheel["X_delta_300"].effective_wavelength, 299792.458 / 300)
    np.testing.assert_allclose(wheel["X_delta_300"](lambda w: np.asarray(w) * 2.0), 2.0 * 299792.458 / 300)


# ---------------------------------------------------------------- likelihood host logic
def _like(**kw):
    from mbb_emcee_amd import likelihood
    return likelihood(**kw)


def test_likelihood_defaults_and_setters():
    """likelihood.py:73, :83-85, :227-229 and the accessor family :378-641"""
    like = _like()
    np.testing.assert_array_equal(like.lowlims, [1, 0.1, 1, 0.1, 1e-3])
    assert like.has_uplims == [False, True, False, True, False, False]
    assert like.uplim("beta") == 20.0 and like.uplim("T") is None
    like.set_phot([250.0, 350.0, 500.0], [30.0, 40.0, 30.0], [3.0, 4.0, 3.0])
    assert like.ndata == 3 and like.has_uplim("lambda0") and like.uplim(2) == 1500.0   # 3 x max wave
    like.set_uplim("lambda_peak", 300.0)
    assert like.has_uplim(5) and like.uplim("peaklam") == 300.0
    like.set_gaussian_prior("T/(1+z)", 20.0, 2.0)
    assert like.has_gaussian_prior("t") and like.get_gaussian_prior(0) == (20.0, 2.0)
    assert like.gprior_ivars[0] == 0.25 and like.get_gaussian_prior("beta") is None
    like.set_lowlim("F500", 0.5)
    assert like.lowlim(4) == 0.5 and like.get_paramindex("alpha") == 3
    assert like._check_lowlim([2, 1, 2, 1, 1]) and not like._check_lowlim([0.5, 1, 2, 1, 1])
    with pytest.raises(ValueError):
        like._check_lowlim([1, 2, 3])
    f, w, o = like.band_tables()
    np.testing.assert_allclose(f, 299792.458 / np.array([250.0, 350.0, 500.0]))
    assert np.all(w == 1) and list(o) == [0, 1, 2, 3]
    with pytest.raises(ValueError):
        like.set_phot([250.0], [1.0, 2.0], [1.0])
    like.set_phot([250.0, 350.0, 500.0], [30.0, 40.0, 30.0], [3.0, 4.0, 3.0])
    with pytest.raises(ValueError):
        like.set_cov(np.eye(2))                      # ValueError, not NameError
    like.set_cov(np.diag([9.0, 16.0, 9.0]))
    np.testing.assert_allclose(like.data_flux_unc, [3, 4, 3])
    np.testing.assert_allclose(like.data_invcovmatrix, np.diag([1 / 9.0, 1 / 16.0, 1 / 9.0]))


def test_likelihood_response_mode_and_files(tmp_path):
    like = _like(response=True)
    with pytest.raises(ValueError):
        like.set_phot([250.0], [1.0], [1.0])          # names expected
    with pytest.raises(ValueError):
        like.set_phot(["NoSuch_band"], [1.0], [1.0])
    like.set_phot(["SPIRE_250um", "ALMA_alma_343", "X_box_850um_50"], [30, 5, 8], [3, 1, 1])
    assert like.response_names == ["SPIRE_250um", "ALMA_alma_343", "X_box_850um_50"]
    assert like.has_response("ALMA_alma_343")
    np.testing.assert_allclose(like.data_wave[0], 247.268656, atol=1e-4)
    f, w, o = like.band_tables()
    assert list(o) == [0, 189, 189 + 29, 189 + 29 + 11]
    # photometry file; lowlim[4] follows the data only through the constructor (Q8)
    pf = tmp_path / "phot.txt"
    pf.write_text("# wave flux err\n250 30.0 3.0\n350 40.0 4.0\n500 20.0 3.0\n")
    like2 = _like(photfile=str(pf))
    assert like2.ndata == 3 and like2.lowlim("fnorm") == pytest.approx(1e-3 * 20.0)
    like3 = _like()
    like3.read_phot(str(pf))
    assert like3.lowlim("fnorm") == 1e-3
    s = pickle.dumps(like2)                            # device context is not pickled
    like4 = pickle.loads(s)
    assert like4._ctx is None and like4.ndata == 3


def test_map_applies_a_foreign_function_row_by_row():
    """likelihood.map is a pool adaptor: only this object's own call (bare, bound, or inside an
    emcee-style wrapper without extra arguments) may be turned into one launch; any other function
    is the caller's posterior and is applied to every row (no GPU needed for that)."""
    like = _like()
    like.set_phot([250.0, 350.0], [30.0, 40.0], [3.0, 4.0])
    rows = [np.array([10.0 + i, 2.0, 600.0, 3.0, 40.0]) for i in range(4)]
    assert like.map(lambda r: -0.5 * r[0], rows) == [-5.0, -5.5, -6.0, -6.5]

    class Wrapper(object):                      # emcee 2's _function_wrapper with args=
        def __init__(self, f, args):
            self.f, self.args, self.kwargs = f, args, {}
        def __call__(self, x):
            return self.f(x, *self.args)
    calls = []
    w = Wrapper(lambda x, offset: calls.append(1) or x[0] + offset, (100.0,))
    assert like.map(w, rows) == [110.0, 111.0, 112.0, 113.0] and len(calls) == 4
    assert like.map(like, []) == []


def test_no_gpu_fails_loudly():
    """There is no CPU fallback: without a device the call raises."""
    from mbb_emcee_amd import _native
    lib = _native.load()
    if lib.mbb_device_count() > 0:
        pytest.skip("a GPU is present")
    like = _like()
    like.set_phot([250.0, 350.0], [30.0, 40.0], [3.0, 4.0])
    with pytest.raises(_native.NativeError):
        like(np.array([10.0, 2.0, 600.0, 3.0, 40.0]))


def _run_bench(argv, env_extra=None):
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MBB_BENCH_WORKER")}
    env.update(env_extra or {})
    if "MBB_BENCH_FAKE_WORKER" in env:
        env["MBB_BENCH_RANK_HOOK"] = os.path.join(ROOT, "tests", "_fake_bench_rank.py")
    pr = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env,
                        stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    out = [ln for ln in pr.stdout.decode().splitlines() if ln.strip()]
    return pr.returncode, out


@pytest.mark.parametrize("argv", [["--gpus", "2"], ["--gpus", "2", "--oversubscribe"], ["--gpus", "1"]])
def test_bench_starts_its_own_ranks_and_always_prints_one_line(argv):
    """`python bench.py --gpus N` as the driver runs it, without a launcher: the ranks are started as
    child processes (the role of emcee's pool, reference mbb_fit.py:80-81).  Without a GPU the run
    fails -- with exactly one JSON line on stdout and a non-zero status, never a bare traceback."""
    import json
    from mbb_emcee_amd import _native
    if _native.load().mbb_device_count() > 0:
        pytest.skip("a GPU is present: the GPU suite runs the real thing")
    rc, out = _run_bench(argv + ["--steps", "3", "--warmup", "1"])
    assert rc != 0
    assert len(out) == 1, out
    line = json.loads(out[0])
    assert line["value"] is None and line["n_gpus"] == int(argv[1]) and "error" in line
    assert line["metric"].startswith("walker-likelihood")


@pytest.mark.parametrize("how,rc_want,checks", [
    ("part", 0, lambda d: d["value"] == 2.0e6 and d["exchange_validation"]["rccl"] == "ok" and "ranks_ended_badly" not in d),
    ("part,crash1", 0, lambda d: d["value"] == 2.0e6 and d["ranks_ended_badly"] == {"1": 7}),
    ("hang1", 0, lambda d: d["value"] == 2.0e6 and d["ranks_ended_badly"] == {"1": -9}),
    ("noline,fail0", 3, lambda d: d["value"] is None and "without a line" in d["error"]),
])
def test_bench_supervisor_collects_merges_and_ends_ranks(how, rc_want, checks):
    """The process the driver starts for --gpus N supervises its ranks (no GPU touched; fake ranks here): rank
    0's measured line and the parts printed after it are merged into ONE line; a rank that crashes or hangs after
    the measurement is named beside the line and ended, and does not turn the measurement into a failure; a
    rank 0 that leaves without a line gives an error line and its status."""
    import json
    rc, out = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1"],
                         {"MBB_BENCH_FAKE_WORKER": how, "MBB_BENCH_GRACE_S": "1"})
    assert len(out) == 1, out
    line = json.loads(out[0])
    assert rc == rc_want and checks(line), (rc, line)
    assert line["n_gpus"] == 2


@pytest.mark.parametrize("how,rc_zero,checks", [
    # a peer wedged after the measurement: the value stands
    ("hang1", True, lambda d: d["value"] == 2.0e6 and d["supervisor_timeout"] and d["ranks_ended_badly"] == {"1": -9}),
    # rank 0 itself wedged in something optional after its line (the parts it had printed are kept)
    ("part,hang0", True, lambda d: d["value"] == 2.0e6 and d["supervisor_timeout"] and d["exchange_validation"]["rccl"] == "ok"),
    # wedged before anything was measured: an error line, not silence
    ("noline,hang0", False, lambda d: d["value"] is None and d["supervisor_timeout"] and "deadline" in d["error"]),
])
def test_bench_supervisor_deadline_ends_the_ranks_and_still_prints_the_line(how, rc_zero, checks):
    """Round 3's supervisor waited 1500 s by default -- longer than the driver's limit for the whole command --
    and printed only after every rank had ended.  Now the deadline (480 s by default, 6 s here) ends the ranks
    and the ONE line is printed from whatever rank 0 had handed over: the measured value if there was one, an
    error otherwise.  Worst cases of the defaults stay inside the driver's 600 s."""
    import json
    import time
    import bench
    assert bench.SUPERVISOR_DEADLINE_S <= 480 and bench.GUARD_S <= 45 and bench.OPTIONAL_UNTIL_S <= 240
    t0 = time.time()
    rc, out = _run_bench(["--gpus", "2", "--steps", "3", "--warmup", "1"],
                         {"MBB_BENCH_FAKE_WORKER": how, "MBB_BENCH_GRACE_S": "300", "MBB_BENCH_DEADLINE_S": "6"})
    assert time.time() - t0 < 60
    assert len(out) == 1, out
    line = json.loads(out[0])
    assert (rc == 0) == rc_zero and checks(line), (rc, line)
    assert line["supervisor_deadline_s"] == 6.0


REQUIRED_LINE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                      "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "full")


def _check_driver_line(raw, n_gpus):
    """What the driver must be able to parse: ONE line, strict JSON, far below the size that lost round 4's record."""
    import json
    import bench
    assert "\n" not in raw.strip() and len(raw.encode()) < bench.LINE_LIMIT, len(raw)

    def strict(c):
        raise ValueError("not strict JSON: " + c)
    d = json.loads(raw, parse_constant=strict)
    for k in REQUIRED_LINE_KEYS:
        assert k in d, k
    assert d["n_gpus"] == n_gpus and d["config"]["workload"] and d["dtype"].startswith("f64")
    assert d["metric"].startswith(json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"])
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in d["roofline"], k
    assert len(d["roofline"]["kernel"]) <= 80
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in d["cpu_baseline"], k
    return d


def test_bench_line_stays_short_whatever_was_measured():
    """Round 4's driver record has `parsed: null`: the one JSON line had grown to 20.7 KB.  The line is now built by
    `short_line` from everything measured and everything else goes to gpurun_out/bench_full.json.  Here: round 4's
    two whole lines (every leg populated) and the same with every string blown up and every number made awkward come
    out as one strict-JSON line below 4 KB with the keys the contract names."""
    import json
    import bench
    for f in ("bench_v5.json", "bench_driver_command.json"):
        full = json.load(open(os.path.join(ROOT, "profiles", "r04", f)))
        assert len(json.dumps(full)) > 15000
        d = _check_driver_line(json.dumps(bench.short_line(full), allow_nan=False, separators=(",", ":")), 1)
        assert d["value"] == pytest.approx(full["value"], rel=1e-6) and d["roofline"]["frac"] == pytest.approx(full["roofline"]["frac"], rel=1e-5)
        assert d["boundary_M1"]["p50_us"] == pytest.approx(full["boundary_M1"]["p50_us"], rel=1e-5)
        assert d["cpu_baseline"]["value"] == pytest.approx(full["cpu_baseline"]["value"], rel=1e-5)

        def blow(o):
            if isinstance(o, dict):
                return {k: blow(v) for k, v in o.items()}
            if isinstance(o, list):
                return [blow(v) for v in o] * 3
            if isinstance(o, str):
                return o * 40
            if isinstance(o, float):
                return o * (1.0 + 1.0 / 3.0)
            return o
        big = blow(full)
        big["roofline"]["counted"]["frac"] = float("nan")           # a non-finite number must not reach the line
        big["error"] = "x" * 100000
        big["exchange_validation"] = {m: {"ok": False, "why": "y" * 5000} for m in ("ipc", "ipc-launches", "rccl")}
        raw = json.dumps(bench.short_line(big), allow_nan=False, separators=(",", ":"))
        d = _check_driver_line(raw, 1)
        assert d["roofline"]["counted_frac"] is None


@pytest.mark.parametrize("world", [1, 8])
def test_bench_prints_the_short_line_and_the_side_file(world, tmp_path):
    """bench.py end to end with fake ranks that hand over a fully populated line (round 4's) and long parts: the last
    thing on stdout is the short line; the whole record is in the side file.  8 ranks: the supervisor keeps the parts
    out of the line (`exchange_validation` one word per exchange)."""
    import json
    import shutil
    # (the side file lands under the repo copy's gpurun_out/: run a copy so a test never touches the real one)
    root = tmp_path / "repo"
    root.mkdir()
    shutil.copy(os.path.join(ROOT, "bench.py"), root / "bench.py")
    shutil.copy(os.path.join(ROOT, "BASELINE.json"), root / "BASELINE.json")
    (root / "profiles" / "r04").mkdir(parents=True)
    shutil.copy(os.path.join(ROOT, "profiles", "r04", "bench_v5.json"), root / "profiles" / "r04" / "bench_v5.json")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MBB_BENCH_WORKER")}
    env.update({"MBB_BENCH_FAKE_WORKER": "full,part", "MBB_BENCH_GRACE_S": "5", "PYTHONPATH": ROOT,
                "MBB_BENCH_RANK_HOOK": os.path.join(ROOT, "tests", "_fake_bench_rank.py")})
    if world == 1:
        env["MBB_BENCH_WORKER_FAKE_TOP"] = "1"
    pr = subprocess.run([sys.executable, str(root / "bench.py"), "--gpus", str(world), "--steps", "20", "--warmup", "5"], env=env,
                        stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=300)
    assert pr.returncode == 0, pr.stderr.decode()[-2000:]
    out = [ln for ln in pr.stdout.decode().splitlines() if ln.strip()]
    assert len(out) == 1, out
    d = _check_driver_line(out[0], world)
    assert d["full"] == "gpurun_out/bench_full.json"
    full = json.load(open(root / "gpurun_out" / "bench_full.json"))
    assert "large_ensembles" in full and "postprocess" in full and "user_runs" in full and "configs" in full
    assert "large_ensembles" not in d and "postprocess" not in d
    if world > 1:
        assert d["exchange_validation"]["rccl"] == "ok" and d["exchange_validation"]["ipc"].startswith("rehearsal")
        assert len(d["exchange_validation"]["ipc"]) <= 70
        assert d["boundary_sharded"]["ok"] is True and d["boundary_sharded"]["rows_%d" % (125 * world)] == pytest.approx(31.123, rel=1e-4)
        assert len(full["exchange_validation"]["ipc"]["why"]) > 1000          # the side file keeps the detail


def test_bench_under_a_launcher_with_the_wrong_world_size_says_so():
    import json
    rc, out = _run_bench(["--gpus", "3"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0",
                                           "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29999"})
    assert rc == 2 and len(out) == 1
    assert "WORLD_SIZE=2" in json.loads(out[0])["error"]


def test_bench_roofline_lookups_name_kernels_the_library_has_and_the_committed_summaries_hold():
    """Round 3's driver line carried `roofline.frac = null`: a template parameter added to k_flowm after bench.py's
    literal kernel name was written.  For every kernel bench.py looks up in a committed counter summary: the name
    it derives is an instantiation the built library really has (nm), the newest committed summaries hold it
    where the default bench command needs them (form 7, the plain launch, cfg5, cfg1, cfg4), and the roofline
    objects come out finite; where a summary lacks a kernel the object says so in `error`."""
    import bench
    lib = os.path.join(ROOT, "mbb_emcee_amd", "libmbb_hip.so")
    syms = subprocess.run(["nm", "-C", lib], capture_output=True, text=True, check=True).stdout
    have = set(re.findall(r"\bvoid (k_(?:lnlike|flowm|flowa)<[^>]*>)\(LikeArgs\)", syms))
    assert len(have) >= 40
    keys = [bench.kernel_key(7, pairs=1), bench.kernel_key(9), bench.kernel_key(1),
            bench.kernel_key(0), bench.kernel_key(0, staged=False), bench.kernel_key(7, opthin=True, pairs=1),
            bench.kernel_key(0, opthin=True)]
    for key in keys:
        assert any(key in h for h in have), "bench.py would look for %r, the library has no such kernel" % key
    # the default command's headline: form 7, one pair per workgroup, staged tables -- nothing may be missing
    roof, hbm = bench.dominant_kernel_roofline(7, 1, True, 3.2, 20, "label", 2209, 8, 125)
    assert roof.get("error") is None, roof.get("error")
    for v in (roof["frac"], roof["achieved"], roof["counted"]["frac"], roof["traffic"], roof["valu_issue_frac"],
              roof["valu_issue_frac_useful"], hbm["traffic_ratio"]):
        assert v is not None and np.isfinite(v) and v > 0
    assert 0.02 < roof["counted"]["frac"] < roof["frac"] < 0.5
    assert os.path.exists(os.path.join(ROOT, roof["counters_source"]))
    assert os.path.exists(os.path.join(ROOT, hbm["traffic_source"]))
    # the other sampler forms: the algorithmic figure is always there; a missing summary is said, not hidden
    for form in (9, 1):
        r, h = bench.dominant_kernel_roofline(form, 1, True, 5.0, 20, "label", 2209, 8, 125)
        assert np.isfinite(r["frac"]) and r["frac"] > 0
        assert (r["counted"]["frac"] is not None) or r.get("error")
    r, h = bench.dominant_kernel_roofline(7, 3, True, 5.0, 20, "label", 2209, 8, 125)      # no such instantiation
    assert np.isfinite(r["frac"]) and r["traffic"] is None and "no kernel matching" in r["error"]
    # cfg5, cfg1, cfg4
    pm5, src5, err5 = bench.measured_valu("pmc_valu_cfg5*.json", bench.kernel_key(0, staged=False))
    assert pm5 and not err5 and bench.valu_roofline(pm5, src5, 1.3e-3, "cfg5")["frac"] > 0
    for cfg, thin in (("cfg1", True), ("cfg4", False)):
        for key in (bench.kernel_key(0, opthin=thin), bench.kernel_key(7, opthin=thin, pairs=1)):
            pm, src, err = bench.measured_valu("pmc_valu_%s.json" % cfg, key)
            assert pm and not err, (cfg, key, err)


def test_committed_parity_report_is_of_these_sources():
    """profiles/rNN/parity_report.json (the observed maxima DESIGN.md section 2 quotes) must have been
    measured on the kernel sources of the tree it is committed in: the report carries a hash of
    everything the device library is built from, and a change to any of it without a fresh GPU run
    of the suite (copy gpurun_out/parity_report.json over it) fails here."""
    import glob
    import json
    from conftest import kernel_sources_sha16
    reports = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "parity_report.json")))
    assert reports, "no committed parity report"
    rep = json.load(open(reports[-1]))
    assert rep.get("exitstatus") == 0 and rep.get("tests_failed", 0) == 0, reports[-1]
    assert rep.get("kernel_sources_sha16") == kernel_sources_sha16(), \
        "%s was measured on other kernel sources than this tree's: rerun `pytest -m gpu` on the GPU box and " \
        "commit its gpurun_out/parity_report.json" % os.path.relpath(reports[-1], ROOT)
    assert rep.get("tests_run", 0) >= 140 and rep.get("tests_with_tolerance_checks", 0) >= 50


def test_c_abi_exports_every_declared_symbol():
    """Every function declared in include/mbb_hip.h is exported by the in-tree
    library and bound with a signature in _native.SIGNATURES."""
    from mbb_emcee_amd import _native
    hdr = open(os.path.join(ROOT, "include", "mbb_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(mbb_[a-z0-9_]+)\s*\(", hdr))
    assert len(declared) >= 30
    lib = _native.load()
    for name in declared:
        assert hasattr(lib, name), name
    assert declared == set(_native.SIGNATURES), declared ^ set(_native.SIGNATURES)
    assert lib.mbb_last_error() is not None


def test_oracle_is_not_reachable_from_the_product():
    """The product package never imports the oracle."""
    pkg = os.path.join(ROOT, "mbb_emcee_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, fn)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), fn
                assert "libmbb_oracle" not in src and "mbbo_" not in src, fn


# ---------------------------------------------------------------- fit driver
def test_generate_initial_values_respects_limits():
    """mbb_fit.py:362-479"""
    from mbb_emcee_amd import mbb_fitter
    fit = mbb_fitter(nwalkers=200, seed=5)
    fit.set_data([250.0, 350.0, 500.0], [30.0, 40.0, 30.0], [3.0, 4.0, 3.0])
    fit.fix_param("alpha")
    fit.set_uplim("T", 30.0)
    p0 = fit.generate_initial_values(np.array([10.0, 2.0, 2500.0, 4.0, 40.0]),
                                     np.array([2, 0.2, 100, 0.3, 5.0]))
    assert p0.shape == (200, 5)
    assert np.all(p0[:, 3] == 4.0)                       # fixed: no scatter
    assert np.all(p0[:, 2] <= 1500.0) and np.all(p0[:, 2] >= 1)   # lambda0 pulled inside 3x500
    assert abs(np.median(p0[:, 2]) - 1300.0) < 40        # upper limit - 2 sigma
    assert np.all(p0[:, 0] <= 30.0) and np.all(p0[:, 0] >= 1) and np.all(p0[:, 1] >= 0.1)
    fit.fix_param("lambda0")
    with pytest.raises(ValueError):                      # fixed and outside limits
        fit.generate_initial_values(np.array([10.0, 2.0, 2500.0, 4.0, 40.0]), np.ones(5))
    fit.unfix_param(2)
    with pytest.raises(ValueError):
        fit.generate_initial_values(np.ones(4), np.ones(5))
    # run() validation happens before any lnprob evaluation (mbb_fit.py:510-522)
    bad = p0.copy(); bad[0, 1] = 25.0
    with pytest.raises(ValueError):
        fit.run(1, 1, bad)
    fit2 = mbb_fitter(nwalkers=10)
    with pytest.raises(Exception):
        fit2.run(1, 1, p0[:10])                          # no data


def test_ensemble_sampler_gaussian():
    """Stretch move samples a correlated Gaussian; row-wise, vectorised and
    pool-mapped lnprob give identical chains for the same seed."""
    from mbb_emcee_amd import EnsembleSampler
    icov = np.linalg.inv(np.array([[2.0, 0.6], [0.6, 0.5]]))

    def lnp_row(x):
        return -0.5 * x.dot(icov).dot(x)

    def lnp_vec(x):
        return -0.5 * np.einsum("ni,ij,nj->n", x, icov, x)

    class Pool(object):
        def map(self, f, rows):
            return [f(r) for r in rows]

    p0 = np.random.RandomState(2).normal(0, 1, (40, 2))
    chains = []
    for kw in (dict(), dict(vectorize=True), dict(pool=Pool())):
        s = EnsembleSampler(40, 2, lnp_vec if kw.get("vectorize") else lnp_row, seed=9, **kw)
        pos, lnp, state = s.run_mcmc(p0, 400)
        chains.append(s.chain.copy())
        assert s.chain.shape == (40, 400, 2) and s.lnprobability.shape == (40, 400)
        assert np.allclose(lnp, lnp_vec(pos))
    assert np.allclose(chains[0], chains[1]) and np.allclose(chains[0], chains[2])
    flat = chains[0][:, 100:, :].reshape(-1, 2)
    assert np.allclose(np.cov(flat.T), [[2.0, 0.6], [0.6, 0.5]], atol=0.25)
    assert 0.3 < s.acceptance_fraction.mean() < 0.9
    assert np.all(np.isfinite(s.acor))
    s.reset()
    assert s.chain.shape == (40, 0, 2)
    with pytest.raises(ValueError):
        EnsembleSampler(3, 2, lnp_row)
    with pytest.raises(ValueError):
        EnsembleSampler(40, 2, lambda x: np.nan).run_mcmc(p0, 1)


def test_block_bounds():
    from mbb_emcee_amd.parallel import block_bounds
    per, b = block_bounds(2000, 8)
    assert per == 250 and b[0] == (0, 250) and b[7] == (1750, 2000)
    per, b = block_bounds(10, 4)
    assert per == 3 and b == [(0, 3), (3, 6), (6, 9), (9, 10)]
    per, b = block_bounds(2, 4)
    assert per == 1 and b == [(0, 1), (1, 2), (2, 2), (2, 2)]


def test_sharded_two_ranks_gloo():
    """world_size 2 over gloo: partition, pad, all-gather, reassemble and a whole
    sharded sampler run equal the single-process result bit for bit."""
    import tempfile
    d = tempfile.mkdtemp(prefix="mbb_gloo_")
    env = dict(os.environ, OMP_NUM_THREADS="1", WORLD_SIZE="2", MBB_TEST_RDZV_FILE=os.path.join(d, "rdzv"),
               GLOO_SOCKET_IFNAME="lo")
    for k in ("MASTER_PORT", "MASTER_ADDR", "LOCAL_RANK"):
        env.pop(k, None)
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dist_worker.py")], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    texts = []
    for pr in procs:
        try:
            texts.append(pr.communicate(timeout=300)[0].decode(errors="replace"))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()                                              # exactly the processes started above
            raise AssertionError("a rank did not finish within 300 s")
    assert all(pr.returncode == 0 for pr in procs), "\n".join(t[-3000:] for t in texts)
    assert "DIST_OK" in texts[0]


def test_sharded_three_ranks_file_side_channel():
    """The same sharding logic with THREE ranks (ragged last block, fewer rows than ranks, a -inf row in the last
    block, a whole sharded sampler run) over the side channel the multi-process GPU tests use -- ranks started by the
    test itself, meeting through files, no torch and no port (tests/_filecomm.py) -- and that harness's own
    behaviour: a rank that fails ends the others' waits and the launcher reports it."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from _filecomm import launch
    ok, text = launch(os.path.join(ROOT, "tests", "_filecomm_worker.py"), 3, extra_env={"OMP_NUM_THREADS": "1"}, timeout=300.0)
    assert ok and all("FILECOMM_OK %d" % r in text for r in range(3)), text[-4000:]
    # a rank that dies: the others are told (the abort mark), nobody waits for the timeout
    import tempfile, textwrap, time
    d = tempfile.mkdtemp()
    bad = os.path.join(d, "bad_rank.py")
    open(bad, "w").write(textwrap.dedent("""
        import os, sys
        sys.path.insert(0, %r)
        from _filecomm import FileComm
        c = FileComm(timeout=60.0)
        c.barrier()
        if c.rank == 1:
            os._exit(5)
        c.barrier()
    """ % os.path.join(ROOT, "tests")))
    t0 = time.time()
    ok, text = launch(bad, 3, timeout=120.0)
    assert not ok and "status 5" in text and "another rank gave up" in text and time.time() - t0 < 40.0, text[-3000:]


# ------------------------------------------------- host-only table builders (C++), sanitizers
def _hosttables_lib():
    import _layout_checks as LC
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "libmbb_hosttables.so"])
    return LC, LC.load(os.path.join(ROOT, "oracle", "libmbb_hosttables.so"))


def test_band_layout_properties():
    """build_band_layout (mbb_emcee_amd/csrc/mbb_host_tables.cpp; what mbb_set_bands uploads):
    the 18 band sizes around the row / chunk boundaries of the GPU test, delta-function
    photometry, and 1000 random layouts -- every sample appears exactly once with its weight
    and log, the units tile the chunks, the bands' slot ranges tile [0, npart), and every
    band's slots sum to exactly the band's samples when the table is walked as the kernel
    walks it."""
    LC, lib = _hosttables_lib()
    assert LC.run_all(lib, nrandom=1000) >= 1040


def test_band_layout_rejects_bad_tables():
    LC, lib = _hosttables_lib()
    import ctypes as C
    f = np.array([10.0, 20.0, -1.0]); w = np.ones(3)
    cnt = np.zeros(9, dtype=np.int32); buf = np.zeros(640); ib = np.zeros(64, dtype=np.int32)

    def call(freq, offsets, nb):
        o = np.asarray(offsets, dtype=np.int32)
        return lib.mbbh_band_layout(freq.ctypes.data_as(LC._dp), w.ctypes.data_as(LC._dp), o.ctypes.data_as(LC._ip),
                                    nb, 4, 1, cnt.ctypes.data_as(LC._ip), buf.ctypes.data_as(LC._dp),
                                    buf.ctypes.data_as(LC._dp), buf.ctypes.data_as(LC._dp), ib.ctypes.data_as(LC._ip),
                                    ib.ctypes.data_as(LC._ip), ib.ctypes.data_as(LC._ip), 3, 4)
    assert call(f, [0, 3], 1) == -4                 # non-positive frequency
    assert call(f, [1, 3], 1) == -2                 # offsets[0] != 0
    assert call(f, [0, 2, 2], 2) == -3              # empty band
    assert call(f, [0, 2], 0) == -1
    assert call(np.array([5.0, 6.0, 7.0]), [0, 3], 1) == 0


def test_poly_tables_accuracy():
    """The piecewise degree-7 polynomials of b(x) = x/expm1(x) and C(y) = 1 - e^-y the sample loop evaluates
    (mbb_math.hip.h, polyrow_eval: row = int(8x), t = fract(8x), rows kPolyStride doubles apart), rebuilt on the host
    and evaluated by the same Horner recurrence in float64, against numpy longdouble: <= 4 ulp everywhere on [0, 48]
    and [0, 37], interval edges included -- RELATIVE error, down to arguments of 1e-15, where C ~ y (row 0 of C is t
    times the degree-6 interpolant of C/y: its coefficient of t^0 is an exact zero, and it is good to 3e-15)."""
    LC, lib = _hosttables_lib()
    import ctypes as C
    nb_, nc_, k_ = C.c_int(), C.c_int(), C.c_int()
    lib.mbbh_poly_counts(C.byref(nb_), C.byref(nc_), C.byref(k_))
    assert (nb_.value, nc_.value, k_.value) == (385, 297, 10)
    b = np.zeros((nb_.value, k_.value)); c = np.zeros((nc_.value, k_.value))
    lib.mbbh_poly_tables(b.ctypes.data_as(LC._dp), c.ctypes.data_as(LC._dp))
    assert not b[:, 8:].any() and not c[:, 8:].any()                      # (the padding)
    assert b[0, 0] == 1.0 and c[0, 0] == 0.0
    LD = np.longdouble
    rng = np.random.RandomState(9)
    ulp = 1.1102230246251565e-16

    def horner(tab, x):
        X = 8.0 * x
        i = np.floor(X).astype(np.int64)
        t = X - np.floor(X)
        p = tab[i, 7]
        for k in range(6, -1, -1):
            p = p * t + tab[i, k]
        return p, i
    for tab, xmax, fun, row0 in ((b, 48.0, lambda v: v / np.expm1(v), 4 * ulp), (c, 37.0, lambda v: -np.expm1(-v), 3e-15)):
        edges = np.arange(1, int(8 * xmax)) / 8.0
        x = np.concatenate([rng.uniform(0, xmax, 20000), 10.0 ** rng.uniform(-15, 0, 4000), edges,
                            np.nextafter(edges, 0), np.nextafter(edges, 100), [xmax, 1e-300]])
        ref = fun(x.astype(LD))
        got, row = horner(tab, x)
        err = np.abs((got.astype(LD) - ref) / ref).astype(np.float64)
        rest = row > 0
        assert err[rest].max() < 4 * ulp, (xmax, err[rest].max(), x[rest][err[rest].argmax()])
        assert err[~rest].max() < row0, (xmax, err[~rest].max(), x[~rest][err[~rest].argmax()])


def test_sanitizers():
    """AddressSanitizer + UBSan over the host-only C++ of the product (band layout, polynomial
    tables) and over the oracle: `make -C oracle asan`, then tests/_asan_worker.py with the
    sanitizer runtimes preloaded (sanitizers run on the CPU build only)."""
    libs = [subprocess.check_output(["gcc", "-print-file-name=" + n]).decode().strip()
            for n in ("libasan.so", "libubsan.so")]
    if not all(os.path.isabs(p) and os.path.exists(p) for p in libs):
        pytest.skip("gcc's sanitizer runtimes are not installed")
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "asan"])
    env = dict(os.environ, LD_PRELOAD=":".join(libs), ASAN_OPTIONS="detect_leaks=0:abort_on_error=0",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1", MBB_ASAN_LAYOUTS="300")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_asan_worker.py")], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    text = out.stdout.decode(errors="replace")
    assert out.returncode == 0 and "ASAN_OK" in text and "ERROR: AddressSanitizer" not in text \
        and "runtime error" not in text, text[-3000:]


# ------------------------------------------------- dust mass (results.py:726-801), parity unpinned
def test_dustmass_identities():
    """postprocess.dustmass restates results.py:746-801, which cannot be imported here
    (astropy.units): PARITY UNPINNED.  Held instead to what the formula means, computed by
    a different route: (1) optically thin, the mass reproduces the flux density it was
    derived from, S = M kappa_nu B_nu(T_rest) (1+z) / D_L^2, with the Planck function written
    with exp, cgs constants spelled out here; (2) thick / thin = tau / (1 - e^-tau) with
    tau = (lambda0 / wavenorm)^beta, -> 1 for tau -> 0 and -> tau for large tau;
    (3) scalings with D_L, fnorm and kappa."""
    import mbb_emcee_amd as mbb
    from mbb_emcee_amd import postprocess as pp
    rng = np.random.RandomState(8)
    chain = np.column_stack([rng.uniform(15, 60, 200), rng.uniform(1.0, 2.5, 200), rng.uniform(50, 400, 200),
                             rng.uniform(2, 4, 200), rng.uniform(5, 80, 200)]).reshape(20, 10, 5)
    z, dl, kappa, kw = 2.3, 18900.0, 2.64, 125.0
    thin = mbb.likelihood(opthin=True, wavenorm=850.0)
    thick = mbb.likelihood(opthin=False, wavenorm=850.0)
    m_thin = pp.dustmass(thin, chain, z, dl, kappa=kappa, kappa_wave=kw)
    m_thick = pp.dustmass(thick, chain, z, dl, kappa=kappa, kappa_wave=kw)
    assert m_thin.shape == (20, 10)
    # (1) implied flux density, cgs
    h, k, c = 6.6260693e-27, 1.38065e-16, 2.99792458e10
    nu = c / (850.0e-4 / (1 + z))                                 # rest-frame Hz
    T = chain[..., 0] * (1 + z)
    B = 2 * h * nu ** 3 / c ** 2 / (np.exp(h * nu / (k * T)) - 1.0)
    kap = 10.0 * kappa * (kw / (850.0 / (1 + z))) ** chain[..., 1]        # cm^2/g at nu, kappa ~ nu^beta
    S = m_thin * 1.97792e41 * kap * B * (1 + z) / (dl * 3.0856775814913673e24) ** 2
    np.testing.assert_allclose(S / 1e-26, chain[..., 4], rtol=1e-12)
    # (2) the optical-depth correction
    tau = (chain[..., 2] / 850.0) ** chain[..., 1]
    np.testing.assert_allclose(m_thick / m_thin, tau / (1.0 - np.exp(-tau)), rtol=1e-12)
    small = chain.copy(); small[..., 2] = 1e-3                    # tau -> 0
    np.testing.assert_allclose(pp.dustmass(thick, small, z, dl), pp.dustmass(thin, small, z, dl), rtol=1e-4)
    big = chain.copy(); big[..., 2] = 850.0 * 60.0                # tau >> 1
    tb = 60.0 ** big[..., 1]
    np.testing.assert_allclose(pp.dustmass(thick, big, z, dl) / pp.dustmass(thin, big, z, dl), tb, rtol=1e-12)
    # (3) scalings
    np.testing.assert_allclose(pp.dustmass(thin, chain, z, 2 * dl), 4 * m_thin, rtol=1e-14)
    c2 = chain.copy(); c2[..., 4] *= 3.0
    np.testing.assert_allclose(pp.dustmass(thin, c2, z, dl), 3 * m_thin, rtol=1e-14)
    np.testing.assert_allclose(pp.dustmass(thin, chain, z, dl, kappa=2 * kappa), m_thin / 2, rtol=1e-14)
    with pytest.raises(ValueError):
        pp.dustmass(thin, chain, z, dl, kappa=0.0)


def test_flow_protocol_model():
    """The hand-over protocol of the one-launch look-ahead sampler run (k_lnlike SMODE 5), restated
    on the host with the kernel's own index arithmetic (mbb_flow_index.h) and run in random and in
    adversarial order: with the lag guard no slot is ever reused under a reader and every run
    completes; a stalled worker stops what depends on it and nothing else goes wrong; without the
    guard the same adversary does get a slot overwritten -- the model can see what the guard is for."""
    import random
    import _flow_model as FM
    LC, lib = _hosttables_lib()
    ix = FM.Index(lib)
    assert ix.slots == 4 and ix.lag == 4
    # index arithmetic: moves completed before half-step j, and the word after move m
    for h in (0, 1):
        for j in range(-1, 40):
            assert ix.cnt(h, j) == len([g for g in range(h, j, 2)]) if j > h else ix.cnt(h, j) == 0
        for m in range(1, 20):
            assert ix.seq(h, m) == (h + 2 * (m - 1)) + 1
    # random schedules, random partners
    for seed in range(12):
        rng = random.Random(seed)
        lead, events, finished = FM.run(ix, n2=rng.choice((3, 5, 8)), nsteps=14, rng=rng)
        assert finished and lead <= ix.lag + 2, (seed, lead)
    # adversary: nobody ever needs row 0 as a partner, and row 0's worker sleeps through half-step 4
    # for a long time: the others run ahead of row 0 until the guard stops them; when the sleeper
    # wakes up everything it reads is still there, and the run completes
    n2 = 4
    avoid0 = lambda r, j: 1 + (r + j) % (n2 - 1)
    lead, events, finished = FM.run(ix, n2, 14, random.Random(1), partner=avoid0, stall=(0, 4, 5000))
    assert finished and lead <= ix.lag + 2
    # the same without the guard: they run on, and the sleeper wakes up to slots that were reused
    with pytest.raises(FM.Violation):
        FM.run(ix, n2, 14, random.Random(1), guard=False, partner=avoid0, stall=(0, 4, 5000))


def test_flowm_protocol_model():
    """The hand-over protocol of sampler form 7 (k_flowm: one workgroup per pair of walkers and candidate,
    constructor / quadrature / accept test in waves of their own), restated on the host with the kernel's
    index arithmetic and constants (mbb_flow_index.h) and run in random and adversarial order, every store
    landing at a random later time: with the lag guard no slot -- proposal, row, decision word -- is
    reused under a reader and every run completes; a stalled constructor wave stops what depends on it
    and nothing else; without the guard the same adversary gets a slot overwritten."""
    import random
    import _flow_model as FM
    import _flowm_model as F7
    LC, lib = _hosttables_lib()
    ix = FM.Index(lib)
    slots, lag, ring, nc, nb = F7.consts(lib)
    assert (slots, lag, ring, nc, nb) == (4, 4, 8, 3, 4)
    assert lag <= 2 * slots - 4 and ring >= 2 * lag and (ring & (ring - 1)) == 0     # what the kernel's comments derive
    for seed in range(8):
        rng = random.Random(seed)
        lead, events, finished = F7.run(ix, lib, n2=rng.choice((2, 3, 5)), nsteps=10, rng=rng)
        assert finished and lead <= lag + nb, (seed, lead)
    # adversary: nobody needs pair 0 as a partner; its candidate-0 workgroup's constructor sleeps through
    # half-step 4: the others run ahead until the guard stops them, the sleeper wakes up to intact slots
    n2 = 4
    avoid0 = lambda r, j: 1 + (r + j) % (n2 - 1)
    lead, events, finished = F7.run(ix, lib, n2, 10, random.Random(1), partner=avoid0, stall=((0, 0), 4, 20000))
    assert finished and lead <= lag + nb
    with pytest.raises(FM.Violation):
        F7.run(ix, lib, n2, 10, random.Random(1), guard=False, partner=avoid0, stall=((0, 0), 4, 20000))


def test_resident_forms_protocol_model():
    """The hand-over protocol of the resident sampler forms (k_flowa, form 9 -- and round 4's form 8, which shared it: workgroups owning
    several walkers of each half, rows / proposals / decision words filed under the move number mod four slots),
    restated on the host with the kernel's index arithmetic and constants and run in random and adversarial order,
    every store landing at a random later time: with the lag guard no slot is reused under a reader and every run
    completes, whatever the number of walkers per workgroup; without it a stalled workgroup gets one overwritten."""
    import random
    import _flow_model as FM
    import _flowa_model as FA
    import _flowm_model as F7
    LC, lib = _hosttables_lib()
    ix = FM.Index(lib)
    slots, lag, ring, _, _ = F7.consts(lib)
    # a row version written in half-step j is read (form 9) by the constructors of j + 3 and rewritten in j + 2 slots
    assert lag <= 2 * slots - 3 and ring >= 2 * lag
    for ahead in (True, False):
        for seed in range(8):
            rng = random.Random(seed)
            n2, W = rng.choice(((2, 1), (3, 2), (5, 2), (7, 3), (9, 8)))
            lead, events, finished = FA.run(ix, lib, n2, W, nsteps=10, rng=rng, ahead=ahead)
            assert finished and lead <= lag + 2, (ahead, seed, lead)
        # adversary: nobody draws workgroup 0's walkers as partners, and its constructor sleeps through half-step 4:
        # the others run ahead until the guard stops them, the sleeper wakes up to intact slots
        n2 = 4
        avoid0 = lambda r, j: 1 + (r + j) % (n2 - 1)
        lead, events, finished = FA.run(ix, lib, n2, 1, 10, random.Random(1), ahead=ahead, partner=avoid0, stall=(0, 4, 20000))
        assert finished and lead <= lag + 2
        with pytest.raises(FM.Violation):
            FA.run(ix, lib, n2, 1, 10, random.Random(1), ahead=ahead, guard=False, partner=avoid0, stall=(0, 4, 20000))


def test_every_option_is_documented_in_the_header():
    """mbb_set_option's names (the strcmp chain in mbb_hip.hip) all appear in include/mbb_hip.h."""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, "mbb_emcee_amd", "csrc", "mbb_hip.hip")).read()
    hdr = open(os.path.join(root, "include", "mbb_hip.h")).read()
    opts = set(re.findall(r'!strcmp\(name, "([a-z_0-9]+)"\)\) c->opt', src))
    assert len(opts) >= 20
    assert [o for o in sorted(opts) if '"%s"' % o not in hdr] == []


def test_fastcall_extension_with_a_stand_in_for_the_native_call():
    """mbb_emcee_amd/csrc/mbb_fastcall.c (the `_mbbfast` extension: likelihood.__call__'s boundary call as one C-level
    callable): rows into the block, the native entry called with the row count, results out into a fresh array -- with a
    ctypes callback standing in for mbb_lnlike_call (no GPU here); everything it does not take gives None."""
    import ctypes as C
    from mbb_emcee_amd import build as B
    if B.build_fastcall() is None:
        pytest.skip("no host compiler or no Python / numpy headers here")
    import importlib
    lk = importlib.import_module("mbb_emcee_amd.likelihood")
    ext = lk._mbbfast if lk._mbbfast is not None else importlib.import_module("mbb_emcee_amd._mbbfast")
    rows = np.zeros((256, 5)); lnl = np.zeros(256); calls = []

    def native(h, n):
        calls.append((h, n)); lnl[:n] = rows[:n].sum(axis=1)
        return 0 if rows[0, 0] >= 0 else 7
    cb = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int)(native)
    gen = np.array([5], dtype=np.uint64)
    fc = ext.FastCall(C.cast(cb, C.c_void_p).value, 1234, rows.ctypes.data, lnl.ctypes.data, 256, gen.ctypes.data, 5)
    p = np.random.RandomState(0).rand(40, 5)
    r = fc(p)
    assert type(r) is np.ndarray and r.dtype == np.float64 and r.base is None and np.array_equal(r, p.sum(axis=1))
    assert calls == [(1234, 40)]
    one = fc(p[3])
    assert type(one) is float and one == p[3].sum() and calls[-1] == (1234, 1)
    lnl[:] = -1.0
    assert np.array_equal(r, p.sum(axis=1))                           # (the result was a copy)
    n_before = len(calls)
    for other in (p.astype(np.float32), np.asfortranarray(p), p[::2], list(p[0]), np.zeros((300, 5)), np.zeros((3, 4)),
                  np.zeros((0, 5)), np.zeros(4), p.reshape(2, 20, 5), None):
        assert fc(other) is None
    assert len(calls) == n_before                                     # none of them reached the native call
    bad = p.copy(); bad[0, 0] = -1.0
    assert fc(bad) is None and calls[-1] == (1234, 40)                # a non-zero return: the caller takes the general path
    # the blocks made anew by some other call on the context (the generation word moved): nothing is written through the
    # stale addresses, nothing is called, None sends the caller for the buffers again
    rows[:] = 0.0; n_before = len(calls)
    gen[0] = 6
    assert fc(p) is None and len(calls) == n_before and not rows.any()
    gen[0] = 5
    assert np.array_equal(fc(p), p.sum(axis=1))
    with pytest.raises(TypeError):
        fc()
    with pytest.raises(ValueError):
        ext.FastCall(0, 1, 2, 3, 4, 5, 6)
    with pytest.raises(ValueError):
        ext.FastCall(1, 1, 2, 3, 4, 0, 6)



_REGISTRY_CHILD = r"""
import ctypes, os, sys
lib = ctypes.CDLL(sys.argv[1])
lib.mbbh_registry_join.argtypes = [ctypes.c_uint32]; lib.mbbh_registry_leave.argtypes = [ctypes.c_uint32]
lib.mbbh_registry_peers.argtypes = [ctypes.c_uint32, ctypes.c_int]
lib.mbbh_registry_busy.argtypes = [ctypes.c_uint32, ctypes.c_uint64, ctypes.c_uint64]
for line in sys.stdin:
    cmd, _, arg = line.strip().partition(" ")
    if cmd == "join": print(lib.mbbh_registry_join(int(arg)), flush=True)
    elif cmd == "leave": print(lib.mbbh_registry_leave(int(arg)), flush=True)
    elif cmd == "peers": print(lib.mbbh_registry_peers(int(arg), 0), flush=True)
    elif cmd == "busy":
        key, now = arg.split()
        print(lib.mbbh_registry_busy(int(key), int(now), 250), flush=True)
    elif cmd == "fork":
        # a child of fork() starts afresh: it is a process of its own in the table
        pid = os.fork()
        if pid == 0:
            lib.mbbh_registry_join(int(arg))
            os._exit(lib.mbbh_registry_peers(int(arg), 1))
        print(os.waitpid(pid, 0)[1] >> 8, flush=True)      # (answered once the child is gone for good: not even a zombie)
    elif cmd == "quit": break
"""


def test_device_registry_sees_other_processes_and_forgets_dead_ones(tmp_path):
    """mbb_emcee_amd/csrc/mbb_registry.cpp: which processes of this library hold contexts on which GPU, so that a
    resident server (k_serve) is started only by a process that has the device to itself (emcee's pool pattern,
    reference mbb_fit.py:80-81 threads > 1).  Real processes, a table of the test's own in /dev/shm: a second process
    on the same device is seen at once (the generation word), one on another device is not, a process that leaves is
    gone at once, one that is killed is gone at the next recount and its slot is free again, a fork()ed child is a
    process of its own."""
    import ctypes
    _, _ = _hosttables_lib()
    path = os.path.join(ROOT, "oracle", "libmbb_hosttables.so")
    name = "/mbb_hip_registry_test_%d" % os.getpid()
    os.environ["MBB_REGISTRY_NAME"] = name
    try:
        lib = ctypes.CDLL(path)
        lib.mbbh_registry_join.argtypes = [ctypes.c_uint32]; lib.mbbh_registry_leave.argtypes = [ctypes.c_uint32]
        lib.mbbh_registry_peers.argtypes = [ctypes.c_uint32, ctypes.c_int]
        lib.mbbh_registry_name.restype = ctypes.c_char_p
        K, K2 = 0x80000300, 0x80000400
        assert lib.mbbh_registry_join(K) == 1 and lib.mbbh_registry_join(K) == 2          # (two contexts of this process)
        assert lib.mbbh_registry_name().decode() == name and os.path.exists("/dev/shm" + name)
        assert lib.mbbh_registry_peers(K, 0) == 0

        def child():
            pr = subprocess.Popen([sys.executable, "-c", _REGISTRY_CHILD, path], stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                                  env=dict(os.environ), text=True)

            def ask(line):
                pr.stdin.write(line + "\n"); pr.stdin.flush()
                return int(pr.stdout.readline())
            return pr, ask
        b, ask_b = child()
        assert ask_b("join %d" % K2) == 1
        assert lib.mbbh_registry_peers(K, 0) == 0                                         # another device: not a peer
        assert ask_b("join %d" % K) == 1
        assert lib.mbbh_registry_peers(K, 0) == 1 and ask_b("peers %d" % K) == 1           # seen at once, both ways
        c, ask_c = child()
        assert ask_c("join %d" % K) == 1
        assert lib.mbbh_registry_peers(K, 0) == 2 and ask_b("peers %d" % K) == 2
        assert ask_b("fork %d" % K) == 3                                                   # the child of a fork: a fourth process
        assert lib.mbbh_registry_peers(K, 1) == 2                                          # ... and gone again when it exits
        assert ask_b("leave %d" % K) == 0
        assert lib.mbbh_registry_peers(K, 0) == 1                                          # left: gone at once
        c.kill(); c.wait()                                                                 # killed: no word from it
        assert lib.mbbh_registry_peers(K, 0) == 1                                          # (the cached answer still believes it)
        assert lib.mbbh_registry_peers(K, 1) == 0                                          # a recount does not
        assert lib.mbbh_registry_leave(K) == 1 and lib.mbbh_registry_leave(K) == 0 and lib.mbbh_registry_leave(K) == 0
        assert ask_b("join %d" % K) == 1 and ask_b("peers %d" % K) == 0                    # this process has withdrawn the key
        b.stdin.write("quit\n"); b.stdin.flush(); b.wait(timeout=30)
        # many processes come and go: slots of the dead are taken again, the table (256 slots) does not fill up
        for _ in range(3):
            kids = [child() for _ in range(8)]
            for pr, ask in kids:
                assert ask("join %d" % K) == 1
            assert lib.mbbh_registry_join(K) == 1 and lib.mbbh_registry_peers(K, 0) == 8
            for pr, ask in kids:
                pr.kill(); pr.wait()
            assert lib.mbbh_registry_peers(K, 1) == 0 and lib.mbbh_registry_leave(K) == 0
        # who is BUSY on the device (what a server's width goes by): processes that made a boundary call within the last
        # 250 ms of the clock they all read -- a process that holds a context and does not call (a pool's parent) is not
        lib.mbbh_registry_busy.argtypes = [ctypes.c_uint32, ctypes.c_uint64, ctypes.c_uint64]
        e, ask_e = child()
        f, ask_f = child()
        assert ask_e("join %d" % K) == 1 and ask_f("join %d" % K) == 1 and lib.mbbh_registry_join(K) == 1
        assert lib.mbbh_registry_busy(K, 1000, 250) == 0                                   # registered, none has called
        assert ask_e("busy %d 1001" % K) == 1                                              # (this process, a millisecond ago)
        assert lib.mbbh_registry_busy(K, 1002, 250) == 1 and ask_f("busy %d 1003" % K) == 2
        assert lib.mbbh_registry_busy(K, 1100, 250) == 2
        assert lib.mbbh_registry_busy(K, 1260, 250) == 0                                   # their calls are 250 ms old
        assert ask_e("busy %d 1270" % K) == 1 and lib.mbbh_registry_busy(K, 1271, 250) == 1
        assert lib.mbbh_registry_busy(K2, 1272, 250) == 0                                  # (not a device of this process)
        e.kill(); e.wait()
        assert lib.mbbh_registry_busy(K, 1600, 250) == 0                                   # gone: it stopped calling
        f.stdin.write("quit\n"); f.stdin.flush(); f.wait(timeout=30)
        assert lib.mbbh_registry_leave(K) == 0
        # a pid handed out again: a slot written by a process that is gone, whose pid now belongs to somebody else (here:
        # to a live child of this test, with a birth time that is not the one the slot holds) is not a peer and is free again
        import mmap, struct
        d, ask_d = child()
        assert ask_d("join %d" % K2) == 1 and lib.mbbh_registry_join(K) == 1
        with open("/dev/shm" + name, "r+b") as f:
            mm = mmap.mmap(f.fileno(), 0)
            slot_bytes, header, born_at = 208, 16, 200      # {pid i32, key u32[16], pad, beat u64[16], born u64}; {magic u32, pad, gen u64}
            assert (len(mm) - header) // slot_bytes == 256
            free = next(i for i in range(256) if struct.unpack_from("<i", mm, header + i * slot_bytes)[0] == 0)
            off = header + free * slot_bytes
            struct.pack_into("<i", mm, off, d.pid)
            struct.pack_into("<I", mm, off + 4, K)
            struct.pack_into("<Q", mm, off + born_at, 12345)     # (nobody alive was born then)
            assert lib.mbbh_registry_peers(K, 1) == 0
            assert struct.unpack_from("<i", mm, off)[0] == 0
            # ... the same slot with the birth time unknown: the pid alone answers
            struct.pack_into("<Q", mm, off + born_at, 0)
            struct.pack_into("<I", mm, off + 4, K)
            struct.pack_into("<i", mm, off, d.pid)
            assert lib.mbbh_registry_peers(K, 1) == 1
            struct.pack_into("<i", mm, off, 0); struct.pack_into("<I", mm, off + 4, 0)
            mm.close()
        d.kill(); d.wait()
        assert lib.mbbh_registry_leave(K) == 0
    finally:
        del os.environ["MBB_REGISTRY_NAME"]
        try:
            os.unlink("/dev/shm" + name)
        except OSError:
            pass


def test_package_level_helpers_of_the_reference():
    """The reference exports `alpha_merge_eqn` (modified_blackbody.py:122-151) and `isiterable` (utility.py) beside its
    classes; a caller that imports the names finds them here.  The merge equation on the host is zero at the merge
    points the reference itself found (tests/golden/sed.npz, made by the imported reference), follows the reference's
    formula term by term elsewhere, and takes the overflow of (x / x0)^beta as the reference does (b -> 0)."""
    import math
    import mbb_emcee_amd as mbb
    g = np.load(os.path.join(ROOT, "tests", "golden", "sed.npz"))
    pars, scal = g["pars"], g["thick_walpha/scalars"]
    for p, s in zip(pars, scal):
        xmerge, x0 = s[1], s[3]
        assert abs(mbb.alpha_merge_eqn(xmerge, p[3], p[1], x0)) < 1e-9 * max(1.0, xmerge)
        assert mbb.alpha_merge_eqn(0.5 * xmerge, p[3], p[1], x0) < 0 < mbb.alpha_merge_eqn(2.0 * xmerge, p[3], p[1], x0)
    rng = np.random.RandomState(4)
    for _ in range(2000):
        x, a, b, x0 = 10 ** rng.uniform(-2, 2), rng.uniform(0.1, 6), rng.uniform(0, 4), 10 ** rng.uniform(-2, 2)
        y = (x / x0) ** b
        try:
            bterm = y / math.expm1(y)
        except OverflowError:                     # (the reference: bterm = 0, :144-150)
            bterm = 0.0
        want = x - (1.0 - math.exp(-x)) * (3.0 + a + b * bterm)
        assert abs(mbb.alpha_merge_eqn(x, a, b, x0) - want) <= 2e-15 * max(1.0, abs(want))     # (1 - e^-x by expm1 here: a few ulp)
    assert mbb.alpha_merge_eqn(50.0, 3.0, 400.0, 0.01) == 50.0 - (1.0 - math.exp(-50.0)) * 6.0      # (x / x0)^beta overflows
    assert mbb.alpha_merge_eqn(2.0, 3.0, 1.5, 1.0, opthin=True) == mbb.alpha_merge_eqn(2.0, 3.0, 1.5, 1.0)
    assert [mbb.isiterable(o) for o in (np.array(3.0), [1], 3, np.zeros(3), "ab", (x for x in ()), None)] == \
        [False, True, False, True, True, True, False]
