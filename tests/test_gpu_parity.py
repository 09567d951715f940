"""Parity of the HIP path (through the C-ABI) with the golden fixtures made by
the reference and with the CPU oracle.  Needs an MI355X: `pytest -m gpu`.

Tolerances: band flux / f_nu rtol 1e-12, xmerge atol 1e-10,
lnL |d| <= 1e-10 max(1, |lnL|), -inf must match exactly.
"""
import os
import numpy as np
import pytest

from conftest import VARIANTS, golden_bands, lnl_close, rec_allclose, parity_record

pytestmark = pytest.mark.gpu

FLUX_RTOL = 1e-12
SED_RTOL = 1e-12


@pytest.fixture(scope="module")
def mbb():
    import mbb_emcee_amd
    return mbb_emcee_amd


@pytest.fixture(scope="module")
def ctx(mbb):
    from mbb_emcee_amd import _native
    return _native.default_context()


def relerr(a, b):
    return np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300))


def rel_below(kind, a, b, tol):
    """max relative error below tol, recorded in the parity report"""
    e = relerr(np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64))
    parity_record(kind + " (rel)", e, tol)
    assert e < tol, "%s: max relative error %g (tolerance %g)" % (kind, e, tol)


PATHS = ["launched", "served"]


def boundary_eval(like, pars, path):
    """likelihood.__call__ on float64 rows by one of the two ways the boundary call is evaluated:
    `launched` -- what a first call gets; `served` -- what a sampler's LOOP of calls gets after three in a row: the rows
    go to the resident kernel (k_serve) and no launch is made.  The served way needs several calls before it starts, so
    a test that calls once never sees it (round 4: a wrong-sums bug hid there): here it is switched on from the first
    boundary call (`serve` 2, `serve_after` 1), the rows go in batches of at most a row per CU, and the counters must
    say that every batch was answered by the resident kernel -- a silent fall-back to a launch cannot pass."""
    pars = np.ascontiguousarray(pars, dtype=np.float64)
    if path == "launched":
        # (k_lnlike: the general path's kernel -- model_flux, row status, batches beyond a row per CU)
        lnl, st = like._sync_device().lnlike_batch(np.atleast_2d(pars))
        return float(lnl[0]) if pars.ndim == 1 else lnl
    ctx = like._sync_device()
    saved_after = 3
    ctx.set_option("serve", 2); ctx.set_option("serve_after", 1)
    ctx.set_option("serve_budget_us", 100000)        # (a busy box must not turn a served request into a launch)
    cus = ctx.info("cu_count")
    req0, fb0 = ctx.info("serve_requests"), ctx.info("serve_fallbacks")
    one = pars.ndim == 1
    rows = pars[None, :] if one else pars
    out, batches = [], 0
    for i in range(0, rows.shape[0], cus):
        chunk = np.ascontiguousarray(rows[i:i + cus])
        got = like(chunk[0]) if one else like(chunk)
        out.append(np.atleast_1d(got))
        batches += 1
        assert ctx.info("serving") == 1, "batch %d was not served" % batches
    assert ctx.info("serve_requests") - req0 == batches and ctx.info("serve_fallbacks") == fb0, \
        (ctx.info("serve_requests") - req0, batches, ctx.info("serve_fallbacks") - fb0)
    assert ctx.info("last_kernel_form") == 10        # (k_serve's launch was the context's last)
    ctx.set_option("serve", 1); ctx.set_option("serve_after", saved_after); ctx.set_option("serve_budget_us", 400)
    res = np.concatenate(out) if out else np.zeros(0)
    return float(res[0]) if one else res


# ------------------------------------------------------------------ G5: KATs
def test_kat_thick(mbb):
    """reference tests/test_modified_blackbody.py:6-20"""
    m = mbb.modified_blackbody(10.0, 2.0, 800.0, 2.0, 45.0)
    assert m.has_alpha and not m.optically_thin
    rec_allclose(m(500), 45.0, atol=1e-4, kind="reference KAT")
    wave = np.array([250.0, 350.0, 500.0, 850.0])
    rec_allclose(m(wave), [21.96268738, 39.53249977, 45.0, 22.06274444], rtol=1e-4, kind="reference KAT")


def test_kat_thin(mbb):
    """reference tests/test_modified_blackbody.py:23-36"""
    m = mbb.modified_blackbody(15.0, 1.8, 200.0, 3.0, 50.0, opthin=True)
    assert m.has_alpha and m.optically_thin and m.lambda0 is None
    wave = np.array([250.0, 350.0, 500.0, 850.0])
    rec_allclose(m(wave), [178.34976, 111.03026, 50.0, 10.880588], rtol=1e-4, kind="reference KAT")


def test_kat_thinthick(mbb):
    """reference tests/test_modified_blackbody.py:39-47"""
    wave = np.array([500.0, 850.0, 1100.0, 2500.0])
    a = mbb.modified_blackbody(15.0, 1.8, 5.0, 3.0, 50.0, opthin=True)
    b = mbb.modified_blackbody(15.0, 1.8, 5.0, 3.0, 50.0, opthin=False)
    rec_allclose(a(wave), b(wave), rtol=1e-3, kind="reference KAT")


def test_kat_merge(mbb):
    """reference tests/test_modified_blackbody.py:50-69"""
    mb = mbb.modified_blackbody
    assert mb(20.0, 1.9, None, 3.5, 50.0, noalpha=True, opthin=True).wavemerge is None
    rec_allclose(mb(20.0, 1.9, None, 3.5, 50.0, opthin=True).wavemerge, 85.66065, rtol=1e-3, kind="reference KAT")
    rec_allclose(mb(35.0, 2.2, None, 2.8, 50.0, opthin=True).wavemerge, 51.40211, rtol=1e-3, kind="reference KAT")
    assert mb(20.0, 1.9, 250.0, 3.5, 50.0, noalpha=True).wavemerge is None
    rec_allclose(mb(20.0, 1.9, 250.0, 3.5, 50.0).wavemerge, 109.5506829, rtol=1e-3, kind="reference KAT")
    rec_allclose(mb(40.0, 1.5, 600.0, 3.0, 50.0).wavemerge, 60.10021595, rtol=1e-3, kind="reference KAT")


def test_kat_spire250_flat(mbb):
    """reference tests/test_response.py:19-29 with the SED evaluated on the GPU"""
    wheel = mbb.response_set()
    # a (nearly) flat SED: alpha-law side of a hot thin body is not flat, so use
    # the callable path with a constant, which is host-only, plus a GPU SED check
    rec_allclose(wheel["SPIRE_250um"](lambda x: 1), 1.011046, atol=1e-4, kind="reference KAT")


def test_ctor_errors(mbb):
    """modified_blackbody.py:219-224"""
    with pytest.raises(ValueError):
        mbb.modified_blackbody(10.0, 2.0, 800.0, -1.0, 45.0)
    with pytest.raises(ValueError):
        mbb.modified_blackbody(10.0, -0.5, 800.0, 2.0, 45.0)
    mbb.modified_blackbody(10.0, 2.0, 800.0, -1.0, 45.0, noalpha=True)   # alpha ignored


# ------------------------------------------------- G2: constructor scalars
@pytest.mark.parametrize("name,opthin,noalpha", VARIANTS)
def test_prologue_scalars(ctx, g_sed, name, opthin, noalpha):
    pars = g_sed["pars"]
    ref = g_sed[name + "/scalars"]
    out, st = ctx.sed_prologue(pars, opthin, noalpha, 500.0, want_peak=True)
    assert np.all(st == 0)
    rel_below("normfac", out[:, 0], ref[:, 0], 1e-12)
    if not noalpha:
        parity_record("xmerge (abs)", np.max(np.abs(out[:, 1] - ref[:, 1])), 1e-10)
        assert np.max(np.abs(out[:, 1] - ref[:, 1])) < 1e-10    # xmerge (atol)
        rel_below("kappa", out[:, 2], ref[:, 2], 1e-10)
        rel_below("wavemerge", out[:, 4], ref[:, 4], 1e-10)
    else:
        assert np.all(np.isnan(out[:, 1])) and np.all(np.isnan(out[:, 4]))
    if not opthin:
        rel_below("x0", out[:, 3], ref[:, 3], 1e-14)
    rel_below("max_wave", out[:, 5], ref[:, 5], 1e-10)                 # (brentq xtol)


# ------------------------------------------------------- G3: f_nu on a grid
@pytest.mark.parametrize("name,opthin,noalpha", VARIANTS)
def test_fnu_grid(ctx, g_sed, name, opthin, noalpha):
    from mbb_emcee_amd.modified_blackbody import um_to_GHz
    pars, grid = g_sed["pars"], g_sed["wave_grid"]
    ref = g_sed[name + "/fnu_grid"]
    out, st = ctx.sed_eval(pars, opthin, noalpha, 500.0, um_to_GHz / grid)
    assert np.all(st == 0)
    rel_below("f_nu", out, ref, SED_RTOL)
    out1, _ = ctx.sed_eval(pars, opthin, noalpha, 500.0, np.array([um_to_GHz / 433.0]))
    rel_below("f_nu", out1[:, 0], g_sed[name + "/fnu_scalar433"], SED_RTOL)


def test_fnu_wavenorm850(mbb, g_sed):
    m = mbb.modified_blackbody(25.0, 1.6, 150.0, 2.5, 12.0, wavenorm=850.0)
    rel_below("f_nu", m(g_sed["wave_grid"]), g_sed["wn850/fnu_grid"], SED_RTOL)
    rec_allclose(m(850.0), 12.0, rtol=1e-13, kind="f_nu(wavenorm) = fnorm")


@pytest.mark.parametrize("name,opthin,noalpha", VARIANTS)
def test_fnu_explicit_matches_pyx_signature(ctx, g_sed, name, opthin, noalpha):
    """fnu.pyx:9-108 entry points with explicit scalars"""
    from mbb_emcee_amd.modified_blackbody import um_to_GHz
    pars, grid = g_sed["pars"], g_sed["wave_grid"]
    sc = g_sed[name + "/scalars"]
    for i in (0, 7, 40, 60, 90):
        out = ctx.fnu_eval(opthin, noalpha, um_to_GHz / grid, pars[i, 0], pars[i, 1],
                           sc[i, 3] if not opthin else 1.0, pars[i, 3], sc[i, 0],
                           sc[i, 1] if not noalpha else 0.0, sc[i, 2] if not noalpha else 0.0)
        rel_below("f_nu", out, g_sed[name + "/fnu_grid"][i], SED_RTOL)


# ----------------------------------------------------- G4: lnL, cfg 1/2/4
def _like_cfg1(mbb, g, name, opthin, noalpha):
    k = "cfg1/" + name
    like = mbb.likelihood(noalpha=noalpha, opthin=opthin)
    like.set_phot(g[k + "/wave"], g[k + "/flux"], g[k + "/unc"])
    return like, k


@pytest.mark.parametrize("path", PATHS)
@pytest.mark.parametrize("name,opthin,noalpha", VARIANTS)
def test_lnlike_cfg1_delta(mbb, g_lnl, name, opthin, noalpha, path):
    like, k = _like_cfg1(mbb, g_lnl, name, opthin, noalpha)
    np.testing.assert_array_equal(like.uplims, g_lnl[k + "/uplim"])
    pars = g_lnl[k + "/pars"]
    got = boundary_eval(like, pars, path)
    lnl_close(got, g_lnl[k + "/lnl"], kind="lnL" if path == "launched" else "lnL (%s)" % path)
    fin = np.isfinite(g_lnl[k + "/lnl"])
    mf = like.model_flux(pars[fin])
    rel_below("band flux", mf, g_lnl[k + "/model_flux"][fin], FLUX_RTOL)
    # scalar call returns a Python float, row by row identical to the batch
    for i in (0, 3, 50, 55):
        v = boundary_eval(like, pars[i], path)
        assert isinstance(v, float)
        assert (v == got[i]) or (np.isneginf(v) and np.isneginf(got[i]))


@pytest.mark.parametrize("path", PATHS)
@pytest.mark.parametrize("name,opthin,noalpha", VARIANTS)
def test_lnlike_cfg2_passbands(mbb, g_lnl, name, opthin, noalpha, path):
    k = "cfg2/" + name
    like = mbb.likelihood(noalpha=noalpha, opthin=opthin, response=True)
    like.set_phot([str(b) for b in g_lnl["cfg2/bands"]], g_lnl[k + "/flux"], g_lnl[k + "/unc"])
    rec_allclose(like.uplims, g_lnl[k + "/uplim"], rtol=1e-15, kind="host bookkeeping")
    pars = g_lnl[k + "/pars"]
    lnl_close(boundary_eval(like, pars, path), g_lnl[k + "/lnl"], kind="lnL" if path == "launched" else "lnL (%s)" % path)
    fin = np.isfinite(g_lnl[k + "/lnl"])
    rel_below("band flux", like.model_flux(pars[fin]), g_lnl[k + "/model_flux"][fin], FLUX_RTOL)


@pytest.mark.parametrize("path", PATHS)
@pytest.mark.parametrize("name,opthin,noalpha", VARIANTS)
def test_lnlike_cfg4_covariance(mbb, g_lnl, name, opthin, noalpha, path):
    k = "cfg4/" + name
    like = mbb.likelihood(noalpha=noalpha, opthin=opthin, response=True)
    like.set_phot([str(b) for b in g_lnl["cfg4/bands"]], g_lnl[k + "/flux"], g_lnl[k + "/unc"])
    like.set_cov(g_lnl[k + "/cov"])
    rec_allclose(like.data_wave, g_lnl[k + "/eff_wave"], rtol=1e-14, kind="host bookkeeping")
    pars = g_lnl[k + "/pars"]
    lnl_close(boundary_eval(like, pars, path), g_lnl[k + "/lnl"],
              kind="lnL (covariance)" if path == "launched" else "lnL (covariance, %s)" % path)
    fin = np.isfinite(g_lnl[k + "/lnl"])
    rel_below("band flux", like.model_flux(pars[fin]), g_lnl[k + "/model_flux"][fin], FLUX_RTOL)


@pytest.mark.parametrize("path", PATHS)
def test_lnlike_priors_and_peak(mbb, g_lnl, path):
    """Gaussian priors on all five parameters, extra upper/lower limits, and the
    lambda_peak ghost parameter (likelihood.py:672-752)."""
    bands = [str(b) for b in g_lnl["cfg2/bands"]]
    like = mbb.likelihood(response=True)
    like.set_phot(bands, g_lnl["cfg2/thick_walpha/flux"], g_lnl["cfg2/thick_walpha/unc"])
    like.set_gaussian_prior("T", 11.0, 2.0)
    like.set_gaussian_prior("beta", 1.9, 0.3)
    like.set_gaussian_prior("lambda0", 550.0, 80.0)
    like.set_gaussian_prior("alpha", 3.2, 0.5)
    like.set_gaussian_prior("fnorm", 42.0, 4.0)
    like.set_uplim("T", 12.5)
    like.set_uplim("fnorm", 41.0)
    like.set_lowlim("beta", 1.5)
    pars = g_lnl["cfg2/priors/pars"]
    kind = "lnL" if path == "launched" else "lnL (%s)" % path
    lnl_close(boundary_eval(like, pars, path), g_lnl["cfg2/priors/lnl"], kind=kind)
    like.set_gaussian_prior("lambda_peak", 260.0, 15.0)
    like.set_uplim("lambda_peak", 265.0)
    lnl_close(boundary_eval(like, pars, path), g_lnl["cfg2/priors_peak/lnl"], kind=kind)
    # thin + alpha with a peak prior
    like = mbb.likelihood(response=True, opthin=True)
    like.set_phot(bands, g_lnl["cfg2/thin_walpha/flux"], g_lnl["cfg2/thin_walpha/unc"])
    like.set_gaussian_prior("peaklam", 250.0, 20.0)
    lnl_close(boundary_eval(like, pars, path), g_lnl["cfg2/thin_peak/lnl"], kind=kind)


def test_max_wave(mbb, g_lnl):
    pars = g_lnl["cfg2/priors/pars"]
    ref = g_lnl["cfg2/priors_peak/max_wave"]
    for i in (0, 11, 50):
        rec_allclose(mbb.modified_blackbody(*pars[i]).max_wave(), ref[i], rtol=1e-10, kind="max_wave")


# ------------------------------------------- HIP vs oracle on fresh inputs
@pytest.mark.parametrize("name,opthin,noalpha", VARIANTS)
@pytest.mark.parametrize("n", [1, 2, 63, 125, 250, 1000, 4099])
def test_vs_oracle_batch_sizes(mbb, oracle, g_lnl, name, opthin, noalpha, n):
    """Every batch geometry (walkers per block 1..64, ragged last block) gives
    the oracle's answer, and a row's value does not depend on the batch it is in."""
    bands = [str(b) for b in g_lnl["cfg2/bands"]]
    k = "cfg2/" + name
    like = mbb.likelihood(noalpha=noalpha, opthin=opthin, response=True)
    like.set_phot(bands, g_lnl[k + "/flux"], g_lnl[k + "/unc"])
    rng = np.random.RandomState(1000 + n)
    pars = np.column_stack([rng.normal(12, 1, n), rng.normal(1.8, 0.2, n),
                            rng.normal(600, 50, n), rng.normal(3, 0.3, n), rng.normal(40, 3, n)])
    got = like(pars)
    orc = oracle.OracleLikelihood(
        g_lnl[k + "/flux"], g_lnl[k + "/unc"],
        bands=[(r.wavelength, r._sedmult, r._normfac) for r in like._responses],
        opthin=opthin, noalpha=noalpha, has_uplim=[int(b) for b in like.has_uplims],
        uplim=like.uplims)
    ref = orc(pars, nthreads=8)
    lnl_close(got, ref)
    # batch-composition independence: bitwise
    sub = like(pars[: max(1, n // 3)])
    assert np.array_equal(sub, got[: max(1, n // 3)])


def test_host_path_modes_identical(mbb, g_lnl):
    """mbb_lnlike_batch gives bitwise the same answers however the rows travel and however
    the host waits: explicit copies or zero-copy, parameter rows pushed through the BAR or
    pulled from pinned memory, blocking on the stream (spin_wait 0), polling it (1) or
    watching the result slots (2, the default).  The watch must actually see the results
    (a silent fall-back would cost the whole spin budget per call), and when it is forced
    to give up (spin_budget 0) the stream wait takes over with the same results.
    Reference behaviour kept: likelihood.__call__ returns synchronously (likelihood.py:834)."""
    bands = [str(b) for b in g_lnl["cfg2/bands"]]
    like = mbb.likelihood(response=True)
    like.set_phot(bands, g_lnl["cfg2/thick_walpha/flux"], g_lnl["cfg2/thick_walpha/unc"])
    pars = g_lnl["cfg2/thick_walpha/pars"]
    ctx = like.context
    ref = like(pars)
    assert np.isneginf(ref).any() and np.isfinite(ref).any()
    assert ctx.info("last_watch_seen") == 1            # default path: the watch saw every slot
    try:
        for zero_copy in (1, 0):
            for bar in (1, 0):
                for spin in (2, 1, 0):
                    ctx.set_option("zero_copy", zero_copy); ctx.set_option("bar_params", bar)
                    ctx.set_option("spin_wait", spin)
                    for n in (1, 125, len(pars)):
                        got = like(pars[:n])
                        assert np.array_equal(got, ref[:n], equal_nan=True), (zero_copy, bar, spin, n)
                    seen = ctx.info("last_watch_seen")
                    assert seen == (1 if (zero_copy and spin == 2) else -1), (zero_copy, bar, spin, seen)
        # forced fall-back: no polls allowed, the results must come from the stream wait
        ctx.set_option("zero_copy", 1); ctx.set_option("bar_params", 1); ctx.set_option("spin_wait", 2)
        ctx.set_option("spin_budget", 0)
        for n in (1, 125, len(pars)):
            got = like(pars[:n])
            assert np.array_equal(got, ref[:n], equal_nan=True)
            assert ctx.info("last_watch_seen") == 0 and ctx.info("last_wait_ns") > 0
        # a budget too small to outlast the kernel gives up as well, or sees everything: same numbers
        ctx.set_option("spin_budget", 3)
        assert np.array_equal(like(pars[:125]), ref[:125], equal_nan=True)
        assert ctx.info("last_watch_seen") in (0, 1)
    finally:
        ctx.set_option("spin_budget", 20000000); ctx.set_option("spin_wait", 2)
        ctx.set_option("zero_copy", 1); ctx.set_option("bar_params", 1)
    assert np.array_equal(like(pars), ref, equal_nan=True)
    assert ctx.info("last_watch_seen") == 1


@pytest.mark.parametrize("wpb,threads", [(1, 64), (1, 1024), (3, 256), (16, 128), (64, 512)])
def test_geometry_invariance(mbb, g_lnl, wpb, threads):
    """lnL is bitwise independent of the launch geometry."""
    bands = [str(b) for b in g_lnl["cfg4/bands"]]
    k = "cfg4/thick_walpha"
    like = mbb.likelihood(response=True)
    like.set_phot(bands, g_lnl[k + "/flux"], g_lnl[k + "/unc"])
    like.set_cov(g_lnl[k + "/cov"])
    pars = g_lnl[k + "/pars"]
    ref = like(pars)
    like.context.set_option("walkers_per_group", wpb)
    like.context.set_option("block_threads", threads)
    got = like(pars)
    # a walker's prologue takes one row of 16 lanes, so a workgroup has >= 16 wpb threads
    assert like.context.info("last_wpb") == wpb
    assert like.context.info("last_threads") == max(threads, 16 * wpb)
    assert np.array_equal(ref, got, equal_nan=True)


@pytest.mark.parametrize("name,opthin,noalpha", VARIANTS)
def test_prepass_lane_per_walker_equals_rows_of_lanes(mbb, g_lnl, name, opthin, noalpha):
    """Big batches work out gate, SED constructor and penalties in a pass of their own with a LANE per walker
    (k_walker_pre: mbb_walker_consts.inc with every exp evaluated where the row form deals them to its 16 lanes) and
    k_lnlike fetches the records.  The same values bit for bit as with a row of lanes per walker: lnL, row status (rows
    below a limit, NaN rows, rows the reference raises for), model fluxes; with priors and the peak wavelength's terms
    (a second root); every model variant; batch sizes around the workgroup sizes.  And it is what a 250 000-row launch
    takes by itself (`prepass` -1: from 64 rows per CU)."""
    k = "cfg2/" + name
    like = mbb.likelihood(noalpha=noalpha, opthin=opthin, response=True)
    like.set_phot([str(b) for b in g_lnl["cfg2/bands"]], g_lnl[k + "/flux"], g_lnl[k + "/unc"])
    like.set_gaussian_prior("T", 11.0, 2.0)
    like.set_gaussian_prior("lambda_peak", 260.0, 15.0)
    like.set_uplim("lambda_peak", 265.0)
    like.set_uplim("fnorm", 41.0)
    ctx = like._sync_device()
    base = g_lnl[k + "/pars"]
    pars = np.tile(base, (14, 1))[:4099].copy()
    pars[7, 0] = np.nan; pars[11, 2] = np.inf
    like.set_lowlim("alpha", -5.0); like.set_lowlim("beta", -5.0)     # (so that rows the reference raises for get past the gate)
    pars[13, 3] = -1.0; pars[17, 1] = -0.5
    ctx = like._sync_device()
    for n in (1, 63, 257, 4099):
        ctx.set_option("prepass", 0)
        lnl0, st0, fl0 = ctx.lnlike_batch(pars[:n], want_flux=True)
        assert ctx.info("last_prepass") == 0
        ctx.set_option("prepass", 1)
        lnl1, st1, fl1 = ctx.lnlike_batch(pars[:n], want_flux=True)
        assert ctx.info("last_prepass") == 1
        assert np.array_equal(st0, st1), n
        assert np.array_equal(lnl0, lnl1, equal_nan=True) and np.array_equal(fl0, fl1, equal_nan=True), n
    assert set(np.unique(st1)) >= {0, 1, 7} and (noalpha or 2 in st1) and 3 in st1
    # other geometries fetch the same records
    for wpb, thr in ((3, 256), (16, 512), (64, 1024)):
        ctx.set_option("walkers_per_group", wpb); ctx.set_option("block_threads", thr)
        assert np.array_equal(ctx.lnlike_batch(pars)[0], lnl1, equal_nan=True)
    ctx.set_option("walkers_per_group", 0); ctx.set_option("block_threads", 0)
    # by itself: only for big batches
    ctx.set_option("prepass", -1)
    ctx.lnlike_batch(pars)
    assert ctx.info("last_prepass") == 0
    big = np.tile(base, (64 * ctx.info("cu_count") // base.shape[0] + 1, 1))
    lb = ctx.lnlike_batch(big)[0]
    assert ctx.info("last_prepass") == 1
    ctx.set_option("prepass", 0)
    assert np.array_equal(ctx.lnlike_batch(big)[0], lb, equal_nan=True)
    ctx.set_option("prepass", -1)


def test_empty_and_bad_shapes(mbb, g_lnl):
    like, k = _like_cfg1(mbb, g_lnl, "thick_walpha", False, False)
    assert like(np.empty((0, 5))).shape == (0,)
    with pytest.raises(ValueError):
        like(np.ones(4))
    with pytest.raises(ValueError):
        like(np.ones((3, 4)))
    # NaN parameter rows pass the `<` gate like the reference and come out NaN
    p = g_lnl[k + "/pars"][:4].copy()
    p[1, 0] = np.nan
    out = like(p)
    assert np.isnan(out[1]) and np.all(np.isfinite(out[[0, 2, 3]]))


def test_bad_alpha_raises_like_reference(mbb, g_lnl):
    """alpha <= 0 reaches the SED constructor only when the lower limit allows it;
    the reference then raises ValueError (modified_blackbody.py:219-221)."""
    like, k = _like_cfg1(mbb, g_lnl, "thick_walpha", False, False)
    like.set_lowlim("alpha", -5.0)
    p = g_lnl[k + "/pars"][:3].copy()
    p[2, 3] = -1.0
    with pytest.raises(ValueError):
        like(p)


def test_low_temperature_overflow_gives_zero_not_nan(mbb):
    """expm1(x) overflows for x > 709.78 (MIPS 24um at T ~ 1 K): the band flux
    must be 0, not NaN (SURVEY.md 8a numerical-range notes)."""
    like = mbb.likelihood(response=True, opthin=True, noalpha=True)
    like.set_phot(["MIPS_24um", "SPIRE_500um"], [1.0, 40.0], [1.0, 4.0])
    p = np.array([[1.0, 1.5, 100.0, 2.0, 40.0]])
    fl = like.model_flux(p)
    assert np.all(np.isfinite(fl)) and 0.0 <= fl[0, 0] < 1e-100
    assert np.isfinite(like(p[0]))
    # the delta-function version sits entirely beyond the overflow point
    like = mbb.likelihood(opthin=True, noalpha=True)
    like.set_phot([19.0, 500.0], [1.0, 40.0], [1.0, 4.0])
    fl = like.model_flux(p)
    assert 0.0 <= fl[0, 0] < 1e-280 and np.isfinite(like(p[0]))


def test_special_passbands_vs_oracle(mbb, oracle):
    names = ["SCUBA2_box_850um_85", "SMA_gauss_345_8", "X_dsb_230_16_8", "ALMA_alma_343",
             "Y_delta_880um", "SPIRE_350um"]
    like = mbb.likelihood(response=True)
    flux = np.array([9.0, 14.0, 3.0, 12.0, 8.0, 75.0])
    like.set_phot(names, flux, 0.1 * flux + 0.5)
    rng = np.random.RandomState(5)
    pars = np.column_stack([rng.normal(12, 1, 97), rng.normal(1.8, 0.2, 97),
                            rng.normal(600, 50, 97), rng.normal(3, 0.3, 97), rng.normal(40, 3, 97)])
    bands = []
    for r in like._responses:
        if r.isdelta:
            bands.append((r.wavelength, np.ones(1), 1.0))
        else:
            bands.append((r.wavelength, r._sedmult, r._normfac))
    orc = oracle.OracleLikelihood(flux, 0.1 * flux + 0.5, bands=bands,
                                  has_uplim=[int(b) for b in like.has_uplims], uplim=like.uplims)
    lnl_close(like(pars), orc(pars))


def test_sampler_recovers_truth(mbb):
    """End-to-end: mbb_fitter on synthetic cfg1-like data recovers the truth
    (statistical parity only: emcee itself is not available)."""
    wave = np.array([100.0, 160.0, 250.0, 350.0, 500.0, 850.0])
    truth = mbb.modified_blackbody(14.0, 1.8, None, None, 40.0, opthin=True, noalpha=True)
    flux = truth(wave)
    fit = mbb.mbb_fitter(nwalkers=64, opthin=True, noalpha=True, seed=3)
    fit.set_data(wave, flux, 0.05 * flux)
    fit.fix_param("lambda0")
    fit.fix_param("alpha")
    p0 = fit.generate_initial_values(np.array([12.0, 2.0, 600.0, 3.0, 35.0]),
                                     np.array([2, 0.2, 100, 0.3, 5.0]))
    assert np.all(p0[:, 2] == 600.0) and np.all(p0[:, 3] == 3.0)
    fit.run(150, 150, p0)
    ch = fit.sampler.chain
    assert ch.shape == (64, 150, 5) and fit.sampler.lnprobability.shape == (64, 150)
    assert np.all(ch[:, :, 2] == 600.0)                # fixed by zero initial scatter
    med = np.median(ch.reshape(-1, 5), axis=0)
    assert abs(med[0] - 14.0) < 1.0 and abs(med[1] - 1.8) < 0.3 and abs(med[4] - 40.0) < 2.0
    assert 0.2 < fit.sampler.acceptance_fraction.mean() < 0.9


# ------------------------------------------------ device-resident sampler
def _philox4x32(c, k0, k1):
    """numpy replica of the kernel's Philox4x32-10 (Salmon et al. 2011)."""
    c = [np.uint64(x) for x in c]
    k0, k1 = np.uint64(k0), np.uint64(k1)
    M = np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c[0]
        p1 = np.uint64(0xCD9E8D57) * c[2]
        hi0, lo0 = p0 >> np.uint64(32), p0 & M
        hi1, lo1 = p1 >> np.uint64(32), p1 & M
        c = [(hi1 ^ c[1] ^ k0) & M, lo1, (hi0 ^ c[3] ^ k1) & M, lo0]
        k0 = (k0 + np.uint64(0x9E3779B9)) & M
        k1 = (k1 + np.uint64(0xBB67AE85)) & M
    return [int(x) for x in c]


def _host_stretch_step(like, pos, lnp, seed, steps_done, t, a=2.0):
    """One full step with exactly the kernel's random stream and accept rule;
    lnprob from the likelihood's batch call."""
    pos, lnp = pos.copy(), lnp.copy()
    nw = pos.shape[0]; half = nw // 2
    key = (seed + 0x9E3779B97F4A7C15 * (steps_done + t + 1)) & 0xFFFFFFFFFFFFFFFF
    acc = np.zeros(nw, dtype=bool)
    for h in range(2):
        s_begin, c_begin = (half, 0) if h else (0, half)
        q = np.empty((half, 5)); lz = np.empty(half); lu = np.empty(half)
        for w in range(half):
            row = s_begin + w
            r = _philox4x32([row, h, 0, 0], key & 0xFFFFFFFF, key >> 32)      # (the key carries the step's number)
            u1 = ((r[0] >> 5) * 67108864.0 + (r[1] >> 6)) / 9007199254740992.0
            u2 = r[2] / 4294967296.0
            u3 = (r[3] + 0.5) / 4294967296.0
            zz = ((a - 1.0) * u1 + 1.0) ** 2 / a
            pj = min(int(u2 * half), half - 1)
            c, s = pos[c_begin + pj], pos[row]
            q[w] = c - zz * (c - s)
            lz[w] = 4.0 * np.log(zz); lu[w] = np.log(u3)
        new = like(q)
        for w in range(half):
            row = s_begin + w
            if lz[w] + new[w] - lnp[row] > lu[w]:
                pos[row] = q[w]; lnp[row] = new[w]; acc[row] = True
    return pos, lnp, acc


def _cfg2_like(mbb, g_lnl):
    like = mbb.likelihood(response=True)
    like.set_phot([str(b) for b in g_lnl["cfg2/bands"]], g_lnl["cfg2/thick_walpha/flux"],
                  g_lnl["cfg2/thick_walpha/unc"])
    return like


def test_device_sampler_matches_host_emulation(mbb, g_lnl):
    """Three steps of the device sampler equal a host emulation that uses the same
    Philox stream, the same proposal/accept rule and lnprob from the batch call."""
    like = _cfg2_like(mbb, g_lnl)
    p0 = g_lnl["cfg2/thick_walpha/pars"][:64].copy()
    p0[5, 1] = 0.12                                       # near the beta lower limit: -inf proposals
    seed = 1234567
    s = mbb.DeviceEnsembleSampler(64, 5, like, seed=seed)
    pos, lnp, _ = s.run_mcmc(p0, 3)
    hp, hl = p0.copy(), like(p0)
    nacc = np.zeros(64)
    for t in range(3):
        hp, hl, acc = _host_stretch_step(like, hp, hl, seed, 0, t)
        nacc += acc
        rec_allclose(s.chain[:, t, :], hp, rtol=1e-13, kind="sampler vs host emulation")
        rec_allclose(s.lnprobability[:, t], hl, rtol=1e-11, atol=1e-11, kind="sampler vs host emulation")
    rec_allclose(pos, hp, rtol=1e-13, kind="sampler vs host emulation")
    assert np.array_equal(s.naccepted, nacc) and 0 < nacc.sum() < 3 * 64
    # the chain's lnprob is the likelihood of the chain's positions
    lnl_close(like(s.chain[:, -1, :]), s.lnprobability[:, -1])
    # continuing uses a fresh part of the random stream
    pos2, _, _ = s.run_mcmc(None, 2)
    hp2, hl2 = hp, hl
    for t in range(2):
        hp2, hl2, _ = _host_stretch_step(like, hp2, hl2, seed, 3, t)
    rec_allclose(pos2, hp2, rtol=1e-13, kind="sampler vs host emulation")
    assert s.chain.shape == (64, 5, 5)


def test_device_sampler_statistics_match_host_sampler(mbb):
    """Posterior means/widths from the device sampler agree with the host
    stretch-move sampler on the same problem; fixed columns stay fixed; limits hold."""
    wave = np.array([100.0, 160.0, 250.0, 350.0, 500.0, 850.0])
    truth = mbb.modified_blackbody(14.0, 1.8, None, None, 40.0, opthin=True, noalpha=True)
    flux = truth(wave)
    stats = {}
    for kind in ("native", "device"):
        fit = mbb.mbb_fitter(nwalkers=100, opthin=True, noalpha=True, seed=21, sampler=kind)
        fit.set_data(wave, flux, 0.05 * flux)
        fit.fix_param("lambda0"); fit.fix_param("alpha")
        p0 = fit.generate_initial_values(np.array([12.0, 2.0, 600.0, 3.0, 35.0]),
                                         np.array([2, 0.2, 100, 0.3, 5.0]))
        fit.run(300, 400, p0)
        ch = fit.sampler.chain
        assert ch.shape == (100, 400, 5) and fit.sampler.lnprobability.shape == (100, 400)
        assert np.all(ch[:, :, 2] == 600.0) and np.all(ch[:, :, 3] == 3.0)
        assert np.all(ch[:, :, 1] >= 0.1) and np.all(ch[:, :, 0] >= 1.0)
        flat = ch.reshape(-1, 5)
        stats[kind] = (flat.mean(axis=0), flat.std(axis=0), fit.sampler.acceptance_fraction.mean())
        lnl_close(fit.like(ch[:, -1, :]), fit.sampler.lnprobability[:, -1])
    (m0, s0, a0), (m1, s1, a1) = stats["native"], stats["device"]
    for i in (0, 1, 4):
        assert abs(m0[i] - m1[i]) < 0.25 * s0[i], (i, m0[i], m1[i], s0[i])
        assert 0.75 < s1[i] / s0[i] < 1.33
    assert abs(m1[0] - 14.0) < 3 * s1[0] and abs(m1[1] - 1.8) < 3 * s1[1]
    assert abs(a0 - a1) < 0.08 and 0.2 < a1 < 0.9


def test_device_sampler_generator_form_equals_run_mcmc(mbb, g_lnl):
    """emcee's generator form, ``for pos, lnprob, rstate in sampler.sample(p0, iterations=N)`` (the host sampler has it:
    ensemble.py), on the device sampler: the steps are made a chunk at a time in one launch and handed out one by one.  The
    chain is run_mcmc(p0, N)'s bit for bit WHATEVER the chunk (round 6: a step's draws are keyed by its number in the
    sampler's life, not by its place in a launch) -- stored or not, continued by run_mcmc, broken out of early; and while
    a step is out, chain / lnprobability / iterations / naccepted are those of the steps handed out so far."""
    like = _cfg2_like(mbb, g_lnl)
    rng = np.random.RandomState(8)
    p0 = np.array([12.0, 1.8, 600.0, 3.0, 40.0]) * (1.0 + 0.02 * rng.normal(size=(64, 5)))
    whole = mbb.DeviceEnsembleSampler(64, 5, like, seed=21)
    whole.run_mcmc(p0, 100)
    whole.run_mcmc(None, 50)
    acc_by_step = np.cumsum(np.any(np.diff(np.concatenate((p0[:, None, :], whole.chain), axis=1), axis=1) != 0, axis=-1), axis=1)
    for chunk in (1, 7, 64, 1000):
        ref = mbb.DeviceEnsembleSampler(64, 5, like, seed=21)
        pos, left = p0, 100
        while left > 0:                                              # the same steps by run_mcmc, chunk by chunk
            ref.run_mcmc(pos, min(chunk, left)); pos = None; left -= min(chunk, left)
        ref.run_mcmc(None, 50)
        # (a step's draws are keyed by its number in the sampler's life: the grouping into launches does not show)
        assert np.array_equal(ref.chain, whole.chain) and np.array_equal(ref.naccepted, whole.naccepted), chunk
        s = mbb.DeviceEnsembleSampler(64, 5, like, seed=21)
        seen = 0
        for pos, lnp, _ in s.sample(p0, iterations=100, chunk=chunk, rstate0=np.random.RandomState(1)):
            assert np.array_equal(pos, ref.chain[:, seen, :]) and np.array_equal(lnp, ref.lnprobability[:, seen]), (chunk, seen)
            seen += 1
            # while a step is out, the sampler shows the steps handed out so far and nothing of those the device has
            # made ahead of them
            assert s.chain.shape == (64, seen, 5) and s.lnprobability.shape == (64, seen) and s.iterations == seen
            assert np.array_equal(s.chain[:, -1], pos) and np.array_equal(s.naccepted, acc_by_step[:, seen - 1]), (chunk, seen)
        assert seen == 100 and s.chain.shape == (64, 100, 5) and s.iterations == 100
        s.run_mcmc(None, 50)                                          # ... and goes on as run_mcmc does
        assert np.array_equal(s.chain, ref.chain) and np.array_equal(s.lnprobability, ref.lnprobability), chunk
        assert np.array_equal(s.naccepted, ref.naccepted)
    # nothing stored: the same ensembles, the chain untouched
    ref = mbb.DeviceEnsembleSampler(64, 5, like, seed=21)
    ref.run_mcmc(p0, 16); ref.run_mcmc(None, 16); ref.run_mcmc(None, 8)
    s = mbb.DeviceEnsembleSampler(64, 5, like, seed=21)
    last = None
    for k, (pos, lnp, _) in enumerate(s.sample(p0, iterations=40, storechain=False, chunk=16)):
        assert np.array_equal(pos, ref.chain[:, k, :])
        last = pos.copy()
    assert s.chain.shape == (64, 0, 5) and s.iterations == 40 and np.array_equal(last, ref.chain[:, 39, :])
    # broken out of early: the chain holds whole chunks
    s = mbb.DeviceEnsembleSampler(64, 5, like, seed=21)
    for k, (pos, lnp, _) in enumerate(s.sample(p0, iterations=100, chunk=10)):
        if k == 24:
            break
    assert s.chain.shape[1] == 30 and s.iterations == 30
    # the posterior is the same whatever the chunk (means of the second half of longer runs, in units of the spread)
    a, b = mbb.DeviceEnsembleSampler(64, 5, like, seed=5), mbb.DeviceEnsembleSampler(64, 5, like, seed=5)
    a.run_mcmc(p0, 1200)
    for _ in b.sample(p0, iterations=1200, chunk=13):
        pass
    fa, fb = a.chain[:, 600:].reshape(-1, 5), b.chain[:, 600:].reshape(-1, 5)
    assert np.all(np.abs(fa.mean(0) - fb.mean(0)) < 0.25 * fa.std(0)), (fa.mean(0), fb.mean(0), fa.std(0))
    del s, ref, whole, a, b


def test_device_sampler_errors(mbb, g_lnl):
    like = _cfg2_like(mbb, g_lnl)
    with pytest.raises(ValueError):
        mbb.DeviceEnsembleSampler(7, 5, like)
    with pytest.raises(ValueError):
        mbb.DeviceEnsembleSampler(8, 5, like)
    with pytest.raises(TypeError):
        mbb.DeviceEnsembleSampler(20, 5, lambda p: 0.0)
    s = mbb.DeviceEnsembleSampler(20, 5, like, seed=1)
    p0 = g_lnl["cfg2/thick_walpha/pars"][:20].copy()
    bad = p0.copy(); bad[3, 0] = np.nan
    with pytest.raises(ValueError):
        s.run_mcmc(bad, 1)
    with pytest.raises(ValueError):
        s.run_mcmc(None, 1)
    # alpha <= 0 reachable (lower limit relaxed): the SED constructor's ValueError surfaces
    like.set_lowlim("alpha", -50.0)
    p1 = p0.copy(); p1[:, 3] = np.linspace(0.01, 0.5, 20)
    s2 = mbb.DeviceEnsembleSampler(20, 5, like, seed=2)
    with pytest.raises(ValueError):
        s2.run_mcmc(p1, 50)


# ------------------------------------------------------------- RCCL plumbing
def test_rccl_single_rank_allgather(mbb, g_lnl):
    """The dlopen'ed RCCL path end to end with a 1-rank communicator: unique id,
    ncclCommInitRank, ncclAllGather on the context's stream, destroy.  (More ranks
    need more GPUs; the sharding logic itself is covered by the gloo CPU test.)"""
    from mbb_emcee_amd.parallel import RcclComm, ShardedLikelihood
    like = _cfg2_like(mbb, g_lnl)
    ctx = like._sync_device()
    uid = ctx.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    ctx.comm_init(1, 0, uid)
    assert ctx.info("nranks") == 1
    x = np.arange(125, dtype=np.float64) * 0.5
    a, b = ctx.alloc(x.nbytes), ctx.alloc(x.nbytes)
    a.upload(x)
    ctx.allgather_f64(a, b, 125)
    ctx.sync()
    assert np.array_equal(b.download(np.float64, 125), x)
    # the fused launch + gather call used by bench.py --gpus N and ShardedLikelihood
    pars = g_lnl["cfg2/thick_walpha/pars"][:125]
    dp, dl, ds, da = ctx.alloc(pars.nbytes), ctx.alloc(1000), ctx.alloc(500), ctx.alloc(1000)
    dp.upload(pars)
    ctx.lnlike_allgather_device(dp, 125, dl, ds, da)
    ctx.sync()
    assert np.array_equal(da.download(np.float64, 125), like(pars))
    ctx.comm_destroy()
    comm = RcclComm(ctx, 0, 1, uid)
    sharded = ShardedLikelihood(like, comm)
    allp = g_lnl["cfg2/thick_walpha/pars"]
    want = like(allp)
    assert np.array_equal(sharded(allp), want, equal_nan=True)
    # the boundary call of a sharded run (mbb_lnlike_allgather: host rows in, every rank's lnprob out, one native
    # call) without a communicator, through a real one-rank RCCL communicator, and with and without zero-copy
    for with_comm in (False, True):
        if with_comm:
            ctx.comm_init(1, 0, ctx.comm_unique_id())         # (a unique id makes one communicator)
        for zc in (1, 0):
            ctx.set_option("zero_copy", zc)
            for n in (1, 125, 250, allp.shape[0]):
                full, st = ctx.lnlike_allgather(allp[:n], 1)
                assert np.array_equal(full, want[:n], equal_nan=True) and st.shape == (n,), (with_comm, zc, n)
            assert np.array_equal(sharded(allp), want, equal_nan=True)
        ctx.set_option("zero_copy", 1)
    ctx.comm_destroy()


def test_bench_sharded_boundary_leg_with_one_rank(mbb, g_lnl):
    """bench.py's `boundary_sharded` leg (the host-driven split of north_star: ShardedLikelihood over RcclComm, one
    ncclAllGather of lnprob per call) has never seen more than one GPU; here its code runs end to end with one rank
    and the side channel's two operations as identities (no torch in the test process): set-up, the timed loop, the bitwise check against the unsharded evaluation,
    the reduction of the medians, tear-down."""
    import bench

    like = _cfg2_like(mbb, g_lnl)
    ctx = like._sync_device()
    allw = bench.walkers(1)
    res = bench.sharded_boundary(ctx, like, 0, 1, allw, 250, lambda: None, lambda ok: bool(ok), lambda obj: obj, lambda v: v)
    assert res["ok"] is True, res
    for key, n in (("rows_125", 125), ("rows_250", 250)):
        r = res[key]
        assert r["rows_per_rank"] == n and r["equals_unsharded_bitwise"] and 2.0 < r["median_us"] < 500.0 and r["calls"] == 200
    assert ctx.info("nranks") == 1


# --------------------------------------------- batched multi-source mode (cfg5)
def _multi_setup(mbb, g_lnl, ns, seed=9):
    bands = [str(b) for b in g_lnl["cfg2/bands"]]
    rng = np.random.RandomState(seed)
    truths = np.column_stack([rng.uniform(8, 20, ns), rng.uniform(1.2, 2.4, ns), rng.uniform(300, 900, ns),
                              rng.uniform(2, 4.5, ns), rng.uniform(10, 80, ns)])
    single = mbb.likelihood(response=True)
    single.set_phot(bands, np.ones(8), np.ones(8))
    flux = single.model_flux(truths)
    unc = 0.1 * flux + 1.0
    multi = mbb.likelihood(response=True)
    multi.set_phot_multi(bands, flux, unc)
    return bands, truths, flux, unc, multi


def test_multi_source_equals_per_source_calls(mbb, g_lnl):
    """Block g of a multi-source batch equals an ordinary single-source likelihood
    built on source g's data, bit for bit."""
    ns, m = 7, 33
    bands, truths, flux, unc, multi = _multi_setup(mbb, g_lnl, ns)
    rng = np.random.RandomState(1)
    pars = truths[:, None, :] * (1.0 + 0.05 * rng.normal(size=(ns, m, 5)))
    pars[2, 5, 0] = 0.3                                   # -inf row
    got = multi(pars)
    assert got.shape == (ns, m) and np.isneginf(got[2, 5])
    assert np.array_equal(multi(pars.reshape(-1, 5)).reshape(ns, m), got)
    for g in range(ns):
        one = mbb.likelihood(response=True)
        one.set_phot(bands, flux[g], unc[g])
        one._uplim[2] = multi.uplims[2]; one._dirty = True     # shared limits: source 0's
        assert np.array_equal(one(pars[g]), got[g])
    with pytest.raises(ValueError):
        multi(np.ones((ns * m + 1, 5)))


def test_multi_source_device_sampler(mbb, g_lnl):
    """Several independent ensembles advance in the same launches; each recovers
    its own truth, and source g's chain equals the chain of a single-source
    sampler given the same rows of the random stream."""
    ns, nw = 6, 40
    bands, truths, flux, unc, multi = _multi_setup(mbb, g_lnl, ns, seed=4)
    rng = np.random.RandomState(2)
    p0 = truths[:, None, :] * (1.0 + 0.02 * rng.normal(size=(ns, nw, 5)))
    s = mbb.DeviceEnsembleSampler(nw, 5, multi, seed=99)
    pos, lnp, _ = s.run_mcmc(p0, 120)
    assert s.chain.shape == (ns, nw, 120, 5) and s.lnprobability.shape == (ns, nw, 120)
    assert pos.shape == (ns, nw, 5) and s.naccepted.shape == (ns, nw)
    lnl_close(multi(s.chain[:, :, -1, :]), s.lnprobability[:, :, -1])
    acc = s.acceptance_fraction
    assert np.all(acc.mean(axis=1) > 0.1) and np.all(acc.mean(axis=1) < 0.9)
    # walkers of one source never mix with another's: the fnorm column stays near its truth
    med = np.median(s.chain[:, :, 60:, 4].reshape(ns, -1), axis=1)
    assert np.all(np.abs(med - truths[:, 4]) < 0.2 * truths[:, 4])
    # one step against the host emulation, per source (rows are numbered src*nw + walker)
    s2 = mbb.DeviceEnsembleSampler(nw, 5, multi, seed=5)
    s2.run_mcmc(p0, 1)
    half = nw // 2
    for g in (0, ns - 1):
        one = mbb.likelihood(response=True)
        one.set_phot(bands, flux[g], unc[g])
        one._uplim[2] = multi.uplims[2]; one._dirty = True
        hp, hl = p0[g].copy(), one(p0[g])
        key = (5 + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
        for h in range(2):
            s_begin, c_begin = (half, 0) if h else (0, half)
            q = np.empty((half, 5)); lz = np.empty(half); lu = np.empty(half)
            for w in range(half):
                r = _philox4x32([g * nw + s_begin + w, h, 0, 0], key & 0xFFFFFFFF, key >> 32)
                u1 = ((r[0] >> 5) * 67108864.0 + (r[1] >> 6)) / 9007199254740992.0
                zz = (u1 + 1.0) ** 2 / 2.0
                pj = min(int(r[2] / 4294967296.0 * half), half - 1)
                c, sv = hp[c_begin + pj], hp[s_begin + w]
                q[w] = c - zz * (c - sv); lz[w] = 4 * np.log(zz); lu[w] = np.log((r[3] + 0.5) / 4294967296.0)
            new = one(q)
            for w in range(half):
                if lz[w] + new[w] - hl[s_begin + w] > lu[w]:
                    hp[s_begin + w] = q[w]; hl[s_begin + w] = new[w]
        rec_allclose(s2.chain[g, :, 0, :], hp, rtol=1e-13, kind="sampler vs host emulation")


def test_multi_source_sampler_in_the_middle_regime(mbb, g_lnl):
    """Many sources: a half-step's launch covers nsrc x nw/2 rows, more than one per CU -- the middle regime's
    geometry (one 1024-thread workgroup per CU, several walkers each, passband tables staged in LDS, per-source
    data read from global memory).  The chain must not depend on it: bitwise that of the same run forced to one
    walker per 256-thread workgroup with the tables read through L2."""
    ns, nw = 24, 40
    bands, truths, flux, unc, multi = _multi_setup(mbb, g_lnl, ns, seed=7)
    p0 = truths[:, None, :] * (1.0 + 0.02 * np.random.RandomState(3).normal(size=(ns, nw, 5)))
    ctx = multi.context
    out = []
    for opts in ((0, 0, -1), (1, 256, 0)):
        ctx.set_option("walkers_per_group", opts[0]); ctx.set_option("block_threads", opts[1]); ctx.set_option("stage_tables", opts[2])
        s = mbb.DeviceEnsembleSampler(nw, 5, multi, seed=31)
        pos, lnp, _ = s.run_mcmc(p0, 25)
        out.append((pos, lnp, s.chain.copy(), s.lnprobability.copy(), s.naccepted.copy(),
                    ctx.info("last_wpb"), ctx.info("last_threads"), ctx.info("last_stage")))
    ctx.set_option("walkers_per_group", 0); ctx.set_option("block_threads", 0); ctx.set_option("stage_tables", -1)
    assert out[0][5:] == (2, 1024, 1) and out[1][5:] == (1, 256, 0), (out[0][5:], out[1][5:])
    for x, y in zip(out[0][:5], out[1][:5]):
        assert np.array_equal(x, y)
    lnl_close(multi(out[0][2][:, :, -1, :]), out[0][3][:, :, -1])


def test_lds_staged_tables_identical(mbb, g_lnl):
    """The LDS-staged variant of the kernel gives bitwise the same lnL as the
    variant that reads the passband tables through L2, at every geometry."""
    bands = [str(b) for b in g_lnl["cfg4/bands"]]
    k = "cfg4/thick_walpha"
    like = mbb.likelihood(response=True)
    like.set_phot(bands, g_lnl[k + "/flux"], g_lnl[k + "/unc"])
    like.set_cov(g_lnl[k + "/cov"])
    pars = g_lnl[k + "/pars"]
    ctx = like.context
    ctx.set_option("stage_tables", 0)
    ref = like(pars)
    assert ctx.info("last_stage") == 0
    for wpb, thr in [(1, 1024), (1, 256), (4, 512), (32, 512)]:
        ctx.set_option("walkers_per_group", wpb); ctx.set_option("block_threads", thr)
        ctx.set_option("stage_tables", 1)
        got = like(pars)
        assert ctx.info("last_stage") == 1
        assert np.array_equal(ref, got, equal_nan=True)
    ctx.set_option("stage_tables", -1); ctx.set_option("walkers_per_group", 0); ctx.set_option("block_threads", 0)
    like(pars[:100])
    assert ctx.info("last_stage") == 1           # auto: staged in the latency regime
    mid = np.tile(pars, (8, 1))
    got = like(mid)
    # auto, up to 256 walkers per CU: one workgroup of 1024 threads per CU, tables staged once for all its walkers
    assert ctx.info("last_stage") == 1 and ctx.info("last_threads") == 1024 and ctx.info("last_grid") <= 256
    assert np.array_equal(got, np.tile(ref, 8), equal_nan=True)
    big = np.tile(pars, (1 + 70000 // pars.shape[0], 1))
    got = like(big)
    assert ctx.info("last_stage") == 0 and ctx.info("last_threads") == 512      # auto: L2 path for big batches
    assert np.array_equal(got, np.tile(ref, big.shape[0] // pars.shape[0]), equal_nan=True)


# ------------------------------------------------------ shapes beyond the configs
@pytest.mark.parametrize("nbands", [100, 150])
def test_many_bands_and_covariance_vs_oracle(mbb, oracle, nbands):
    """100 / 150 plain wavelengths (more bands than lanes in a wave), diagonal and
    full covariance (80 KB: held in LDS; 180 KB: read from global), wavenorm != 500,
    against the oracle."""
    rng = np.random.RandomState(3)
    wave = np.sort(rng.uniform(60.0, 2500.0, nbands))
    for opthin, noalpha in ((False, False), (True, True)):
        like = mbb.likelihood(opthin=opthin, noalpha=noalpha, wavenorm=850.0)
        like.set_phot(wave, np.ones(nbands), np.ones(nbands))
        truth = np.array([18.0, 1.7, 300.0, 2.5, 12.0])
        flux = like.model_flux(truth)[0]
        unc = 0.05 * flux + 0.01
        like.set_phot(wave, flux, unc)
        pars = truth * (1.0 + 0.05 * rng.normal(size=(300, 5)))
        kw = dict(wave=wave, opthin=opthin, noalpha=noalpha, wavenorm=850.0,
                  has_uplim=[int(b) for b in like.has_uplims], uplim=like.uplims)
        lnl_close(like(pars), oracle.OracleLikelihood(flux, unc, **kw)(pars, nthreads=4))
        A = rng.normal(0, 1, (nbands, nbands))
        cov = np.diag(unc ** 2) + 1e-4 * np.median(unc) ** 2 * A.dot(A.T)
        like.set_cov(cov)
        lnl_close(like(pars), oracle.OracleLikelihood(flux, unc, cov=cov, **kw)(pars, nthreads=4),
                  kind="lnL (covariance)")
        rec_allclose(like.get_sed(pars[:7], wave), like.model_flux(pars[:7]), rtol=1e-13, kind="f_nu vs band flux")
        rec_allclose(like.get_sed(pars[3], wave), like.model_flux(pars[3])[0], rtol=1e-13, kind="f_nu vs band flux")


def test_unit_table_balances_the_simds(mbb, g_lnl):
    """cfg2 (2209 samples): 34 full chunks in 11 segments plus 2 tail chunks that hold the
    six leftover rows of five bands (a chunk per leftover would make it 39); the dealing
    order gives each of the four SIMDs of a CU 9 chunks.  Delta-function photometry:
    all bands in one packed chunk."""
    like = _cfg2_like(mbb, g_lnl)
    ctx = like._sync_device()
    assert ctx.info("nchunk") == 36 and ctx.info("nseg") == 11 and ctx.info("nunit") == 13
    assert (ctx.info("simd_chunks_max"), ctx.info("simd_chunks_min")) == (9, 9)
    ref = like(g_lnl["cfg2/thick_walpha/pars"])
    ctx.set_option("pack_tails", 0)
    like._dirty = True
    c0 = like._sync_device()
    assert c0.info("nchunk") == 39 and c0.info("nunit") == 12
    assert np.allclose(like(g_lnl["cfg2/thick_walpha/pars"]), ref, rtol=1e-13, atol=0)
    ctx.set_option("pack_tails", 1)
    like._dirty = True
    d = mbb.likelihood()
    d.set_phot(np.linspace(100.0, 1200.0, 40), np.ones(40), np.ones(40))
    c2 = d._sync_device()
    assert (c2.info("nchunk"), c2.info("nseg"), c2.info("nunit")) == (1, 0, 1)


def test_mixed_delta_and_passband_bands_keep_their_order(mbb, oracle):
    """Single-sample bands are packed 64 to a chunk behind the passband chunks; the
    band fluxes must come back in the caller's band order whatever the mixture."""
    names = ["SPIRE_250um", "Y_delta_1300um", "PACS_100um", "Y_delta_2000um", "Y_delta_70um", "SCUBA2_850um"]
    like = mbb.likelihood(response=True)
    like.set_phot(names, np.ones(6), np.ones(6))
    truth = np.array([22.0, 1.6, 180.0, 2.8, 50.0])
    flux = like.model_flux(truth)[0]
    unc = 0.1 * flux + 0.1
    like.set_phot(names, flux, unc)
    bands = [(r.wavelength, np.ones(1), 1.0) if r.isdelta else (r.wavelength, r._sedmult, r._normfac)
             for r in like._responses]
    assert [r.isdelta for r in like._responses] == [False, True, False, True, True, False]
    rng = np.random.RandomState(12)
    pars = truth * (1.0 + 0.08 * rng.normal(size=(130, 5)))
    orc = oracle.OracleLikelihood(flux, unc, bands=bands, has_uplim=[int(b) for b in like.has_uplims],
                                  uplim=like.uplims)
    ref, rflux = orc(pars, nthreads=4, return_flux=True)
    rec_allclose(like.model_flux(pars), rflux, rtol=1e-12, kind="band flux")
    lnl_close(like(pars), ref)
    for wpb, thr in ((1, 64), (4, 256), (16, 256)):
        like.context.set_option("walkers_per_group", wpb); like.context.set_option("block_threads", thr)
        assert np.array_equal(like(pars), like.__call__(pars))
        lnl_close(like(pars), ref)
    like.context.set_option("walkers_per_group", 0); like.context.set_option("block_threads", 0)


def test_band_sizes_around_the_chunk_and_row_boundaries(mbb, oracle, tmp_path):
    """Passbands of 2 ... 193 samples: nothing but tail rows (< 49 samples), a padded
    chunk (49-63), exact multiples of 16 and 64, and one sample either side of each --
    band fluxes and lnL against the oracle, and the same numbers from the layout with a
    chunk per leftover."""
    from mbb_emcee_amd import response
    sizes = [2, 15, 16, 17, 32, 33, 47, 48, 49, 63, 64, 65, 80, 81, 127, 128, 129, 193]
    rng = np.random.RandomState(21)
    bands = []
    for k, n in enumerate(sizes):
        lo = 60.0 + 45.0 * k
        x = np.linspace(lo, lo + 30.0, n)
        t = 0.2 + rng.rand(n)
        fn = tmp_path / ("sz%d.txt" % n)
        np.savetxt(fn, np.column_stack([x, t]))
        r = response("SZ%d" % n)
        r.setup(str(fn), xtype="wave", xunits="microns", senstype="energy", normtype="power",
                xnorm=lo + 15.0, normparam=-1.0)
        bands.append(r)
    like = mbb.likelihood(response=True)
    for r in bands:
        like._responsewheel._responses[r.name] = r
    names = [r.name for r in bands]
    like.set_phot(names, np.ones(len(sizes)), np.ones(len(sizes)))
    truth = np.array([25.0, 1.7, 150.0, 2.5, 60.0])
    flux = like.model_flux(truth)[0]
    unc = 0.1 * flux + 0.05
    like.set_phot(names, flux, unc)
    pars = truth * (1.0 + 0.07 * rng.normal(size=(150, 5)))
    orc = oracle.OracleLikelihood(flux, unc, bands=[(r.wavelength, r._sedmult, r._normfac) for r in bands],
                                  has_uplim=[int(b) for b in like.has_uplims], uplim=like.uplims)
    ref, rflux = orc(pars, nthreads=4, return_flux=True)
    got, gflux = like(pars), like.model_flux(pars)
    rec_allclose(gflux, rflux, rtol=1e-12, kind="band flux")
    lnl_close(got, ref)
    nfull = sum(n // 64 + (1 if n % 64 > 48 else 0) for n in sizes)
    nrows = sum(-(-(n % 64) // 16) for n in sizes if n % 64 <= 48)
    assert like.context.info("nchunk") == nfull + -(-nrows // 4)
    like.context.set_option("pack_tails", 0)
    like._dirty = True
    rec_allclose(like.model_flux(pars), gflux, rtol=2e-15, kind="band flux")
    assert like.context.info("nchunk") == sum(-(-n // 64) for n in sizes)
    like.context.set_option("pack_tails", 1)
    like._dirty = True


def test_long_passbands_and_segment_length(mbb, oracle, tmp_path):
    """Passbands far longer than the wheel's (5000 and 12000 samples; tables too
    big for LDS) and every segment length give the oracle's answer; a given
    segment length is bitwise reproducible across geometries."""
    from mbb_emcee_amd import response
    rng = np.random.RandomState(8)
    bands = []
    for k, (n, lo, hi) in enumerate([(5000, 180.0, 320.0), (12000, 300.0, 700.0), (77, 60.0, 90.0)]):
        x = np.linspace(lo, hi, n)
        t = np.exp(-0.5 * ((x - 0.5 * (lo + hi)) / (0.2 * (hi - lo))) ** 2) * (1 + 0.1 * rng.rand(n))
        fn = tmp_path / ("band%d.txt" % k)
        np.savetxt(fn, np.column_stack([x, t]))
        r = response("B%d" % k)
        r.setup(str(fn), xtype="wave", xunits="microns", senstype="energy", normtype="power",
                xnorm=0.5 * (lo + hi), normparam=-1.0)
        bands.append(r)
    like = mbb.likelihood(response=True)
    for r in bands:
        like._responsewheel._responses[r.name] = r
    names = [r.name for r in bands]
    like.set_phot(names, np.ones(3), np.ones(3))
    truth = np.array([14.0, 1.9, 500.0, 3.2, 30.0])
    flux = like.model_flux(truth)[0]
    like.set_phot(names, flux, 0.1 * flux)
    pars = truth * (1.0 + 0.05 * rng.normal(size=(200, 5)))
    ref = oracle.OracleLikelihood(flux, 0.1 * flux, bands=[(r.wavelength, r._sedmult, r._normfac) for r in bands],
                                  has_uplim=[int(b) for b in like.has_uplims], uplim=like.uplims)(pars, nthreads=8)
    base = like(pars)
    lnl_close(base, ref)
    # 5000, 12000 and (third band) samples: full chunks, their leftovers share tail chunks
    nfull = sum(len(r._sedmult) // 64 + (1 if len(r._sedmult) % 64 > 48 else 0) for r in bands)
    nrows = sum(-(-(len(r._sedmult) % 64) // 16) if len(r._sedmult) % 64 <= 48 else 0 for r in bands)
    assert like.context.info("nchunk") == nfull + -(-nrows // 4)
    like.context.set_option("pack_tails", 0)                # one chunk per leftover, as a check
    like._dirty = True
    assert np.allclose(like(pars), base, rtol=1e-13, atol=0) and like.context.info("nchunk") == 79 + 188 + 2
    like.context.set_option("pack_tails", 1)
    like._dirty = True
    for seg in (1, 2, 8, 64):
        like.context.set_option("seg_chunks", seg)
        like._dirty = True
        got = like(pars)
        lnl_close(got, ref)
        like.context.set_option("walkers_per_group", 7); like.context.set_option("block_threads", 192)
        assert np.array_equal(got, like(pars))
        like.context.set_option("walkers_per_group", 0); like.context.set_option("block_threads", 0)
    like.context.set_option("seg_chunks", 4)


# ----------------------------------------------- chain post-processing kernels
@pytest.mark.parametrize("name,opthin,noalpha", VARIANTS)
def test_freq_integrate_vs_reference_quad(mbb, ctx, g_sed, name, opthin, noalpha):
    """modified_blackbody.freq_integrate (scipy quad, epsrel 1.5e-8 in the
    reference) against the split Gauss-Legendre kernel."""
    from mbb_emcee_amd.modified_blackbody import um_to_GHz
    pars = g_sed["pars"]
    ref = g_sed[name + "/freq_integrate"]
    sel = np.nonzero(np.isfinite(ref[:, 0]))[0]
    for col, (lo, hi) in enumerate(((24.0, 3000.0), (42.5, 122.5))):
        out, st = ctx.sed_integrate(pars[sel], opthin, noalpha, 500.0, um_to_GHz / hi, um_to_GHz / lo)
        assert np.all(st == 0)
        rec_allclose(1e-17 * out, ref[sel, col], rtol=2e-7, kind="freq_integrate")
    m = mbb.modified_blackbody(*pars[sel[3]], opthin=opthin, noalpha=noalpha)
    rec_allclose(m.freq_integrate(3000.0, 24.0), ref[sel[3], 0], rtol=2e-7, kind="freq_integrate")
    with pytest.raises(ValueError):
        m.freq_integrate(0.0, 10.0)


def test_postprocess_chain(mbb, g_lnl, oracle):
    from mbb_emcee_amd import postprocess as pp
    like = _cfg2_like(mbb, g_lnl)
    chain = g_lnl["cfg2/thick_walpha/pars"][:60].reshape(6, 10, 5)
    pk = pp.peak_wavelength(like, chain)
    assert pk.shape == (6, 10)
    for i, j in ((0, 0), (3, 7), (5, 9)):
        rec_allclose(pk[i, j], oracle.OracleSED(*chain[i, j]).max_wave(), rtol=1e-10, kind="max_wave")
    L = pp.lir(like, chain, redshift=2.0, lumdist_mpc=15000.0)
    fi = pp.freq_integral(like, chain, 24.0, 3000.0)
    rec_allclose(L, 3.11749657e4 * 15000.0 ** 2 * fi, rtol=1e-14, kind="freq_integrate")
    # against a direct high-accuracy quadrature of the oracle SED
    from scipy.integrate import quad
    sed = oracle.OracleSED(*chain[2, 4])
    wm = sed.wavemerge
    nu0, nu1, num = 299792.458 / 3000.0, 299792.458 / 24.0, 299792.458 / wm
    val = sum(quad(lambda f: float(sed.f_nu(f)[0]), a, b, epsrel=1e-12, limit=200)[0]
              for a, b in ((nu0, num), (num, nu1)))
    rec_allclose(fi[2, 4], 1e-17 * val, rtol=1e-10, kind="freq_integrate")
    md = pp.dustmass(like, chain, redshift=2.0, lumdist_mpc=15000.0)
    assert md.shape == (6, 10) and np.all(md > 0)       # identities: tests/test_host_cpu.py


@pytest.mark.parametrize("name,opthin,noalpha", VARIANTS)
def test_postprocess_vs_reference_results(mbb, g_res, name, opthin, noalpha):
    """SURVEY.md 8f rank 4, pinned: `postprocess.*` on the GPU against what the reference's own mbb_results
    returned for the same 32 x 16 x 5 chain (results.py:570-581, :627-674, :746-801, :895-944;
    tests/golden/make_golden_results.py).  Tolerances: SURVEY 8c's for max_wave (1e-10) and fluxes (1e-12);
    L_IR is held to 2e-7 because the reference's own `quad` stops at epsrel 1.5e-8 of ITS estimate."""
    from mbb_emcee_amd import postprocess as pp
    k = name + "/"
    chain = g_res[k + "chain"]
    z, dl = float(g_res["redshift"]), float(g_res["lumdist_mpc"])
    bands = [str(b) for b in g_res["bands"]]
    like = mbb.likelihood(response=True, opthin=opthin, noalpha=noalpha)
    like.set_phot(bands, g_res[k + "data_flux"], 0.1 * g_res[k + "data_flux"] + 1.0)
    rec_allclose(like(chain.reshape(-1, 5)).reshape(32, 16), g_res[k + "lnprobability"], rtol=1e-10, kind="lnL")
    rec_allclose(pp.peak_wavelength(like, chain, model="reference"), g_res[k + "peaklambda"], rtol=1e-10, kind="max_wave")
    rec_allclose(pp.peak_wavelength(like, chain), g_res[k + "peaklambda_own_model"], rtol=1e-10, kind="max_wave")
    # L_IR: the reference's `quad` (default epsrel 1.5e-8 of its own error ESTIMATE) does not know about the kink
    # of f_nu at the merge frequency: against a quadrature split there, its values for these chains are off by up
    # to 3.9e-7 (thick+alpha, entry [21, 12]; 1.2e-8 thin+alpha, 1e-10 without alpha).  So: 5e-7 against the
    # reference's numbers, and 1e-9 against the split quadrature of the oracle SED for a sample that includes
    # the reference's worst entry.
    got_lir = pp.lir(like, chain, z, dl)
    rec_allclose(got_lir, g_res[k + "lir"], rtol=5e-7, kind="L_IR vs the reference's quad")
    from scipy.integrate import quad
    from oracle import oracle as O
    opz, flat = 1.0 + z, got_lir.reshape(-1)
    for i in sorted(set(range(0, 512, 37)) | {348}):
        sed = O.OracleSED(*chain.reshape(-1, 5)[i], opthin=opthin, noalpha=noalpha)
        pts = [299792.458 / (1000.0 * opz), 299792.458 / (8.0 * opz)]
        if not noalpha and pts[0] < 299792.458 / sed.wavemerge < pts[1]:
            pts.insert(1, 299792.458 / sed.wavemerge)
        val = sum(quad(lambda f: float(sed.f_nu(f)[0]), a, b, epsrel=1e-13, epsabs=0, limit=400)[0]
                  for a, b in zip(pts[:-1], pts[1:]))
        rec_allclose(flat[i], 3.11749657e4 * dl ** 2 * 1e-17 * val, rtol=1e-9, kind="L_IR vs split quadrature")
    rec_allclose(pp.dustmass(like, chain, z, dl), g_res[k + "dustmass"], rtol=1e-13, kind="dust mass")
    specs = [str(b) for b in g_res["pred_bands"]] + [float(w) for w in g_res["pred_waves"]]
    got = pp.predict_flux(like, chain, specs)
    assert got.shape == (32, 16, 4)
    for i, s in enumerate(specs):
        rec_allclose(got[..., i], g_res[k + "predict/" + (s if isinstance(s, str) else "%g" % s)], rtol=1e-12,
                     kind="predicted flux")
    rec_allclose(pp.predict_flux(like, chain, "SCUBA2_450um"), g_res[k + "predict/SCUBA2_450um"], rtol=1e-12,
                 kind="predicted flux")


def test_predict_flux_over_a_chain(mbb, g_lnl, oracle):
    """postprocess.predict_flux (results.py:895-944): predicted band fluxes through any
    passband of the wheel and SED values at any wavelength for every chain entry, in
    batched calls.  results.py cannot be imported (astropy), so this is held against the
    pinned pieces: likelihood.model_flux for the fit's own bands, the oracle's response
    integration for a band that was not fitted, and the oracle SED."""
    from mbb_emcee_amd import postprocess as pp
    like = _cfg2_like(mbb, g_lnl)
    chain = g_lnl["cfg2/thick_walpha/pars"][:60].reshape(6, 10, 5)
    bands = [str(b) for b in g_lnl["cfg2/bands"]]
    own = pp.predict_flux(like, chain, bands)
    assert own.shape == (6, 10, len(bands))
    assert np.array_equal(own.reshape(60, -1), like.model_flux(chain.reshape(60, 5)))
    one = pp.predict_flux(like, chain, "SPIRE_350um")
    assert one.shape == (6, 10) and np.array_equal(one, own[..., bands.index("SPIRE_350um")])
    # a passband of the wheel that was not part of the fit, and two wavelengths, mixed
    mix = pp.predict_flux(like, chain, ["MAMBO2_1.2mm", 433.0, 1300.0])
    r = like._responsewheel["MAMBO2_1.2mm"]
    orc = oracle.OracleLikelihood(np.ones(1), np.ones(1), bands=[(r.wavelength, r._sedmult, r._normfac)],
                                  lowlim=[-np.inf] * 5, has_uplim=[0] * 6, uplim=[np.inf] * 6)
    _, rflux = orc(chain.reshape(60, 5), nthreads=2, return_flux=True)
    rec_allclose(mix[..., 0].reshape(60), rflux[:, 0], rtol=FLUX_RTOL, kind="band flux")
    for j, wv in ((1, 433.0), (2, 1300.0)):
        ref = np.array([oracle.OracleSED(*row)(np.array([wv]))[0] for row in chain.reshape(60, 5)])
        rec_allclose(mix[..., j].reshape(60), ref, rtol=SED_RTOL, kind="f_nu")
    rec_allclose(pp.predict_flux(like, chain, 500.0), chain[..., 4], rtol=1e-13, kind="f_nu(wavenorm) = fnorm")
    with pytest.raises(ValueError):
        pp.predict_flux(like, chain, "NoSuchBand_1um")
    with pytest.raises(ValueError):
        pp.predict_flux(like, chain, -3.0)


def test_boundary_fast_path_equals_the_general_path(mbb, g_lnl):
    """likelihood.__call__ takes a shorter way for what a sampler hands over per half-step (a float array of rows
    or one row, nothing changed since the last call): rows written straight into the block the kernel reads,
    mbb_lnlike_call(ctx, n), the result copied out of the pinned block.  It must give what mbb_lnlike_batch gives,
    bit for bit, for every kind of array, across capacity growth, and must hand everything unusual -- rows the
    reference raises for, changed limits, switched host-path options -- to the general path."""
    import pickle
    like = _cfg2_like(mbb, g_lnl)
    ctx = like._sync_device()
    allp = np.tile(g_lnl["cfg2/thick_walpha/pars"], (20, 1))          # 6620 rows, edge rows (-inf) among them
    want = ctx.lnlike_batch(allp)[0]
    assert np.isneginf(want).any()
    for n in (125, 10, 300, 1, 125, 1000, 6620, 250, 125):           # capacity grows 256 -> 8192 on the way
        got = like(allp[:n])
        assert type(got) is np.ndarray and got.shape == (n,) and np.array_equal(got, want[:n], equal_nan=True), n
        # (the call went the short way: through the C-level callable where the extension is built -- it is, on the GPU box,
        # by __graft_entry__.build() -- else through the views of the two blocks)
        assert like._fast is not None and like._fast[3] >= n and (like._fast[7] is not None or n in like._fast[2])
    assert got.base is None                                           # a copy, not a view of the pinned block
    import importlib
    _lk_mod = importlib.import_module("mbb_emcee_amd.likelihood")      # (the module: the package exports the class under its name)
    if _lk_mod._mbbfast is not None:
        # the same calls with the extension out of the way: the numpy / ctypes form of the short way, bit for bit
        fc = like._fast[7]
        assert fc is not None
        a = like(allp[:125]); b1 = like(allp[7])
        like._fast = like._fast[:7] + (None,) + like._fast[8:]
        assert np.array_equal(like(allp[:125]), a, equal_nan=True) and like(allp[7]) == b1 and 125 in like._fast[2]
        like._fast = None
    keep = like(allp[:125])
    like(allp[125:250])
    assert np.array_equal(keep, want[:125], equal_nan=True)           # ... so a later call does not change it
    # other layouts and float types take the same way; one row gives a Python float
    f32 = allp[:200].astype(np.float32)
    assert np.array_equal(like(f32), ctx.lnlike_batch(f32.astype(np.float64))[0], equal_nan=True)
    assert np.array_equal(like(np.asfortranarray(allp[:200])), want[:200], equal_nan=True)
    assert np.array_equal(like(allp[:400:2]), want[:400:2], equal_nan=True)
    one = like(allp[3])
    assert type(one) is float and one == want[3]
    assert like(allp[3].astype(np.float32)) == ctx.lnlike_batch(allp[3].astype(np.float32).astype(np.float64))[0][0]
    assert like(list(allp[3])) == want[3] and np.array_equal(like(allp[:7].tolist()), want[:7])      # (general path)
    # rows the reference raises for (modified_blackbody.py:219-224) raise here too, and the next call is fine
    # (with the lower limits where they are such rows are gated to -inf first, likelihood.py:643-670; so they are opened)
    bad = allp[:50].copy(); bad[17, 3] = -1.0
    assert np.isneginf(like(bad)[17])
    like.set_lowlim("alpha", -5.0); like.set_lowlim("beta", -5.0)
    with pytest.raises(ValueError, match="alpha"):
        like(bad)                                                     # (first call after the change: the general path)
    with pytest.raises(ValueError, match="alpha"):
        like(bad)                                                     # (the boundary call, which hands over to it)
    bad[17, 3] = 3.0; bad[4, 1] = -0.5
    with pytest.raises(ValueError, match="beta"):
        like(bad[4])
    ok50 = like(allp[:50])
    like.set_lowlim("alpha", 0.1); like.set_lowlim("beta", 0.1)
    fin = np.isfinite(want[:50])
    assert np.array_equal(ok50[fin], want[:50][fin]) and np.array_equal(like(allp[:50]), want[:50], equal_nan=True)
    nan = allp[:50].copy(); nan[9, 0] = np.nan
    assert np.isnan(like(nan)[9]) and np.array_equal(np.delete(like(nan), 9), np.delete(want[:50], 9), equal_nan=True)
    # a changed limit goes through the general path once (upload) and the result is the new one
    like.set_uplim("T", 12.0)
    new = like(allp[:125])
    assert not np.array_equal(new, want[:125], equal_nan=True) and np.array_equal(new, ctx.lnlike_batch(allp[:125])[0], equal_nan=True)
    like._has_uplim[0] = False; like._dirty = True
    assert np.array_equal(like(allp[:125]), want[:125], equal_nan=True)
    # switched host-path options: still right, whichever way it goes
    for opt, val in (("zero_copy", 0), ("zero_copy", 1), ("bar_params", 0), ("bar_params", 1), ("spin_wait", 0), ("spin_wait", 2),
                     ("launch_api", 1), ("launch_api", 0)):
        ctx.set_option(opt, val)
        for _ in range(3):
            assert np.array_equal(like(allp[:125]), want[:125], equal_nan=True), (opt, val)
    # a pickled copy builds its own context and buffers
    twin = pickle.loads(pickle.dumps(like))
    assert twin._fast is None and np.array_equal(twin(allp[:125]), want[:125], equal_nan=True)


@pytest.mark.parametrize("glue", ["extension", "numpy"])
def test_boundary_call_survives_the_blocks_being_made_anew(mbb, g_lnl, glue):
    """likelihood.__call__ caches addresses of the blocks the boundary call works on; any OTHER
    call on the context that needs more rows than they hold -- model_flux of a chain, a list of rows through the general
    path, a 3-D array, the sharded evaluation, ctx.lnlike_batch -- frees them natively.  The next boundary call must see
    that (the context's generation word) before it writes anything: right results, no write into freed memory."""
    import importlib
    lk_mod = importlib.import_module("mbb_emcee_amd.likelihood")
    if glue == "extension" and lk_mod._mbbfast is None:
        pytest.skip("the _mbbfast extension is not built here")
    allp = np.tile(g_lnl["cfg2/thick_walpha/pars"], (20, 1))          # 6620 rows
    a64 = np.ascontiguousarray(allp[:64])

    def fresh():
        like = _cfg2_like(mbb, g_lnl)
        like._sync_device()
        if glue == "numpy":
            saved = lk_mod._mbbfast
            lk_mod._mbbfast = None
            try:
                like(a64)
            finally:
                lk_mod._mbbfast = saved
            assert like._fast[7] is None
        return like
    like0 = _cfg2_like(mbb, g_lnl)
    want = like0._sync_device().lnlike_batch(allp)[0]
    growers = {
        "model_flux": lambda like: like.model_flux(allp),
        "list": lambda like: like([list(r) for r in allp[:3000]]),
        "lnlike_batch": lambda like: like._sync_device().lnlike_batch(allp),
        "3d": lambda like: like(allp[None, :, :]),
    }
    for name, grow in growers.items():
        like = fresh()
        for _ in range(5):                                            # (a run of calls: the served path too)
            assert np.array_equal(like(a64), want[:64], equal_nan=True)
        f_old = like._fast
        assert f_old is not None and f_old[3] == 256
        gen_before = int(f_old[8][0])
        grow(like)
        assert int(f_old[8][0]) != gen_before, name                   # the native side says so ...
        for _ in range(5):
            assert np.array_equal(like(a64), want[:64], equal_nan=True), name
        assert like._fast is not f_old and like._fast[3] >= 256, name  # ... and the cached views were dropped, not written through
        one = like(allp[3])
        assert type(one) is float and one == want[3]


def test_served_boundary_equals_the_launches(mbb, g_lnl):
    """After a few boundary calls in a row the rows go to a kernel that stays on the GPU and is rung through the BAR
    (k_serve) instead of a launch per call.  Same results bit for bit, for every batch size up to a row per CU, for one
    row, with a covariance matrix; anything else on the context makes it leave first; it leaves by itself when nothing
    comes; a request it does not answer in time is evaluated by a launch."""
    import time
    # every model variant (a row's quadrature starts beside its constructor, or after it: option serve_overlap), with and
    # without the merge point's power law, priors incl. the peak wavelength's (a second root in the constructor), rows the
    # gate rejects
    for variant, overlap in (("thick_noalpha", 1), ("thin_noalpha", 1), ("thin_walpha", 0), ("thick_walpha", 0)):
        like = mbb.likelihood(response=True, opthin=variant.startswith("thin"), noalpha=variant.endswith("noalpha"))
        like.set_phot([str(b) for b in g_lnl["cfg2/bands"]], g_lnl["cfg2/%s/flux" % variant], g_lnl["cfg2/%s/unc" % variant])
        if variant == "thin_walpha":
            like.set_gaussian_prior("peaklam", 250.0, 20.0)
            like.set_uplim("lambda_peak", 265.0)
        ctx = like._sync_device()
        pars = np.tile(g_lnl["cfg2/%s/pars" % variant], (2, 1))[:125]
        ctx.set_option("serve", 0)
        want = like(pars).copy()
        ctx.set_option("serve", 2); ctx.set_option("serve_overlap", 2 * overlap)
        for n in (125, 125, 125, 125, 9, 125):
            assert np.array_equal(like(pars[:n]), want[:n], equal_nan=True), (variant, overlap, n)
        assert ctx.info("serving") == 1 and ctx.info("serve_fallbacks") == 0
        # narrow workgroups (options of the launch geometry hold for the served kernel too): one wave, four waves
        for thr in (64, 256, 0):
            ctx.set_option("block_threads", thr)
            for n in (125, 125, 125, 125, 7):
                assert np.array_equal(like(pars[:n]), want[:n], equal_nan=True), (variant, overlap, thr, n)
            assert ctx.info("serving") == 1 and ctx.info("last_threads") == (thr or ctx.info("last_threads"))
        del like
    for variant, cov in (("thick_walpha", False), ("thin_walpha", False), ("thick_walpha", True)):
        opthin = variant.startswith("thin")
        like = mbb.likelihood(response=True, opthin=opthin)
        like.set_phot([str(b) for b in g_lnl["cfg2/bands"]], g_lnl["cfg2/%s/flux" % variant], g_lnl["cfg2/%s/unc" % variant])
        if cov:
            unc = g_lnl["cfg2/%s/unc" % variant]
            like.set_cov(np.diag(unc ** 2) + 0.2 * np.outer(unc, unc))
        ctx = like._sync_device()
        cus = ctx.info("cu_count")
        allp = np.tile(g_lnl["cfg2/%s/pars" % variant], (2, 1))
        ctx.set_option("serve", 0)
        want = like(allp[:cus]).copy()
        w1 = like(allp[3])
        ctx.set_option("serve", 2)                                   # (also beside the test session's other contexts)
        assert ctx.info("serving") == 0
        for n in (125, 125, 125, 125, 7, cus, 1, 125, 64):
            got = like(allp[:n])
            assert np.array_equal(got, want[:n], equal_nan=True), (variant, cov, n)
        assert ctx.info("serving") == 1 and ctx.info("serve_requests") >= 6 and ctx.info("serve_fallbacks") == 0
        assert like(allp[3]) == w1 and type(like(allp[3])) is float and ctx.info("serving") == 1
        # how far ahead of its scan the host asks for the record lines (option serve_prefetch) is the host's business alone
        for ahead in (0, 1, 7, 256, 32):
            ctx.set_option("serve_prefetch", ahead)
            for n in (125, 125, 125, 125, cus, 33, 1):
                assert np.array_equal(like(allp[:n]), want[:n], equal_nan=True), (variant, cov, ahead, n)
            assert ctx.info("serve_fallbacks") == 0
        # more rows than CUs: two rows a workgroup (a 1000-walker ensemble's half-steps through a host sampler); more than
        # twice the CUs: by a launch (the server leaves), then served again after a few calls in a row
        wide = np.tile(allp, (2, 1))
        ctx.set_option("serve", 0)
        want_wide = like(wide[:2 * cus + 8]).copy()
        ctx.set_option("serve", 2)
        for _ in range(3):
            assert np.array_equal(like(allp[:125]), want[:125], equal_nan=True)
        assert ctx.info("serving") == 1
        for n in (cus + 8, 2 * cus, cus + 1, 300):
            assert np.array_equal(like(wide[:n]), want_wide[:n], equal_nan=True), (variant, cov, n)
            assert ctx.info("serving") == 1 and ctx.info("serve_grid") == cus, n
        big = like(wide[:2 * cus + 8])
        assert ctx.info("serving") == 0 and np.array_equal(big, want_wide, equal_nan=True)
        for _ in range(4):
            assert np.array_equal(like(allp[:125]), want[:125], equal_nan=True)
        assert ctx.info("serving") == 1 and ctx.info("serve_fallbacks") == 0
        # anything else on the context: it leaves first, and what was set is what the next calls see
        like.set_uplim("T", 12.0)
        changed = like(allp[:125])
        assert ctx.info("serving") == 0 and not np.array_equal(changed, want[:125], equal_nan=True)
        for _ in range(4):
            assert np.array_equal(like(allp[:125]), changed, equal_nan=True)
        assert ctx.info("serving") == 1
        like._has_uplim[0] = False; like._dirty = True
        assert np.array_equal(like(allp[:125]), want[:125], equal_nan=True)
        # it leaves by itself when nothing comes, and the next request is evaluated all the same
        ctx.set_option("serve_idle_us", 200)
        ctx.set_option("serve_lease_us", 0)                          # (the lease is the next paragraph's)
        for _ in range(4):
            like(allp[:125])
        assert ctx.info("serving") == 1
        time.sleep(0.05)
        nf = ctx.info("serve_fallbacks")
        assert np.array_equal(like(allp[:125]), want[:125], equal_nan=True)
        assert ctx.info("serve_fallbacks") == nf + 1                 # (found gone: by a launch)
        for _ in range(4):
            assert np.array_equal(like(allp[:125]), want[:125], equal_nan=True)
        assert ctx.info("serving") == 1
        # its lease: a server is sent away by the host after `serve_lease_us` in one go -- processes the cross-process
        # registry cannot show get the CUs at least that often -- the rows of that call go by a launch (not a fall-back:
        # nothing was lost), the next server starts after the next few calls in a row
        ctx.set_option("serve_idle_us", 1000); ctx.set_option("serve_lease_us", 3000)
        ny, nf, t0 = ctx.info("serve_lease_yields"), ctx.info("serve_fallbacks"), time.perf_counter()
        ncalls = 0
        while time.perf_counter() - t0 < 0.03:
            assert np.array_equal(like(allp[:125]), want[:125], equal_nan=True)
            ncalls += 1
        assert 3 <= ctx.info("serve_lease_yields") - ny <= 12 and ctx.info("serve_fallbacks") == nf, \
            (ctx.info("serve_lease_yields") - ny, ctx.info("serve_fallbacks") - nf, ncalls)
        ctx.set_option("serve_lease_us", 50000)
        for _ in range(4):
            like(allp[:125])
        assert ctx.info("serving") == 1
        # the device sampler on the same context while a server is resident
        s = mbb.DeviceEnsembleSampler(64, 5, like, seed=3)
        p0 = np.array([12.0, 1.8, 600.0, 3.0, 40.0]) * (1.0 + 0.02 * np.random.RandomState(2).normal(size=(64, 5)))
        pos, lnp, _ = s.run_mcmc(p0, 5)
        assert ctx.info("serving") == 0 and ctx.info("flow_fallbacks") == 0
        lnl_close(lnp, like(pos))
        del s, like


def test_served_boundary_with_several_likelihoods_alive(mbb, g_lnl):
    """One server per device and process: a likelihood that comes to the device while a sibling's kernel is resident
    there tells it to leave first (it holds the CUs), is evaluated, and is served itself after its own few calls in a
    row; the sibling starts again after its next few.  Option "serve" at its default (1), other contexts of the test
    session alive besides."""
    a, b = _cfg2_like(mbb, g_lnl), _cfg2_like(mbb, g_lnl)
    b.set_uplim("T", 12.0)                                            # (so that the two do not give the same numbers)
    ca, cb = a._sync_device(), b._sync_device()
    p = np.tile(g_lnl["cfg2/thick_walpha/pars"], (2, 1))[:125]
    for c in (ca, cb):
        c.set_option("serve", 0)
    wa, wb = a(p).copy(), b(p).copy()
    assert not np.array_equal(wa, wb, equal_nan=True)
    for c in (ca, cb):
        c.set_option("serve", 1)
    for _ in range(5):
        assert np.array_equal(a(p), wa, equal_nan=True)
    assert ca.info("serving") == 1 and cb.info("serving") == 0
    assert np.array_equal(b(p), wb, equal_nan=True)                   # b comes to the device: a's server leaves
    assert ca.info("serving") == 0 and cb.info("serving") == 0
    for _ in range(4):
        assert np.array_equal(b(p), wb, equal_nan=True)
    assert cb.info("serving") == 1 and ca.info("serving") == 0
    for k in range(6):                                                # alternating callers: always right, nobody served
        assert np.array_equal((a if k % 2 == 0 else b)(p), wa if k % 2 == 0 else wb, equal_nan=True)
    assert ca.info("serving") == 0 and cb.info("serving") == 0
    # a context whose server was sent away needs twice as many calls in a row the next time (3 -> 6 -> ... 64): two
    # likelihoods used in turns do not spend their time starting and stopping kernels
    for k in range(5):
        assert np.array_equal(a(p), wa, equal_nan=True)
    assert ca.info("serving") == 0
    assert np.array_equal(a(p), wa, equal_nan=True) and ca.info("serving") == 1
    # any other entry point of a sibling does the same
    s = mbb.DeviceEnsembleSampler(64, 5, b, seed=3)
    p0 = np.array([12.0, 1.8, 600.0, 3.0, 40.0]) * (1.0 + 0.02 * np.random.RandomState(2).normal(size=(64, 5)))
    pos, lnp, _ = s.run_mcmc(p0, 3)
    assert ca.info("serving") == 0 and np.all(np.isfinite(lnp))
    assert ca.info("serve_fallbacks") == 0 and cb.info("serve_fallbacks") == 0
    del s, a, b


def test_two_threads_two_likelihoods_one_device(mbb, g_lnl):
    """contexts of one process that share a device used from different THREADS at once -- ctypes
    and the _mbbfast extension release the GIL around the native call.  A likelihood that comes to the device tells a
    sibling's resident server to leave; that touches the sibling's serve state from the visitor's thread, which is guarded
    by a per-context mutex since round 5.  Two threads, a likelihood each, a loop of boundary calls each, at the same time
    (so that servers start, are sent away and start again all the time), plus a third thread that uses other entry
    points of one of the contexts' SIBLING: every result right, no deadlock, the feature not struck out."""
    import threading, time
    a, b = _cfg2_like(mbb, g_lnl), _cfg2_like(mbb, g_lnl)
    b.set_uplim("T", 12.0)
    p = np.ascontiguousarray(np.tile(g_lnl["cfg2/thick_walpha/pars"], (2, 1))[:125])
    for like in (a, b):
        like._sync_device().set_option("serve", 0)
    wa, wb = a(p).copy(), b(p).copy()
    for like in (a, b):
        like._sync_device().set_option("serve", 2)              # (2: this test session's other processes do not matter)
    bad, done = [], []

    def loop(like, want, ncalls, pause):
        try:
            for i in range(ncalls):
                if not np.array_equal(like(p), want, equal_nan=True):
                    bad.append(i)
                if pause:
                    time.sleep(pause)
            done.append(ncalls)
        except Exception as e:                                    # noqa
            bad.append(repr(e))
    # (the second thread pauses between its calls: the first gets runs of calls long enough to be served, and every call
    # of the second finds a server of the first's to send away -- from its own thread, while the first is inside a call)
    ths = [threading.Thread(target=loop, args=(a, wa, 6000, 0.0)), threading.Thread(target=loop, args=(b, wb, 150, 0.0004))]
    t0 = time.time()
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in ths), "a thread is stuck"
    assert bad == [] and len(done) == 2, bad[:5]
    assert time.time() - t0 < 60
    ca, cb = a._sync_device(), b._sync_device()
    assert ca.info("serve_enabled") == 2 and cb.info("serve_enabled") == 2 and ca.info("serve_rests") == 0 and cb.info("serve_rests") == 0
    # (how often each was served says how the two got along; nothing is asserted about it)
    print("served requests: %d / %d, fall-backs %d / %d" % (ca.info("serve_requests"), cb.info("serve_requests"),
                                                             ca.info("serve_fallbacks"), cb.info("serve_fallbacks")))
    assert ca.info("serve_requests") > 100                      # (the first thread WAS served in between: servers came and went)
    for like in (a, b):
        like._sync_device().set_option("serve", 1)


def test_a_server_is_as_wide_as_the_calls_have_rows(mbb, g_lnl):
    """A resident server (k_serve) holds a CU per workgroup; it has a workgroup per row of the widest call so far, in eights,
    not one per CU -- what it does not hold is there for other processes (test_served_boundary_in_two_processes_on_one_gpu).
    A call of more rows than it has workgroups makes it leave and the next one, wide enough, starts with the same call;
    option "serve_grid" fixes the width; results bit for bit the launches' throughout."""
    like = _cfg2_like(mbb, g_lnl)
    ctx = like._sync_device()
    cus = ctx.info("cu_count")
    allp = np.tile(g_lnl["cfg2/thick_walpha/pars"], (3, 1))[:cus]
    ctx.set_option("serve", 0)
    want = like(allp).copy()
    ctx.set_option("serve", 2); ctx.set_option("serve_after", 1); ctx.set_option("serve_budget_us", 100000)
    ctx.set_option("serve_lease_us", 0)
    assert ctx.info("serve_grid") == 0                                       # (nobody resident)
    for _ in range(3):
        assert like(allp[3]) == want[3]
    assert ctx.info("serving") == 1 and ctx.info("serve_grid") == 8         # a row at a time: eight workgroups
    r0 = ctx.info("serve_resizes")
    for n, grid in ((5, 8), (8, 8), (9, 16), (3, 16), (125, 128), (64, 128), (129, 136), (cus, cus), (1, cus)):
        for _ in range(3):
            assert np.array_equal(like(allp[:n]), want[:n], equal_nan=True), n
        assert ctx.info("serving") == 1 and ctx.info("serve_grid") == min(grid, cus), (n, grid, ctx.info("serve_grid"))
    assert ctx.info("serve_resizes") - r0 == 4 and ctx.info("serve_fallbacks") == 0
    # a width asked for (what the share of a device does to a pool's worker): a server narrower than a call has rows takes two
    # rows a workgroup; calls of more rows than that go by a launch and the server stays away until the calls fit again
    ctx.set_option("serve_grid", 32)
    for n in (20, 32, 33, 64, 47, 20):
        for _ in range(3):
            assert np.array_equal(like(allp[:n]), want[:n], equal_nan=True), n
        assert ctx.info("serving") == 1 and ctx.info("serve_grid") == 32, n
    assert np.array_equal(like(allp[:65]), want[:65], equal_nan=True) and ctx.info("serving") == 0
    assert np.array_equal(like(allp[:cus]), want, equal_nan=True) and ctx.info("serving") == 0
    for _ in range(3):
        assert np.array_equal(like(allp[:61]), want[:61], equal_nan=True)
    assert ctx.info("serving") == 1 and ctx.info("serve_grid") == 32 and ctx.info("serve_fallbacks") == 0


def test_emcee_call_conventions_on_the_gpu_path(mbb, g_lnl, monkeypatch):
    """The three ways an emcee-style sampler can call the likelihood -- row by row
    (emcee's plain map, mbb_fit.py:80-81 with threads=1), through a pool's map with a
    wrapper object around the callable (emcee 2.x `pool=`, what threads>1 selects), and
    vectorised (emcee 3 `vectorize=True`) -- give the same chain, value for value, with the
    fused kernel evaluating a whole half-step per launch in the last two.  Then
    mbb_fitter(sampler="emcee") against stand-ins with emcee 2's and emcee 3's constructor
    signatures (emcee itself is not installed: SURVEY.md 8c)."""
    like = _cfg2_like(mbb, g_lnl)
    rng = np.random.RandomState(5)
    p0 = np.array([12.0, 1.8, 600.0, 3.0, 40.0]) * (1.0 + 0.02 * rng.normal(size=(20, 5)))

    class Wrapper(object):                      # emcee's _function_wrapper
        def __init__(self, f):
            self.f = f

        def __call__(self, x):
            return self.f(x)
    chains = []
    for kw, fn in ((dict(vectorize=False), like), (dict(vectorize=False, pool=like), Wrapper(like)),
                   (dict(vectorize=True), like)):
        s = mbb.EnsembleSampler(20, 5, fn, seed=3, **kw)
        pos, lnp, _ = s.run_mcmc(p0, 12)
        chains.append((s.chain.copy(), s.lnprobability.copy()))
        assert np.array_equal(lnp, like(pos))
    for c, l in chains[1:]:
        assert np.array_equal(c, chains[0][0]) and np.array_equal(l, chains[0][1])
    # the pool adaptor evaluates all rows of a half-step in one launch
    calls = []
    orig = like.context.lnlike_batch
    monkeypatch.setattr(like.context, "lnlike_batch", lambda p, **k: calls.append(len(p) // 5 if p.ndim == 1 else p.shape[0]) or orig(p, **k))
    # (... whichever way the launch is made: the general path above, or the boundary call of __call__'s fast path)
    raw = like.context._lnlike_call_raw
    monkeypatch.setattr(like.context, "_lnlike_call_raw", lambda h, n: calls.append(n) or raw(h, n))
    like._fast = None
    mbb.EnsembleSampler(20, 5, Wrapper(like), seed=3, vectorize=False, pool=like).run_mcmc(p0, 2)
    assert calls == [20, 10, 10, 10, 10]
    # ... but only when the function IS the likelihood: a posterior of the caller's own (a lambda that
    # adds a prior; emcee's wrapper carrying args=) is applied row by row, and its value is what counts
    del calls[:]
    extra_prior = lambda x: like(x) - 0.5 * ((x[0] - 12.0) / 0.5) ** 2
    rows = [p0[i] for i in range(6)]
    got = like.map(extra_prior, rows)
    assert calls == [1] * 6
    assert got == [extra_prior(r) for r in rows] and got != list(like(np.asarray(rows)))
    del calls[:]
    assert like.map(like.__call__, rows) == list(like(np.asarray(rows))) and calls[0] == 6      # the bound method: one launch
    monkeypatch.undo()

    # mbb_fitter(sampler="emcee") with emcee-2 and emcee-3 shaped stand-ins
    import sys, types
    for version in (2, 3):
        fake = types.ModuleType("emcee")

        class FakeSampler(mbb.EnsembleSampler):
            def __init__(self, nwalkers, dim, lnpostfn, a=2.0, args=(), kwargs=None, threads=1, pool=None, **extra):
                if version == 2 and extra:
                    raise TypeError("__init__() got an unexpected keyword argument %r" % list(extra)[0])
                self.how = "vectorize" if extra.get("vectorize") else ("pool" if pool is not None else "rows")
                mbb.EnsembleSampler.__init__(self, nwalkers, dim, lnpostfn, a=a, pool=pool,
                                             vectorize=bool(extra.get("vectorize")), seed=17)
        fake.EnsembleSampler = FakeSampler
        monkeypatch.setitem(sys.modules, "emcee", fake)
        fit = mbb.mbb_fitter(nwalkers=20, response=True, sampler="emcee", seed=4)
        assert fit.sampler.how == ("pool" if version == 2 else "vectorize")
        assert fit.sampler.pool is (fit.like if version == 2 else None)
        bands = [str(b) for b in g_lnl["cfg2/bands"]]
        fit.set_data(bands, g_lnl["cfg2/thick_walpha/flux"], g_lnl["cfg2/thick_walpha/unc"])
        p = fit.generate_initial_values([12.0, 1.8, 600.0, 3.0, 40.0], [1.0, 0.1, 30.0, 0.2, 2.0])
        fit.run(5, 10, p, verbose=False)
        assert fit.sampler.chain.shape == (20, 10, 5) and np.all(np.isfinite(fit.sampler.lnprobability))
        if version == 2:
            ch2 = fit.sampler.chain.copy()
        else:
            assert np.array_equal(ch2, fit.sampler.chain)      # pool-mapped == vectorised


def test_cli_config1(mbb, tmp_path):
    """Config 1 end to end through the command-line harness: 5 plain wavelengths,
    optically thin, 50 walkers."""
    from mbb_emcee_amd import run_mbb_emcee
    wave = np.array([100.0, 160.0, 250.0, 350.0, 500.0])
    truth = mbb.modified_blackbody(12.0, 1.8, None, 3.0, 40.0, opthin=True)
    flux = truth(wave)
    pf = tmp_path / "phot.txt"
    with open(pf, "w") as fh:
        fh.write("# wave flux unc\n")
        for w, f in zip(wave, flux):
            fh.write("%g %.8g %.8g\n" % (w, f, 0.1 * f + 1.0))
    out = tmp_path / "fit.npz"
    rc = run_mbb_emcee.main([str(pf), str(out), "--opthin", "-n", "50", "-b", "100", "-N", "200",
                             "--initT", "12", "--initBeta", "1.8", "--initAlpha", "3", "--fixLambda0",
                             "--initLambda0", "600",
                             "--seed", "5", "--get_peaklambda"])
    assert rc == 0
    d = np.load(out)
    assert d["chain"].shape == (50, 200, 5) and d["peaklambda"].shape == (50, 200)
    med = np.median(d["chain"].reshape(-1, 5), axis=0)
    # 5 points leave T and beta strongly degenerate: check the fit, not the marginals
    assert 6.0 < med[0] < 20.0 and abs(med[4] - 40.0) < 8.0
    assert d["lnprobability"].max() > -1.0
    assert np.all(d["chain"][:, :, 2] == d["chain"][0, 0, 2])


# ------------------------------------ full BASELINE sizes: size-independent properties
def test_full_size_cfg3_and_cfg5_properties(mbb, g_lnl, oracle):
    """At the full sizes of configs 3 and 5 (2000 walkers; 1000 sources x 250
    walkers = 250 000 evaluations in one launch) the oracle is too slow to check
    every row, so: (i) a random sample of rows against the oracle, (ii) properties
    that hold for any size -- idempotence, row-permutation equivariance (bitwise),
    band fluxes linear in fnorm, lnL = -chi^2/2 rebuilt from the returned band
    fluxes, the soft-wall penalty added exactly once."""
    from tools.bench_cfg5 import setup
    rng = np.random.RandomState(77)
    # ---- cfg3: 2000 walkers, single source -------------------------------------
    like = _cfg2_like(mbb, g_lnl)
    w = np.column_stack([rng.normal(12, 1, 2000), rng.normal(1.8, 0.2, 2000), rng.normal(600, 50, 2000),
                         rng.normal(3, 0.3, 2000), rng.normal(40, 3, 2000)])
    l0 = like(w)
    assert np.array_equal(l0, like(w))                               # idempotent
    perm = rng.permutation(2000)
    assert np.array_equal(like(w[perm]), l0[perm])                   # equivariant, bitwise
    mf = like.model_flux(w)
    w3 = w.copy(); w3[:, 4] *= 3.0
    rec_allclose(like.model_flux(w3), 3.0 * mf, rtol=3e-15, kind="band flux")   # linear in fnorm
    chi = -0.5 * np.sum((like.data_flux - mf) ** 2 / like.data_flux_unc ** 2, axis=1)
    rec_allclose(l0, chi, rtol=1e-12, atol=1e-12, kind="lnL identity")   # no wall is hit by these rows
    wb = w.copy(); wb[:, 1] = 20.5                                   # beta above its soft wall
    pen = -0.5 * (20.5 - 20.0) ** 2 / (0.02 * (20.0 - 0.1)) ** 2
    mfb = like.model_flux(wb)
    chib = -0.5 * np.sum((like.data_flux - mfb) ** 2 / like.data_flux_unc ** 2, axis=1)
    rec_allclose(like(wb), chib + pen, rtol=1e-12, kind="lnL identity")
    # ---- cfg5: 1000 sources x 250 walkers ----------------------------------------
    like5, truths, p0 = setup(1000, 250)
    big = like5(p0)
    assert big.shape == (1000, 250) and np.all(np.isfinite(big))
    assert np.array_equal(big, like5(p0))
    sp = rng.permutation(250)
    assert np.array_equal(like5(p0[:, sp, :]), big[:, sp])           # within-source permutation
    bands = [(r.wavelength, r._sedmult, r._normfac) for r in like5._responses]
    for g in rng.choice(1000, 6, replace=False):                     # sampled sources vs the oracle
        orc = oracle.OracleLikelihood(like5._flux_multi[g], 1.0 / np.sqrt(like5._ivar_multi[g]),
                                      bands=bands, has_uplim=[int(b) for b in like5.has_uplims],
                                      uplim=like5.uplims)
        lnl_close(big[g], orc(p0[g], nthreads=8))
    # the truth row of every source has lnL = 0 up to rounding (data were made from it)
    lt = like5(truths[:, None, :])
    assert np.all(np.abs(lt) < 1e-18)


# --------------------------------------------------------------- randomised configurations
@pytest.mark.parametrize("path", PATHS)
@pytest.mark.parametrize("seed", range(24))
def test_random_configurations_vs_oracle(mbb, oracle, seed, path):
    """Random band subsets (wheel bands, specials, repeats, a single band), random
    model variant, random limits / priors / lambda_peak terms, diagonal or full
    covariance, random batch size and launch geometry -- always the oracle's answer."""
    rng = np.random.RandomState(1000 + seed)
    pool = ["PACS_70um", "PACS_100um", "PACS_160um", "SPIRE_250um", "SPIRE_350um", "SPIRE_500um",
            "SCUBA2_450um", "SCUBA2_850um", "Bolocam_1.1mm", "MAMBO2_1.2mm", "GISMO_2mm",
            "LABOCA_870um", "MIPS_160um", "X_box_850um_60", "ALMA_alma_343", "ALMA_alma_230",
            "S_gauss_345_10", "D_dsb_300_20_6", "Y_delta_1300um"]
    nb = int(rng.choice([1, 2, 3, 5, 8, 13]))
    names = [str(x) for x in rng.choice(pool, nb, replace=(nb > 8))]
    opthin, noalpha = bool(rng.randint(2)), bool(rng.randint(2))
    wavenorm = float(rng.choice([500.0, 350.0, 850.0]))
    like = mbb.likelihood(opthin=opthin, noalpha=noalpha, response=True, wavenorm=wavenorm)
    like.set_phot(names, np.ones(nb), np.ones(nb))
    truth = np.array([rng.uniform(8, 40), rng.uniform(1.0, 2.6), rng.uniform(80, 900),
                      rng.uniform(1.5, 5.0), rng.uniform(5, 90)])
    flux = like.model_flux(truth)[0]
    unc = 0.08 * flux + 0.3
    like.set_phot(names, flux, unc)
    cov = None
    if nb > 1 and rng.rand() < 0.5:
        A = rng.normal(0, 1, (nb, nb))
        cov = np.diag(unc ** 2) + 0.02 * np.median(unc) ** 2 * A.dot(A.T)
        like.set_cov(cov)
    has_g, gm, gs = [0] * 6, [0.0] * 6, [1.0] * 6
    for i in range(6):
        if rng.rand() < 0.35:
            centre = truth[i] if i < 5 else 200.0
            has_g[i], gm[i], gs[i] = 1, centre * rng.uniform(0.9, 1.1), abs(centre) * rng.uniform(0.05, 0.3)
            like.set_gaussian_prior(i, gm[i], gs[i])
    if rng.rand() < 0.4:
        like.set_uplim("T", truth[0] * 1.02)
    if rng.rand() < 0.3:
        like.set_uplim("lambda_peak", 180.0)
    if rng.rand() < 0.3:
        like.set_lowlim("beta", truth[1] * 0.97)
    n = int(rng.choice([1, 3, 64, 65, 257, 900]))
    pars = truth * (1.0 + 0.06 * rng.normal(size=(n, 5)))
    bands = []
    for r in like._responses:
        bands.append((r.wavelength, np.ones(1), 1.0) if r.isdelta else (r.wavelength, r._sedmult, r._normfac))
    orc = oracle.OracleLikelihood(flux, unc, bands=bands, cov=cov, opthin=opthin, noalpha=noalpha,
                                  wavenorm=wavenorm, lowlim=like.lowlims,
                                  has_uplim=[int(b) for b in like.has_uplims], uplim=like.uplims,
                                  has_gprior=has_g, gprior_mean=gm, gprior_sigma=gs)
    ref = orc(pars, nthreads=4)
    like.context.set_option("walkers_per_group", int(rng.choice([0, 1, 2, 5, 16])))
    like.context.set_option("block_threads", int(rng.choice([0, 64, 256, 640, 1024])))
    got = boundary_eval(like, pars, path)
    like.context.set_option("walkers_per_group", 0); like.context.set_option("block_threads", 0)
    lnl_close(got, ref, kind=("lnL (covariance%s)" % ("" if path == "launched" else ", " + path)) if cov is not None else
              ("lnL" if path == "launched" else "lnL (%s)" % path))
    # (and the ways agree bit for bit: another one now, default geometry)
    assert np.array_equal(got, boundary_eval(like, pars, "served" if path == "launched" else "launched"), equal_nan=True)


@pytest.mark.parametrize("opthin,noalpha", [(False, False), (False, True), (True, False), (True, True)])
def test_wide_parameter_ranges_vs_oracle(mbb, oracle, opthin, noalpha):
    """The whole prior volume a fit can wander through, not just the neighbourhood of a
    good fit: T 3-200 K, beta 0-4.5, lambda0 5-3000 um, alpha 0.1-10, fnorm 0.01-1000 mJy,
    with a lambda_peak prior so that the second root find runs too.  Band fluxes and
    lnL against the oracle; the merge-point solve must converge on every row."""
    rng = np.random.RandomState(4242)
    n = 600
    pars = np.column_stack([np.exp(rng.uniform(np.log(3), np.log(200), n)), rng.uniform(0.0, 4.5, n),
                            np.exp(rng.uniform(np.log(5), np.log(3000), n)),
                            np.exp(rng.uniform(np.log(0.1), np.log(10), n)),
                            np.exp(rng.uniform(np.log(0.01), np.log(1000), n))])
    pars[:8, 1] = 0.0                                        # beta = 0 exactly
    names = ["PACS_100um", "SPIRE_250um", "SPIRE_500um", "SCUBA2_850um", "GISMO_2mm"]
    like = mbb.likelihood(opthin=opthin, noalpha=noalpha, response=True)
    flux = np.array([30.0, 60.0, 25.0, 6.0, 0.5])
    unc = 0.1 * flux + 0.2
    like.set_phot(names, flux, unc)
    like.set_gaussian_prior("lambda_peak", 120.0, 40.0)
    has_g, gm, gs = [0] * 6, [0.0] * 6, [1.0] * 6
    has_g[5], gm[5], gs[5] = 1, 120.0, 40.0
    bands = [(r.wavelength, r._sedmult, r._normfac) for r in like._responses]
    orc = oracle.OracleLikelihood(flux, unc, bands=bands, opthin=opthin, noalpha=noalpha, wavenorm=500.0,
                                  lowlim=like.lowlims, has_uplim=[int(b) for b in like.has_uplims],
                                  uplim=like.uplims, has_gprior=has_g, gprior_mean=gm, gprior_sigma=gs)
    ref, rflux = orc(pars, nthreads=4, return_flux=True)
    got = like(pars)
    gflux = like.model_flux(pars)
    fin = np.isfinite(ref)
    assert fin.sum() > 0.9 * n
    assert np.array_equal(np.isfinite(got), fin)
    ok = fin[:, None] & (rflux > 1e-280)
    ferr = np.max(np.abs(gflux[ok] / rflux[ok] - 1.0))
    parity_record("band flux (rel)", ferr, FLUX_RTOL)
    assert ferr < FLUX_RTOL
    lnl_close(got, ref)                     # (a row whose root find failed would have raised)


@pytest.mark.parametrize("opthin,noalpha", [(False, False), (False, True), (True, False), (True, True)])
def test_table_ends_and_first_rows_vs_oracle(mbb, oracle, opthin, noalpha):
    """Where the sample loop's tables begin and end (round 6: b(x) on [0, 48], beyond it the far branch b = x e^-x; C(y) on
    [0, 37], y held there; row 0 of C built as y times an interpolant so that it keeps its relative accuracy down to 0):
    temperatures of 1 to 8 K put whole passbands beyond x = 48 (fluxes down to 1e-48: never NaN), thousands of kelvin put them into the first rows (x < 1/8); lambda0 of 1 um makes y tiny everywhere
    (row 0 of C: the flux is proportional to y), 4500 um makes it huge (C = 1).  Every band flux against the oracle at the
    stated 1e-12, lnL likewise; -inf rows must agree."""
    Ts = np.array([1.0, 1.2, 1.5, 2.0, 2.5, 3.0, 3.2, 3.6, 4.0, 5.0, 8.0, 150.0, 300.0, 1000.0, 5000.0])
    l0s = np.array([1.0, 30.0, 600.0, 3000.0, 4500.0])
    betas = np.array([0.5, 1.8, 3.5])
    T, L0, B = (x.ravel() for x in np.meshgrid(Ts, l0s, betas, indexing="ij"))
    n = T.size
    pars = np.column_stack([T, B, L0, np.full(n, 2.5), np.full(n, 40.0)])
    names = ["PACS_70um", "PACS_160um", "SPIRE_250um", "SPIRE_500um", "SCUBA2_850um", "Bolocam_1.1mm"]
    like = mbb.likelihood(opthin=opthin, noalpha=noalpha, response=True)
    flux = np.array([20.0, 60.0, 50.0, 25.0, 6.0, 2.0])
    unc = 0.1 * flux + 0.2
    like.set_phot(names, flux, unc)
    like.set_uplim("lambda0", 5000.0)                                       # (3 x max wavelength would cut the grid short)
    bands = [(r.wavelength, r._sedmult, r._normfac) for r in like._responses]
    orc = oracle.OracleLikelihood(flux, unc, bands=bands, opthin=opthin, noalpha=noalpha, wavenorm=500.0,
                                  lowlim=like.lowlims, has_uplim=[int(b) for b in like.has_uplims], uplim=like.uplims)
    ref, rflux = orc(pars, nthreads=4, return_flux=True)
    got = like(pars)
    gflux = like.model_flux(pars)
    assert not np.isnan(got).any() and not np.isnan(gflux).any()
    assert np.array_equal(np.isneginf(got), np.isneginf(ref))
    big = rflux > 1e-280
    # (the far branch is the models' without alpha: with the Wien-side power law x > 48 lies beyond the merge point;
    # their cold rows' PACS fluxes go down to 1e-48 of the normalisation here)
    assert big.sum() > 0.9 * rflux.size and (rflux.min() < 1e-30) == bool(noalpha)
    ferr = np.max(np.abs(gflux[big] / rflux[big] - 1.0))
    parity_record("band flux (rel)", ferr, FLUX_RTOL)
    assert ferr < FLUX_RTOL
    assert np.all(gflux[~big] < 1e-270)                                     # what the reference makes 0 or denormal is no larger here
    lnl_close(got, ref)


def test_sharded_device_sampler_equals_unsharded(mbb, g_lnl):
    """The sharded form of the device sampler (each rank moves a contiguous block of
    the half-ensemble, in-place all-gather of the state rows) is exercised on one
    GPU by running the G shards one after another ('virtual_ranks'): chain, lnprob,
    acceptance counts and final state are bitwise those of the unsharded run."""
    like = _cfg2_like(mbb, g_lnl)
    p0 = g_lnl["cfg2/thick_walpha/pars"][:64].copy()
    ref = mbb.DeviceEnsembleSampler(64, 5, like, seed=77)
    rp, rl, _ = ref.run_mcmc(p0, 12)
    rp2, _, _ = ref.run_mcmc(None, 5)
    ctx = like.context
    for g in (2, 4, 8):
        ctx.set_option("virtual_ranks", g)
        s = mbb.DeviceEnsembleSampler(64, 5, like, seed=77)
        sp, slp, _ = s.run_mcmc(p0, 12)
        assert np.array_equal(s.chain, ref.chain[:, :12]) and np.array_equal(s.lnprobability, ref.lnprobability[:, :12])
        assert np.array_equal(sp, rp) and np.array_equal(slp, rl)
        sp2, _, _ = s.run_mcmc(None, 5)
        assert np.array_equal(sp2, rp2) and np.array_equal(s.naccepted, ref.naccepted)
        assert np.array_equal(s.chain, ref.chain)
    ctx.set_option("virtual_ranks", 3)                     # 32 is not a multiple of 3
    with pytest.raises(Exception):
        mbb.DeviceEnsembleSampler(64, 5, like, seed=1).run_mcmc(p0, 1)
    ctx.set_option("virtual_ranks", 0)
    # a real 1-rank communicator takes the ordinary path
    ctx.comm_init(1, 0, ctx.comm_unique_id())
    s = mbb.DeviceEnsembleSampler(64, 5, like, seed=77)
    s.run_mcmc(p0, 12)
    assert np.array_equal(s.chain, ref.chain[:, :12])
    ctx.comm_destroy()


FORM_RESIDENT = 9      # what "last_kernel_form" says for the default resident run of a large ensemble (k_flowa)


def _sampler_forms(ctx):
    """(name, options) of the single-GPU device sampler's forms; the first is the reference."""
    return [("plain", {"lookahead_sampler": 0, "flow_sampler": 0}),
            ("one launch, quadrature ahead (form 7)", {"lookahead_sampler": 1, "flow_sampler": 1, "merged_flow_sampler": 1,
                                                       "resident_sampler": 1}),
            ("one launch, resident, constructor a half-step ahead (form 9), one walker of each half per workgroup",
             {"lookahead_sampler": 1, "flow_sampler": 1, "resident_sampler": 2}),
            ("one launch, resident, constructor a half-step ahead (form 9), three walkers of each half per workgroup",
             {"lookahead_sampler": 1, "flow_sampler": 1, "resident_sampler": 2, "resident_walkers": 3}),
            ("one launch, resident, constructor a half-step ahead (form 9), seven walkers of each half per workgroup",
             {"lookahead_sampler": 1, "flow_sampler": 1, "resident_sampler": 2, "resident_walkers": 7})]


def test_lookahead_sampler_forms_equal_the_plain_launch_train(mbb, g_lnl):
    """The one-launch forms of the device sampler -- form 7 (k_flowm: one workgroup per pair of walkers and candidate,
    quadrature and constructor ahead of the decisions they depend on) and form 9 (k_flowa: a workgroup owns several
    walkers of each half, the constructor a half-step ahead for both outcomes of each partner's pending move) -- make the same draws and do the same arithmetic per proposal as the
    plain train of one launch per half-step, so chain, lnprob, final state and acceptance counts must be bitwise its:
    every model variant, stored and unstored runs in sequence."""
    bands = [str(b) for b in g_lnl["cfg2/bands"]]
    for name, opthin, noalpha in VARIANTS:
        k = "cfg2/" + name
        res = []
        for form, opts in _sampler_forms(None):
            like = mbb.likelihood(response=True, opthin=opthin, noalpha=noalpha)
            like.set_phot(bands, g_lnl[k + "/flux"], g_lnl[k + "/unc"])
            for o, v in opts.items():
                like.context.set_option(o, v)
            p0 = np.array([12.0, 1.8, 600.0, 3.0, 40.0]) * (1.0 + 0.02 * np.random.RandomState(4).normal(size=(250, 5)))
            s = mbb.DeviceEnsembleSampler(250, 5, like, seed=77)
            a = s.run_mcmc(p0, 40)
            b = s.run_mcmc(None, 25, storechain=False)
            s.advance_async(30); like.context.sync()
            c = s.run_mcmc(None, 7)
            res.append((a[0], a[1], b[0], b[1], c[0], c[1], s.chain.copy(), s.lnprobability.copy(), s.naccepted.copy()))
        for form, r in zip(_sampler_forms(None)[1:], res[1:]):
            for x, y in zip(res[0], r):
                assert np.array_equal(x, y), (name, form[0])
        assert res[0][6].shape == (250, 47, 5) and 0.1 < res[0][8].mean() / 102 < 0.9


@pytest.mark.parametrize("nw,W", [(514, 0), (1000, 0), (2000, 0), (2000, 8), (4096, 0), (300, 5), (36, 8), (26, 7), (300, 2)])
def test_resident_sampler_form_for_large_ensembles_equals_the_launch_train(mbb, g_lnl, nw, W):
    """Sampler form 9 (k_flowa): ensembles beyond one pair of walkers per CU -- cfg3's 2000 walkers on one GPU
    among them -- run as ONE launch per run with several walkers of each half per workgroup, the rows handed over
    through check words instead of a launch boundary, the constructor a half-step ahead.  The chain, the final
    state and the counts are bitwise the launch train's, whatever the number of walkers per workgroup (the
    host's choice, a forced one, a last workgroup that owns fewer; up to the eight of 4096 walkers: round 4's form 8,
    which took 3073-4096, is gone)."""
    ahead = 1
    like = _cfg2_like(mbb, g_lnl)
    ctx = like.context
    p0 = np.array([12.0, 1.8, 600.0, 3.0, 40.0]) * (1.0 + 0.02 * np.random.RandomState(nw).normal(size=(nw, 5)))
    out = []
    for resident in (0, 1):
        ctx.set_option("lookahead_sampler", resident); ctx.set_option("resident_sampler", 2 if W else 1); ctx.set_option("resident_walkers", W)
        s = mbb.DeviceEnsembleSampler(nw, 5, like, seed=nw + 1)
        a = s.run_mcmc(p0, 9)
        if resident:
            assert ctx.info("last_kernel_form") == (9 if ahead else 8) and ctx.info("flow_fallbacks") == 0
            assert ctx.info("last_wpb") == (W or -(-(nw // 2) // ctx.info("cu_count")))
        b = s.run_mcmc(None, 6, storechain=False)
        s.advance_async(5); ctx.sync()
        c = s.run_mcmc(None, 4)
        out.append((a[0], a[1], b[0], b[1], c[0], c[1], s.chain.copy(), s.lnprobability.copy(), s.naccepted.copy()))
        assert ctx.info("last_kernel_form") == ((9 if ahead else 8) if resident else 1)
    for x, y in zip(*out):
        assert np.array_equal(x, y), (nw, W, ahead)
    assert out[0][6].shape == (nw, 13, 5) and 0.1 < out[0][8].mean() / 24 < 0.9
    lnl_close(out[1][5], like(out[1][4]))


def test_lookahead_sampler_forms_with_priors_limits_and_long_runs(mbb, g_lnl):
    """The same equality where the working-ahead path does more than the constructor: upper
    limits and Gaussian priors on parameters and on the peak wavelength (its root solve runs in
    the look-ahead rows), walkers driven against a lower limit (rejected proposals, -inf), a
    covariance matrix, an ensemble of 10, one too large for either look-ahead form (fall-back),
    and a run longer than the 4096 steps one launch covers."""
    bands = [str(b) for b in g_lnl["cfg2/bands"]]
    flux, unc = g_lnl["cfg2/thick_walpha/flux"], g_lnl["cfg2/thick_walpha/unc"]

    def run(configure, nw, nsteps, p_center, spread, seed):
        out = []
        for form, opts in _sampler_forms(None):
            like = mbb.likelihood(response=True)
            like.set_phot(bands, flux, unc)
            configure(like)
            for o, v in opts.items():
                like.context.set_option(o, v)
            p0 = np.array(p_center) * (1.0 + spread * np.random.RandomState(seed).normal(size=(nw, 5)))
            s = mbb.DeviceEnsembleSampler(nw, 5, like, seed=seed)
            pos, lnp, _ = s.run_mcmc(p0, nsteps)
            out.append((pos, lnp, s.chain.copy(), s.lnprobability.copy(), s.naccepted.copy()))
        for form, r in zip(_sampler_forms(None)[1:], out[1:]):
            for x, y in zip(out[0], r):
                assert np.array_equal(x, y), form[0]
        return out[0]

    def priors(like):
        like.set_uplim("T", 14.0); like.set_uplim("beta", 2.2); like.set_uplim("peaklam", 260.0)
        like.set_gaussian_prior("beta", 1.9, 0.2); like.set_gaussian_prior("peaklam", 240.0, 15.0)
        like.set_gaussian_prior("alpha", 3.2, 0.5)
    r = run(priors, 60, 60, [12.0, 1.8, 600.0, 3.0, 40.0], 0.03, 21)
    assert np.isfinite(r[1]).all()

    def lowlim(like):
        like.set_lowlim("T", 11.9); like.set_lowlim("beta", 1.75)
    r = run(lowlim, 40, 50, [12.0, 1.8, 600.0, 3.0, 40.0], 0.02, 22)
    assert 0.0 < r[4].mean() / 50 < 0.6                     # many proposals fall below the limits

    def cov(like):
        nb = len(bands)
        c = np.diag(np.asarray(unc) ** 2) + 0.2 * np.outer(unc, unc)
        like.set_cov(c)
    run(cov, 30, 40, [12.0, 1.8, 600.0, 3.0, 40.0], 0.02, 23)
    run(lambda like: None, 10, 80, [12.0, 1.8, 600.0, 3.0, 40.0], 0.02, 24)
    run(lambda like: None, 520, 5, [12.0, 1.8, 600.0, 3.0, 40.0], 0.02, 25)      # 260 movers: too many, plain train
    r = run(lambda like: None, 12, 4200, [12.0, 1.8, 600.0, 3.0, 40.0], 0.02, 26)
    assert r[2].shape == (12, 4200, 5)


def test_lookahead_sampler_forms_other_band_sets_and_geometries(mbb, g_lnl):
    """The look-ahead forms against the plain launch train where the band layout and the launch
    geometry are not the bench's: delta-function photometry (five bands in one shared chunk,
    fewer units than waves), the 12-band set with its covariance matrix, passband tables read
    through L2 instead of LDS, narrower workgroups, units of one and of six chunks."""
    def compare(make, nw, nsteps, p_center, seed, options=()):
        out = []
        for form, opts in _sampler_forms(None):
            like = make()
            for o, v in tuple(opts.items()) + tuple(options):
                like.context.set_option(o, v)
            if any(o in ("seg_chunks", "pack_tails") for o, _ in options):
                like._dirty = True                                   # the layout options act at the next set_bands
            p0 = np.array(p_center) * (1.0 + 0.02 * np.random.RandomState(seed).normal(size=(nw, 5)))
            s = mbb.DeviceEnsembleSampler(nw, 5, like, seed=seed)
            pos, lnp, _ = s.run_mcmc(p0, nsteps)
            pos2, lnp2, _ = s.run_mcmc(None, 3)
            out.append((pos, lnp, pos2, lnp2, s.chain.copy(), s.lnprobability.copy(), s.naccepted.copy()))
            assert like.context.info("last_kernel_form") == (1 if form == "plain" else int(form[form.index("(form ") + 6]))
        for form, r in zip(_sampler_forms(None)[1:], out[1:]):
            for x, y in zip(out[0], r):
                assert np.array_equal(x, y), (form[0], options)
        assert np.isfinite(out[0][1]).all()

    for name, opthin, noalpha in VARIANTS:
        compare(lambda: _like_cfg1(mbb, g_lnl, name, opthin, noalpha)[0], 40, 25, [14.0, 1.8, 500.0, 3.0, 30.0], 31)

    def cfg4():
        like = mbb.likelihood(response=True)
        k = "cfg4/thick_walpha"
        like.set_phot([str(b) for b in g_lnl["cfg4/bands"]], g_lnl[k + "/flux"], g_lnl[k + "/unc"])
        like.set_cov(g_lnl[k + "/cov"])
        return like
    compare(cfg4, 60, 20, [12.0, 1.8, 600.0, 3.0, 40.0], 32)
    for options in ((("stage_tables", 0),), (("block_threads", 512),), (("seg_chunks", 1),), (("seg_chunks", 6), ("pack_tails", 0))):
        compare(lambda: _cfg2_like(mbb, g_lnl), 50, 20, [12.0, 1.8, 600.0, 3.0, 40.0], 33, options)


# (MBB_TEST_RANDOM_SAMPLER_SEEDS=n: more of them, for a one-off sweep)
@pytest.mark.parametrize("seed", range(int(os.environ.get("MBB_TEST_RANDOM_SAMPLER_SEEDS", "10"))))
def test_random_sampler_configurations_all_forms_equal(mbb, seed):
    """The device sampler's forms on random configurations -- band subsets from single delta bands to
    13 bands (more units than quadrature waves, bands sharing chunks, more than 8 bands), random model
    variant, priors, limits, peak-wavelength terms, diagonal or full covariance, random ensemble size
    and run lengths, random workgroup width and unit size: every look-ahead form that the size allows
    gives the chain of the plain launch train bit for bit, stored and unstored runs in sequence."""
    rng = np.random.RandomState(7000 + seed)
    pool = ["PACS_70um", "PACS_100um", "PACS_160um", "SPIRE_250um", "SPIRE_350um", "SPIRE_500um",
            "SCUBA2_450um", "SCUBA2_850um", "Bolocam_1.1mm", "MAMBO2_1.2mm", "GISMO_2mm",
            "LABOCA_870um", "MIPS_160um", "X_box_850um_60", "ALMA_alma_343", "ALMA_alma_230",
            "S_gauss_345_10", "D_dsb_300_20_6", "Y_delta_1300um"]
    nb = int(rng.choice([1, 2, 3, 5, 8, 13]))
    names = [str(x) for x in rng.choice(pool, nb, replace=(nb > 8))]
    opthin, noalpha = bool(rng.randint(2)), bool(rng.randint(2))
    truth = np.array([rng.uniform(10, 30), rng.uniform(1.2, 2.4), rng.uniform(100, 700),
                      rng.uniform(2.0, 4.5), rng.uniform(10, 80)])
    nw = int(rng.choice([10, 12, 34, 128, 250, 256, 300]))
    nsteps = int(rng.choice([3, 11, 40]))
    opts = {"block_threads": int(rng.choice([0, 0, 512, 768])), "seg_chunks": int(rng.choice([0, 0, 2, 6]))}
    cov_seed, prior_seed = rng.randint(1 << 30), rng.randint(1 << 30)
    p0 = truth * (1.0 + 0.02 * rng.normal(size=(nw, 5)))
    res, forms = [], []
    for form, fopts in _sampler_forms(None):
        like = mbb.likelihood(opthin=opthin, noalpha=noalpha, response=True)
        for o, v in opts.items():
            if v:
                like.context.set_option(o, v)
        like.set_phot(names, np.ones(nb), np.ones(nb))
        flux = like.model_flux(truth)[0]
        unc = 0.08 * flux + 0.3
        like.set_phot(names, flux, unc)
        r2 = np.random.RandomState(cov_seed)
        if nb > 1 and r2.rand() < 0.5:
            A = r2.normal(0, 1, (nb, nb))
            like.set_cov(np.diag(unc ** 2) + 0.02 * np.median(unc) ** 2 * A.dot(A.T))
        r3 = np.random.RandomState(prior_seed)
        for i in range(6):
            if r3.rand() < 0.3:
                centre = truth[i] if i < 5 else 200.0
                like.set_gaussian_prior(i, centre * r3.uniform(0.95, 1.05), abs(centre) * r3.uniform(0.05, 0.3))
        if r3.rand() < 0.4:
            like.set_uplim("T", truth[0] * 1.02)
        if r3.rand() < 0.3:
            like.set_uplim("lambda_peak", 220.0)
        if r3.rand() < 0.3:
            like.set_lowlim("beta", truth[1] * 0.97)
        for o, v in fopts.items():
            like.context.set_option(o, v)
        s = mbb.DeviceEnsembleSampler(nw, 5, like, seed=seed)
        a = s.run_mcmc(p0, nsteps)
        b = s.run_mcmc(None, 7, storechain=False)
        c = s.run_mcmc(None, 2)
        res.append((a[0], a[1], b[0], b[1], c[0], c[1], s.chain.copy(), s.lnprobability.copy(), s.naccepted.copy()))
        forms.append(like.context.info("last_kernel_form"))
    for form, r in zip(_sampler_forms(None)[1:], res[1:]):
        for x, y in zip(res[0], r):
            assert np.array_equal(x, y, equal_nan=True), (seed, form[0], names, nw, opts)
    # (the forced forms ran as asked; the default one-launch entry: form 7 while every pair and candidate has a CU
    # of its own, the resident form with the constructor ahead beyond)
    assert forms[0] == 1 and forms[2:] == [9, 9, 9], forms
    assert forms[1] == (7 if nw <= 256 else 9), (nw, forms)


def test_stored_chain_comes_back_in_emcees_layout_small_and_large(mbb, g_lnl):
    """A stored run's chain is re-ordered on the device (k_chain_reorder) and comes back either straight
    into the caller's arrays (under 8 MB) or through a pinned landing buffer (above): chain[walker, step]
    must be the walker's position after that step and lnprobability[walker, step] its lnprob -- the last
    step equals the returned state, every entry's lnprob is the likelihood of its position, and a step
    where a walker did not move repeats the entry before (250 walkers x 40 steps: 0.5 MB; x 800: 9.6 MB;
    the launch train and the one-launch form)."""
    like = _cfg2_like(mbb, g_lnl)
    ctx = like.context
    p0 = np.array([12.0, 1.8, 600.0, 3.0, 40.0]) * (1.0 + 0.02 * np.random.RandomState(3).normal(size=(250, 5)))
    rng = np.random.RandomState(4)
    for look, nsteps in ((1, 40), (1, 800), (0, 40), (0, 800)):
        ctx.set_option("lookahead_sampler", look)
        s = mbb.DeviceEnsembleSampler(250, 5, like, seed=21)
        pos, lnp, _ = s.run_mcmc(p0, nsteps)
        ch, lp = s.chain, s.lnprobability
        assert ch.shape == (250, nsteps, 5) and lp.shape == (250, nsteps)
        assert np.array_equal(ch[:, -1, :], pos) and np.array_equal(lp[:, -1], lnp)
        w, t = rng.randint(250, size=400), rng.randint(nsteps, size=400)
        assert np.array_equal(like(ch[w, t]), lp[w, t])
        moved = np.any(ch[:, 1:, :] != ch[:, :-1, :], axis=2)
        assert np.array_equal(lp[:, 1:][~moved], lp[:, :-1][~moved])
        assert abs(moved.mean() - s.acceptance_fraction.mean()) < 0.02 and 0.2 < moved.mean() < 0.7
        if nsteps == 800 and look == 1:
            big = (ch.copy(), lp.copy())
        elif nsteps == 800:
            assert np.array_equal(big[0], ch) and np.array_equal(big[1], lp)      # same chain whichever form stored it
    ctx.set_option("lookahead_sampler", 1)


def test_advance_timed_is_advance_async_with_a_clock(mbb, g_lnl):
    """mbb_sampler_advance_timed (bench.py's timed region on one GPU) enqueues exactly what advance_async does --
    the same chain afterwards, whatever form the run takes (form 7; form 9 with one, two, four and six walkers of each half
    per workgroup, and seven; the launch train for an ensemble too large for any) -- and returns a wall time that
    covers the stream time."""
    like = _cfg2_like(mbb, g_lnl)
    ctx = like.context
    for nw, form in ((60, 7), (300, 9), (600, 9), (2000, 9), (2600, 9), (3600, 9), (4200, 1)):
        p0 = np.array([12.0, 1.8, 600.0, 3.0, 40.0]) * (1.0 + 0.02 * np.random.RandomState(nw).normal(size=(nw, 5)))
        a = mbb.DeviceEnsembleSampler(nw, 5, like, seed=8)
        b = mbb.DeviceEnsembleSampler(nw, 5, like, seed=8)
        a.run_mcmc(p0, 5, storechain=False); b.run_mcmc(p0, 5, storechain=False)
        a.advance_async(40); ctx.sync()
        wall, ms = b.advance_timed(40)
        assert ctx.info("last_kernel_form") == form
        assert 0.0 < ms * 1e-3 <= wall < 1.0
        ra, rb = a.run_mcmc(None, 3), b.run_mcmc(None, 3)
        assert np.array_equal(ra[0], rb[0]) and np.array_equal(ra[1], rb[1])


def test_one_launch_sampler_falls_back_when_it_cannot_be_resident(mbb, g_lnl):
    """The one-launch forms need every workgroup on the GPU at once.  An ensemble with no form that fits -- more
    than 8 walkers per CU and half, or the resident forms switched off beyond form 7's range -- is not started and
    left to time out: the run takes the launch-per-half-step form, same chain; and each form ends where the next
    begins."""
    like = _cfg2_like(mbb, g_lnl)
    ctx = like.context
    cus = ctx.info("cu_count")
    for nw, opts, form in ((cus, {}, 7), (cus + 2, {}, 9), (cus + 2, {"resident_sampler": 0}, 1), (12 * cus, {}, 9),
                           (12 * cus + 2, {}, 9), (16 * cus, {}, 9), (16 * cus + 2, {}, 1), (cus, {"merged_flow_sampler": 0}, 9),
                           (60, {"resident_sampler": 2, "resident_walkers": 9}, 9)):
        for o in ("merged_flow_sampler", "resident_sampler"):
            ctx.set_option(o, 1)
        ctx.set_option("resident_walkers", 0)
        for o, v in opts.items():
            ctx.set_option(o, v)
        p0 = np.array([12.0, 1.8, 600.0, 3.0, 40.0]) * (1.0 + 0.02 * np.random.RandomState(8).normal(size=(nw, 5)))
        ctx.set_option("lookahead_sampler", 1)
        s = mbb.DeviceEnsembleSampler(nw, 5, like, seed=9)
        pos, lnp, _ = s.run_mcmc(p0, 4)
        assert ctx.info("last_kernel_form") == form and np.isfinite(lnp).all(), (nw, opts, ctx.info("last_kernel_form"))
        if nw > 8 * cus:
            continue                                          # (the large ones are compared in the test above this one's size)
        ctx.set_option("lookahead_sampler", 0)
        s2 = mbb.DeviceEnsembleSampler(nw, 5, like, seed=9)
        pos2, lnp2, _ = s2.run_mcmc(p0, 4)
        assert np.array_equal(pos, pos2) and np.array_equal(lnp, lnp2) and np.array_equal(s.chain, s2.chain), (nw, opts)


def test_one_launch_sampler_run_that_times_out_is_redone_as_a_launch_train(mbb, g_lnl):
    """A one-launch run whose waits give up (here: no poll budget, the first word that is not there; in the field: a workgroup
    that is not resident because another process holds CUs) ends with an error flag, not a hang;
    mbb_sampler_run then restores the state the run started from and does the same steps as a train
    of launches -- the caller gets the chain it would have got, with a warning.  A give-up is a
    property of the moment: the next run takes the one-launch form again; only three give-ups in a
    row rest it, for sixteen runs."""
    for merged, form in ((1, 7), (0, 9)):
        like = _cfg2_like(mbb, g_lnl)
        ctx = like.context
        p0 = np.array([12.0, 1.8, 600.0, 3.0, 40.0]) * (1.0 + 0.02 * np.random.RandomState(8).normal(size=(100, 5)))
        ctx.set_option("lookahead_sampler", 0)
        s0 = mbb.DeviceEnsembleSampler(100, 5, like, seed=4)
        a = s0.run_mcmc(p0, 6)[:2] + (s0.run_mcmc(None, 5)[:2]) + (s0.chain.copy(), s0.naccepted.copy())
        ctx.set_option("lookahead_sampler", 1); ctx.set_option("flow_sampler", 1); ctx.set_option("merged_flow_sampler", merged)
        s1 = mbb.DeviceEnsembleSampler(100, 5, like, seed=4)
        first = s1.run_mcmc(p0, 6)[:2]
        assert ctx.info("last_kernel_form") == form and ctx.info("flow_fallbacks") == 0
        ctx.set_option("flow_spin_log2", 63)                      # the second run gives up at once ...
        with pytest.warns(RuntimeWarning, match="redone as a train"):
            second = s1.run_mcmc(None, 5)[:2]
        assert ctx.info("flow_fallbacks") == 1 and ctx.info("last_kernel_form") == 1      # ... and was redone
        b = first + second + (s1.chain.copy(), s1.naccepted.copy())
        for x, y in zip(a, b):
            assert np.array_equal(x, y), form
        ctx.set_option("flow_spin_log2", 0)
        # (the reference sampler makes the same runs as a launch train: a run's length is part of its
        # random stream)
        def ref_run(n):
            ctx.set_option("lookahead_sampler", 0)
            out = s0.run_mcmc(None, n)[:2]
            ctx.set_option("lookahead_sampler", 1)
            return out
        # the next run is a one-launch run again, and the same chain goes on
        c1 = s1.run_mcmc(None, 4)[:2]
        assert ctx.info("last_kernel_form") == form and ctx.info("flow_fallbacks") == 1
        c0 = ref_run(4)
        assert np.array_equal(c0[0], c1[0]) and np.array_equal(c0[1], c1[1])
        # three give-ups in a row: the form rests (16 runs on the train, no give-ups), then comes back
        ctx.set_option("flow_spin_log2", 63)
        for k in range(3):
            with pytest.warns(RuntimeWarning):
                c1 = s1.run_mcmc(None, 2)[:2]
            ctx.set_option("flow_spin_log2", 0)
            c0 = ref_run(2)                                       # (a train: neither a give-up nor a rest)
            ctx.set_option("flow_spin_log2", 63)
            assert np.array_equal(c0[0], c1[0]) and np.array_equal(c0[1], c1[1])
        assert ctx.info("flow_fallbacks") == 4 and ctx.info("flow_resting") > 0
        while ctx.info("flow_resting") > 0:
            c1 = s1.run_mcmc(None, 2)[:2]
            assert ctx.info("last_kernel_form") == 1 and ctx.info("flow_fallbacks") == 4
            c0 = ref_run(2)
            assert np.array_equal(c0[0], c1[0]) and np.array_equal(c0[1], c1[1])
        ctx.set_option("flow_spin_log2", 0)
        c1 = s1.run_mcmc(None, 2)[:2]
        assert ctx.info("last_kernel_form") == form and ctx.info("flow_resting") == 0
        c0 = ref_run(2)
        assert np.array_equal(c0[0], c1[0]) and np.array_equal(c0[1], c1[1])
        # an asynchronous advance keeps nothing to redo a run from: the give-up surfaces at the next
        # synchronous call as an error, and the sampler wants its state set again
        ctx.set_option("flow_spin_log2", 63)
        s1.advance_async(5); ctx.sync()
        ctx.set_option("flow_spin_log2", 0)
        nfall = ctx.info("flow_fallbacks")
        # (a run long enough to take the one-launch form itself: round 3 backed the lost rows up, found the stale
        # flag after its own launch, "redid" the run from them and returned a chain; a reset must not wipe the
        # give-up either, and a further asynchronous advance is refused as well)
        for attempt in (lambda: s1.run_mcmc(None, 5), lambda: s1.run_mcmc(None, 1), lambda: (ctx.lib.mbb_sampler_reset(ctx.h, s1._h), s1.run_mcmc(None, 5)),
                        lambda: s1.advance_async(5), lambda: s1.advance_timed(5)):
            with pytest.raises(Exception, match="set the sampler's state again"):
                attempt()
        assert ctx.info("flow_fallbacks") == nfall
        # with its state set again the sampler goes on, in the one-launch form, on a valid chain
        pos3, lnp3 = s1.run_mcmc(p0, 3)[:2]
        assert ctx.info("last_kernel_form") == form and ctx.info("flow_fallbacks") == nfall
        lnl_close(lnp3, like(pos3))


def test_one_launch_sampler_ignores_records_left_in_reused_memory(mbb, g_lnl):
    """A one-launch run takes its proposal records by their check words.  Samplers of short runs on
    context after context get the same device memory back from the allocator, with another run's
    records still in it: they must not be mistaken for this run's (the check words carry a
    process-wide launch number, and a sampler's record memory starts zeroed)."""
    for k in range(8):
        res = []
        for look, merged in ((1, 1), (1, 0), (0, 0)):
            like = _cfg2_like(mbb, g_lnl)                         # a context of its own each time
            like.context.set_option("lookahead_sampler", look); like.context.set_option("merged_flow_sampler", merged)
            p0 = np.array([12.0, 1.8, 600.0, 3.0, 40.0]) * (1.0 + 0.02 * np.random.RandomState(100 + k).normal(size=(64, 5)))
            s = mbb.DeviceEnsembleSampler(64, 5, like, seed=4)     # the same seed: the same check-word arithmetic
            res.append(s.run_mcmc(p0, 3)[:2] + (s.chain.copy(),))
            del s, like
        for r in res[:2]:
            for x, y in zip(r, res[2]):
                assert np.array_equal(x, y), k
