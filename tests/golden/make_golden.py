#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ from the *reference itself*.

TEST INFRASTRUCTURE.  Runs only in the build container, where the upstream
reference is mounted read-only at /root/reference.  It imports the reference's
hot-path modules unmodified (modified_blackbody.py, response.py, likelihood.py
and the Cython fnu.pyx, compiled under /tmp) through the import shim described
in SURVEY.md Appendix A, evaluates them on seeded inputs and writes only
*numbers* (arrays + scalars) to tests/golden/*.npz.  Nothing from the reference
(source, bytecode, compiled objects) is written into the repository.

The raw transmission curves (two numeric columns per instrument) are packed
into mbb_emcee_amd/resources/passband_curves.npz -- they are the input data of
the path, not code.

Usage:  python tests/golden/make_golden.py          (about 20 s)
"""
import collections
import collections.abc
import importlib
import os
import shutil
import subprocess
import sys
import types

import numpy as np

REF = "/root/reference"
REFPKG = os.path.join(REF, "mbb_emcee")
SCRATCH = "/tmp/mbb_ref_oracle"
HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
RESDIR = os.path.join(REPO, "mbb_emcee_amd", "resources")


# --------------------------------------------------------------------------
# 1. build the reference's Cython kernel under /tmp (never under the repo)
# --------------------------------------------------------------------------
def build_fnu():
    os.makedirs(SCRATCH, exist_ok=True)
    so = [f for f in os.listdir(SCRATCH) if f.startswith("fnu.") and f.endswith(".so")]
    if so:
        return
    shutil.copy(os.path.join(REFPKG, "fnu.pyx"), SCRATCH)
    with open(os.path.join(SCRATCH, "setup.py"), "w") as fh:
        fh.write(
            "from setuptools import setup, Extension\n"
            "from Cython.Build import cythonize\n"
            "import numpy\n"
            "setup(ext_modules=cythonize([Extension('fnu', ['fnu.pyx'],"
            " include_dirs=[numpy.get_include()], libraries=['m'])],"
            " language_level=3))\n")
    subprocess.check_call([sys.executable, "setup.py", "build_ext", "--inplace"],
                          cwd=SCRATCH, stdout=subprocess.DEVNULL,
                          stderr=subprocess.DEVNULL)


# --------------------------------------------------------------------------
# 2. import shim (compatibility only; no reference file is modified)
# --------------------------------------------------------------------------
def _ascii_read(fn, comment=None, **kw):
    rows = []
    with open(fn) as fh:
        for line in fh:
            s = line.strip()
            if not s or s.startswith("#"):
                continue
            row = []
            for tok in s.split():
                try:
                    row.append(int(tok))
                except ValueError:
                    try:
                        row.append(float(tok))
                    except ValueError:
                        row.append(tok)
            rows.append(tuple(row))
    return rows


def install_shim():
    sys.dont_write_bytecode = True
    sys.path.insert(0, SCRATCH)
    collections.Iterable = collections.abc.Iterable      # utility.py:14
    np.float = float                                      # modified_blackbody.py:457
    _lin = np.linspace

    def linspace(start, stop, *a, **k):                   # response.py:477-487
        if isinstance(start, np.ndarray) and start.size == 1:
            start = float(start.reshape(-1)[0])
        if isinstance(stop, np.ndarray) and stop.size == 1:
            stop = float(stop.reshape(-1)[0])
        return _lin(start, stop, *a, **k)
    np.linspace = linspace

    astropy = types.ModuleType("astropy")
    aio = types.ModuleType("astropy.io")
    aascii = types.ModuleType("astropy.io.ascii")
    aascii.read = _ascii_read
    astropy.io = aio
    aio.ascii = aascii
    sys.modules.update({"astropy": astropy, "astropy.io": aio,
                        "astropy.io.ascii": aascii,
                        "h5py": types.ModuleType("h5py")})
    pkg = types.ModuleType("mbb_emcee")
    pkg.__path__ = [REFPKG]
    pkg.__file__ = os.path.join(REFPKG, "__init__.py")
    sys.modules["mbb_emcee"] = pkg
    mb = importlib.import_module("mbb_emcee.modified_blackbody")
    rs = importlib.import_module("mbb_emcee.response")
    lk = importlib.import_module("mbb_emcee.likelihood")
    return mb, rs, lk


# --------------------------------------------------------------------------
# 3. sanity gate: the reference's own known-answer tests must reproduce
#    (mbb_emcee/tests/test_modified_blackbody.py, test_response.py)
# --------------------------------------------------------------------------
def kat_gate(mb, rs):
    from numpy.testing import assert_allclose
    wave = np.array([250.0, 350.0, 500.0, 850.0])
    m = mb.modified_blackbody(10.0, 2.0, 800.0, 2.0, 45.0)
    assert_allclose(m(wave), [21.96268738, 39.53249977, 45.0, 22.06274444], rtol=1e-4)
    m = mb.modified_blackbody(15.0, 1.8, 200.0, 3.0, 50.0, opthin=True)
    assert_allclose(m(wave), [178.34976, 111.03026, 50.0, 10.880588], rtol=1e-4)
    assert_allclose(mb.modified_blackbody(20.0, 1.9, None, 3.5, 50.0, opthin=True).wavemerge,
                    85.66065, rtol=1e-3)
    assert_allclose(mb.modified_blackbody(40.0, 1.5, 600.0, 3.0, 50.0).wavemerge,
                    60.10021595, rtol=1e-3)
    wheel = rs.response_set()
    assert_allclose(wheel["SPIRE_250um"].normfac, 3.0796e-3, atol=1e-4)
    assert_allclose(wheel["SPIRE_250um"].effective_wavelength, 247.268656, atol=1e-4)
    assert_allclose(wheel["SPIRE_250um"](lambda x: 1), 1.011046, atol=1e-4)
    wheel.add_special("ZSpec_box_1050um_100")
    assert_allclose(wheel["ZSpec_box_1050um_100"].effective_frequency, 286.1655, atol=1e-3)
    return wheel


VARIANTS = [("thin_noalpha", True, True), ("thin_walpha", True, False),
            ("thick_noalpha", False, True), ("thick_walpha", False, False)]

CFG2_BANDS = ["PACS_70um", "PACS_100um", "PACS_160um", "SPIRE_250um",
              "SPIRE_350um", "SPIRE_500um", "SCUBA2_850um", "Bolocam_1.1mm"]
CFG4_BANDS = ["PACS_70um", "PACS_100um", "PACS_160um", "SPIRE_250um",
              "SPIRE_350um", "SPIRE_500um", "SCUBA2_450um", "SCUBA2_850um",
              "SCUBA2_box_850um_85", "ALMA_alma_343", "ALMA_alma_230",
              "ALMA_box_145_7.5"]
SPECIALS = ["ZSpec_box_1050um_100", "SMA_gauss_345_8", "X_dsb_230_16_8",
            "SCUBA2_box_850um_85", "ALMA_alma_343", "ALMA_alma_230",
            "ALMA_alma_100", "ALMA_alma_145", "ALMA_alma_450",
            "ALMA_box_145_7.5", "Y_delta_880um", "Z_gauss_450um_30"]


# --------------------------------------------------------------------------
# fixtures
# --------------------------------------------------------------------------
def pack_raw_curves():
    """Raw two-column transmission curves -> resources/passband_curves.npz."""
    out = {}
    for fn in sorted(os.listdir(os.path.join(REFPKG, "resources"))):
        if fn == "mbb_filterwheel.txt" or not fn.endswith(".txt"):
            continue
        rows = _ascii_read(os.path.join(REFPKG, "resources", fn))
        arr = np.array([[r[0], r[1]] for r in rows], dtype=np.float64)
        out[fn] = arr
    os.makedirs(RESDIR, exist_ok=True)
    np.savez_compressed(os.path.join(RESDIR, "passband_curves.npz"), **out)
    return out


def g1_passbands(rs, wheel):
    """G1: processed passband tables (response.py:252-332)."""
    out = {}
    names = list(wheel.keys())
    for nm in SPECIALS:
        if nm not in wheel:
            wheel.add_special(nm)
    names = sorted(wheel.keys())
    for nm in names:
        r = wheel[nm]
        out[nm + "/wave"] = np.asarray(r._wave, dtype=np.float64)
        out[nm + "/freq"] = np.asarray(r._freq, dtype=np.float64)
        out[nm + "/resp"] = np.asarray(r._resp, dtype=np.float64)
        if not r._isdelta:
            out[nm + "/sedmult"] = np.asarray(r._sedmult, dtype=np.float64)
        out[nm + "/scalars"] = np.array(
            [r._normfac, r._effective_wave, r._effective_freq,
             r._normwave, r._normfreq, float(r._nresp), float(r._isdelta),
             float(r(lambda x: np.ones_like(np.atleast_1d(x), dtype=float))
                   if not r._isdelta else 1.0)], dtype=np.float64)
    out["names"] = np.array(names)
    # counts-type sensitivity and bb / none normalisations through setup()
    extra = [("SPIRE_250.txt", "counts", "power", 250.0, -1.0),
             ("MIPS_70.txt", "energy", "none", 71.44, 0.0),
             ("PACS_100.txt", "counts", "flat", 100.0, 0.0)]
    for spec, sens, ntype, xn, npar in extra:
        r = rs.response("x")
        r.setup(spec, xtype="wave", xunits="microns", senstype=sens,
                normtype=ntype, xnorm=xn, normparam=npar, dir="!package-dir!")
        key = "setup:%s:%s:%s" % (spec, sens, ntype)
        out[key + "/sedmult"] = np.asarray(r._sedmult)
        out[key + "/scalars"] = np.array([r._normfac, r._effective_wave,
                                          r._effective_freq], dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "passbands.npz"), **out)


def _param_sets(rng, n):
    """Parameter rows: half near astrophysical values, half over the prior box."""
    near = np.column_stack([rng.normal(14, 4, n // 2).clip(3, 60),
                            rng.normal(1.8, 0.5, n // 2).clip(0.2, 4),
                            rng.uniform(20, 1500, n // 2),
                            rng.normal(3.0, 1.0, n // 2).clip(0.3, 8),
                            rng.uniform(1, 100, n // 2)])
    box = np.column_stack([rng.uniform(1, 80, n - n // 2),
                           rng.uniform(0.1, 21, n - n // 2),
                           rng.uniform(1, 4500, n - n // 2),
                           rng.uniform(0.1, 21, n - n // 2),
                           rng.uniform(1e-3, 100, n - n // 2)])
    return np.vstack([near, box])


def g2_g3_sed(mb):
    """G2 prologue scalars, G3 f_nu vectors (modified_blackbody.py:168-337,
    :493-554; fnu.pyx:9-108), plus max_wave (:581-637)."""
    rng = np.random.RandomState(20260301)
    pars = _param_sets(rng, 96)
    # a few rows that force xnorm > xmerge (normalisation on the power law side)
    pars[:6, 0] = [2.0, 2.5, 3.0, 1.5, 4.0, 2.2]
    pars[:6, 3] = [1.0, 0.8, 1.5, 0.5, 0.6, 2.0]
    pars[:6, 1] = [1.0, 1.5, 0.5, 2.0, 1.2, 0.3]
    grid = np.exp(np.linspace(np.log(20.0), np.log(3000.0), 160))
    out = {"pars": pars, "wave_grid": grid}
    for nm, opthin, noalpha in VARIANTS:
        scal = np.full((len(pars), 6), np.nan)
        fnu_grid = np.full((len(pars), len(grid)), np.nan)
        fnu_scalar = np.full(len(pars), np.nan)
        fint = np.full((len(pars), 2), np.nan)
        for i, p in enumerate(pars):
            m = mb.modified_blackbody(p[0], p[1], p[2], p[3], p[4],
                                      wavenorm=500.0, noalpha=noalpha, opthin=opthin)
            scal[i, 0] = m._normfac
            scal[i, 1] = m._xmerge if not noalpha else np.nan
            scal[i, 2] = m._kappa if not noalpha else np.nan
            scal[i, 3] = m._x0 if not opthin else np.nan
            scal[i, 4] = m.wavemerge if not noalpha else np.nan
            scal[i, 5] = m.max_wave()
            fnu_grid[i] = m(grid)
            fnu_scalar[i] = float(m(433.0)[0])     # numpy f_nu path (:441-491)
            if i % 3 == 0:                         # scipy quad of f_nu (:639-674)
                fint[i, 0] = m.freq_integrate(24.0, 3000.0)
                fint[i, 1] = m.freq_integrate(42.5, 122.5)
        out[nm + "/scalars"] = scal
        out[nm + "/fnu_grid"] = fnu_grid
        out[nm + "/fnu_scalar433"] = fnu_scalar
        out[nm + "/freq_integrate"] = fint
    # wavenorm != 500
    m = mb.modified_blackbody(25.0, 1.6, 150.0, 2.5, 12.0, wavenorm=850.0)
    out["wn850/fnu_grid"] = m(grid)
    out["wn850/scalars"] = np.array([m._normfac, m._xmerge, m._kappa, m._x0])
    np.savez_compressed(os.path.join(HERE, "sed.npz"), **out)


def _edge_rows(base):
    """Rows that exercise the limit gate and the soft upper walls."""
    rows = []
    for idx, v in [(0, 0.5), (1, 0.05), (2, 0.5), (3, 0.05), (4, 1e-5)]:
        r = base.copy(); r[idx] = v; rows.append(r)          # below lower limit
    for idx, v in [(1, 20.4), (3, 20.7), (2, 5000.0), (1, 21.5), (3, 23.0)]:
        r = base.copy(); r[idx] = v; rows.append(r)          # above soft wall
    r = base.copy(); r[1] = 20.2; r[3] = 20.1; rows.append(r)
    return np.array(rows)


def _eval(like, pars):
    lnl = np.empty(len(pars))
    flux = np.full((len(pars), like._ndata), np.nan)
    for i, p in enumerate(pars):
        lnl[i] = like(p)
        if np.isfinite(lnl[i]):
            if like._response_integrate:
                flux[i] = [float(r(like._sed)) for r in like._responses]
            else:
                flux[i] = like._sed(like._wave)
    return lnl, flux


def g4_lnlike(mb, lk):
    """G4: model fluxes and lnL for configs 1, 2 and 4 (likelihood.py:790-834)."""
    out = {}
    # ---------------- cfg1: 5 delta bands, 50 walkers -------------------
    wave1 = np.array([100.0, 160.0, 250.0, 350.0, 500.0])
    rng = np.random.RandomState(1)
    for nm, opthin, noalpha in VARIANTS:
        truth = np.array([12.0, 1.8, 600.0, 3.0, 40.0])
        sed = mb.modified_blackbody(*truth, noalpha=noalpha, opthin=opthin)
        flux = np.asarray(sed(wave1))
        unc = 0.1 * flux + 1.0
        like = lk.likelihood(noalpha=noalpha, opthin=opthin)
        like.set_phot(wave1, flux, unc)
        pars = np.column_stack([rng.normal(12, 1, 50), rng.normal(1.8, 0.2, 50),
                                rng.normal(600, 50, 50), rng.normal(3, 0.3, 50),
                                rng.normal(40, 3, 50)])
        pars = np.vstack([pars, _edge_rows(truth)])
        lnl, mflux = _eval(like, pars)
        k = "cfg1/" + nm
        out[k + "/wave"] = wave1; out[k + "/flux"] = flux; out[k + "/unc"] = unc
        out[k + "/pars"] = pars; out[k + "/lnl"] = lnl; out[k + "/model_flux"] = mflux
        out[k + "/lowlim"] = like._lowlim.copy()
        out[k + "/uplim"] = like._uplim.copy()
        out[k + "/has_uplim"] = np.array(like._has_uplim, dtype=np.int32)

    # ---------------- cfg2: 8 bands with response, 256+edge rows --------
    rng = np.random.RandomState(0)
    walkers = np.column_stack([rng.normal(12, 1, 2000), rng.normal(1.8, 0.2, 2000),
                               rng.normal(600, 50, 2000), rng.normal(3, 0.3, 2000),
                               rng.normal(40, 3, 2000)])[:256]
    box = _param_sets(np.random.RandomState(77), 64)
    for nm, opthin, noalpha in VARIANTS:
        truth = np.array([12.0, 1.8, 600.0, 3.0, 40.0])
        like = lk.likelihood(noalpha=noalpha, opthin=opthin, response=True)
        like.set_phot(CFG2_BANDS, np.ones(8), np.ones(8))
        like._set_sed(truth)
        flux = np.array([float(r(like._sed)) for r in like._responses])
        unc = 0.1 * flux + 1.0
        like = lk.likelihood(noalpha=noalpha, opthin=opthin, response=True)
        like.set_phot(CFG2_BANDS, flux, unc)
        pars = np.vstack([walkers, _edge_rows(truth), box])
        lnl, mflux = _eval(like, pars)
        k = "cfg2/" + nm
        out[k + "/flux"] = flux; out[k + "/unc"] = unc
        out[k + "/pars"] = pars; out[k + "/lnl"] = lnl; out[k + "/model_flux"] = mflux
        out[k + "/uplim"] = like._uplim.copy()
    out["cfg2/bands"] = np.array(CFG2_BANDS)

    # cfg2 thick+alpha with Gaussian priors and lambda_peak prior + wall
    truth = np.array([12.0, 1.8, 600.0, 3.0, 40.0])
    like = lk.likelihood(response=True)
    like.set_phot(CFG2_BANDS, out["cfg2/thick_walpha/flux"], out["cfg2/thick_walpha/unc"])
    like.set_gaussian_prior("T", 11.0, 2.0)
    like.set_gaussian_prior("beta", 1.9, 0.3)
    like.set_gaussian_prior("lambda0", 550.0, 80.0)
    like.set_gaussian_prior("alpha", 3.2, 0.5)
    like.set_gaussian_prior("fnorm", 42.0, 4.0)
    like.set_uplim("T", 12.5)
    like.set_uplim("fnorm", 41.0)
    like.set_lowlim("beta", 1.5)
    pars = walkers[:96]
    lnl, mflux = _eval(like, pars)
    out["cfg2/priors/pars"] = pars; out["cfg2/priors/lnl"] = lnl
    like.set_gaussian_prior("lambda_peak", 260.0, 15.0)
    like.set_uplim("lambda_peak", 265.0)
    lnl, _ = _eval(like, pars)
    out["cfg2/priors_peak/lnl"] = lnl
    out["cfg2/priors_peak/max_wave"] = np.array(
        [mb.modified_blackbody(*p).max_wave() for p in pars])
    # thin+alpha lambda_peak prior
    like = lk.likelihood(response=True, opthin=True)
    like.set_phot(CFG2_BANDS, out["cfg2/thin_walpha/flux"], out["cfg2/thin_walpha/unc"])
    like.set_gaussian_prior("peaklam", 250.0, 20.0)
    lnl, _ = _eval(like, pars)
    out["cfg2/thin_peak/lnl"] = lnl

    # ---------------- cfg4: 12 bands, covariance ------------------------
    rng = np.random.RandomState(4)
    for nm, opthin, noalpha in VARIANTS:
        like = lk.likelihood(noalpha=noalpha, opthin=opthin, response=True)
        like.set_phot(CFG4_BANDS, np.ones(12), np.ones(12))
        like._set_sed(truth)
        flux = np.array([float(r(like._sed)) for r in like._responses])
        unc = 0.1 * flux + 1.0
        A = np.random.RandomState(44).normal(0, 1, (12, 12))
        cov = np.diag(unc ** 2) + 0.01 * A.dot(A.T)
        like = lk.likelihood(noalpha=noalpha, opthin=opthin, response=True)
        like.set_phot(CFG4_BANDS, flux, unc)
        like.set_cov(cov)
        pars = np.vstack([walkers, _edge_rows(truth), box])
        lnl, mflux = _eval(like, pars)
        k = "cfg4/" + nm
        out[k + "/flux"] = flux; out[k + "/unc"] = unc; out[k + "/cov"] = cov
        out[k + "/pars"] = pars; out[k + "/lnl"] = lnl; out[k + "/model_flux"] = mflux
        out[k + "/uplim"] = like._uplim.copy()
        out[k + "/eff_wave"] = like._wave.copy()
    out["cfg4/bands"] = np.array(CFG4_BANDS)
    np.savez_compressed(os.path.join(HERE, "lnlike.npz"), **out)


def main():
    if not os.path.isdir(REFPKG):
        raise SystemExit("reference not mounted at %s" % REF)
    build_fnu()
    mb, rs, lk = install_shim()
    wheel = kat_gate(mb, rs)
    print("reference KATs reproduce")
    pack_raw_curves()
    g1_passbands(rs, wheel)
    g2_g3_sed(mb)
    g4_lnlike(mb, lk)
    for fn in sorted(os.listdir(HERE)):
        if fn.endswith(".npz"):
            print("  %-16s %8d B" % (fn, os.path.getsize(os.path.join(HERE, fn))))


if __name__ == "__main__":
    main()
