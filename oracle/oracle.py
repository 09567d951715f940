"""ctypes front-end of the CPU parity oracle (oracle/mbb_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package never imports this module.
Parity status: pinned against tests/golden/*.npz (made by the reference itself).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# MBB_ORACLE_LIB: an alternative build of the same source (the sanitizer build, tests/_asan_worker.py)
_LIB = os.environ.get("MBB_ORACLE_LIB", os.path.join(_HERE, "libmbb_oracle.so"))

STATUS = {0: "ok", 1: "below lower limit", 2: "alpha must be positive",
          3: "beta must be non-negative", 4: "couldn't bracket low alpha merge point",
          5: "couldn't bracket high alpha merge point", 6: "root finder did not converge",
          7: "couldn't bracket SED maximum"}


def build(force=False):
    src = os.path.join(_HERE, "mbb_oracle.c")
    if "MBB_ORACLE_LIB" in os.environ:
        return _LIB
    if force or not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libmbb_oracle.so"])
    return _LIB


class Sed(C.Structure):
    _fields_ = [("opthin", C.c_int), ("noalpha", C.c_int),
                ("T", C.c_double), ("beta", C.c_double), ("lambda0", C.c_double),
                ("alpha", C.c_double), ("fnorm", C.c_double), ("wavenorm", C.c_double),
                ("hcokt", C.c_double), ("x0", C.c_double), ("xnorm", C.c_double),
                ("normfac", C.c_double), ("xmerge", C.c_double), ("kappa", C.c_double)]


_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


class Like(C.Structure):
    _fields_ = [("opthin", C.c_int), ("noalpha", C.c_int), ("wavenorm", C.c_double),
                ("nb", C.c_int), ("response_integrate", C.c_int),
                ("wave", _dp), ("sedmult", _dp), ("normfac", _dp), ("offsets", _ip),
                ("flux", _dp), ("ivar", _dp), ("invcov", _dp),
                ("lowlim", C.c_double * 5), ("has_uplim", C.c_int32 * 6),
                ("uplim", C.c_double * 6), ("has_gprior", C.c_int32 * 6),
                ("gprior_mean", C.c_double * 6), ("gprior_ivar", C.c_double * 6)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB)
        L.mbbo_sed_init.argtypes = [C.POINTER(Sed)] + [C.c_double] * 6 + [C.c_int, C.c_int]
        L.mbbo_sed_init.restype = C.c_int
        L.mbbo_fnu.argtypes = [C.POINTER(Sed), _dp, C.c_int, _dp]
        L.mbbo_fnu.restype = None
        L.mbbo_sed_call.argtypes = [C.POINTER(Sed), _dp, C.c_int, _dp, _dp]
        L.mbbo_sed_call.restype = None
        L.mbbo_max_wave.argtypes = [C.POINTER(Sed), _dp]
        L.mbbo_max_wave.restype = C.c_int
        L.mbbo_lnlike.argtypes = [C.POINTER(Like), _dp, _dp, _dp, _dp]
        L.mbbo_lnlike.restype = C.c_int
        L.mbbo_lnlike_batch.argtypes = [C.POINTER(Like), _dp, C.c_int, _dp, _ip, _dp, C.c_int]
        L.mbbo_lnlike_batch.restype = C.c_int
        L.mbbo_num_threads.restype = C.c_int
        L.mbbo_sizeof_sed.restype = C.c_size_t
        L.mbbo_sizeof_like.restype = C.c_size_t
        assert L.mbbo_sizeof_sed() == C.sizeof(Sed)
        assert L.mbbo_sizeof_like() == C.sizeof(Like)
        _lib = L
    return _lib


def _d(a):
    return a.ctypes.data_as(_dp)


class OracleSED(object):
    """modified_blackbody restated (modified_blackbody.py:154-674)."""

    def __init__(self, T, beta, lambda0, alpha, fnorm, wavenorm=500.0,
                 noalpha=False, opthin=False):
        self.s = Sed()
        self.status = lib().mbbo_sed_init(
            C.byref(self.s), float(T), float(beta),
            float(lambda0 if lambda0 is not None else 0.0),
            float(alpha if alpha is not None else 0.0), float(fnorm),
            float(wavenorm), int(bool(noalpha)), int(bool(opthin)))
        if self.status:
            raise ValueError(STATUS[self.status])

    def f_nu(self, freq):
        f = np.ascontiguousarray(np.atleast_1d(freq), dtype=np.float64)
        out = np.empty_like(f)
        lib().mbbo_fnu(C.byref(self.s), _d(f), f.size, _d(out))
        return out

    def __call__(self, wave):
        w = np.ascontiguousarray(np.atleast_1d(wave), dtype=np.float64)
        out = np.empty_like(w)
        scr = np.empty_like(w)
        lib().mbbo_sed_call(C.byref(self.s), _d(w), w.size, _d(out), _d(scr))
        return out

    def max_wave(self):
        out = C.c_double()
        st = lib().mbbo_max_wave(C.byref(self.s), C.byref(out))
        if st:
            raise ValueError(STATUS[st])
        return out.value

    @property
    def wavemerge(self):
        if self.s.noalpha:
            return None
        return self.s.hcokt / self.s.xmerge

    def freq_integrate(self, minwave, maxwave):
        """modified_blackbody.freq_integrate (modified_blackbody.py:639-674): scipy's `quad` (QUADPACK QAGS,
        the third-party routine the reference itself calls; scipy is part of this image) over f_nu in GHz,
        in erg/s/cm^2."""
        from scipy.integrate import quad
        minwave, maxwave = float(minwave), float(maxwave)
        if minwave <= 0.0:
            raise ValueError("Minimum wavelength must be > 0.0")
        if minwave > maxwave:
            minwave, maxwave = maxwave, minwave
        um_to_GHz = 299792458e-3
        return 1e-17 * quad(lambda f: float(self.f_nu(f)[0]), um_to_GHz / maxwave, um_to_GHz / minwave)[0]


# ---- chain post-processing restated (results.py:534-944), one chain entry at a time as the reference does.
# Checked against fixtures the reference's own mbb_results produced (tests/golden/results.npz).
MPC_IN_CM = 3.0856775814913673e24          # astropy's Mpc (results.py:778 `lumdist.to(u.cm)`)


def post_peaklambda(chain, opthin=False, noalpha=False, as_reference=True):
    """results.py:570-581.  The reference maps `peaklambda_inner(step, opthin=False, noalpha=False)` without
    keyword arguments: whatever the fit's model, the peak is that of the optically thick model with alpha
    (`as_reference=True`); False gives the fit's own model (what likelihood.py:748-750 uses for its prior)."""
    c = np.asarray(chain, dtype=np.float64).reshape(-1, 5)
    ot, na = (False, False) if as_reference else (opthin, noalpha)
    out = np.array([OracleSED(*p, opthin=ot, noalpha=na).max_wave() for p in c])
    return out.reshape(np.shape(chain)[:-1])


def post_lir(chain, redshift, lumdist_mpc, opthin=False, noalpha=False, wavemin=8.0, wavemax=1000.0):
    """results.py:627-674 with mbb_freqint (results.py:1271-1330): 10^12 L_sun."""
    c = np.asarray(chain, dtype=np.float64).reshape(-1, 5)
    opz = 1.0 + float(redshift)
    pre = 3.11749657e4 * float(lumdist_mpc) ** 2
    out = np.array([OracleSED(*p, opthin=opthin, noalpha=noalpha).freq_integrate(wavemin * opz, wavemax * opz) for p in c])
    return pre * out.reshape(np.shape(chain)[:-1])


def post_dustmass(chain, redshift, lumdist_mpc, wavenorm=500.0, opthin=False, kappa=2.64, kappa_wave=125.0):
    """results.py:726-801 (_dmass_calc + compute_dustmass), scalar arithmetic with math.expm1 as there: 10^8 M_sun."""
    import math
    c = np.asarray(chain, dtype=np.float64).reshape(-1, 5)
    dl2 = (float(lumdist_mpc) * MPC_IN_CM) ** 2
    opz = 1.0 + float(redshift)
    wavenorm_rest = wavenorm / opz
    nunorm_rest = 299792458e6 / wavenorm_rest
    temp_fac = 6.6260693e-27 * nunorm_rest / 1.38065e-16
    bnu_fac = 2 * 6.6260693e-27 * nunorm_rest ** 3 / 299792458e2 ** 2
    knu_fac = wavenorm_rest / kappa_wave
    msolar8 = 1.97792e41
    out = np.empty(c.shape[0])
    for i, step in enumerate(c):
        T = step[0] * opz
        beta = step[1]
        S_nu = step[4] * 1e-26
        B_nu = bnu_fac / math.expm1(temp_fac / T)
        K_nu = 10.0 * kappa * knu_fac ** (-beta)
        m = dl2 * S_nu / (opz * K_nu * B_nu * msolar8)
        if not opthin:
            tau_nu = (step[2] / wavenorm) ** beta
            m *= -tau_nu / math.expm1(-tau_nu)
        out[i] = m
    return out.reshape(np.shape(chain)[:-1])


def post_predict_flux(chain, spec, opthin=False, noalpha=False):
    """results.py:895-944: `spec` a wavelength in um (the SED there) or one passband as (wave, sedmult, normfac)
    (response.py:572-576).  The SED is built with the modified_blackbody default wavenorm, as there."""
    c = np.asarray(chain, dtype=np.float64).reshape(-1, 5)
    if isinstance(spec, tuple):
        L = OracleLikelihood(np.ones(1), np.ones(1), bands=[spec], opthin=opthin, noalpha=noalpha,
                             lowlim=np.full(5, -np.inf))
        out = L(c, return_flux=True)[1][:, 0]
    else:
        out = np.array([OracleSED(*p, opthin=opthin, noalpha=noalpha)(float(spec))[0] for p in c])
    return out.reshape(np.shape(chain)[:-1])


class OracleLikelihood(object):
    """likelihood.__call__ restated (likelihood.py:790-834) on explicit tables.

    bands: list of (wave[n], sedmult[n], normfac) per passband
           (response.py:252-332 quantities), or None for plain wavelengths.
    """

    def __init__(self, flux, unc, bands=None, wave=None, cov=None,
                 opthin=False, noalpha=False, wavenorm=500.0, lowlim=None,
                 has_uplim=None, uplim=None, has_gprior=None, gprior_mean=None,
                 gprior_sigma=None):
        self.flux = np.ascontiguousarray(flux, dtype=np.float64)
        nb = self.flux.size
        self.ivar = np.ascontiguousarray(1.0 / np.asarray(unc, dtype=np.float64) ** 2)
        L = Like()
        L.opthin = int(bool(opthin)); L.noalpha = int(bool(noalpha))
        L.wavenorm = float(wavenorm); L.nb = nb
        if bands is not None:
            assert len(bands) == nb
            self.wave = np.ascontiguousarray(np.concatenate([np.atleast_1d(b[0]) for b in bands]),
                                             dtype=np.float64)
            self.sedmult = np.ascontiguousarray(np.concatenate([np.atleast_1d(b[1]) for b in bands]),
                                                dtype=np.float64)
            self.normfac = np.array([b[2] for b in bands], dtype=np.float64)
            self.offsets = np.zeros(nb + 1, dtype=np.int32)
            self.offsets[1:] = np.cumsum([np.atleast_1d(b[0]).size for b in bands])
            L.response_integrate = 1
            L.sedmult = _d(self.sedmult); L.normfac = _d(self.normfac)
            L.offsets = self.offsets.ctypes.data_as(_ip)
            maxwave = None
        else:
            self.wave = np.ascontiguousarray(wave, dtype=np.float64)
            assert self.wave.size == nb
            L.response_integrate = 0
        L.wave = _d(self.wave)
        L.flux = _d(self.flux); L.ivar = _d(self.ivar)
        if cov is not None:
            self.invcov = np.ascontiguousarray(np.linalg.inv(np.asarray(cov, dtype=np.float64)))
            L.invcov = _d(self.invcov)
        # likelihood.py:73, :83-85 defaults
        ll = np.array([1, 0.1, 1, 0.1, 1e-3]) if lowlim is None else np.asarray(lowlim, float)
        hu = [0, 1, 0, 1, 0, 0] if has_uplim is None else list(has_uplim)
        inf = float("inf")
        ul = [inf, 20.0, inf, 20.0, inf, inf] if uplim is None else list(uplim)
        for i in range(5):
            L.lowlim[i] = ll[i]
        for i in range(6):
            L.has_uplim[i] = int(hu[i]); L.uplim[i] = float(ul[i])
            L.has_gprior[i] = int(has_gprior[i]) if has_gprior is not None else 0
            L.gprior_mean[i] = float(gprior_mean[i]) if gprior_mean is not None else 0.0
            sig = float(gprior_sigma[i]) if gprior_sigma is not None else 1.0
            L.gprior_ivar[i] = 1.0 / sig ** 2 if sig != 0 else 1.0
        self.L = L
        self.nb = nb

    def __call__(self, pars, nthreads=1, return_flux=False):
        p = np.ascontiguousarray(np.atleast_2d(pars), dtype=np.float64)
        n = p.shape[0]
        lnl = np.empty(n)
        st = np.zeros(n, dtype=np.int32)
        fl = np.full((n, self.nb), np.nan)
        lib().mbbo_lnlike_batch(C.byref(self.L), _d(p), n, _d(lnl),
                                st.ctypes.data_as(_ip), _d(fl), int(nthreads))
        self.status = st
        if return_flux:
            return lnl, fl
        return lnl


def num_threads():
    return lib().mbbo_num_threads()
