"""Modified blackbody SED, evaluated on the MI355X.

Host-side mirror of the reference's ``modified_blackbody`` class
(reference mbb_emcee/modified_blackbody.py:154-674): same constructor, same
properties, same call conventions.  All arithmetic (the constructor's
normalisation / merge-point solve, f_nu on a grid, the SED peak) runs in HIP
kernels through the C-ABI; there is no CPU implementation here.
"""
import math

import numpy as np

from . import _native
from .utility import isiterable as _is_iterable

__all__ = ["modified_blackbody", "alpha_merge_eqn"]

# modified_blackbody.py:15-18
c = 299792458e6  # um / s
h = 6.6260693e-34  # J s
k = 1.3806505e-23  # J / K
um_to_GHz = 299792458e-3


def alpha_merge_eqn(x, alpha, beta, x0, opthin=False):
    """The function whose root x = h nu / k T is where the blue-side power law nu^-alpha takes over from the modified
    blackbody (reference modified_blackbody.py:122-151):

        x - (1 - e^-x) (3 + alpha + beta b),   b = y / (e^y - 1),   y = (x / x0)^beta   (b = 0 where y overflows).

    `opthin` is accepted and, as in the reference, plays no part here (the optically thin merge point has its own
    fixed-point form, :262).  The constructor on the device finds the root itself (csrc/mbb_device.hip.h
    `thick_merge_root`, parity: tests/golden/sed.npz xmerge to 1e-10); this host form is for callers that import the name."""
    try:
        y = math.pow(x / x0, beta)
        b = y / math.expm1(y) if y > 0.0 else 1.0
    except OverflowError:
        b = 0.0
    return x - (3.0 + alpha + beta * b) * -math.expm1(-x)


class modified_blackbody(object):
    """f_nu ~ (1 - exp(-(nu/nu0)^beta)) B_nu(T), optionally optically thin,
    optionally joined to a nu^-alpha power law on the Wien side.

    Parameters as the reference (modified_blackbody.py:168-196):
    T [K], beta, lambda0 [um], alpha, fnorm [mJy] at wavenorm [um].
    """

    def __init__(self, T, beta, lambda0, alpha, fnorm, wavenorm=500.0,
                 noalpha=False, opthin=False, context=None):
        self._T = float(T)
        self._beta = float(beta)
        self._hasalpha = not bool(noalpha)
        self._alpha = float(alpha) if self._hasalpha else None
        self._fnorm = float(fnorm)
        self._wavenorm = float(wavenorm)
        self._opthin = bool(opthin)
        self._lambda0 = None if self._opthin else float(lambda0)
        # modified_blackbody.py:219-224
        if self._hasalpha and self._alpha <= 0.0:
            raise ValueError("alpha must be positive.  You gave: {:.5g}".format(self._alpha))
        if self._beta < 0.0:
            raise ValueError("beta must be non-negative.  You gave: {:.5g}".format(self._beta))
        self._ctx = context if context is not None else _native.default_context()
        self._pars = np.array([[self._T, self._beta,
                                self._lambda0 if self._lambda0 is not None else 1.0,
                                self._alpha if self._alpha is not None else 1.0,
                                self._fnorm]])
        out, st = self._ctx.sed_prologue(self._pars, self._opthin, not self._hasalpha,
                                         self._wavenorm, want_peak=False)
        _native.raise_for_status(st)
        self._normfac, xm, kap, x0 = out[0, 0], out[0, 1], out[0, 2], out[0, 3]
        self._hcokt = h * c / (k * self._T)
        self._xnorm = self._hcokt / self._wavenorm
        if self._hasalpha:
            self._xmerge = xm
            self._kappa = kap
        if not self._opthin:
            self._x0 = x0

    # ---- properties (modified_blackbody.py:339-388) --------------------------
    @property
    def T(self):
        """Temperature / (1+z) in K"""
        return self._T

    @property
    def beta(self):
        return self._beta

    @property
    def lambda0(self):
        """lambda_0 (1+z) in microns, None if optically thin"""
        return None if self._opthin else self._lambda0

    @property
    def alpha(self):
        return self._alpha if self._hasalpha else None

    @property
    def fnorm(self):
        """Normalisation flux at wavenorm in mJy"""
        return self._fnorm

    @property
    def wavenorm(self):
        return self._wavenorm

    @property
    def has_alpha(self):
        return self._hasalpha

    @property
    def optically_thin(self):
        return self._opthin

    @property
    def wavemerge(self):
        """Wavelength (um) where the power law takes over; None without alpha"""
        if not self._hasalpha:
            return None
        return self._hcokt / self._xmerge

    def __repr__(self):
        return ("modified_blackbody({:.2g}, {:.2g}, {}, {}, {:.2g}, wavenorm={:.2g}, "
                "noalpha={}, opthin={})").format(
                    self._T, self._beta,
                    "None" if self._opthin else "{:.2g}".format(self._lambda0),
                    "None" if not self._hasalpha else "{:.2g}".format(self._alpha),
                    self._fnorm, self._wavenorm, not self._hasalpha, self._opthin)

    def __str__(self):
        parts = ["T: {:.2g}".format(self._T), "beta: {:.2g}".format(self._beta)]
        if not self._opthin:
            parts.append("lambda0: {:.2g}".format(self._lambda0))
        if self._hasalpha:
            parts.append("alpha: {:.2g}".format(self._alpha))
        parts += ["fnorm: {:.2g}".format(self._fnorm), "wavenorm: {:.2g}".format(self._wavenorm)]
        return "modified_blackbody(" + " ".join(parts) + ")"

    # ---- evaluation ----------------------------------------------------------
    def f_nu(self, freq):
        """f_nu in mJy at frequencies in GHz (modified_blackbody.py:441-491)."""
        fr = np.atleast_1d(np.asarray(freq, dtype=np.float64))
        shape = fr.shape
        out, st = self._ctx.sed_eval(self._pars, self._opthin, not self._hasalpha,
                                     self._wavenorm, fr.reshape(-1))
        _native.raise_for_status(st)
        return out[0].reshape(shape)

    _f_nu_c = f_nu        # modified_blackbody.py:493-533: same kernel here

    def __call__(self, wave):
        """f_nu in mJy at wavelengths in microns (modified_blackbody.py:535-554).
        A scalar wavelength gives a length-1 array, as the reference does."""
        if _is_iterable(wave):
            return self.f_nu(um_to_GHz / np.asanyarray(wave, dtype=np.float64))
        return self.f_nu(um_to_GHz / float(wave))

    def max_wave(self):
        """Wavelength of maximum f_nu in microns (modified_blackbody.py:581-637)."""
        out, st = self._ctx.sed_prologue(self._pars, self._opthin, not self._hasalpha,
                                         self._wavenorm, want_peak=True)
        _native.raise_for_status(st)
        return float(out[0, 5])

    def freq_integrate(self, minwave, maxwave):
        """Integral of f_nu over [minwave, maxwave] microns, in erg/s/cm^2
        (modified_blackbody.py:639-674)."""
        minwave, maxwave = float(minwave), float(maxwave)
        if minwave <= 0.0:
            raise ValueError("Minimum wavelength must be > 0.0")
        if minwave > maxwave:
            minwave, maxwave = maxwave, minwave
        out, st = self._ctx.sed_integrate(self._pars, self._opthin, not self._hasalpha,
                                          self._wavenorm, um_to_GHz / maxwave, um_to_GHz / minwave)
        _native.raise_for_status(st)
        return 1e-17 * float(out[0])

    def __getstate__(self):
        d = dict(self.__dict__)
        d["_ctx"] = None
        return d

    def __setstate__(self, d):
        self.__dict__.update(d)
        self._ctx = _native.default_context()
