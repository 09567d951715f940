"""Derived quantities over a whole chain, batched on the MI355X.

The reference computes these one chain step at a time in ``mbb_results``
(reference mbb_emcee/results.py:534-801): peak wavelength (``max_wave`` per
step), L_IR (``freq_integrate`` = scipy quad per step) and dust mass, skipping
steps that repeat the previous one.  Here every step of every walker is one row
of a batched kernel call, so no de-duplication is needed.  Only the arithmetic
is provided; the results object, its HDF5 layout and the cosmology
(astropy) stay out of scope -- pass the luminosity distance in.
"""
import numpy as np

from . import _native
from .modified_blackbody import um_to_GHz

__all__ = ["peak_wavelength", "freq_integral", "lir", "dustmass"]


def _rows(chain):
    c = np.asarray(chain, dtype=np.float64)
    if c.shape[-1] != 5:
        raise ValueError("chain must end in the 5 parameters")
    return c.reshape(-1, 5), c.shape[:-1]


def peak_wavelength(like, chain):
    """Observer-frame wavelength of peak f_nu [um] for every chain entry
    (results.py:570-581 -> modified_blackbody.max_wave)."""
    rows, shape = _rows(chain)
    out, st = like.context.sed_prologue(rows, like.opthin, like.noalpha, like.wavenorm,
                                        want_peak=True)
    _native.raise_for_status(st)
    return out[:, 5].reshape(shape)


def freq_integral(like, chain, minwave, maxwave):
    """Integral of f_nu over observer-frame [minwave, maxwave] um in erg/s/cm^2
    (modified_blackbody.py:639-674) for every chain entry."""
    minwave, maxwave = float(minwave), float(maxwave)
    if minwave <= 0 or maxwave <= 0:
        raise ValueError("wavelengths must be positive")
    if minwave > maxwave:
        minwave, maxwave = maxwave, minwave
    rows, shape = _rows(chain)
    out, st = like.context.sed_integrate(rows, like.opthin, like.noalpha, like.wavenorm,
                                         um_to_GHz / maxwave, um_to_GHz / minwave)
    _native.raise_for_status(st)
    return 1e-17 * out.reshape(shape)


def lir(like, chain, redshift, lumdist_mpc, wavemin=8.0, wavemax=1000.0):
    """L_IR in 10^12 L_sun between rest-frame wavemin and wavemax um
    (results.py:627-674; prefactor 4 pi Mpc^2 / L_sun in cgs, :661)."""
    opz = 1.0 + float(redshift)
    prefac = 3.11749657e4 * float(lumdist_mpc) ** 2
    return prefac * freq_integral(like, chain, wavemin * opz, wavemax * opz)


def dustmass(like, chain, redshift, lumdist_mpc, kappa=2.64, kappa_wave=125.0):
    """Dust mass in 10^8 M_sun (results.py:726-801): closed form in the chain
    values, evaluated on the host."""
    if kappa <= 0 or kappa_wave <= 0:
        raise ValueError("kappa and kappa_wave must be positive")
    c = np.asarray(chain, dtype=np.float64)
    opz = 1.0 + float(redshift)
    dl2 = (float(lumdist_mpc) * 3.0856775814913673e24) ** 2          # cm^2
    wavenorm_rest = like.wavenorm / opz
    nunorm_rest = 299792458e6 / wavenorm_rest                        # Hz
    temp_fac = 6.6260693e-27 * nunorm_rest / 1.38065e-16             # h nu / k   (:779)
    bnu_fac = 2 * 6.6260693e-27 * nunorm_rest ** 3 / 299792458e2 ** 2
    knu_fac = wavenorm_rest / float(kappa_wave)
    msolar8 = 1.97792e41
    T = c[..., 0] * opz
    beta = c[..., 1]
    S_nu = c[..., 4] * 1e-26
    B_nu = bnu_fac / np.expm1(temp_fac / T)
    K_nu = 10.0 * float(kappa) * knu_fac ** (-beta)
    m = dl2 * S_nu / (opz * K_nu * B_nu * msolar8)
    if not like.opthin:
        tau = (c[..., 2] / like.wavenorm) ** beta
        m = m * (-tau / np.expm1(-tau))
    return m
