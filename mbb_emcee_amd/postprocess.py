"""Derived quantities over a whole chain, batched on the MI355X.

The reference computes these one chain step at a time in ``mbb_results``
(reference mbb_emcee/results.py:534-801): peak wavelength (``max_wave`` per
step), L_IR (``freq_integrate`` = scipy quad per step) and dust mass, skipping
steps that repeat the previous one.  Here every step of every walker is one row
of a batched kernel call, so no de-duplication is needed.  Only the arithmetic
is provided; the results object, its HDF5 layout and the cosmology
(astropy) stay out of scope -- pass the luminosity distance in.

Parity: PINNED.  Every function here is checked against what the reference's own
``mbb_results`` returns for a 32 x 16 x 5 chain of each model variant
(tests/golden/results.npz, made by tests/golden/make_golden_results.py from the
imported ``results.py``): ``compute_peaklambda``, ``compute_lir``,
``compute_dustmass``, ``_predict_flux`` for passbands and wavelengths.  Two things
the reference does that a caller should know about:

* ``compute_peaklambda`` maps ``peaklambda_inner(step, opthin=False, noalpha=False)``
  without keyword arguments (results.py:574-581), so the peak of a chain from an
  optically THIN fit is computed with the optically thick model.  ``peak_wavelength``
  uses the fit's own model -- what the likelihood's lambda_peak prior uses
  (likelihood.py:748-750) -- unless asked for ``model="reference"``.
* ``_map_chain`` re-uses the previous step's value when ``numpy.allclose(prev, curr)``
  (rtol 1e-5; results.py:556-562), so distinct steps closer than that share a value
  there.  Here every chain entry is computed.
"""
import numpy as np

from . import _native
from .modified_blackbody import um_to_GHz

__all__ = ["peak_wavelength", "freq_integral", "lir", "dustmass", "predict_flux"]


def _rows(chain):
    c = np.asarray(chain, dtype=np.float64)
    if c.shape[-1] != 5:
        raise ValueError("chain must end in the 5 parameters")
    return c.reshape(-1, 5), c.shape[:-1]


def peak_wavelength(like, chain, model="fit"):
    """Observer-frame wavelength of peak f_nu [um] for every chain entry
    (results.py:570-581 -> modified_blackbody.max_wave, modified_blackbody.py:581-637).
    ``model="fit"``: the fit's own model; ``model="reference"``: optically thick with
    alpha whatever the fit's, which is what ``mbb_results.compute_peaklambda`` returns
    (see the module docstring)."""
    if model not in ("fit", "reference"):
        raise ValueError("model must be 'fit' or 'reference'")
    rows, shape = _rows(chain)
    opthin, noalpha = (like.opthin, like.noalpha) if model == "fit" else (False, False)
    out, st = like.context.sed_prologue(rows, opthin, noalpha, like.wavenorm,
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


def predict_flux(like, chain, spec, wavenorm=None):
    """Predicted flux density [mJy] for every chain entry (results.py:895-944,
    ``mbb_results._predict_flux``): ``spec`` is a wavelength in um (the SED there), the name
    of a passband of the fit's filter wheel (the band flux through that response,
    response.py:544-576), or a list mixing both -- the result then has a trailing axis of
    that length.  All entries of the chain go through one batched call per kind.

    The reference builds its SED here without passing the fit's ``wavenorm`` on
    (results.py:931-934: the modified_blackbody default, 500 um, is used whatever the
    fit was normalised at).  ``wavenorm=None`` uses the fit's own normalisation
    wavelength, which is what the chain's fnorm refers to; pass 500.0 to reproduce the
    reference literally for a fit with another wavenorm."""
    rows, shape = _rows(chain)
    single = isinstance(spec, str) or np.isscalar(spec)
    specs = [spec] if single else list(spec)
    wn = like.wavenorm if wavenorm is None else float(wavenorm)
    out = np.empty((rows.shape[0], len(specs)))
    names = [(i, str(s)) for i, s in enumerate(specs) if isinstance(s, str)]
    waves = [(i, float(s)) for i, s in enumerate(specs) if not isinstance(s, str)]
    for _, wv in waves:
        if wv <= 0:
            raise ValueError("Invalid wavelength {:f}".format(wv))
    if names:
        if not like.response_integrate:
            raise Exception("Asked for response integration, but no response functions "
                            "available from original fit")
        for _, nm in names:
            if nm not in like._responsewheel:
                raise ValueError("Do not have response function matching {:s}".format(nm))
        # the fused kernel's band-flux output for a one-off set of bands: a private
        # likelihood on the same wheel and device, with dummy data
        from .likelihood import likelihood
        tmp = likelihood(opthin=like.opthin, noalpha=like.noalpha, wavenorm=wn, device=like._device)
        tmp._responsewheel = like._responsewheel
        tmp._response_integrate = True
        tmp._lowlim = np.full(5, -np.inf)          # a chain entry is never gated here
        tmp.set_phot([nm for _, nm in names], np.ones(len(names)), np.ones(len(names)))
        fl = np.empty((rows.shape[0], len(names)))
        for i0 in range(0, rows.shape[0], 1 << 20):
            fl[i0:i0 + (1 << 20)] = tmp.model_flux(rows[i0:i0 + (1 << 20)])
        for k, (i, _) in enumerate(names):
            out[:, i] = fl[:, k]
    if waves:
        sed, st = like.context.sed_eval(rows, like.opthin, like.noalpha, wn,
                                        um_to_GHz / np.array([wv for _, wv in waves]))
        _native.raise_for_status(st)
        for k, (i, _) in enumerate(waves):
            out[:, i] = sed[:, k]
    return out[:, 0].reshape(shape) if single else out.reshape(shape + (len(specs),))
