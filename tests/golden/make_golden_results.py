#!/usr/bin/env python3
"""Golden fixtures for the chain post-processing from the *reference itself*.

TEST INFRASTRUCTURE.  Runs only in the build container (reference mounted read-only at
/root/reference).  It imports the reference's `mbb_emcee/results.py` and `mbb_emcee/mbb_fit.py`
unmodified, on top of the import shim of make_golden.py, with stand-ins for
the three third-party modules those two files import and this image lacks:

  emcee              `EnsembleSampler(nwalkers, dim, lnpostfn, threads=)`: a holder for `chain` and
                     `lnprobability`.  Nothing here samples: the chain is made below, seeded, with
                     numpy, and handed to the reference's own `mbb_fitter` (mbb_fit.py:80-81) so that
                     `mbb_results.process_fit` (results.py:107-181) runs as written.
  astropy.units      `Quantity(value, unit)`, `.to(unit)`, `.value`, and the two units the code
                     touches, `Mpc` and `cm` (results.py:95-98, :656, :778).  The one number the
                     stand-in supplies is the conversion 1 Mpc = 3.0856775814913673e24 cm
                     (astropy's: 1 pc = 648000/pi au, 1 au = 149 597 870 700 m).
  astropy.cosmology  never evaluated: `lumdist` is passed explicitly (results.py:90-98), so
                     `mbb_results.cosmology` (results.py:221-226) is not reached.

Writes only numbers to tests/golden/results.npz: the chain, and per model variant
`compute_peaklambda` (results.py:570-581), `compute_lir` (:627-674), `compute_dustmass` (:746-801)
and `_predict_flux` (:895-944) for two passbands and two wavelengths.

Reference behaviour the fixtures are arranged around (recorded in DESIGN.md section 2):
  * `_map_chain` (results.py:534-568) re-uses the previous step's value when
    `numpy.allclose(prev, curr)` -- rtol 1e-5 -- so two *different* consecutive steps closer than that
    get the same output.  The chain below repeats steps exactly (rejected moves) or moves them by
    percents, never in between; `max_close_but_unequal` in the fixture records that no such pair exists.
  * `compute_peaklambda` calls its inner function without the fit's model flags (results.py:574-581:
    `peaklambda_inner(step, opthin=False, noalpha=False)` mapped with no keyword arguments), so the peak of
    a chain from an optically thin fit is computed with the optically THICK model, lambda0 and all.  The
    fixture holds what the reference returns; `peaklambda_own_model` holds `max_wave()` of the fit's own
    model for the same chain (modified_blackbody.py:581-637), which is what the likelihood's lambda_peak
    prior uses (likelihood.py:748-750).
  * `_predict_flux` builds its SED without the fit's `wavenorm` (results.py:934-937: the
    modified_blackbody default of 500 um).  The fit here is normalised at 500 um, where both agree.

Usage:  python tests/golden/make_golden_results.py          (about 40 s: scipy quad per chain entry)
"""
import importlib
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as G          # noqa: E402

MPC_IN_CM = 3.0856775814913673e24

NWALK, NSTEP = 32, 16
REDSHIFT, LUMDIST_MPC = 2.3, 18700.0
BANDS = ["PACS_100um", "PACS_160um", "SPIRE_250um", "SPIRE_350um", "SPIRE_500um", "SCUBA2_850um"]
PRED_BANDS = ["SPIRE_250um", "SCUBA2_450um"]          # one fitted band, one that is not
PRED_WAVES = [70.0, 1100.0]


def install_results_shim():
    emcee = types.ModuleType("emcee")

    class EnsembleSampler(object):
        def __init__(self, nwalkers, dim, lnpostfn, threads=1, **kw):
            self.k, self.dim, self.lnprobfn = nwalkers, dim, lnpostfn
            self.chain = np.empty((nwalkers, 0, dim))
            self.lnprobability = np.empty((nwalkers, 0))
    emcee.EnsembleSampler = EnsembleSampler

    units = types.ModuleType("astropy.units")

    class Unit(object):
        def __init__(self, name, in_cm):
            self.name, self.in_cm = name, in_cm

    class Quantity(object):
        def __init__(self, value, unit):
            self.value, self.unit = float(value), unit

        def to(self, unit):
            return Quantity(self.value * self.unit.in_cm / unit.in_cm, unit)
    units.Quantity, units.Mpc, units.cm = Quantity, Unit("Mpc", MPC_IN_CM), Unit("cm", 1.0)

    cosmology = types.ModuleType("astropy.cosmology")
    astropy = sys.modules["astropy"]
    astropy.units, astropy.cosmology = units, cosmology
    sys.modules.update({"emcee": emcee, "astropy.units": units, "astropy.cosmology": cosmology})
    fit = importlib.import_module("mbb_emcee.mbb_fit")
    res = importlib.import_module("mbb_emcee.results")
    return fit, res


def make_chain(seed):
    """A chain shaped like a sampler's: a walker either repeats its previous step exactly (a rejected
    move) or moves by a few percent."""
    rng = np.random.RandomState(seed)
    start = np.column_stack([rng.normal(14, 2.5, NWALK).clip(6, 40), rng.normal(1.8, 0.3, NWALK).clip(0.5, 3.5),
                             rng.uniform(150, 900, NWALK), rng.normal(3.0, 0.6, NWALK).clip(1.0, 6.0),
                             rng.uniform(10, 80, NWALK)])
    chain = np.empty((NWALK, NSTEP, 5))
    chain[:, 0] = start
    for t in range(1, NSTEP):
        move = rng.rand(NWALK) < 0.6
        step = chain[:, t - 1] * (1.0 + rng.normal(0, 0.03, (NWALK, 5)))
        chain[:, t] = np.where(move[:, None], step, chain[:, t - 1])
    return chain


def main():
    if not os.path.isdir(G.REFPKG):
        raise SystemExit("reference not mounted at %s" % G.REF)
    G.build_fnu()
    mb, rs, lk = G.install_shim()
    G.kat_gate(mb, rs)
    fitmod, resmod = install_results_shim()
    out = {"redshift": REDSHIFT, "lumdist_mpc": LUMDIST_MPC, "mpc_in_cm": MPC_IN_CM,
           "bands": np.array(BANDS), "pred_bands": np.array(PRED_BANDS), "pred_waves": np.array(PRED_WAVES),
           "kappa": 2.64, "kappa_wave": 125.0, "lir_range": np.array([8.0, 1000.0])}
    for seed, (nm, opthin, noalpha) in enumerate((("thin_walpha", True, False), ("thick_walpha", False, False),
                                                  ("thick_noalpha", False, True), ("thin_noalpha", True, True))):
        chain = make_chain(20260404 + seed)
        d = np.abs(np.diff(chain, axis=1))
        moved = d.max(axis=2) > 0
        close = np.all(d <= 1e-8 + 1e-5 * np.abs(chain[:, 1:]), axis=2)
        assert not np.any(moved & close)           # no pair the reference would wrongly treat as a repeat
        fit = fitmod.mbb_fitter(nwalkers=NWALK, response=True, noalpha=noalpha, opthin=opthin)
        truth = np.array([14.0, 1.8, 400.0, 3.0, 40.0])
        fit.like.set_phot(BANDS, np.ones(len(BANDS)), np.ones(len(BANDS)))
        fit.like._set_sed(truth)
        flux = np.array([float(r(fit.like._sed)) for r in fit.like._responses])
        fit.like.set_phot(BANDS, flux, 0.1 * flux + 1.0)
        fit.sampler.chain = chain
        fit.sampler.lnprobability = np.array([[fit.like(p) for p in w] for w in chain])
        res = resmod.mbb_results(fit=fit, redshift=REDSHIFT, lumdist=LUMDIST_MPC)
        res.compute_peaklambda()
        res.compute_lir()
        res.compute_dustmass()
        k = nm + "/"
        out[k + "chain"] = chain
        out[k + "lnprobability"] = fit.sampler.lnprobability
        out[k + "data_flux"] = flux
        out[k + "max_close_but_unequal"] = float(np.sum(moved & close))
        out[k + "peaklambda"] = np.asarray(res.peaklambda, dtype=np.float64)
        out[k + "peaklambda_own_model"] = np.array(
            [[mb.modified_blackbody(*p, opthin=opthin, noalpha=noalpha).max_wave() for p in w] for w in chain])
        out[k + "lir"] = np.asarray(res.lir, dtype=np.float64)
        out[k + "dustmass"] = np.asarray(res.dustmass, dtype=np.float64)
        for b in PRED_BANDS:
            if b not in res._responsewheel:
                res._responsewheel.add_special(b)
            out[k + "predict/" + b] = np.asarray(res._predict_flux(b), dtype=np.float64)
        for wv in PRED_WAVES:
            out[k + "predict/%g" % wv] = np.asarray(res._predict_flux(wv), dtype=np.float64)
        out[k + "best_fit_lnprob"] = float(res._best_fit[1])
        print("  %s: lambda_peak %.1f-%.1f um, L_IR %.3g-%.3g e12 Lsun, M_d %.3g-%.3g e8 Msun"
              % (nm, res.peaklambda.min(), res.peaklambda.max(), res.lir.min(), res.lir.max(),
                 res.dustmass.min(), res.dustmass.max()))
    np.savez_compressed(os.path.join(HERE, "results.npz"), **out)
    print("  results.npz %d B" % os.path.getsize(os.path.join(HERE, "results.npz")))


if __name__ == "__main__":
    main()
