#!/usr/bin/env python3
"""Command-line fit: photometry file in, chain out.

A compact harness with the reference CLI's main flags and defaults
(reference mbb_emcee/run_mbb_emcee.py:66-193: 250 walkers, 50 burn-in and 250
main steps, initial values (10, 2, 2500, 4, 40) with scatter (2, 0.2, 100, 0.3,
5)).  The reference's results object / HDF5 output and the cosmology-dependent
quantities are out of scope: the chain is written as a NumPy .npz.

    python -m mbb_emcee_amd.run_mbb_emcee phot.txt out.npz --opthin -v
"""
from __future__ import print_function

import argparse

import numpy as np

from . import mbb_fitter, postprocess

NAMES = ["T", "Beta", "Lambda0", "Alpha", "Fnorm"]


def build_parser():
    p = argparse.ArgumentParser(description="Fit a modified blackbody to photometry with an "
                                            "ensemble MCMC running on an MI355X.")
    p.add_argument("photfile", help="text file: wavelength [um] (or passband name with "
                                    "--response), flux [mJy], uncertainty [mJy]")
    p.add_argument("outfile", help="output .npz")
    p.add_argument("-b", "--burn", type=int, default=50)
    p.add_argument("-n", "--nwalkers", type=int, default=250)
    p.add_argument("-N", "--nsteps", type=int, default=250)
    p.add_argument("--noalpha", action="store_true")
    p.add_argument("--opthin", action="store_true")
    p.add_argument("-r", "--response", action="store_true")
    p.add_argument("--responsefile", default=None)
    p.add_argument("--responsedir", default=None)
    p.add_argument("-w", "--wavenorm", type=float, default=500.0)
    p.add_argument("-t", "--threads", type=int, default=1, help="accepted and ignored")
    p.add_argument("--sampler", choices=["device", "native"], default="device")
    p.add_argument("--seed", type=int, default=None)
    p.add_argument("--get_peaklambda", action="store_true")
    p.add_argument("-v", "--verbose", action="store_true")
    for nm, dflt in zip(NAMES, (10.0, 2.0, 2500.0, 4.0, 40.0)):
        p.add_argument("--init" + nm, type=float, default=dflt)
        p.add_argument("--fix" + nm, action="store_true")
        p.add_argument("--low" + nm, type=float, default=None)
        p.add_argument("--up" + nm, type=float, default=None)
        p.add_argument("--prior" + nm, type=float, nargs=2, default=None, metavar=("MEAN", "SIGMA"))
    p.add_argument("--upLambdaPeak", type=float, default=None)
    p.add_argument("--priorLambdaPeak", type=float, nargs=2, default=None)
    return p


def main(argv=None):
    a = build_parser().parse_args(argv)
    if a.nwalkers <= 0:
        raise ValueError("Invalid (non-positive) nwalkers: %d" % a.nwalkers)
    fit = mbb_fitter(nwalkers=a.nwalkers, photfile=a.photfile, wavenorm=a.wavenorm,
                     noalpha=a.noalpha, opthin=a.opthin, nthreads=a.threads, response=a.response,
                     responsefile=a.responsefile, responsedir=a.responsedir, sampler=a.sampler,
                     seed=a.seed)
    unused = {"Lambda0": a.opthin, "Alpha": a.noalpha}
    for nm in NAMES:                                  # run_mbb_emcee.py:226-283
        if getattr(a, "fix" + nm) or (nm == "Alpha" and a.noalpha):
            fit.fix_param(nm.lower())
        if unused.get(nm):
            continue
        if getattr(a, "low" + nm) is not None:
            fit.set_lowlim(nm.lower(), getattr(a, "low" + nm))
        if getattr(a, "up" + nm) is not None:
            fit.set_uplim(nm.lower(), getattr(a, "up" + nm))
        pr = getattr(a, "prior" + nm)
        if pr is not None:
            fit.set_gaussian_prior(nm.lower(), pr[0], pr[1])
    if a.upLambdaPeak is not None:
        fit.set_uplim("lambda_peak", a.upLambdaPeak)
    if a.priorLambdaPeak is not None:
        fit.set_gaussian_prior("lambda_peak", a.priorLambdaPeak[0], a.priorLambdaPeak[1])

    p0init = np.array([getattr(a, "init" + nm) for nm in NAMES])
    p0 = fit.generate_initial_values(p0init, np.array([2, 0.2, 100, 0.3, 5.0]))   # :285-291
    fit.run(a.burn, a.nsteps, p0, verbose=a.verbose)
    chain, lnp = fit.sampler.chain, fit.sampler.lnprobability
    out = dict(chain=chain, lnprobability=lnp, acceptance_fraction=fit.sampler.acceptance_fraction,
               parnames=np.array(NAMES), noalpha=a.noalpha, opthin=a.opthin, wavenorm=a.wavenorm,
               data_wave=fit.like.data_wave, data_flux=fit.like.data_flux,
               data_flux_unc=fit.like.data_flux_unc)
    if a.get_peaklambda:
        out["peaklambda"] = postprocess.peak_wavelength(fit.like, chain)
    np.savez_compressed(a.outfile, **out)
    if a.verbose:
        flat = chain.reshape(-1, 5)
        print("Fit results (median, +/- 68.3%):")
        for i, nm in enumerate(NAMES):
            if (i == 2 and a.opthin) or (i == 3 and a.noalpha):
                continue
            lo, med, hi = np.percentile(flat[:, i], [15.85, 50.0, 84.15])
            print("  {:8s} {:10.4g} +{:.3g} -{:.3g}".format(nm, med, hi - med, med - lo))
        best = np.unravel_index(np.argmax(lnp), lnp.shape)
        print("  best lnP {:.3f} at".format(lnp[best]), chain[best])
        print("Saved", a.outfile)
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
