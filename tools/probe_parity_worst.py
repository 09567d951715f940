"""Where the parity tests' largest errors come from: worst rows of the wide-range test
per model variant (band flux and lnL against the oracle)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import mbb_emcee_amd as mbb
from oracle import oracle as O
O.build()
for opthin, noalpha in [(False, False), (False, True), (True, False), (True, True)]:
    rng = np.random.RandomState(4242)
    n = 600
    pars = np.column_stack([np.exp(rng.uniform(np.log(3), np.log(200), n)), rng.uniform(0.0, 4.5, n),
                            np.exp(rng.uniform(np.log(5), np.log(3000), n)),
                            np.exp(rng.uniform(np.log(0.1), np.log(10), n)),
                            np.exp(rng.uniform(np.log(0.01), np.log(1000), n))])
    pars[:8, 1] = 0.0
    names = ["PACS_100um", "SPIRE_250um", "SPIRE_500um", "SCUBA2_850um", "GISMO_2mm"]
    like = mbb.likelihood(opthin=opthin, noalpha=noalpha, response=True)
    flux = np.array([30.0, 60.0, 25.0, 6.0, 0.5]); unc = 0.1 * flux + 0.2
    like.set_phot(names, flux, unc)
    like.set_gaussian_prior("lambda_peak", 120.0, 40.0)
    has_g, gm, gs = [0] * 6, [0.0] * 6, [1.0] * 6
    has_g[5], gm[5], gs[5] = 1, 120.0, 40.0
    bands = [(r.wavelength, r._sedmult, r._normfac) for r in like._responses]
    orc = O.OracleLikelihood(flux, unc, bands=bands, opthin=opthin, noalpha=noalpha, wavenorm=500.0,
                             lowlim=like.lowlims, has_uplim=[int(b) for b in like.has_uplims],
                             uplim=like.uplims, has_gprior=has_g, gprior_mean=gm, gprior_sigma=gs)
    ref, rflux = orc(pars, nthreads=4, return_flux=True)
    got = like(pars); gflux = like.model_flux(pars)
    fin = np.isfinite(ref)
    ok = fin[:, None] & (rflux > 1e-280)
    rel = np.where(ok, np.abs(gflux / np.where(ok, rflux, 1) - 1.0), 0)
    lerr = np.where(fin, np.abs(got - ref) / np.maximum(1, np.abs(ref)), 0)
    print("opthin=%d noalpha=%d: max flux rel %.2e, max lnL err %.2e" % (opthin, noalpha, rel.max(), lerr.max()))
    # without the lambda_peak prior
    like.set_gaussian_prior("lambda_peak", 120.0, 40.0)
    for idx in np.argsort(rel.max(axis=1))[-3:][::-1]:
        sc, st = like.context.sed_prologue(pars[idx], opthin, noalpha, 500.0, want_peak=True)
        print("   flux row %d pars %s rel %s  xmerge %.6g" % (idx, np.array2string(pars[idx], precision=4), np.array2string(rel[idx], precision=2), sc[0, 1]))
    for idx in np.argsort(lerr)[-3:][::-1]:
        print("   lnL row %d pars %s err %.2e lnL %.6g  flux rel max %.2e" % (idx, np.array2string(pars[idx], precision=4), lerr[idx], ref[idx], rel[idx].max()))
