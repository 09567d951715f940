"""Fit driver: owns the likelihood and the ensemble sampler.

Host-side mirror of the reference's ``mbb_fitter``
(reference mbb_emcee/mbb_fit.py:13-563): same constructor keywords, the same
forwarding setters, the same limit-respecting initial positions and the same
burn-in + main-chain ``run``.  The sampler evaluates each half-step of the
ensemble in one kernel launch through ``likelihood.__call__`` on an (n, 5) array.
"""
from __future__ import print_function

import numpy as np

from .ensemble import EnsembleSampler
from .likelihood import likelihood

__all__ = ["mbb_fitter"]


class mbb_fitter(object):
    """Modified-blackbody MCMC fit."""

    # mbb_fit.py:17-24
    _param_order = {'t': 0, 't/(1+z)': 0, 'beta': 1, 'lambda0': 2,
                    'lambda0*(1+z)': 2, 'lambda_0': 2, 'lambda_0*(1+z)': 2,
                    'alpha': 3, 'fnorm': 4, 'f500': 4, 'lambda_peak': 5, 'peaklam': 5}
    _parnames = np.array(['T/(1+z)', 'Beta', 'Lambda0*(1+z)', 'Alpha', 'Fnorm'])

    def __init__(self, nwalkers=250, photfile=None, covfile=None, covextn=0,
                 response=False, responsefile=None, responsedir=None, wavenorm=500.0,
                 noalpha=False, opthin=False, nthreads=1, device=None, seed=None,
                 sampler="device"):
        """Keywords as mbb_fit.py:26-72.  nthreads is accepted and ignored: the
        walkers of a half-step are evaluated together on the GPU.
        sampler="device" (the default): the whole stretch-move step runs on the GPU
        (DeviceEnsembleSampler; a fit of 50 + 250 steps of 250 walkers takes milliseconds,
        bench.py `fit_wall_s`); sampler="native": host stretch move, one launch per half-step
        through likelihood.__call__ -- the path an external sampler takes;
        sampler="emcee" uses emcee.EnsembleSampler when that package is installed
        (vectorised when it supports it, else through like.map)."""
        self._noalpha = noalpha
        self._opthin = opthin
        self._wavenorm = float(wavenorm)
        self._nwalkers = int(nwalkers)
        self._nthreads = int(nthreads)
        self.like = likelihood(photfile=photfile, covfile=covfile, covextn=covextn,
                               wavenorm=wavenorm, noalpha=noalpha, opthin=opthin,
                               response=response, responsefile=responsefile,
                               responsedir=responsedir, device=device)
        if sampler == "device":
            from .device_sampler import DeviceEnsembleSampler
            self.sampler = DeviceEnsembleSampler(self._nwalkers, 5, self.like, seed=seed)
        elif sampler == "emcee":
            import emcee
            try:
                self.sampler = emcee.EnsembleSampler(self._nwalkers, 5, self.like,
                                                     vectorize=True)
            except TypeError:
                self.sampler = emcee.EnsembleSampler(self._nwalkers, 5, self.like,
                                                     pool=self.like)
        else:
            self.sampler = EnsembleSampler(self._nwalkers, 5, self.like,
                                           threads=self._nthreads, vectorize=True, seed=seed)
        self._random = np.random.RandomState(seed) if seed is not None else np.random
        self._sampled = False
        self._fixed = [False, False, False, False, False]

    # ---- read-only views (mbb_fit.py:87-125) are attached below the class -----

    # ---- data (mbb_fit.py:127-185) --------------------------------------------
    def read_data(self, photfile, covfile=None, covextn=0, responsefile=None,
                  responsedir=None):
        if responsefile is not None:
            self.like.read_responses(responsefile, responsedir=responsedir)
        self.like.read_phot(photfile)
        if covfile is not None:
            self.like.read_cov(covfile, extn=covextn)

    def set_data(self, wave, flux, flux_unc, covmatrix=None):
        self.like.set_phot(wave, flux, flux_unc)
        if covmatrix is not None:
            self.like.set_cov(covmatrix)

    # ---- fixed parameters (mbb_fit.py:187-219); the limit / prior calls of
    # mbb_fit.py:221-360 are forwarded to the likelihood, see _FORWARDED below ----
    def _pidx(self, param):
        return self._param_order[param.lower()] if isinstance(param, str) else int(param)

    def fix_param(self, param):
        """Hold a parameter (index or name) at its initial value."""
        self._fixed[self._pidx(param)] = True

    def unfix_param(self, param):
        self._fixed[self._pidx(param)] = False

    # ---- initial positions (behaviour of mbb_fit.py:362-479) ---------------------
    def generate_initial_values(self, initvals, initsigma):
        """Starting positions [nwalkers, 5]: a Gaussian ball of widths ``initsigma`` around
        ``initvals`` in which every walker respects the parameter limits (the peak-wavelength
        limit is not looked at).  A requested centre that lies outside a parameter's limits is moved
        inside by two sigma, or to the middle of the allowed range when that range is narrower than
        four sigma; a fixed parameter gets no scatter and must itself lie inside its limits."""
        centre = np.array(initvals, dtype=np.float64)
        sigma = np.array(initsigma, dtype=np.float64)
        if centre.shape != (5,):
            raise ValueError("Initial values not expected length")
        if sigma.shape != (5,):
            raise ValueError("Initial sigma values not expected length")
        lo = np.array([self.lowlim(i) for i in range(5)], dtype=np.float64)
        hi = np.array([self.uplim(i) if self.has_uplim(i) else np.inf for i in range(5)], dtype=np.float64)
        fixed = np.asarray(self._fixed, dtype=bool)
        below, above = centre < lo, centre > hi
        stuck = fixed & (below | above)
        if stuck.any():
            raise ValueError("Some fixed parameters outside limits: {:s}".format(', '.join(self._parnames[stuck])))
        width = hi - lo
        crossed = (below | above) & (width <= 0)
        if crossed.any():
            raise ValueError("Limits on parameter {:d} cross".format(int(np.nonzero(crossed)[0][0])))
        # where the ball is centred: as asked for, or two sigma inside the limit that was violated,
        # or mid-range when the range cannot hold that
        narrow = np.isfinite(width) & (2.0 * sigma >= width)
        centre = np.where(below, np.where(narrow, lo + 0.5 * np.where(np.isfinite(width), width, 0.0), lo + 2.0 * sigma), centre)
        centre = np.where(above, np.where(narrow, lo + 0.5 * np.where(np.isfinite(width), width, 0.0), hi - 2.0 * sigma), centre)
        scatter = np.where(fixed, 0.0, sigma)
        p0 = centre + scatter * self._random.randn(self._nwalkers, 5)
        # one rejection loop over the whole array: redraw the entries that fell outside
        for attempt in range(101):
            out = (p0 < lo) | (p0 > hi)
            nout = int(out.sum())
            if nout == 0:
                return p0
            rows, cols = np.nonzero(out)
            p0[rows, cols] = centre[cols] + scatter[cols] * self._random.randn(nout)
        worst = int(np.argmax(out.sum(axis=0)))
        raise Exception("Too many iterations initializing param {:d}: {:d} of {:d} walkers still outside "
                        "[{:g}, {:g}] after 100 redraws".format(worst, int(out[:, worst].sum()), self._nwalkers,
                                                                lo[worst], hi[worst]))

    # ---- run (mbb_fit.py:481-563) ------------------------------------------------
    def run(self, nburn, nsteps, p0, verbose=False):
        """Burn in for nburn steps, reset, then sample nsteps steps per walker."""
        if not self.like.data_read:
            raise Exception("Data not read, needed to do fit")
        if verbose:
            print("Starting fit")
            if self.response_integrate:
                print("  Using response integration")
        p0 = np.asarray(p0, dtype=np.float64)
        for i in range(5):
            if (i == 2 and self._opthin) or (i == 3 and self._noalpha):
                continue
            if self.has_uplim(i) and p0[:, i].max() > self.uplim(i):
                raise ValueError("Upper limit initial value violation for "
                                 "{:s}".format(self._parnames[i]))
            if p0[:, i].min() < self.lowlim(i):
                raise ValueError("Lower limit initial value violation for "
                                 "{:s}".format(self._parnames[i]))

        self.sampler.reset()
        self._sampled = False
        if nburn <= 0:
            raise ValueError("Invalid (non-positive) number of burn in steps: {:d}".format(nburn))
        if verbose:
            print("  Doing burn in with {:d} steps".format(nburn))
        pos, prob, rstate = self.sampler.run_mcmc(p0, nburn)[:3]

        self.sampler.reset()
        if nsteps <= 0:
            raise ValueError("Invalid (non-positive) number of main chain steps: "
                             "{:d}".format(nsteps))
        if verbose:
            print("  Doing main chain with {:d} steps".format(nsteps))
        self.sampler.run_mcmc(pos, nsteps, rstate0=rstate)
        self._sampled = True

        if verbose:
            print("  Fit complete")
            print("   Mean acceptance fraction:", np.mean(self.sampler.acceptance_fraction))
            try:
                acor = self.sampler.acor
                print("   Autocorrelation time: ")
                print("    Number of burn in steps ({:d}) should be larger "
                      "than these".format(nburn))
                print("\tT:        {:f}".format(acor[0]))
                print("\tbeta:     {:f}".format(acor[1]))
                if not self._opthin:
                    print("\tlambda0:  {:f}".format(acor[2]))
                if not self._noalpha:
                    print("\talpha:    {:f}".format(acor[3]))
                print("\tfnorm:    {:f}".format(acor[4]))
            except Exception:
                pass


def _view(attr, doc):
    return property(lambda self: getattr(self, attr), doc=doc)


for _name, _attr, _doc in (("noalpha", "_noalpha", "Not using the blue side power law?"),
                           ("opthin", "_opthin", "Assuming an optically thin model?"),
                           ("wavenorm", "_wavenorm", "Normalisation wavelength [um]"),
                           ("nwalkers", "_nwalkers", "Number of walkers"),
                           ("nthreads", "_nthreads", "Accepted for compatibility, unused"),
                           ("sampled", "_sampled", "Has the distribution been sampled?"),
                           ("fixed", "_fixed", "Which parameters are held fixed")):
    setattr(mbb_fitter, _name, _view(_attr, _doc))
mbb_fitter.response_integrate = property(lambda self: self.like.response_integrate,
                                         doc="Is passband integration in use?")


def _forward(name):
    def call(self, *args):
        return getattr(self.like, name)(*args)
    call.__name__ = name
    call.__doc__ = "Forwards to likelihood.%s (param: index or name, incl. 'lambda_peak')." % name
    return call


# mbb_fit.py:221-360: the fitter exposes the likelihood's limit and prior calls
_FORWARDED = ("set_lowlim", "lowlim", "set_uplim", "has_uplim", "uplim",
              "set_gaussian_prior", "has_gaussian_prior", "get_gaussian_prior")
for _name in _FORWARDED:
    setattr(mbb_fitter, _name, _forward(_name))
