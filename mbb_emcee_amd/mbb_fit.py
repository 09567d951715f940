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
                 sampler="native"):
        """Keywords as mbb_fit.py:26-72.  nthreads is accepted and ignored: the
        walkers of a half-step are evaluated together on the GPU.
        sampler="native": host stretch move, one launch per half-step;
        sampler="device": the whole stretch-move step runs on the GPU
        (DeviceEnsembleSampler); sampler="emcee" uses emcee.EnsembleSampler when
        that package is installed (vectorised when it supports it, else through
        like.map)."""
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

    # ---- initial positions (mbb_fit.py:362-479) ---------------------------------
    def generate_initial_values(self, initvals, initsigma):
        """nwalkers x 5 starting positions: Gaussian balls around initvals that
        respect the parameter limits; fixed parameters get zero scatter."""
        if len(initvals) != 5:
            raise ValueError("Initial values not expected length")
        if len(initsigma) != 5:
            raise ValueError("Initial sigma values not expected length")

        outside = [False] * 5
        for i, val in enumerate(initvals):
            if val < self.lowlim(i):
                outside[i] = True
            elif self.has_uplim(i) and val > self.uplim(i):
                outside[i] = True
        fixed_and_outside = np.logical_and(self._fixed, outside)
        if fixed_and_outside.any():
            bad = ', '.join(self._parnames[fixed_and_outside.nonzero()[0]])
            raise ValueError("Some fixed parameters outside limits: {:s}".format(bad))

        # centre of each ball: the user's value, pulled inside the limits by two
        # sigma, or to the middle of a narrow range
        centre = np.zeros(5)
        for i in range(5):
            if not outside[i]:
                centre[i] = initvals[i]
            elif self.has_uplim(i):
                par_range = self.uplim(i) - self.lowlim(i)
                if par_range <= 0:
                    raise ValueError("Limits on parameter {:d} cross".format(i))
                if 2.0 * initsigma[i] >= par_range:
                    centre[i] = self.lowlim(i) + 0.5 * par_range
                elif initvals[i] < self.lowlim(i):
                    centre[i] = self.lowlim(i) + 2 * initsigma[i]
                else:
                    centre[i] = self.uplim(i) - 2 * initsigma[i]
            else:
                centre[i] = self.lowlim(i) + 2 * initsigma[i]

        p0 = np.zeros((self._nwalkers, 5))
        for i in range(5):
            if self._fixed[i]:
                p0[:, i] = centre[i]
                continue
            lo = self.lowlim(i)
            hi = self.uplim(i) if self.has_uplim(i) else np.inf
            vec = initsigma[i] * self._random.randn(self._nwalkers) + centre[i]
            bad = np.nonzero((vec > hi) | (vec < lo))[0]
            iters = 0
            while len(bad) > 0:
                vec[bad] = initsigma[i] * self._random.randn(len(bad)) + centre[i]
                iters += 1
                bad = np.nonzero((vec > hi) | (vec < lo))[0]
                if iters > 100:
                    raise Exception("Too many iterations initializing param {:d}".format(i))
            p0[:, i] = vec
        return p0

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
