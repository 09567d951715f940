"""The emcee log-probability callable, evaluated on the MI355X.

Host-side mirror of the reference's ``likelihood`` class
(reference mbb_emcee/likelihood.py:17-834): same constructor keywords, same
setters/getters, same parameter order (T, beta, lambda0, alpha, fnorm).

``like(pars)`` with a length-5 vector returns a Python float exactly as the
reference does (so the object drops into ``emcee.EnsembleSampler`` as lnpostfn);
``like(pars)`` with an ``(n, 5)`` array evaluates all n walkers in one fused
kernel launch and returns ``float64[n]`` (emcee-3 ``vectorize=True``, or this
package's own sampler).  Data, limits and priors live in a device-side constant
block that is rebuilt lazily after any setter call.
"""
import copy
import ctypes as C

import numpy as np

from . import _native
try:                                    # (optional: the boundary call in C, built by mbb_emcee_amd.build.build_fastcall)
    from .build import fastcall_is_current as _fc_ok
    if not _fc_ok():                    # (built for another Python / numpy, or older than its source: not loaded)
        raise ImportError("_mbbfast is not this interpreter's")
    from . import _mbbfast
except Exception:                       # noqa -- never a reason to fail: the same call goes through numpy and ctypes
    _mbbfast = None
from .modified_blackbody import modified_blackbody, um_to_GHz
from .response import response_set, special_types, read_table

__all__ = ["likelihood"]


def _fastcall(ctx, rows, lnl, cap, gen, gen0):
    """The boundary call as one C-level callable (the `_mbbfast` extension, mbb_emcee_amd/csrc/mbb_fastcall.c), or None
    where it is not built or the native entry has been replaced by something else (a test's wrapper)."""
    raw = ctx._lnlike_call_raw
    if _mbbfast is None or not isinstance(raw, C._CFuncPtr):
        return None
    try:
        return _mbbfast.FastCall(C.cast(raw, C.c_void_p).value, ctx.h.value, rows.ctypes.data, lnl.ctypes.data, cap,
                                 gen.ctypes.data, gen0)
    except Exception:           # noqa
        return None


class likelihood(object):
    """Holds the photometry, limits and priors and evaluates ln P(pars | data)."""

    # likelihood.py:20-22
    _param_order = {'t': 0, 't/(1+z)': 0, 'beta': 1, 'lambda0': 2,
                    'lambda0*(1+z)': 2, 'lambda_0': 2, 'lambda_0*(1+z)': 2,
                    'alpha': 3, 'fnorm': 4, 'f500': 4}

    def __init__(self, photfile=None, covfile=None, covextn=0, wavenorm=500.0,
                 noalpha=False, opthin=False, response=False, responsefile=None,
                 responsedir=None, device=None):
        self._wavenorm = float(wavenorm)
        self._noalpha = bool(noalpha)
        self._opthin = bool(opthin)
        self._device = device
        self._ctx = None
        self._dirty = True
        self._fast = None      # the boundary call's buffers and views (see __call__)

        self._lowlim = np.array([1, 0.1, 1, 0.1, 1e-3])               # :73
        self._limprior_order = copy.copy(self._param_order)            # :77-79
        self._limprior_order.update({'lambda_peak': 5, 'peaklam': 5,
                                     'lambdapeak': 5, 'peak_lambda': 5})
        self._has_uplim = [False, True, False, True, False, False]     # :83-85
        inf = float("inf")
        self._uplim = np.array([inf, 20.0, inf, 20.0, inf, inf])
        self._any_gprior = False                                       # :88-92
        self._has_gprior = [False] * 6
        self._gprior_mean = np.zeros(6)
        self._gprior_sigma = np.zeros(6)
        self._gprior_ivar = np.ones(6)

        self._response_integrate = False
        if response:
            self.read_responses(responsefile, responsedir=responsedir)

        self._data_read = False
        self._has_covmatrix = False
        self._nsources = 1
        if photfile is not None:
            self.read_phot(photfile)
            if covfile is not None:
                if not isinstance(covfile, str):
                    raise TypeError("covfile must be string-like")
                self.read_cov(covfile, extn=covextn)
            self._lowlim[4] = 1e-3 * self._flux.min()                  # :110-111
        elif covfile is not None:
            raise Exception("Can't pass in covfile if no photfile")
        self._badval = float("-inf")

    # ---- data ----------------------------------------------------------------
    def read_responses(self, responsefile=None, responsedir=None):
        """Load a filter wheel and turn on passband integration (:139-156)."""
        self._responsewheel = response_set(responsefile, dir=responsedir)
        self._response_integrate = True
        self._dirty = True

    def _passbands_for(self, names):
        """The wheel's responses in the order of the photometry; a name the wheel does not hold is made on the spot when
        its second field spells one of the special passbands (box, gauss, dsb, alma, delta: likelihood.py:182-203)."""
        if not isinstance(names[0], str):
            raise ValueError("Expecting response string name")
        wheel, found = self._responsewheel, []
        for name in map(str, names):
            if name not in wheel:
                kind = name.split('_')[1].lower() if '_' in name else None
                if kind not in special_types:
                    raise ValueError("Unknown filter response {:s}".format(name))
                wheel.add_special(name)
            found.append(wheel[name])
        return found

    def set_phot(self, firstarg, flux, flux_unc):
        """Set photometry: passband names (response mode) or wavelengths [um],
        flux densities and uncertainties [mJy] (likelihood.py:158-232).  A covariance matrix set before is dropped
        (:231); nothing of the object changes when an argument is refused."""
        bands = self._passbands_for(firstarg) if self._response_integrate else None
        wave = (np.array([r.effective_wavelength for r in bands]) if bands is not None
                else np.asarray(firstarg, dtype=np.float64))
        npts = len(wave)
        if npts == 0:
            raise ValueError("No elements in wavelength vector")
        values = {}
        for what, given in (("flux", flux), ("flux_unc", flux_unc)):
            values[what] = np.asarray(given, dtype=np.float64)
            if len(values[what]) != npts:
                raise ValueError("wave not same length as " + what)
        if bands is not None:
            self._responses, self._response_names = bands, [r.name for r in bands]
        self._wave, self._ndata = wave, npts
        self._flux, self._flux_unc = values["flux"], values["flux_unc"]
        self._ivar = 1.0 / self._flux_unc ** 2
        # lambda0 three times beyond the longest wavelength says nothing the data can test (:224-229)
        if not self._has_uplim[2]:
            self._has_uplim[2], self._uplim[2] = True, 3.0 * wave.max()
        self._data_read, self._has_covmatrix, self._nsources = True, False, 1
        self._dirty = True

    def set_phot_multi(self, firstarg, flux, flux_unc):
        """Batched multi-source mode: ``flux`` and ``flux_unc`` are [nsources, ndata]
        arrays of independent SEDs observed through the same bands (``firstarg`` as
        in set_phot).  Afterwards ``like(pars)`` takes [nsources, m, 5] (or the same
        rows flattened) and returns [nsources, m]; row block g is compared with
        source g.  Diagonal uncertainties only.  Limits and priors are shared."""
        flux = np.asarray(flux, dtype=np.float64)
        flux_unc = np.asarray(flux_unc, dtype=np.float64)
        if flux.ndim != 2 or flux.shape != flux_unc.shape:
            raise ValueError("flux and flux_unc must both be [nsources, ndata]")
        self.set_phot(firstarg, flux[0], flux_unc[0])
        self._flux_multi = flux
        self._ivar_multi = 1.0 / flux_unc ** 2
        self._nsources = flux.shape[0]
        self._dirty = True

    def read_phot(self, filename):
        """Three-column text file: wavelength [um] (or passband name), flux,
        uncertainty [mJy] (likelihood.py:234-262)."""
        if not isinstance(filename, str):
            raise TypeError("filename must be string-like")
        data = read_table(filename)
        if len(data) == 0:
            raise IOError("No data read from %s" % filename)
        self.set_phot([d[0] for d in data], [d[1] for d in data], [d[2] for d in data])

    def has_response(self, name):
        return hasattr(self, '_responsewheel') and name in self._responsewheel

    def get_response(self, name):
        if not hasattr(self, '_responsewheel'):
            return None
        return self._responsewheel[name]

    @property
    def data_flux_unc(self):
        if not self._data_read:
            return None
        if self._has_covmatrix:
            return np.sqrt(np.diag(self._covmatrix))
        return self._flux_unc

    def set_cov(self, covmatrix):
        """Set the flux covariance matrix [mJy^2] (likelihood.py:330-357): square, one row per photometry point."""
        if not self._data_read:
            raise Exception("Can't set covariance matrix without photometry")
        cov = np.asarray(covmatrix, dtype=np.float64)
        if cov.ndim != 2:
            raise ValueError("Covariance matrix is not 2 dimensional")
        nrow, ncol = cov.shape
        if nrow != ncol:
            raise ValueError("Covariance matrix from is not square: %d by %d" % (nrow, ncol))
        if nrow != self._ndata:
            raise ValueError("Covariance matrix doesn't have same number of datapoints as "
                             "photometry; {0:d} vs. {1:d}".format(nrow, self._ndata))
        self._covmatrix, self._invcovmatrix = cov, np.linalg.inv(cov)
        self._has_covmatrix = self._dirty = True

    def read_cov(self, filename, extn=0):
        """Covariance matrix from a FITS file (likelihood.py:359-376); needs astropy."""
        if not self._data_read:
            raise Exception("Can't read in covaraince matrix without phot")
        try:
            import astropy.io.fits as fits
        except ImportError:
            raise ImportError("read_cov needs astropy.io.fits; use set_cov(array) instead")
        self.set_cov(fits.open(filename)[extn].data)

    # ---- limits and priors (likelihood.py:378-641) ----------------------------
    def get_paramindex(self, paramname):
        return self._param_order[paramname]

    def _pidx(self, param, table):
        return table[param.lower()] if isinstance(param, str) else int(param)

    def set_lowlim(self, param, val):
        self._lowlim[self._pidx(param, self._param_order)] = val
        self._dirty = True

    def lowlim(self, param):
        return self._lowlim[self._pidx(param, self._param_order)]

    def set_uplim(self, param, val):
        i = self._pidx(param, self._limprior_order)
        self._has_uplim[i] = True
        self._uplim[i] = val
        self._dirty = True

    def has_uplim(self, param):
        return self._has_uplim[self._pidx(param, self._limprior_order)]

    def uplim(self, param):
        i = self._pidx(param, self._limprior_order)
        return self._uplim[i] if self._has_uplim[i] else None

    def set_gaussian_prior(self, param, mean, sigma):
        i = self._pidx(param, self._limprior_order)
        self._any_gprior = True
        self._has_gprior[i] = True
        self._gprior_mean[i] = float(mean)
        self._gprior_sigma[i] = float(sigma)
        self._gprior_ivar[i] = 1.0 / (float(sigma) ** 2)
        self._dirty = True

    def has_gaussian_prior(self, param):
        return self._has_gprior[self._pidx(param, self._limprior_order)]

    def get_gaussian_prior(self, param):
        if not self._any_gprior:
            return None
        i = self._pidx(param, self._limprior_order)
        if not self._has_gprior[i]:
            return None
        return (self._gprior_mean[i], self._gprior_sigma[i])

    def _check_lowlim(self, pars):
        """True when every parameter is at or above its lower limit (:643-670)."""
        if len(pars) != 5:
            raise ValueError("pars is not of expected length 5")
        return not any(v < self._lowlim[i] for i, v in enumerate(pars))

    # ---- device state ---------------------------------------------------------
    @property
    def context(self):
        """The native context (device + stream + constant block), created lazily."""
        if self._ctx is None:
            self._ctx = _native.Context(self._device)
            self._dirty = True
        return self._ctx

    def band_tables(self):
        """(freq_GHz, weight, offsets): the quadrature uploaded to the GPU.
        Passband mode: response.quadrature() per band (response.py:572-576);
        plain photometry: one sample per wavelength, weight 1 (likelihood.py:817)."""
        if not self._data_read:
            raise Exception("Data not read")
        if self._response_integrate:
            quads = [r.quadrature() for r in self._responses]
            freq = np.concatenate([q[0] for q in quads])
            wt = np.concatenate([q[1] for q in quads])
            offs = np.concatenate(([0], np.cumsum([q[0].size for q in quads]))).astype(np.int32)
        else:
            freq = um_to_GHz / self._wave
            wt = np.ones(self._ndata)
            offs = np.arange(self._ndata + 1, dtype=np.int32)
        return freq, wt, offs

    def _sync_device(self):
        ctx = self.context
        if not self._dirty:
            return ctx
        if not self._data_read:
            raise Exception("Data not read, can't evaluate the likelihood")
        ctx.set_model(self._opthin, self._noalpha, self._wavenorm)
        ctx.set_bands(*self.band_tables())
        if self._nsources > 1:
            if self._has_covmatrix:
                raise ValueError("multi-source mode supports diagonal uncertainties only")
            ctx.set_data_multi(self._flux_multi, self._ivar_multi)
        elif self._has_covmatrix:
            ctx.set_data(self._flux, invcov=self._invcovmatrix)
        else:
            ctx.set_data(self._flux, ivar=self._ivar)
        ctx.set_limits(self._lowlim, [int(b) for b in self._has_uplim], self._uplim)
        ctx.set_gpriors([int(b) for b in self._has_gprior], self._gprior_mean, self._gprior_ivar)
        self._dirty = False
        return ctx

    # ---- evaluation -------------------------------------------------------------
    def _set_sed(self, pars):
        """SED object for a parameter vector (likelihood.py:754-768)."""
        if len(pars) != 5:
            raise ValueError("pars is not of expected length 5")
        self._sed = modified_blackbody(pars[0], pars[1], pars[2], pars[3], pars[4],
                                       wavenorm=self._wavenorm, noalpha=self._noalpha,
                                       opthin=self._opthin, context=self.context)

    def get_sed(self, pars, wave):
        """Model SED [mJy] at wavelengths [um] (likelihood.py:770-788).  pars may
        be (5,) -> [nwave] or (n, 5) -> [n, nwave]."""
        p = np.asarray(pars, dtype=np.float64)
        if p.ndim == 1:
            self._set_sed(p)
            return self._sed(wave)
        w = np.atleast_1d(np.asarray(wave, dtype=np.float64))
        out, st = self.context.sed_eval(p, self._opthin, self._noalpha, self._wavenorm,
                                        um_to_GHz / w)
        _native.raise_for_status(st)
        return out

    def model_flux(self, pars):
        """Model band fluxes [n, ndata] for (n, 5) parameter rows."""
        p = np.atleast_2d(np.asarray(pars, dtype=np.float64))
        if p.shape[1] != 5:
            raise ValueError("pars is not of expected length 5")
        lnl, st, fl = self._sync_device().lnlike_batch(p, want_flux=True)
        _native.raise_for_status(st)
        return fl

    def _fast_views(self, n):
        """Views for the boundary call of n rows: the rows of the block the kernel reads its parameters from
        (device memory behind the PCIe BAR, or pinned memory) and of the pinned block it writes lnprob to.
        One native call when the capacity has to grow, a dictionary look-up afterwards."""
        ctx = self._sync_device()
        f = self._fast
        # (f[8][0] != f[9]: some other call on the context -- model_flux of a chain, a list of rows through the general
        # path, a sharded evaluation -- needed more rows than the blocks held and the native side made them anew: the
        # views are of freed memory and must never be written through again)
        if f is None or n > f[3] or f[6] is not ctx or f[8][0] != f[9]:
            cap, rows, lnl, st, gen, gen0 = ctx.boundary_views(n)
            self._fast = f = (ctx._lnlike_call_raw, ctx.h.value, {}, cap, rows, lnl, ctx,
                              _fastcall(ctx, rows, lnl, cap, gen, gen0), gen, gen0)
        if len(f[2]) > 64:
            f[2].clear()
        v = f[2][n] = (f[4][:n], f[5][:n])
        return v

    def __call__(self, pars):
        """ln P(pars | data) including limits and priors (likelihood.py:790-834).

        (5,) -> float;  (n, 5) -> float64[n], all rows in one kernel launch.
        Rows below a lower limit give -inf."""
        # What a sampler calls once per half-step (emcee: mbb_fit.py:80-81) or once per walker: a float array
        # of rows, or one row, with nothing changed since the last call.  The rows are written straight into
        # the block the kernel reads (no copy of them anywhere else), the native call takes a row count and
        # nothing else, the result is copied out of the pinned block the kernel wrote.  Whatever is unusual --
        # first call, changed data or limits, a row the reference raises for, options that switch the host
        # path -- takes the general path below, which also does the raising.
        if not self._dirty and type(pars) is np.ndarray and pars.dtype.kind == "f" and self._nsources == 1:
            f = self._fast
            # (the same in C where the extension is built -- csrc/mbb_fastcall.c: rows in, native call, results out, ~1 us
            # less of Python per call; None for anything it leaves to the lines below)
            if f is not None and f[7] is not None:
                r = f[7](pars)
                if r is not None:
                    return r
            if f is not None and f[8][0] != f[9]:
                self._fast = f = None           # the blocks were made anew under the cached views (see _fast_views)
            if pars.ndim == 2 and pars.shape[1] == 5 and pars.shape[0] > 0:
                n = pars.shape[0]
                v = f[2].get(n) if f is not None else None
                if v is None:
                    v = self._fast_views(n)
                    f = self._fast
                v[0][...] = pars
                rc = f[0](f[1], n)
                if rc == 0:
                    return v[1].copy()
                self._fast = None
            elif pars.ndim == 1 and pars.shape[0] == 5:
                v = f[2].get(1) if f is not None else None
                if v is None:
                    v = self._fast_views(1)
                    f = self._fast
                v[0][0] = pars
                rc = f[0](f[1], 1)
                if rc == 0:
                    return float(v[1][0])
                self._fast = None
        p = np.asarray(pars, dtype=np.float64)
        if p.ndim == 1:
            if p.shape[0] != 5:
                raise ValueError("pars is not of expected length 5")
            lnl, st = self._sync_device().lnlike_batch(p[None, :])
            _native.raise_for_status(st)
            return float(lnl[0])
        if p.ndim == 3 and p.shape[2] == 5 and p.shape[0] == self._nsources:
            lnl, st = self._sync_device().lnlike_batch(p.reshape(-1, 5))
            _native.raise_for_status(st)
            return lnl.reshape(p.shape[0], p.shape[1])
        if p.ndim != 2 or p.shape[1] != 5:
            raise ValueError("pars is not of expected length 5")
        if self._nsources > 1 and p.shape[0] % self._nsources != 0:
            raise ValueError("in multi-source mode the rows must be nsources equal blocks")
        lnl, st = self._sync_device().lnlike_batch(p)
        _native.raise_for_status(st)
        return lnl

    def map(self, func, rows):
        """pool-style adaptor: ``EnsembleSampler(..., pool=like)`` makes an emcee-2 style sampler
        hand a whole half-step over at once.  When ``func`` is this object -- bare, as a bound
        ``__call__``, or inside emcee's function wrapper (attribute ``f``) without extra
        arguments -- the rows are evaluated in ONE launch.  Any other function (a lambda that adds
        a prior, a wrapper carrying ``args=``) is what the caller wants evaluated: it is applied row
        by row, as ``map`` would."""
        rows = list(rows)
        target = getattr(func, "f", func)
        extra = bool(getattr(func, "args", None)) or bool(getattr(func, "kwargs", None))
        own = target is self or (getattr(target, "__self__", None) is self and
                                 getattr(target, "__name__", "") == "__call__")
        if own and not extra:
            if not rows:
                return []
            return list(self(np.asarray(rows, dtype=np.float64)))
        return [func(r) for r in rows]

    # ---- pickling: the device context is rebuilt on demand ---------------------
    def __getstate__(self):
        d = dict(self.__dict__)
        d["_ctx"] = None
        d["_fast"] = None
        d["_dirty"] = True
        d.pop("_sed", None)
        return d

    def __setstate__(self, d):
        self.__dict__.update(d)


def _ro(attr, doc, needs=None, default=None):
    """Read-only view of an attribute, optionally gated on a flag attribute
    (the reference returns None / 0 before data are set, likelihood.py:264-397)."""
    def get(self):
        if needs is not None and not getattr(self, needs, False):
            return default
        return getattr(self, attr, default)
    return property(get, doc=doc)


for _n, _a, _d, _needs, _dflt in (
        ("wavenorm", "_wavenorm", "Normalisation wavelength [um]", None, None),
        ("noalpha", "_noalpha", "Not including a blue side power law?", None, None),
        ("opthin", "_opthin", "Assuming an optically thin model?", None, None),
        ("response_integrate", "_response_integrate", "Is passband integration in use?", None, None),
        ("nsources", "_nsources", "Number of sources (multi-source mode)", None, 1),
        ("data_read", "_data_read", "Has the photometry been set?", None, False),
        ("ndata", "_ndata", "Number of data points", "_data_read", 0),
        ("data_wave", "_wave", "Data (effective) wavelengths [um]", "_data_read", None),
        ("data_flux", "_flux", "Flux densities [mJy]", "_data_read", None),
        ("response_names", "_response_names", "Passband names of the data", None, None),
        ("has_data_covmatrix", "_has_covmatrix", "Is there a flux covariance matrix?", None, False),
        ("data_covmatrix", "_covmatrix", "Flux covariance matrix [mJy^2]", "_has_covmatrix", None),
        ("data_invcovmatrix", "_invcovmatrix", "Inverse covariance matrix", "_has_covmatrix", None),
        ("lowlims", "_lowlim", "Lower limits: T, beta, lambda0, alpha, fnorm", None, None),
        ("has_uplims", "_has_uplim", "Which upper limits are set (6th: lambda_peak)", None, None),
        ("uplims", "_uplim", "Upper limits", None, None),
        ("has_gpriors", "_has_gprior", "Which Gaussian priors are set", None, None),
        ("gprior_means", "_gprior_mean", "Gaussian prior means", None, None),
        ("gprior_sigmas", "_gprior_sigma", "Gaussian prior sigmas", None, None),
        ("gprior_ivars", "_gprior_ivar", "Gaussian prior inverse variances", None, None)):
    setattr(likelihood, _n, _ro(_a, _d, _needs, _dflt))
