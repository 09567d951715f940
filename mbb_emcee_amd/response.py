"""Instrument passbands: the quadrature grids of the likelihood hot path.

Host-side mirror of the reference's ``response`` / ``response_set``
(reference mbb_emcee/response.py:51-840).  This is the cold path: it runs once
per fit and produces, per band, the arrays the GPU kernel integrates over
(frequencies in GHz and trapezoid x transmission weights, response.py:252-332).
Plain numpy, no astropy/pkg_resources/h5py.

Deliberate departures from the reference, all in code the reference cannot
run on current numpy or that is internally inconsistent:
  Q3  alma_* passbands are built as 1-D grids (the reference makes (13,1) arrays);
  Q4  delta_<val> given in frequency units gets normwave = c/nu for every unit;
  Q5  dsb/alma grids are always in GHz; wavelength-unit specs are converted
      once instead of being re-interpreted.
"""
import math
import copy
import os
import re

import numpy as np

__all__ = ["response", "response_set", "special_types", "default_wheel"]

special_types = ["delta", "box", "gauss", "dsb", "alma"]

_C_GHZ_UM = 299792458e-3        # GHz * um
_H = 6.6260693e-34
_K = 1.3806505e-23
_RESDIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "resources")
_PKGDIR = "!package-dir!"

# The built-in filter wheel (reference resources/mbb_filterwheel.txt:2-19):
# name, curve, xtype, xunits, senstype, normtype, xnorm, normparam
default_wheel = [
    ("MIPS_24um", "MIPS_24.txt", "wave", "microns", "energy", "bb", 23.675, 10000.0),
    ("MIPS_70um", "MIPS_70.txt", "wave", "microns", "energy", "bb", 71.440, 10000.0),
    ("MIPS_160um", "MIPS_160.txt", "wave", "microns", "energy", "bb", 155.899, 10000.0),
    ("PACS_70um", "PACS_70.txt", "wave", "microns", "energy", "power", 70.0, -1.0),
    ("PACS_100um", "PACS_100.txt", "wave", "microns", "energy", "power", 100.0, -1.0),
    ("PACS_160um", "PACS_160.txt", "wave", "microns", "energy", "power", 160.0, -1.0),
    ("SPIRE_250um", "SPIRE_250.txt", "wave", "microns", "energy", "power", 250.0, -1.0),
    ("SPIRE_350um", "SPIRE_350.txt", "wave", "microns", "energy", "power", 350.0, -1.0),
    ("SPIRE_500um", "SPIRE_500.txt", "wave", "microns", "energy", "power", 500.0, -1.0),
    ("SABOCA_350um", "SABOCA_350um.txt", "wave", "microns", "energy", "flat", 350.0, 0.0),
    ("LABOCA_870um", "LABOCA_870um.txt", "wave", "microns", "energy", "flat", 870.0, 0.0),
    ("SCUBA_450um", "SCUBA_450um.txt", "wave", "microns", "energy", "flat", 450.0, 0.0),
    ("SCUBA_850um", "SCUBA_850um.txt", "wave", "microns", "energy", "flat", 850.0, 0.0),
    ("SCUBA2_450um", "SCUBA2_450um.txt", "wave", "microns", "energy", "flat", 450.0, 0.0),
    ("SCUBA2_850um", "SCUBA2_850um.txt", "wave", "microns", "energy", "flat", 850.0, 0.0),
    ("Bolocam_1.1mm", "Bolocam_1100.txt", "wave", "microns", "energy", "flat", 1100.0, 0.0),
    ("MAMBO2_1.2mm", "MAMBO2.txt", "wave", "microns", "energy", "flat", 1200.0, 0.0),
    ("GISMO_2mm", "GISMO_2000.txt", "wave", "microns", "energy", "flat", 2000.0, 0.0),
]

_curves = None


def _packaged_curve(filename):
    """Two-column transmission curve shipped with the package."""
    global _curves
    if _curves is None:
        _curves = np.load(os.path.join(_RESDIR, "passband_curves.npz"))
    if filename not in _curves.files:
        raise IOError("No packaged passband curve named {:s}".format(filename))
    arr = _curves[filename]
    return arr[:, 0].copy(), arr[:, 1].copy()


def read_table(filename):
    """Whitespace-separated text table; '#' starts a comment.  Returns rows of
    tokens converted to int, float or str (the wire format of passband curves,
    filter wheels and photometry files)."""
    rows = []
    with open(filename) as fh:
        for line in fh:
            line = line.split("#", 1)[0].strip()
            if not line:
                continue
            row = []
            for tok in line.split():
                try:
                    row.append(int(tok))
                except ValueError:
                    try:
                        row.append(float(tok))
                    except ValueError:
                        row.append(tok)
            rows.append(row)
    return rows


_WAVE_TO_UM = {"angstroms": 1e-4, "a": 1e-4, "microns": 1.0, "um": 1.0,
               "meters": 1e6, "m": 1e6}
_FREQ_TO_GHZ = {"hz": 1e-9, "mhz": 1e-3, "ghz": 1.0, "thz": 1e3}


def _to_ghz(vals, xtype, xunits):
    """Convert scalar/array in (xtype, xunits) to GHz."""
    if xtype == "wave":
        if xunits not in _WAVE_TO_UM:
            raise ValueError("Unrecognized wavelength unit {:s}".format(xunits))
        return _C_GHZ_UM / (np.asarray(vals, dtype=float) * _WAVE_TO_UM[xunits])
    if xtype == "freq":
        if xunits not in _FREQ_TO_GHZ:
            raise ValueError("Unrecognized frequency unit {:s}".format(xunits))
        return np.asarray(vals, dtype=float) * _FREQ_TO_GHZ[xunits]
    raise ValueError("Unknown unit type {:s}".format(xtype))


def response_bb(freq, temperature):
    """Unnormalised blackbody f_nu at freq [GHz] (response.py:25-48)."""
    hokt = 1e9 * _H / (_K * float(temperature))
    return freq ** 3 / np.expm1(hokt * freq)


class response(object):
    """Response of one instrument passband plus its pipeline normalisation
    convention (response.py:51-66)."""

    def __init__(self, name):
        self._name = str(name)
        self._data_read = False

    # ------------------------------------------------------------------ setup
    def setup(self, inputspec, xtype="wave", xunits="microns", senstype="energy",
              normtype="power", xnorm=250.0, normparam=-1.0, dir=None):
        """Build the passband from a text file or from a special spec
        (delta_v, box_c_w, gauss_c_fwhm, dsb_c_w_gap, alma_c); arguments as
        response.py:68-139."""
        ntyp, xtyp = normtype.lower(), xtype.lower()
        xun, styp = xunits.lower(), senstype.lower()
        if not isinstance(inputspec, str):
            raise TypeError("filename must be string-like")
        self._isdelta = False
        parts = inputspec.split("_")
        kind = parts[0].lower()
        grid_is_ghz = False
        if kind == "delta":
            if len(parts) < 2:
                raise ValueError("delta needs central frequency")
            self._setup_delta(float(parts[1]), xtyp, xun)
            return
        elif kind == "box":
            if len(parts) < 3:
                raise ValueError("box car needs 2 params in {:s}".format(inputspec))
            cent, width = float(parts[1]), float(parts[2])
            xvals = np.linspace(cent - 0.5 * width, cent + 0.5 * width, 11)   # :376-382
            resp = np.ones(11)
        elif kind == "gauss":
            if len(parts) < 3:
                raise ValueError("gaussian needs 2 params in {:s}".format(inputspec))
            cent, fwhm = float(parts[1]), float(parts[2])
            sig = fwhm / math.sqrt(8 * math.log(2))                            # :384-393
            xvals = np.linspace(cent - 3.0 * fwhm, cent + 3.0 * fwhm, 43)
            resp = np.exp(-0.5 * ((xvals - cent) / sig) ** 2)
        elif kind == "dsb":
            if len(parts) < 4:
                raise ValueError("dsb needs 3 params in {:s}".format(inputspec))
            cent, width, gap = float(parts[1]), float(parts[2]), float(parts[3])
            nodes = np.array([cent - width / 2, cent - gap / 2, cent + gap / 2, cent + width / 2])
            xvals, resp = self._two_sidebands(np.sort(_to_ghz(nodes, xtyp, xun)))  # :395-440
            grid_is_ghz = True
        elif kind == "alma":
            if len(parts) < 2:
                raise ValueError("alma needs 1 params in {:s}".format(inputspec))
            xvals, resp = self._alma_grid(float(_to_ghz(float(parts[1]), xtyp, xun)))  # :443-491
            grid_is_ghz = True
        else:
            if dir is None:
                x, r = self._read_curve(inputspec)
            elif dir == _PKGDIR:
                x, r = _packaged_curve(inputspec)
            else:
                x, r = self._read_curve(os.path.join(dir, inputspec))
            xvals, resp = x, r

        xvals = np.asarray(xvals, dtype=np.float64)
        resp = np.asarray(resp, dtype=np.float64)
        if xvals.min() <= 0:
            raise ValueError("Non-positive x value encountered")
        if resp.min() < 0:
            raise ValueError("Negative response encountered")
        if xnorm <= 0:
            raise ValueError("Non-positive xnorm")

        # wavelengths in um, frequencies in GHz (response.py:215-249)
        if xtyp == "wave":
            if xun not in _WAVE_TO_UM:
                raise ValueError("Unrecognized wavelength unit {:s}".format(xun))
            self._normwave = _WAVE_TO_UM[xun] * xnorm
            self._normfreq = _C_GHZ_UM / self._normwave
        elif xtyp == "freq":
            if xun not in _FREQ_TO_GHZ:
                raise ValueError("Unrecognized frequency unit {:s}".format(xun))
            self._normfreq = _FREQ_TO_GHZ[xun] * xnorm
            self._normwave = _C_GHZ_UM / self._normfreq
        else:
            raise ValueError("Unknown unit type {:s}".format(xtype))
        if grid_is_ghz:
            freq = xvals
            wave = _C_GHZ_UM / freq
        elif xtyp == "wave":
            wave = _WAVE_TO_UM[xun] * xvals if _WAVE_TO_UM[xun] != 1.0 else xvals
            freq = _C_GHZ_UM / wave
        else:
            freq = _FREQ_TO_GHZ[xun] * xvals if _FREQ_TO_GHZ[xun] != 1.0 else xvals
            wave = _C_GHZ_UM / freq

        order = wave.argsort()                                   # :252-256
        self._wave, self._freq = wave[order], freq[order]
        self._resp = resp[order] / resp.max()                    # :260
        self._nresp = n = len(self._resp)
        if n < 2:
            raise ValueError("A passband needs at least two samples")

        if styp == "energy":
            self._sens_energy = True
        elif styp == "counts":
            self._sens_energy = False
        else:
            raise ValueError("Unknown sensitivity type {:s}".format(senstype))

        # trapezoid weights in frequency times transmission (response.py:274-279);
        # negative because frequency descends along the arrays
        self._dnu = self._freq[1:] - self._freq[:-1]
        w = np.empty(n)
        w[:-1] = 0.5 * self._dnu
        w[-1] = 0.5 * self._dnu[-1]
        w[1:-1] += 0.5 * self._dnu[:-1]
        w *= self._resp
        if not self._sens_energy:                                # :284-289
            w *= self._freq[n // 2] / self._freq
        self._sedmult = w

        self._normtype = str(ntyp)
        if ntyp == "none":                                       # :293-301
            self._normparam = None
            self._normfac = -1.0
            eff = (self._freq * w).sum() / w.sum()
        else:
            if ntyp == "power":                                  # :304-306
                self._normparam = float(normparam)
                sed = (self._freq / self._normfreq) ** self._normparam
            elif ntyp == "flat":
                self._normparam = None
                sed = np.ones(n)
            elif ntyp == "bb":                                   # :310-319
                self._normparam = float(normparam)
                if self._normparam <= 0.0:
                    raise ValueError("Invalid (non-positive) blackbody temperature "
                                     "{:f}".format(self._normparam))
                sed = response_bb(self._freq, self._normparam) / \
                    response_bb(self._normfreq, self._normparam)
            else:
                raise ValueError("Unknown normalization type {:s}".format(normtype))
            self._normfac = 1.0 / (sed * w).sum()                # :327
            eff = (self._freq * sed * w).sum() * self._normfac   # :330
        self._effective_freq = eff
        self._effective_wave = _C_GHZ_UM / eff
        self._data_read = True

    @staticmethod
    def _read_curve(path):
        rows = read_table(path)
        if len(rows) == 0:
            raise IOError("No data read from {:s}".format(path))
        return (np.array([r[0] for r in rows], dtype=np.float64),
                np.array([r[1] for r in rows], dtype=np.float64))

    @staticmethod
    def _two_sidebands(f):
        """Two boxes f[0]..f[1] and f[2]..f[3] (GHz) with a zero-weight gap."""
        grid = np.concatenate((np.linspace(f[0], f[1], 13),
                               np.linspace(f[1] + 0.0001, f[2] - 0.0001, 3),
                               np.linspace(f[2], f[3], 13)))
        resp = np.concatenate((np.ones(13), np.zeros(3), np.ones(13)))
        return grid, resp

    @classmethod
    def _alma_grid(cls, cen_freq):
        """ALMA 2SB setup, bands 3/4/6/7/8 (response.py:472-491)."""
        bands = [(92.0, 108.0, 4.0), (125.0, 163.0, 4.0), (221.0, 265.0, 6.0),
                 (283.0, 365.0, 4.0), (385.0, 500.0, 4.0)]
        for lo, hi, if_bot in bands:
            if lo <= cen_freq <= hi:
                return cls._two_sidebands(np.array([cen_freq - if_bot - 3.75, cen_freq - if_bot,
                                                    cen_freq + if_bot, cen_freq + if_bot + 3.75]))
        raise ValueError("Unable to identify ALMA band with central freq "
                         "{:0.1f}".format(cen_freq))

    def _setup_delta(self, val, xtyp, xun):
        """Delta-function response (response.py:336-374)."""
        if val <= 0:
            raise ValueError("Non-positive value")
        self._normfreq = float(_to_ghz(val, xtyp, xun))
        self._normwave = _C_GHZ_UM / self._normfreq
        if xtyp == "wave":
            self._normwave = _WAVE_TO_UM[xun] * val
            self._normfreq = _C_GHZ_UM / self._normwave
        self._isdelta = True
        self._effective_wave = self._normwave
        self._effective_freq = _C_GHZ_UM / self._effective_wave
        self._wave = np.array([self._normwave])
        self._freq = np.array([self._effective_freq])
        self._resp = np.array([1.0])
        self._nresp = 1
        self._normtype = "delta"
        self._normparam = None
        self._normfac = 1.0
        self._sens_energy = True
        self._data_read = True

    # ------------------------------------------------------------- properties
    @property
    def name(self):
        return self._name

    @property
    def data_read(self):
        return self._data_read

    @property
    def isdelta(self):
        return self._isdelta if self._data_read else None

    @property
    def wavelength(self):
        """Wavelengths of the samples in microns (ascending)"""
        return self._wave if self._data_read else None

    @property
    def frequency(self):
        """Frequencies of the samples in GHz"""
        return self._freq if self._data_read else None

    @property
    def response(self):
        return self._resp if self._data_read else None

    @property
    def effective_wavelength(self):
        return self._effective_wave if self._data_read else None

    @property
    def effective_frequency(self):
        return self._effective_freq if self._data_read else None

    @property
    def normfac(self):
        """Normalisation value, sign flipped to be positive (response.py:536-542)"""
        return -1.0 * self._normfac if self._data_read else None

    def quadrature(self):
        """(freq_GHz[n], weight[n]) such that the band flux of an SED f is
        sum(f(freq) * weight): weight = sedmult * normfac (response.py:575-576).
        This is what is uploaded to the GPU."""
        if not self._data_read:
            raise Exception("Data not read yet, can't get response")
        if self._isdelta:
            return self._freq.copy(), np.ones(1)
        return self._freq.copy(), self._sedmult * self._normfac

    def __call__(self, fnufunc, freq=False):
        """Instrument response to an SED given as a callable of wavelength [um]
        (or of frequency [GHz] with freq=True); response.py:544-576."""
        if not self._data_read:
            raise Exception("Data not read yet, can't get response")
        if self._isdelta:
            return fnufunc(self._normfreq if freq else self._normwave)
        x = self._freq if freq else self._wave
        return (fnufunc(x) * self._sedmult).sum() * self._normfac

    def __str__(self):
        return "{0:s} lambda_eff: {1:0.1f} [um]".format(self._name, self._effective_wave)


_BUILTIN_WHEEL = {}


def _own_arrays(resp, freeze=False):
    """A copy of a passband object whose ndarray attributes are copies too (read-only ones when `freeze`)."""
    new = copy.copy(resp)
    for k, v in list(vars(new).items()):
        if isinstance(v, np.ndarray):
            w = v.copy()
            w.setflags(write=not freeze)
            setattr(new, k, w)
    return new


class response_set(object):
    """A named set of passbands -- the filter wheel (response.py:642-840)."""

    def __init__(self, inputfile=None, dir=None):
        self._responses = {}
        self.read(inputfile=inputfile, dir=dir)

    def read(self, inputfile=None, dir=None):
        """Load a filter-wheel file (8 columns: Name File Xtype Xunit Sens
        NormType XNorm NormPar); None loads the built-in wheel.  Clears what
        was loaded before (response.py:661-710)."""
        if inputfile is None:
            rows, indir = default_wheel, _PKGDIR
        else:
            if not isinstance(inputfile, str):
                raise TypeError("filename must be string-like")
            if dir is not None and not isinstance(dir, str):
                raise TypeError("dir must be string-like")
            infile = inputfile if dir is None else os.path.join(dir, inputfile)
            rows, indir = read_table(infile), dir
            if len(rows) == 0:
                raise IOError("No data read from {:s}".format(inputfile))
        self._responses.clear()
        if inputfile is None and _BUILTIN_WHEEL:
            # the built-in wheel is set up once per process (18 curves: 3 ms); every wheel gets arrays of its own, writable,
            # as in the reference (response.py:252-332) -- copies of the frozen master's, a few hundred KB
            for name, resp in _BUILTIN_WHEEL.items():
                self._responses[name] = _own_arrays(resp)
            return
        for r in rows:
            self.add(str(r[0]), str(r[1]), str(r[2]).lower(), str(r[3]).lower(),
                     str(r[4]).lower(), str(r[5]).lower(), float(r[6]), float(r[7]),
                     dir=indir)
        if inputfile is None:
            # the master keeps frozen copies of its own: what a user does to the arrays of THIS wheel (the reference's
            # pattern `rs[name].response[:] *= k` works on it) never reaches a later wheel of the process
            _BUILTIN_WHEEL.update((name, _own_arrays(resp, freeze=True)) for name, resp in self._responses.items())

    def add(self, name, spec, xtype, xunits, senstype, normtype, xnorm, normparam, dir=None):
        resp = response(name)
        resp.setup(spec, xtype=xtype, xunits=xunits, senstype=senstype, normtype=normtype,
                   xnorm=xnorm, normparam=normparam, dir=dir)
        self._responses[name] = resp

    def add_special(self, name):
        """Add name_type_values (box, gauss, dsb, alma, delta) on the fly; the
        first value may carry a unit suffix 'um' or 'ghz' (default GHz).  Energy
        sensitivity and flat normalisation (response.py:722-780)."""
        spl = name.split("_")
        if len(spl) < 2 or spl[1].lower() not in special_types:
            raise ValueError("Unknown 'special' response type in {:s}".format(name))
        kind = spl[1].lower()
        if len(spl) < 3:
            raise ValueError("Special type has no numerical spec")
        first = spl[2].lower()
        pat = re.compile(r"\d*\.\d+|\d+")
        num = pat.findall(first)
        if len(num) == 0:
            raise ValueError("Special type needs numeric specification")
        unit = pat.sub("", first)
        if unit in ("", "ghz"):
            xtype, xunit = "freq", "ghz"
        elif unit == "um":
            xtype, xunit = "wave", "microns"
        else:
            raise ValueError("Unable to understand unit specification {:s}".format(unit))
        newspec = "_".join([kind, num[0]] + spl[3:])
        resp = response(name)
        resp.setup(newspec, xtype=xtype, xunits=xunit, senstype="energy", normtype="flat",
                   xnorm=float(num[0]), normparam=0)
        self._responses[name] = resp

    def __getitem__(self, name):
        return self._responses[name]

    def keys(self):
        return self._responses.keys()

    def __contains__(self, val):
        return val in self._responses

    def items(self):
        return self._responses.items()

    def values(self):
        return self._responses.values()

    def __delitem__(self, val):
        del self._responses[val]

    def __len__(self):
        return len(self._responses)

    def __str__(self):
        return "\n".join(str(r) for r in self._responses.values())
