"""ctypes binding of libmbb_hip.so (include/mbb_hip.h).

There is no CPU fallback: if the HIP library is missing, or no MI355X is
visible, creating a Context raises.  The library is built in-tree by
``python -m mbb_emcee_amd.build`` (hipcc, gfx950).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmbb_hip.so")

ROW_OK, ROW_BELOW_LOWLIM, ROW_BAD_ALPHA, ROW_BAD_BETA, ROW_NOCONV, ROW_NONFINITE = 0, 1, 2, 3, 6, 7

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_vp = C.c_void_p

# name -> (restype, argtypes); mirrors include/mbb_hip.h one to one
SIGNATURES = {
    "mbb_last_error": (C.c_char_p, []),
    "mbb_device_count": (C.c_int, []),
    "mbb_ctx_create": (C.c_int, [C.c_int, C.POINTER(_vp)]),
    "mbb_ctx_destroy": (None, [_vp]),
    "mbb_set_model": (C.c_int, [_vp, C.c_int, C.c_int, C.c_double]),
    "mbb_set_bands": (C.c_int, [_vp, _dp, _dp, _ip, C.c_int]),
    "mbb_set_data": (C.c_int, [_vp, _dp, _dp, C.c_int, C.c_int]),
    "mbb_set_data_multi": (C.c_int, [_vp, _dp, _dp, C.c_int, C.c_int]),
    "mbb_set_limits": (C.c_int, [_vp, _dp, _ip, _dp]),
    "mbb_set_gpriors": (C.c_int, [_vp, _ip, _dp, _dp]),
    "mbb_lnlike_batch": (C.c_int, [_vp, _dp, C.c_int, _dp, _ip, _dp]),
    "mbb_boundary_buffers": (C.c_int, [_vp, C.c_int, C.POINTER(_vp), C.POINTER(_vp), C.POINTER(_vp)]),
    "mbb_lnlike_call": (C.c_int, [_vp, C.c_int]),
    "mbb_boundary_generation": (_vp, [_vp]),
    "mbb_lnlike_batch_device": (C.c_int, [_vp, _vp, C.c_int, _vp, _vp, _vp]),
    "mbb_lnlike_repeat_device": (C.c_int, [_vp, _vp, C.c_int, _vp, _vp, C.c_int]),
    "mbb_roof_probe": (C.c_int, [_vp, _dp, C.c_int, _dp, _dp, _dp]),
    "mbb_sampler_create": (C.c_int, [_vp, C.c_int, C.c_ulonglong, C.POINTER(_vp)]),
    "mbb_sampler_destroy": (C.c_int, [_vp, _vp]),
    "mbb_sampler_reset": (C.c_int, [_vp, _vp]),
    "mbb_sampler_set_state": (C.c_int, [_vp, _vp, _dp, _dp]),
    "mbb_sampler_run": (C.c_int, [_vp, _vp, C.c_int, C.c_double, _dp, _dp, _dp, _dp, _dp]),
    "mbb_sampler_advance_async": (C.c_int, [_vp, _vp, C.c_int, C.c_double]),
    "mbb_sampler_advance_timed": (C.c_int, [_vp, _vp, C.c_int, C.c_double, C.POINTER(C.c_double), C.POINTER(C.c_float)]),
    "mbb_sed_prologue_batch": (C.c_int, [_vp, _dp, C.c_int, C.c_int, C.c_int, C.c_double,
                                         C.c_int, _dp, _ip]),
    "mbb_sed_eval_batch": (C.c_int, [_vp, _dp, C.c_int, C.c_int, C.c_int, C.c_double, _dp,
                                     C.c_int, _dp, _ip]),
    "mbb_sed_integrate_batch": (C.c_int, [_vp, _dp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double,
                                          C.c_double, _dp, _ip]),
    "mbb_fnu_eval": (C.c_int, [_vp, C.c_int, C.c_int, _dp, C.c_int] + [C.c_double] * 7 + [_dp]),
    "mbb_malloc": (C.c_int, [_vp, C.c_size_t, C.POINTER(_vp)]),
    "mbb_free": (C.c_int, [_vp, _vp]),
    "mbb_memcpy_h2d": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    "mbb_memcpy_d2h": (C.c_int, [_vp, _vp, _vp, C.c_size_t]),
    "mbb_sync": (C.c_int, [_vp]),
    "mbb_stream": (_vp, [_vp]),
    "mbb_event_create": (C.c_int, [_vp, C.POINTER(_vp)]),
    "mbb_event_record": (C.c_int, [_vp, _vp]),
    "mbb_event_elapsed_ms": (C.c_int, [_vp, _vp, _vp, C.POINTER(C.c_float)]),
    "mbb_event_destroy": (C.c_int, [_vp, _vp]),
    "mbb_set_option": (C.c_int, [_vp, C.c_char_p, C.c_long]),
    "mbb_get_info": (C.c_int, [_vp, C.c_char_p, C.POINTER(C.c_long)]),
    "mbb_xchg_open": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int, C.c_char_p]),
    "mbb_xchg_connect": (C.c_int, [_vp, C.c_char_p]),
    "mbb_xchg_close": (C.c_int, [_vp]),
    "mbb_comm_unique_id": (C.c_int, [C.c_char_p]),
    "mbb_comm_init": (C.c_int, [_vp, C.c_int, C.c_int, C.c_char_p]),
    "mbb_comm_destroy": (C.c_int, [_vp]),
    "mbb_allgather_f64": (C.c_int, [_vp, _vp, _vp, C.c_int]),
    "mbb_lnlike_allgather_device": (C.c_int, [_vp, _vp, C.c_int, _vp, _vp, _vp]),
    "mbb_lnlike_allgather": (C.c_int, [_vp, _dp, C.c_int, _dp, _ip]),
}


class NativeError(RuntimeError):
    pass


_lib = None


def load():
    """Load libmbb_hip.so; raise if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise NativeError(
                "%s not found: build it with `python -m mbb_emcee_amd.build` "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)        # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


def _check(rc):
    if rc != 0:
        msg = load().mbb_last_error()
        raise NativeError("mbb_hip error %d: %s" % (rc, msg.decode() if msg else "?"))


def _d(a):
    return a.ctypes.data_as(_dp)


def _i(a):
    return a.ctypes.data_as(_ip)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class DeviceBuffer(object):
    """A device allocation owned by a Context (for callers that keep data resident)."""

    def __init__(self, ctx, nbytes):
        self.ctx = ctx
        self.nbytes = int(nbytes)
        p = _vp()
        _check(ctx.lib.mbb_malloc(ctx.h, self.nbytes, C.byref(p)))
        self.ptr = p

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        _check(self.ctx.lib.mbb_memcpy_h2d(self.ctx.h, self.ptr, arr.ctypes.data_as(_vp), arr.nbytes))

    def download(self, dtype, count):
        out = np.empty(count, dtype=dtype)
        assert out.nbytes <= self.nbytes
        _check(self.ctx.lib.mbb_memcpy_d2h(self.ctx.h, out.ctypes.data_as(_vp), self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr is not None and self.ctx.h:
            self.ctx.lib.mbb_free(self.ctx.h, self.ptr)
        self.ptr = None


class Context(object):
    """One device + one stream + the constant block of one likelihood."""

    def __init__(self, device=None):
        self.lib = load()
        if device is None:
            device = int(os.environ.get("MBB_DEVICE", os.environ.get("LOCAL_RANK", "0")))
            ndev = self.lib.mbb_device_count()
            if ndev > 0:
                device %= ndev
        h = _vp()
        _check(self.lib.mbb_ctx_create(int(device), C.byref(h)))
        self.h = h
        self.device = int(device)
        # the same entry point with untyped pointer arguments (see lnlike_batch)
        self._lnlike_batch_raw = C.CFUNCTYPE(C.c_int, _vp, _vp, C.c_int, _vp, _vp, _vp)(
            ("mbb_lnlike_batch", self.lib))
        self._lnlike_call_raw = C.CFUNCTYPE(C.c_int, _vp, C.c_int)(("mbb_lnlike_call", self.lib))

    def close(self):
        if getattr(self, "h", None):
            self.lib.mbb_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- configuration -----------------------------------------------------
    def set_model(self, opthin, noalpha, wavenorm):
        _check(self.lib.mbb_set_model(self.h, int(bool(opthin)), int(bool(noalpha)), float(wavenorm)))

    def set_bands(self, freq, weight, offsets):
        freq = _f64(freq); weight = _f64(weight)
        offsets = np.ascontiguousarray(offsets, dtype=np.int32)
        assert freq.size == weight.size == offsets[-1]
        _check(self.lib.mbb_set_bands(self.h, _d(freq), _d(weight), _i(offsets), offsets.size - 1))

    def set_data(self, flux, ivar=None, invcov=None):
        flux = _f64(flux)
        if invcov is not None:
            w = _f64(invcov)
            assert w.shape == (flux.size, flux.size)
            _check(self.lib.mbb_set_data(self.h, _d(flux), _d(w), flux.size, 1))
        else:
            w = _f64(ivar)
            assert w.size == flux.size
            _check(self.lib.mbb_set_data(self.h, _d(flux), _d(w), flux.size, 0))

    def set_data_multi(self, flux, ivar):
        flux = _f64(flux); ivar = _f64(ivar)
        assert flux.ndim == 2 and flux.shape == ivar.shape
        _check(self.lib.mbb_set_data_multi(self.h, _d(flux), _d(ivar), flux.shape[1], flux.shape[0]))

    def set_limits(self, lowlim, has_uplim, uplim):
        ll = _f64(lowlim); hu = np.ascontiguousarray(has_uplim, dtype=np.int32); ul = _f64(uplim)
        assert ll.size == 5 and hu.size == 6 and ul.size == 6
        _check(self.lib.mbb_set_limits(self.h, _d(ll), _i(hu), _d(ul)))

    def set_gpriors(self, has, mean, ivar):
        hs = np.ascontiguousarray(has, dtype=np.int32); mn = _f64(mean); iv = _f64(ivar)
        assert hs.size == 6 and mn.size == 6 and iv.size == 6
        _check(self.lib.mbb_set_gpriors(self.h, _i(hs), _d(mn), _d(iv)))

    def set_option(self, name, value):
        _check(self.lib.mbb_set_option(self.h, name.encode(), int(value)))

    def info(self, name):
        v = C.c_long()
        _check(self.lib.mbb_get_info(self.h, name.encode(), C.byref(v)))
        return v.value

    # ---- hot path ------------------------------------------------------------
    def lnlike_batch(self, pars, want_flux=False):
        # this is called once per half-step by host-driven samplers: raw addresses instead
        # of ctypes pointer objects (about 1 us apiece)
        p = pars if (type(pars) is np.ndarray and pars.dtype == np.float64
                     and pars.flags.c_contiguous) else _f64(pars)
        n = p.size // 5
        if p.size != 5 * n:
            raise ValueError("pars is not of expected length 5")
        lnl = np.empty(n)
        st = np.empty(n, dtype=np.int32)          # every entry is written by the library
        fl = np.empty((n, self.info("nb"))) if want_flux else None
        rc = self._lnlike_batch_raw(self.h, p.__array_interface__["data"][0], n,
                                    lnl.__array_interface__["data"][0],
                                    st.__array_interface__["data"][0],
                                    fl.__array_interface__["data"][0] if want_flux else None)
        if rc:
            _check(rc)
        return (lnl, st, fl) if want_flux else (lnl, st)

    def boundary_views(self, nmax):
        """numpy views of the blocks the boundary call works on (mbb_boundary_buffers): `rows` [cap, 5] -- write
        parameter rows straight into it (device memory behind the BAR where there is one) -- and `lnl` [cap],
        `status` [cap], pinned memory the kernel writes.  Valid until ANY call on the context needs more rows than
        they hold (the native side then frees them): `gen`, a one-element view of the context's generation word
        (mbb_boundary_generation), holds `gen0` exactly as long as they are -- compare before writing through them."""
        cap = 256
        while cap < nmax:
            cap *= 2
        pin, pout, pst = _vp(), _vp(), _vp()
        _check(self.lib.mbb_boundary_buffers(self.h, cap, C.byref(pin), C.byref(pout), C.byref(pst)))
        rows = np.ctypeslib.as_array(C.cast(pin, _dp), shape=(cap, 5))
        lnl = np.ctypeslib.as_array(C.cast(pout, _dp), shape=(cap,))
        st = np.ctypeslib.as_array(C.cast(pst, _ip), shape=(cap,))
        gen = np.ctypeslib.as_array(C.cast(self.lib.mbb_boundary_generation(self.h), C.POINTER(C.c_uint64)), shape=(1,))
        return cap, rows, lnl, st, gen, int(gen[0])

    def lnlike_batch_device(self, d_pars, n, d_lnl, d_status=None, d_flux=None):
        _check(self.lib.mbb_lnlike_batch_device(
            self.h, d_pars.ptr if hasattr(d_pars, "ptr") else d_pars, int(n),
            d_lnl.ptr if hasattr(d_lnl, "ptr") else d_lnl,
            (d_status.ptr if hasattr(d_status, "ptr") else d_status),
            (d_flux.ptr if hasattr(d_flux, "ptr") else d_flux)))

    def lnlike_repeat_device(self, d_pars, n, d_lnl, d_status, reps):
        _check(self.lib.mbb_lnlike_repeat_device(
            self.h, d_pars.ptr, int(n), d_lnl.ptr,
            d_status.ptr if d_status is not None else None, int(reps)))

    def roof_probe(self, pars, reps=20):
        """(seconds, lane_slots, clock MHz) of the sample-arithmetic-only kernel (measurement)."""
        p = _f64(pars).reshape(5)
        sec, slots, mhz = C.c_double(), C.c_double(), C.c_double()
        _check(self.lib.mbb_roof_probe(self.h, _d(p), int(reps), C.byref(sec), C.byref(slots), C.byref(mhz)))
        return sec.value, slots.value, mhz.value

    # ---- SED level -------------------------------------------------------------
    def sed_prologue(self, pars, opthin, noalpha, wavenorm, want_peak=False):
        p = _f64(pars).reshape(-1, 5)
        n = p.shape[0]
        out = np.empty((n, 6))
        st = np.zeros(n, dtype=np.int32)
        _check(self.lib.mbb_sed_prologue_batch(self.h, _d(p), n, int(bool(opthin)),
                                               int(bool(noalpha)), float(wavenorm),
                                               int(bool(want_peak)), _d(out), _i(st)))
        return out, st

    def sed_eval(self, pars, opthin, noalpha, wavenorm, freq):
        p = _f64(pars).reshape(-1, 5)
        f = _f64(freq).reshape(-1)
        n, m = p.shape[0], f.size
        out = np.empty((n, m))
        st = np.zeros(n, dtype=np.int32)
        for i0 in range(0, n, 65535):
            i1 = min(n, i0 + 65535)
            _check(self.lib.mbb_sed_eval_batch(self.h, _d(p[i0:i1]), i1 - i0, int(bool(opthin)),
                                               int(bool(noalpha)), float(wavenorm), _d(f), m,
                                               _d(out[i0:i1]), _i(st[i0:i1])))
        return out, st

    def sed_integrate(self, pars, opthin, noalpha, wavenorm, numin, numax):
        p = _f64(pars).reshape(-1, 5)
        n = p.shape[0]
        out = np.empty(n)
        st = np.zeros(n, dtype=np.int32)
        _check(self.lib.mbb_sed_integrate_batch(self.h, _d(p), n, int(bool(opthin)), int(bool(noalpha)),
                                                float(wavenorm), float(numin), float(numax),
                                                _d(out), _i(st)))
        return out, st

    def fnu_eval(self, opthin, noalpha, freq, T, beta, x0, alpha, normfac, xmerge, kappa):
        f = _f64(freq).reshape(-1)
        out = np.empty_like(f)
        _check(self.lib.mbb_fnu_eval(self.h, int(bool(opthin)), int(bool(noalpha)), _d(f), f.size,
                                     float(T), float(beta), float(x0), float(alpha),
                                     float(normfac), float(xmerge), float(kappa), _d(out)))
        return out

    # ---- plumbing --------------------------------------------------------------
    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    def sync(self):
        _check(self.lib.mbb_sync(self.h))

    def event(self):
        e = _vp()
        _check(self.lib.mbb_event_create(self.h, C.byref(e)))
        return e

    def record(self, ev):
        _check(self.lib.mbb_event_record(self.h, ev))

    def elapsed_ms(self, start, stop):
        ms = C.c_float()
        _check(self.lib.mbb_event_elapsed_ms(self.h, start, stop, C.byref(ms)))
        return ms.value

    def event_destroy(self, ev):
        _check(self.lib.mbb_event_destroy(self.h, ev))

    # ---- multi-GPU ---------------------------------------------------------------
    def comm_unique_id(self):
        buf = C.create_string_buffer(128)
        _check(self.lib.mbb_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, nranks, rank, uid):
        assert len(uid) == 128
        _check(self.lib.mbb_comm_init(self.h, int(nranks), int(rank), uid))

    def comm_destroy(self):
        _check(self.lib.mbb_comm_destroy(self.h))

    def xchg_open(self, nranks, rank, max_rows):
        """-> this rank's 64-byte handle (to be handed to every peer)"""
        buf = C.create_string_buffer(64)
        _check(self.lib.mbb_xchg_open(self.h, int(nranks), int(rank), int(max_rows), buf))
        return buf.raw

    def xchg_connect(self, handles):
        blob = b"".join(handles)
        _check(self.lib.mbb_xchg_connect(self.h, blob))

    def xchg_close(self):
        _check(self.lib.mbb_xchg_close(self.h))

    def lnlike_allgather_device(self, d_pars, n, d_lnl, d_status, d_all):
        _check(self.lib.mbb_lnlike_allgather_device(
            self.h, d_pars.ptr, int(n), d_lnl.ptr,
            d_status.ptr if d_status is not None else None, d_all.ptr))

    def lnlike_allgather(self, pars, world):
        """This rank's rows (host, [n, 5]) -> (every rank's lnprob [world * n] rank-major, this rank's row
        status [n]): kernel + one ncclAllGather + one stream wait inside one native call."""
        p = pars if (type(pars) is np.ndarray and pars.dtype == np.float64 and pars.flags.c_contiguous) else _f64(pars)
        n = p.size // 5
        full = np.empty(int(world) * n)
        st = np.empty(n, dtype=np.int32)
        _check(self.lib.mbb_lnlike_allgather(self.h, _d(p), n, _d(full), _i(st)))
        return full, st

    def allgather_f64(self, d_send, d_recv, count):
        _check(self.lib.mbb_allgather_f64(
            self.h, d_send.ptr if hasattr(d_send, "ptr") else d_send,
            d_recv.ptr if hasattr(d_recv, "ptr") else d_recv, int(count)))


_default_ctx = None


def default_context():
    """Process-wide context used by the scalar API (modified_blackbody objects)."""
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context()
    return _default_ctx


def raise_for_status(status):
    """Map per-row status codes onto the reference's exceptions
    (modified_blackbody.py:219-224, :294-316)."""
    if status.size == 0 or status.max() < 2:                    # the usual case, one pass
        return
    bad = status[(status >= 2) & (status != ROW_NONFINITE)]     # NaN in -> NaN out, no raise
    if bad.size:
        code = int(bad[0])
        msg = {ROW_BAD_ALPHA: "alpha must be positive",
               ROW_BAD_BETA: "beta must be non-negative",
               ROW_NOCONV: "couldn't find alpha merge point / SED peak"}.get(code, "bad row")
        raise ValueError(msg)
