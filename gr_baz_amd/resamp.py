"""ctypes binding of include/baz_resamp_hip.h (the fractional resampler engine; no CPU fallback)."""
from __future__ import annotations

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libbaz_resamp_hip.so")

SYMBOLS = ["baz_resamp_create", "baz_resamp_destroy", "baz_resamp_forecast", "baz_resamp_process",
           "baz_resamp_process_device", "baz_resamp_process2", "baz_resamp_process2_device", "baz_resamp_set_mu", "baz_resamp_set_ratio", "baz_resamp_set_ratio_rational",
           "baz_resamp_set_ratio_ppb", "baz_resamp_adjust", "baz_resamp_mu", "baz_resamp_ratio",
           "baz_resamp_phase_exact", "baz_resamp_taps", "baz_resamp_default_taps", "baz_resamp_set_taps", "baz_resamp_set_stream", "baz_resamp_sync",
           "baz_resamp_strerror"]
NTAPS, NSTEPS = 8, 128

_vp = ctypes.c_void_p
_f32p = ctypes.POINTER(ctypes.c_float)
_u64 = ctypes.c_uint64
_lib = None


class ResampError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        super().__init__("%s failed: %s (%d)" % (where, lib().baz_resamp_strerror(int(code)).decode(), code))


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("gr_baz_amd: %s is missing - run `python -m gr_baz_amd.build` (no CPU fallback)" % LIB_PATH)
    L = ctypes.CDLL(LIB_PATH)
    L.baz_resamp_create.restype = ctypes.c_int
    L.baz_resamp_create.argtypes = [ctypes.POINTER(_vp), ctypes.c_uint32, ctypes.c_double, ctypes.c_double, _u64, _u64,
                                    ctypes.c_int]
    L.baz_resamp_destroy.restype = None
    L.baz_resamp_destroy.argtypes = [_vp]
    L.baz_resamp_forecast.restype = ctypes.c_int64
    L.baz_resamp_forecast.argtypes = [_vp, ctypes.c_uint32]
    L.baz_resamp_process.restype = ctypes.c_int64
    L.baz_resamp_process.argtypes = [_vp, _f32p, _u64, _u64, _f32p, _u64, ctypes.c_uint32, ctypes.POINTER(_u64)]
    L.baz_resamp_process_device.restype = ctypes.c_int64
    L.baz_resamp_process_device.argtypes = [_vp, _vp, _u64, _u64, _vp, _u64, ctypes.c_uint32, ctypes.POINTER(_u64)]
    L.baz_resamp_process2.restype = ctypes.c_int64
    L.baz_resamp_process2.argtypes = [_vp, _f32p, _u64, _u64, _f32p, _f32p, _u64, ctypes.c_uint32, ctypes.POINTER(_u64)]
    L.baz_resamp_process2_device.restype = ctypes.c_int64
    L.baz_resamp_process2_device.argtypes = [_vp, _vp, _u64, _u64, _vp, _vp, _u64, ctypes.c_uint32, ctypes.POINTER(_u64)]
    for nm in ("set_mu", "set_ratio", "adjust"):
        f = getattr(L, "baz_resamp_" + nm)
        f.restype = ctypes.c_int
        f.argtypes = [_vp, ctypes.c_double]
    L.baz_resamp_set_ratio_rational.restype = ctypes.c_int
    L.baz_resamp_set_ratio_rational.argtypes = [_vp, _u64, _u64]
    L.baz_resamp_set_ratio_ppb.restype = ctypes.c_int
    L.baz_resamp_set_ratio_ppb.argtypes = [_vp, ctypes.c_long, ctypes.c_double]
    L.baz_resamp_mu.restype = ctypes.c_double
    L.baz_resamp_mu.argtypes = [_vp]
    L.baz_resamp_ratio.restype = ctypes.c_double
    L.baz_resamp_ratio.argtypes = [_vp]
    L.baz_resamp_phase_exact.restype = ctypes.c_int
    L.baz_resamp_phase_exact.argtypes = [_vp]
    L.baz_resamp_taps.restype = _f32p
    L.baz_resamp_taps.argtypes = [_vp]
    L.baz_resamp_default_taps.restype = None
    L.baz_resamp_default_taps.argtypes = [_f32p]
    L.baz_resamp_set_taps.restype = ctypes.c_int
    L.baz_resamp_set_taps.argtypes = [_vp, _f32p]
    L.baz_resamp_set_stream.restype = ctypes.c_int
    L.baz_resamp_set_stream.argtypes = [_vp, _vp]
    L.baz_resamp_sync.restype = ctypes.c_int
    L.baz_resamp_sync.argtypes = [_vp]
    L.baz_resamp_strerror.restype = ctypes.c_char_p
    L.baz_resamp_strerror.argtypes = [ctypes.c_int]
    _lib = L
    return L


class Resampler:
    """gr::baz::fractional_resampler_cc for `nstreams` lock-stepped streams (one shared phase accumulator)."""

    def __init__(self, phase_shift, resamp_ratio, resamp_ratio_num=0, resamp_ratio_denom=0, nstreams=1, device_id=-1):
        h = _vp()
        r = lib().baz_resamp_create(ctypes.byref(h), nstreams, phase_shift, resamp_ratio, resamp_ratio_num,
                                    resamp_ratio_denom, device_id)
        if r != 0:
            raise ResampError(r, "baz_resamp_create")
        self._h = h
        self.nstreams = nstreams

    def close(self):
        if getattr(self, "_h", None):
            lib().baz_resamp_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def forecast(self, noutput):
        return int(lib().baz_resamp_forecast(self._h, noutput))

    def work(self, x, noutput, rr=None):
        """x: (n,) or (nstreams, n) complex64 host window -> (out[..., produced], consumed); state carries over.
        rr: optional (n,) float32 per-sample ratio input = the block's second input port (.cc:205-217)."""
        x = np.ascontiguousarray(x, dtype=np.complex64)
        one = x.ndim == 1
        if one:
            x = x[None, :]
        if x.shape[0] != self.nstreams:
            raise ValueError("expected %d streams" % self.nstreams)
        n = x.shape[1]
        out = np.zeros((self.nstreams, noutput), np.complex64)
        consumed = _u64(0)
        if rr is None:
            r = lib().baz_resamp_process(self._h, x.view(np.float32).ctypes.data_as(_f32p), n, n,
                                         out.view(np.float32).ctypes.data_as(_f32p), noutput, noutput, ctypes.byref(consumed))
        else:
            rr = np.ascontiguousarray(rr, dtype=np.float32)
            if rr.shape != (n,):
                raise ValueError("the ratio input needs one float per input sample")
            r = lib().baz_resamp_process2(self._h, x.view(np.float32).ctypes.data_as(_f32p), n, n, rr.ctypes.data_as(_f32p),
                                          out.view(np.float32).ctypes.data_as(_f32p), noutput, noutput, ctypes.byref(consumed))
        if r < 0:
            raise ResampError(r, "baz_resamp_process")
        out = out[:, :r]
        return (out[0] if one else out), int(consumed.value)

    def process_device(self, d_in, in_stride, ninput, d_out, out_stride, noutput):
        consumed = _u64(0)
        r = lib().baz_resamp_process_device(self._h, _vp(d_in), in_stride, ninput, _vp(d_out), out_stride, noutput,
                                            ctypes.byref(consumed))
        if r < 0:
            raise ResampError(r, "baz_resamp_process_device")
        return int(r), int(consumed.value)

    def process2_device(self, d_in, in_stride, ninput, d_ratio, d_out, out_stride, noutput):
        """Two-input branch on device buffers (synchronises the stream: the counts depend on the data)."""
        consumed = _u64(0)
        r = lib().baz_resamp_process2_device(self._h, _vp(d_in), in_stride, ninput, _vp(d_ratio), _vp(d_out), out_stride,
                                             noutput, ctypes.byref(consumed))
        if r < 0:
            raise ResampError(r, "baz_resamp_process2_device")
        return int(r), int(consumed.value)

    def _chk(self, r, where):
        if r != 0:
            raise ResampError(r, where)

    def set_mu(self, mu): self._chk(lib().baz_resamp_set_mu(self._h, mu), "baz_resamp_set_mu")
    def set_resamp_ratio(self, r): self._chk(lib().baz_resamp_set_ratio(self._h, r), "baz_resamp_set_ratio")
    def set_resamp_ratio_rational(self, n, d): self._chk(lib().baz_resamp_set_ratio_rational(self._h, n, d), "baz_resamp_set_ratio_rational")
    def set_resamp_ratio_ppb(self, whole, frac): self._chk(lib().baz_resamp_set_ratio_ppb(self._h, whole, frac), "baz_resamp_set_ratio_ppb")
    def adjust(self, d): self._chk(lib().baz_resamp_adjust(self._h, d), "baz_resamp_adjust")
    def mu(self): return lib().baz_resamp_mu(self._h)
    def resamp_ratio(self): return lib().baz_resamp_ratio(self._h)
    def phase_exact(self): return bool(lib().baz_resamp_phase_exact(self._h))
    def taps(self): return np.ctypeslib.as_array(lib().baz_resamp_taps(self._h), shape=(NSTEPS + 1, NTAPS)).copy()

    def set_taps(self, taps):
        """Installs a 129 x 8 tap table (the host's gnuradio-filter table, see include/baz_resamp_hip.h)."""
        t = np.ascontiguousarray(np.asarray(taps, dtype=np.float32).reshape(NSTEPS + 1, NTAPS))
        self._chk(lib().baz_resamp_set_taps(self._h, t.ctypes.data_as(_f32p)), "baz_resamp_set_taps")
    def set_stream(self, s): self._chk(lib().baz_resamp_set_stream(self._h, _vp(s) if s else None), "baz_resamp_set_stream")
    def sync(self): self._chk(lib().baz_resamp_sync(self._h), "baz_resamp_sync")


def default_taps():
    """The closed-form MMSE table a fresh context starts with (host arithmetic, no device needed)."""
    t = np.zeros((NSTEPS + 1, NTAPS), np.float32)
    lib().baz_resamp_default_taps(t.ctypes.data_as(_f32p))
    return t
