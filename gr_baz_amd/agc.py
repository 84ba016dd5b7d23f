"""ctypes binding of include/baz_agc_hip.h (the AGC engine; no CPU fallback)."""
from __future__ import annotations

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libbaz_agc_hip.so")

SYMBOLS = ["baz_agc_create", "baz_agc_destroy", "baz_agc_process", "baz_agc_process_device", "baz_agc_process_device_interleaved", "baz_agc_reset", "baz_agc_debug_selfcheck",
           "baz_agc_set_stream", "baz_agc_sync", "baz_agc_count", "baz_agc_strerror"]

_vp = ctypes.c_void_p
_f32p = ctypes.POINTER(ctypes.c_float)
_u64 = ctypes.c_uint64
_lib = None


class AgcError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        super().__init__("%s failed: %s (%d)" % (where, lib().baz_agc_strerror(code).decode(), code))


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("gr_baz_amd: %s is missing - run `python -m gr_baz_amd.build` (no CPU fallback)" % LIB_PATH)
    L = ctypes.CDLL(LIB_PATH)
    L.baz_agc_create.restype = ctypes.c_int
    L.baz_agc_create.argtypes = [ctypes.POINTER(_vp), ctypes.c_uint32, ctypes.c_float, ctypes.c_float,
                                 ctypes.c_float, ctypes.c_float, ctypes.c_int]
    L.baz_agc_destroy.restype = None
    L.baz_agc_destroy.argtypes = [_vp]
    L.baz_agc_process.restype = ctypes.c_int
    L.baz_agc_process.argtypes = [_vp, _f32p, _u64, _u64, _f32p, _f32p, _f32p]
    L.baz_agc_process_device.restype = ctypes.c_int
    L.baz_agc_process_device.argtypes = [_vp, _vp, _u64, _u64, _vp, _vp, _vp]
    L.baz_agc_process_device_interleaved.restype = ctypes.c_int
    L.baz_agc_process_device_interleaved.argtypes = [_vp, _vp, _u64, _u64, _vp]
    L.baz_agc_reset.restype = ctypes.c_int
    L.baz_agc_reset.argtypes = [_vp]
    L.baz_agc_set_stream.restype = ctypes.c_int
    L.baz_agc_set_stream.argtypes = [_vp, _vp]
    L.baz_agc_sync.restype = ctypes.c_int
    L.baz_agc_sync.argtypes = [_vp]
    L.baz_agc_count.restype = _u64
    L.baz_agc_count.argtypes = [_vp]
    L.baz_agc_debug_selfcheck.restype = ctypes.c_int
    L.baz_agc_debug_selfcheck.argtypes = [_vp, _vp, _vp, _u64, ctypes.POINTER(_u64)]
    L.baz_agc_strerror.restype = ctypes.c_char_p
    L.baz_agc_strerror.argtypes = [ctypes.c_int]
    _lib = L
    return L


class Agc:
    """`nstreams` independent baz_agc_cc instances (same parameters) in one context."""

    def __init__(self, rate=1e-4, reference=1.0, gain=1.0, max_gain=0.0, nstreams=1, device_id=-1):
        h = _vp()
        r = lib().baz_agc_create(ctypes.byref(h), nstreams, rate, reference, gain, max_gain, device_id)
        if r != 0:
            raise AgcError(r, "baz_agc_create")
        self._h = h
        self.nstreams = nstreams

    def close(self):
        if getattr(self, "_h", None):
            lib().baz_agc_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def work(self, x, want_env=True, want_mul=True):
        """x: (n,) or (nstreams, n) complex64 host array -> (out, env|None, mul|None), state carries over."""
        x = np.ascontiguousarray(x, dtype=np.complex64)
        shape = x.shape
        if x.ndim == 1:
            x = x[None, :]
        if x.shape[0] != self.nstreams:
            raise ValueError("expected %d streams" % self.nstreams)
        n = x.shape[1]
        out = np.zeros_like(x)
        env = np.zeros(x.shape, np.float32) if want_env else None
        mul = np.zeros(x.shape, np.float32) if want_mul else None
        r = lib().baz_agc_process(self._h, x.view(np.float32).ctypes.data_as(_f32p), n, n,
                                  out.view(np.float32).ctypes.data_as(_f32p),
                                  env.ctypes.data_as(_f32p) if want_env else None,
                                  mul.ctypes.data_as(_f32p) if want_mul else None)
        if r < 0:
            raise AgcError(r, "baz_agc_process")
        rs = lambda a: None if a is None else a.reshape(shape)
        return rs(out), rs(env), rs(mul)

    def process_device(self, d_in, n, stride, d_out, d_env=None, d_mul=None):
        r = lib().baz_agc_process_device(self._h, _vp(d_in), n, stride, _vp(d_out),
                                         _vp(d_env) if d_env else None, _vp(d_mul) if d_mul else None)
        if r < 0:
            raise AgcError(r, "baz_agc_process_device")

    def process_device_interleaved(self, d_in, n, stride, d_items):
        """AGC + interleave: d_items[t * nstreams + s] (MUSIC item layout)."""
        r = lib().baz_agc_process_device_interleaved(self._h, _vp(d_in), n, stride, _vp(d_items))
        if r < 0:
            raise AgcError(r, "baz_agc_process_device_interleaved")

    def set_stream(self, s):
        lib().baz_agc_set_stream(self._h, _vp(s) if s else None)

    def sync(self):
        lib().baz_agc_sync(self._h)

    def reset(self):
        lib().baz_agc_reset(self._h)

    def debug_selfcheck(self, d_a, d_b, n):
        """(differing square roots, differing quotients) of the fast path's sqrt / division against the rounded ones over n
        device-resident doubles a[i] and pairs a[i] / b[i] (baz_agc_debug_selfcheck)."""
        bad = (_u64 * 2)()
        r = lib().baz_agc_debug_selfcheck(self._h, _vp(d_a), _vp(d_b), int(n), bad)
        if r:
            raise AgcError(r, "baz_agc_debug_selfcheck")
        return int(bad[0]), int(bad[1])

    @property
    def count(self):
        return int(lib().baz_agc_count(self._h))
