"""ctypes loader for the fractional-resampler oracle: oracle/resamp_ref.c (plain-C restatement of
lib/baz_fractional_resampler_cc.cc:80-101,152-254 + the published algorithm of gnuradio-filter's MMSE
interpolator) and, when present, oracle/_ref/libbaz_resamp_ref.so (the reference's own
baz_fractional_resampler_cc.cc compiled against oracle/ref_shim).  TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED
(see resamp_ref.c)."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libresamp_ref.so")
_REF = os.path.join(_HERE, "_ref", "libbaz_resamp_ref.so")
_f32p = ctypes.POINTER(ctypes.c_float)
NTAPS, NSTEPS = 8, 128

_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "resamp_ref.c")
        if not os.path.exists(_LIB) or os.path.getmtime(_LIB) < os.path.getmtime(src):
            subprocess.check_call(["make", "-C", _HERE, "libresamp_ref.so"])
        L = ctypes.CDLL(_LIB)
        L.resamp_ref_sizeof.restype = ctypes.c_size_t
        L.resamp_ref_init.restype = ctypes.c_int
        L.resamp_ref_init.argtypes = [ctypes.c_void_p, ctypes.c_double, ctypes.c_double, ctypes.c_ulonglong, ctypes.c_ulonglong]
        L.resamp_ref_work.restype = ctypes.c_int
        L.resamp_ref_work.argtypes = [ctypes.c_void_p, _f32p, ctypes.c_int, _f32p, ctypes.POINTER(ctypes.c_int)]
        L.resamp_ref_work2.restype = ctypes.c_int
        L.resamp_ref_work2.argtypes = [ctypes.c_void_p, _f32p, _f32p, ctypes.c_int, _f32p, ctypes.POINTER(ctypes.c_int)]
        L.resamp_ref_forecast.restype = ctypes.c_int
        L.resamp_ref_forecast.argtypes = [ctypes.c_void_p, ctypes.c_int]
        for nm in ("set_mu", "set_ratio", "adjust"):
            getattr(L, "resamp_ref_" + nm).argtypes = [ctypes.c_void_p, ctypes.c_double]
        L.resamp_ref_set_ratio_rational.argtypes = [ctypes.c_void_p, ctypes.c_ulonglong, ctypes.c_ulonglong]
        L.resamp_ref_mu.restype = ctypes.c_double
        L.resamp_ref_mu.argtypes = [ctypes.c_void_p]
        L.resamp_ref_ratio.restype = ctypes.c_double
        L.resamp_ref_ratio.argtypes = [ctypes.c_void_p]
        L.resamp_ref_table.restype = _f32p
        L.resamp_ref_table.argtypes = [ctypes.c_void_p]
        _lib = L
    return _lib


def have_ref():
    return os.path.exists(_REF)


def ref():
    global _ref
    if _ref is None:
        R = ctypes.CDLL(_REF)
        R.baz_ref_resamp_create.restype = ctypes.c_void_p
        R.baz_ref_resamp_create.argtypes = [ctypes.c_double, ctypes.c_double, ctypes.c_ulonglong, ctypes.c_ulonglong]
        R.baz_ref_resamp_destroy.argtypes = [ctypes.c_void_p]
        R.baz_ref_resamp_forecast.restype = ctypes.c_int
        R.baz_ref_resamp_forecast.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int]
        R.baz_ref_resamp_work.restype = ctypes.c_int
        R.baz_ref_resamp_work.argtypes = [ctypes.c_void_p, _f32p, _f32p, ctypes.c_int, _f32p, ctypes.POINTER(ctypes.c_int)]
        for nm in ("set_mu", "set_ratio", "post_double"):
            getattr(R, "baz_ref_resamp_" + nm).argtypes = [ctypes.c_void_p, ctypes.c_double]
        R.baz_ref_resamp_set_ratio_rational.argtypes = [ctypes.c_void_p, ctypes.c_ulonglong, ctypes.c_ulonglong]
        R.baz_ref_resamp_post_ppb.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_double]
        R.baz_ref_resamp_mu.restype = ctypes.c_double
        R.baz_ref_resamp_mu.argtypes = [ctypes.c_void_p]
        R.baz_ref_resamp_ratio.restype = ctypes.c_double
        R.baz_ref_resamp_ratio.argtypes = [ctypes.c_void_p]
        _ref = R
    return _ref


def _p(a):
    return a.ctypes.data_as(_f32p)


def taps():
    """(129, 8) float32: the closed-form MMSE table of the restatement."""
    r = Resampler(0.0, 1.0)
    return np.ctypeslib.as_array(lib().resamp_ref_table(r._s), shape=(NSTEPS + 1, NTAPS)).copy()


class _Base:
    def forecast(self, noutput):
        raise NotImplementedError

    def work(self, x, noutput, rr=None):
        """x: complex64 input window (must hold forecast(noutput) samples).  -> (out complex64[noutput], consumed)"""
        x = np.ascontiguousarray(x, dtype=np.complex64)
        need = self.forecast(noutput)
        if x.shape[0] < need and rr is None:
            raise ValueError("need %d input samples for %d outputs, have %d" % (need, noutput, x.shape[0]))
        out = np.zeros(noutput, np.complex64)
        consumed = ctypes.c_int(0)
        rrp = None
        if rr is not None:
            rr = np.ascontiguousarray(rr, dtype=np.float32)
            rrp = _p(rr)
        r = self._work(_p(x.view(np.float32)), rrp, noutput, _p(out.view(np.float32)), ctypes.byref(consumed))
        if r != noutput:
            raise RuntimeError("resampler oracle returned %d" % r)
        return out, consumed.value


class Resampler(_Base):
    """Stateful restatement instance (oracle/resamp_ref.c)."""

    def __init__(self, phase_shift, ratio, num=0, denom=0):
        L = lib()
        self._buf = ctypes.create_string_buffer(L.resamp_ref_sizeof() + 64)
        self._s = ctypes.c_void_p((ctypes.addressof(self._buf) + 63) & ~63)
        if L.resamp_ref_init(self._s, phase_shift, ratio, num, denom) != 0:
            raise ValueError("resampling ratio must be > 0 and phase shift in [0, 1]")

    def forecast(self, noutput):
        return lib().resamp_ref_forecast(self._s, noutput)

    def _work(self, inp, rr, n, out, consumed):
        if rr is None:
            return lib().resamp_ref_work(self._s, inp, n, out, consumed)
        return lib().resamp_ref_work2(self._s, inp, rr, n, out, consumed)

    def set_mu(self, mu): lib().resamp_ref_set_mu(self._s, mu)
    def set_resamp_ratio(self, r): lib().resamp_ref_set_ratio(self._s, r)
    def set_resamp_ratio_rational(self, n, d): lib().resamp_ref_set_ratio_rational(self._s, n, d)
    def adjust(self, d): lib().resamp_ref_adjust(self._s, d)
    def mu(self): return lib().resamp_ref_mu(self._s)
    def resamp_ratio(self): return lib().resamp_ref_ratio(self._s)


class RefResampler(_Base):
    """The reference's own block (oracle/_ref), same interface."""

    def __init__(self, phase_shift, ratio, num=0, denom=0):
        self._h = ref().baz_ref_resamp_create(phase_shift, ratio, num, denom)
        if not self._h:
            raise ValueError("the reference constructor threw (ratio <= 0 or phase shift outside [0, 1])")

    def __del__(self):
        try:
            if self._h:
                ref().baz_ref_resamp_destroy(self._h)
                self._h = None
        except Exception:
            pass

    def forecast(self, noutput):
        return ref().baz_ref_resamp_forecast(self._h, noutput, 1)

    def _work(self, inp, rr, n, out, consumed):
        return ref().baz_ref_resamp_work(self._h, inp, rr, n, out, consumed)

    def set_mu(self, mu): ref().baz_ref_resamp_set_mu(self._h, mu)
    def set_resamp_ratio(self, r): ref().baz_ref_resamp_set_ratio(self._h, r)
    def set_resamp_ratio_rational(self, n, d): ref().baz_ref_resamp_set_ratio_rational(self._h, n, d)
    def adjust(self, d): ref().baz_ref_resamp_post_double(self._h, d)
    def post_ppb(self, i, frac): ref().baz_ref_resamp_post_ppb(self._h, i, frac)
    def mu(self): return ref().baz_ref_resamp_mu(self._h)
    def resamp_ratio(self): return ref().baz_ref_resamp_ratio(self._h)
