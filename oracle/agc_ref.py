"""ctypes loader for the AGC oracle: oracle/agc_ref.c (plain-C restatement of
lib/baz_agc_cc.cc:64-102) and, when present, oracle/_ref/libbaz_agc_ref.so (the reference's own
baz_agc_cc.cc compiled against oracle/ref_shim).  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libagc_ref.so")
_REF = os.path.join(_HERE, "_ref", "libbaz_agc_ref.so")
_f32p = ctypes.POINTER(ctypes.c_float)


class _State(ctypes.Structure):
    _fields_ = [("rate", ctypes.c_float), ("reference", ctypes.c_double), ("gain", ctypes.c_double),
                ("count", ctypes.c_ulonglong), ("env", ctypes.c_double)]


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            subprocess.check_call(["make", "-C", _HERE, "libagc_ref.so"])
        _lib = ctypes.CDLL(_LIB)
        _lib.agc_ref_init.argtypes = [ctypes.POINTER(_State), ctypes.c_float, ctypes.c_float, ctypes.c_float]
        _lib.agc_ref_work.restype = ctypes.c_int
        _lib.agc_ref_work.argtypes = [ctypes.POINTER(_State), _f32p, ctypes.c_size_t, _f32p, _f32p, _f32p]
    return _lib


def have_ref():
    return os.path.exists(_REF)


def ref():
    global _ref
    if _ref is None:
        _ref = ctypes.CDLL(_REF)
        _ref.baz_ref_agc_create.restype = ctypes.c_void_p
        _ref.baz_ref_agc_create.argtypes = [ctypes.c_float] * 4
        _ref.baz_ref_agc_destroy.argtypes = [ctypes.c_void_p]
        _ref.baz_ref_agc_work.restype = ctypes.c_int
        _ref.baz_ref_agc_work.argtypes = [ctypes.c_void_p, _f32p, ctypes.c_int, _f32p, _f32p, _f32p]
    return _ref


def _p(a):
    return a.ctypes.data_as(_f32p)


class Agc:
    """Stateful oracle instance: work(x) may be called repeatedly like the scheduler does."""

    def __init__(self, rate=1e-4, reference=1.0, gain=1.0, max_gain=0.0, use_reference_source=False):
        self.use_ref = use_reference_source
        if use_reference_source:
            self.h = ref().baz_ref_agc_create(rate, reference, gain, max_gain)
        else:
            self.s = _State()
            lib().agc_ref_init(ctypes.byref(self.s), rate, reference, gain)

    def work(self, x_c64):
        x = np.ascontiguousarray(x_c64, dtype=np.complex64)
        n = x.shape[0]
        out = np.zeros(n, np.complex64)
        env = np.zeros(n, np.float32)
        mul = np.zeros(n, np.float32)
        if self.use_ref:
            r = ref().baz_ref_agc_work(self.h, _p(x.view(np.float32)), n, _p(out.view(np.float32)), _p(env), _p(mul))
        else:
            r = lib().agc_ref_work(ctypes.byref(self.s), _p(x.view(np.float32)), n, _p(out.view(np.float32)), _p(env), _p(mul))
        assert r == n
        return out, env, mul

    def __del__(self):
        if getattr(self, "use_ref", False) and getattr(self, "h", None):
            ref().baz_ref_agc_destroy(self.h)
            self.h = None
