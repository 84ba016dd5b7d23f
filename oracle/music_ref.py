"""ctypes loader for the plain-C oracle (oracle/music_ref.c -> libmusic_ref.so) and,
when present, oracle/_ref/libbaz_music_ref.so (the reference's own
lib/baz_music_doa.cc compiled against oracle/ref_shim).  TEST INFRASTRUCTURE ONLY."""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "libmusic_ref.so")
_REF = os.path.join(_HERE, "_ref", "libbaz_music_ref.so")

_f32p = ctypes.POINTER(ctypes.c_float)
_f64p = ctypes.POINTER(ctypes.c_double)


def build(force=False):
    """Compile the oracle libraries (gcc/g++ only).  Building the checker is not using it."""
    if force or not os.path.exists(_LIB) or \
            os.path.getmtime(_LIB) < os.path.getmtime(os.path.join(_HERE, "music_ref.c")):
        subprocess.check_call(["make", "-C", _HERE, "libmusic_ref.so", "libagc_ref.so", "libresamp_ref.so"])
    if os.path.isdir("/root/reference") and os.path.isdir(os.path.join(_HERE, "ref_shim")):
        subprocess.check_call(["make", "-C", _HERE, "ref"])


def _load(path):
    lib = ctypes.CDLL(path)
    return lib


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB):
            build()
        _lib = _load(_LIB)
        _lib.music_ref_work.restype = ctypes.c_int
        _lib.music_ref_work.argtypes = [_f32p, _f32p, ctypes.c_uint, ctypes.c_uint, ctypes.c_uint,
                                        ctypes.c_uint, _f32p, _f32p, _f32p, _f64p]
        _lib.music_ref_work_batch.restype = ctypes.c_int
        _lib.music_ref_work_batch.argtypes = [_f32p, ctypes.c_uint, _f32p, ctypes.c_uint,
                                              ctypes.c_uint, ctypes.c_uint, ctypes.c_uint,
                                              _f32p, _f32p, _f32p]
        _lib.music_ref_eig.restype = ctypes.c_int
        _lib.music_ref_eig.argtypes = [ctypes.c_uint, _f64p, _f64p, _f64p]
    return _lib


def have_ref():
    return os.path.exists(_REF)


def ref():
    """oracle/_ref: the reference's own work() (baz_music_doa.cc) behind a C entry point."""
    global _ref
    if _ref is None:
        _ref = _load(_REF)
        _ref.baz_ref_set_lapack.restype = ctypes.c_int
        _ref.baz_ref_set_lapack.argtypes = [ctypes.c_char_p, ctypes.c_char_p]
        _ref.baz_ref_uses_lapack.restype = ctypes.c_int
        _ref.baz_ref_work_batch.restype = ctypes.c_int
        _ref.baz_ref_work_batch.argtypes = [_f32p, ctypes.c_uint, _f32p, ctypes.c_uint,
                                            ctypes.c_uint, ctypes.c_uint, ctypes.c_uint,
                                            _f32p, _f32p, _f32p]
    return _ref


def ref_use_lapack(enable=True):
    """Back the _ref build's arma::eig_sym with LAPACK zheev from scipy's bundled OpenBLAS
    (what Armadillo itself dispatches to); falls back to the shim's Jacobi when unavailable.
    Returns True when LAPACK is in use."""
    r = ref()
    if not enable:
        r.baz_ref_clear_lapack()
        return False
    try:
        import glob
        import scipy
        cands = glob.glob(os.path.join(os.path.dirname(scipy.__file__), "..", "scipy.libs",
                                       "libscipy_openblas*.so"))
        for c in cands:
            if r.baz_ref_set_lapack(os.path.abspath(c).encode(), b"scipy_LAPACKE_zheev") == 0:
                return True
    except Exception:
        pass
    return False


def _p32(a):
    return a.ctypes.data_as(_f32p)


def _run(fn, items_c64, table_c64, m, n, want_spectrum=True):
    items = np.ascontiguousarray(items_c64, dtype=np.complex64)
    if items.ndim == 1:
        items = items[None, :]
    table = np.ascontiguousarray(table_c64, dtype=np.complex64)
    B, N = items.shape
    res = table.shape[0]
    ang = np.zeros((B, n), dtype=np.float32)
    lvl = np.zeros((B, n), dtype=np.float32)
    spec = np.zeros((B, res), dtype=np.float32) if want_spectrum else None
    r = fn(_p32(items.view(np.float32)), B, _p32(table.view(np.float32)), m, n, N, res,
           _p32(ang), _p32(lvl), _p32(spec) if want_spectrum else None)
    if r != B:
        raise RuntimeError("oracle work_batch returned %d" % r)
    return ang, lvl, spec


def work_batch(items_c64, table_c64, m, n, want_spectrum=True):
    """Plain-C restatement, one work() per item."""
    return _run(lib().music_ref_work_batch, items_c64, table_c64, m, n, want_spectrum)


def ref_work_batch(items_c64, table_c64, m, n, want_spectrum=True):
    """The reference's own work() source (oracle/_ref), one work() per item."""
    return _run(ref().baz_ref_work_batch, items_c64, table_c64, m, n, want_spectrum)


def eig(A):
    A = np.ascontiguousarray(A, dtype=np.complex128)
    m = A.shape[0]
    w = np.zeros(m)
    V = np.zeros((m, m), dtype=np.complex128)
    r = lib().music_ref_eig(m, A.view(np.float64).ctypes.data_as(_f64p),
                            w.ctypes.data_as(_f64p), V.view(np.float64).ctypes.data_as(_f64p))
    if r != 0:
        raise RuntimeError("music_ref_eig failed")
    return w, V
