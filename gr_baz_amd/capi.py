"""ctypes binding of the C-ABI declared in include/baz_music_hip.h.

This is the same binding a maintainer of the reference would write on the python side if the
SWIG module were bypassed; the GNU Radio host block binds the same symbols
from C++.  There is no fallback: a missing library raises ImportError at first use, a missing
gfx950 device raises MusicError from Context().
"""
from __future__ import annotations

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libbaz_music_hip.so")
# the same sources built with -DBAZ_MUSIC_LAB: reads the lab switches (ablations, geometry overrides, older kernels) the
# release library does not contain.  Only tests/lab and the A/B tests ask for it (Context(..., lab=True)).
LAB_LIB_PATH = os.path.join(_HERE, "csrc", "libbaz_music_hip_lab.so")

OK = 0
E_INVALID, E_NOMEM, E_HIP, E_UNSUPPORTED, E_NODEVICE = -1, -2, -3, -4, -5
STAGE_COV, STAGE_EVD, STAGE_SCAN, STAGE_MERGE, NUM_STAGES = 0, 1, 2, 3, 4
MAX_M, MAX_N = 64, 63     # BAZ_MUSIC_MAX_M / _MAX_N (specialised kernels up to m = 16, run-time-m kernels above)

# every symbol include/baz_music_hip.h declares (tests/test_abi.py checks the .so exports them all)
SYMBOLS = [
    "baz_music_create", "baz_music_destroy", "baz_music_set_table", "baz_music_process",
    "baz_music_process_device", "baz_music_process_device_on", "baz_music_set_stream", "baz_music_sync", "baz_music_reserve",
    "baz_music_profile", "baz_music_stage_ms", "baz_music_stage_name", "baz_music_debug_cov",
    "baz_music_debug_evd", "baz_music_debug_q", "baz_music_q_stride", "baz_music_bytes_per_item", "baz_music_strerror",
    "baz_music_last_hip_error", "baz_music_version", "baz_music_device_count", "baz_music_device", "baz_music_set_peak_mode", "baz_music_refined_items",
    "baz_music_refined_values", "baz_music_debug_coarse_margin", "baz_music_debug_coarse_fired",
    "baz_music_host_register", "baz_music_set_host_pinning", "baz_music_host_unregister_all", "baz_music_host_pinned_bytes",
    "baz_music_debug_i8_margin", "baz_music_debug_i8_stats", "baz_music_uses_i8_scan", "baz_music_debug_i8_image",
    "baz_music_debug_i8_nsplit", "baz_music_debug_sort_state", "baz_music_last_retune_ms", "baz_music_debug_table_image", "baz_music_debug_host_table_image",
    "baz_music_debug_guard_check", "baz_music_debug_guard_active",
]

_vp = ctypes.c_void_p
_u32 = ctypes.c_uint32
_f32p = ctypes.POINTER(ctypes.c_float)

_lib = None
_lab_lib = None


class MusicError(RuntimeError):
    def __init__(self, code, where, detail=""):
        self.code = code
        msg = "%s failed: %s (%d)" % (where, lib().baz_music_strerror(code).decode(), code)
        if detail:
            msg += " [" + detail + "]"
        super().__init__(msg)


def lib(lab=False):
    """Loads libbaz_music_hip.so (built in-tree by gr_baz_amd.build); never falls back.  lab=True: the -DBAZ_MUSIC_LAB form."""
    global _lib, _lab_lib
    if lab:
        if _lab_lib is None:
            path = LAB_LIB_PATH
            if os.environ.get("BAZ_MUSIC_LAB_LIB") == "quick":      # tests/lab iterations: python -m gr_baz_amd.build --quick
                path = os.path.join(os.path.dirname(LAB_LIB_PATH), "libbaz_music_hip_quick.so")
            if not os.path.exists(path):
                raise ImportError("gr_baz_amd: %s is missing - run `python -m gr_baz_amd.build`" % path)
            _lab_lib = _bind(ctypes.CDLL(path))
        return _lab_lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("gr_baz_amd: %s is missing - run `python -m gr_baz_amd.build` "
                          "(there is no CPU fallback for the MUSIC-DoA path)" % LIB_PATH)
    _lib = _bind(ctypes.CDLL(LIB_PATH))
    return _lib


def _bind(L):
    L.baz_music_create.restype = ctypes.c_int
    L.baz_music_create.argtypes = [ctypes.POINTER(_vp), _u32, _u32, _u32, _u32, _f32p, ctypes.c_int]
    L.baz_music_destroy.restype = None
    L.baz_music_destroy.argtypes = [_vp]
    L.baz_music_set_table.restype = ctypes.c_int
    L.baz_music_set_table.argtypes = [_vp, _f32p]
    L.baz_music_process.restype = ctypes.c_int
    L.baz_music_process.argtypes = [_vp, _f32p, _u32, _f32p, _f32p, _f32p]
    L.baz_music_process_device.restype = ctypes.c_int
    L.baz_music_process_device.argtypes = [_vp, _vp, _u32, _vp, _vp, _vp]
    L.baz_music_process_device_on.restype = ctypes.c_int
    L.baz_music_process_device_on.argtypes = [_vp, _vp, _vp, _u32, _vp, _vp, _vp]
    L.baz_music_set_stream.restype = ctypes.c_int
    L.baz_music_set_stream.argtypes = [_vp, _vp]
    L.baz_music_sync.restype = ctypes.c_int
    L.baz_music_sync.argtypes = [_vp]
    L.baz_music_reserve.restype = ctypes.c_int
    L.baz_music_reserve.argtypes = [_vp, _u32]
    L.baz_music_profile.restype = ctypes.c_int
    L.baz_music_profile.argtypes = [_vp, ctypes.c_int]
    L.baz_music_stage_ms.restype = ctypes.c_int
    L.baz_music_stage_ms.argtypes = [_vp, ctypes.c_int, ctypes.POINTER(ctypes.c_double),
                                     ctypes.POINTER(ctypes.c_uint64)]
    L.baz_music_stage_name.restype = ctypes.c_char_p
    L.baz_music_stage_name.argtypes = [_vp, ctypes.c_int]
    L.baz_music_debug_cov.restype = ctypes.c_int
    L.baz_music_debug_cov.argtypes = [_vp, _vp, _u32, _vp]
    L.baz_music_debug_evd.restype = ctypes.c_int
    L.baz_music_debug_evd.argtypes = [_vp, _vp, _u32, _vp]
    L.baz_music_debug_q.restype = ctypes.c_int
    L.baz_music_debug_q.argtypes = [_vp, _vp, _u32, _vp]
    L.baz_music_q_stride.restype = _u32
    L.baz_music_q_stride.argtypes = [_u32]
    L.baz_music_bytes_per_item.restype = ctypes.c_uint64
    L.baz_music_bytes_per_item.argtypes = [_vp, ctypes.c_int]
    L.baz_music_strerror.restype = ctypes.c_char_p
    L.baz_music_strerror.argtypes = [ctypes.c_int]
    L.baz_music_last_hip_error.restype = ctypes.c_char_p
    L.baz_music_last_hip_error.argtypes = [_vp]
    L.baz_music_version.restype = ctypes.c_char_p
    L.baz_music_version.argtypes = []
    L.baz_music_device_count.restype = ctypes.c_int
    L.baz_music_device_count.argtypes = []
    L.baz_music_device.restype = ctypes.c_int
    L.baz_music_device.argtypes = [_vp]
    L.baz_music_refined_items.restype = ctypes.c_int64
    L.baz_music_refined_items.argtypes = [_vp]
    L.baz_music_refined_values.restype = ctypes.c_int64
    L.baz_music_refined_values.argtypes = [_vp]
    L.baz_music_debug_coarse_fired.restype = ctypes.c_int64
    L.baz_music_debug_coarse_fired.argtypes = [_vp]
    L.baz_music_debug_coarse_margin.restype = ctypes.c_int
    L.baz_music_debug_coarse_margin.argtypes = [_vp, _vp, _u32, ctypes.POINTER(ctypes.c_float)]
    L.baz_music_set_peak_mode.restype = ctypes.c_int
    L.baz_music_set_peak_mode.argtypes = [_vp, ctypes.c_int]
    L.baz_music_host_register.restype = ctypes.c_int
    L.baz_music_host_register.argtypes = [_vp, _vp, ctypes.c_size_t]
    L.baz_music_set_host_pinning.restype = ctypes.c_int
    L.baz_music_set_host_pinning.argtypes = [_vp, ctypes.c_int]
    L.baz_music_host_unregister_all.restype = ctypes.c_int
    L.baz_music_host_unregister_all.argtypes = [_vp]
    L.baz_music_host_pinned_bytes.restype = ctypes.c_uint64
    L.baz_music_host_pinned_bytes.argtypes = [_vp]
    L.baz_music_debug_i8_margin.restype = ctypes.c_int
    L.baz_music_debug_i8_margin.argtypes = [_vp, _vp, _u32, ctypes.POINTER(ctypes.c_float)]   # float worst[3]
    L.baz_music_debug_i8_stats.restype = ctypes.c_int
    L.baz_music_debug_i8_stats.argtypes = [_vp, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
    L.baz_music_uses_i8_scan.restype = ctypes.c_int
    L.baz_music_uses_i8_scan.argtypes = [_vp]
    L.baz_music_debug_i8_nsplit.restype = _u32
    L.baz_music_debug_i8_nsplit.argtypes = [_u32, _u32, _u32]
    L.baz_music_debug_i8_image.restype = ctypes.c_size_t
    L.baz_music_debug_i8_image.argtypes = [_u32, _u32, _f32p, ctypes.POINTER(ctypes.c_uint8), ctypes.c_size_t,
                                           ctypes.POINTER(ctypes.c_double)]
    L.baz_music_debug_guard_check.restype = ctypes.c_int
    L.baz_music_debug_guard_check.argtypes = []
    L.baz_music_debug_guard_active.restype = ctypes.c_int
    L.baz_music_debug_guard_active.argtypes = []
    L.baz_music_debug_sort_state.restype = ctypes.c_int
    L.baz_music_debug_sort_state.argtypes = [_vp, ctypes.POINTER(ctypes.c_uint64)]
    L.baz_music_last_retune_ms.restype = ctypes.c_int
    L.baz_music_last_retune_ms.argtypes = [_vp, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
    L.baz_music_debug_table_image.restype = ctypes.c_size_t
    L.baz_music_debug_table_image.argtypes = [_vp, ctypes.c_int, _vp, ctypes.c_size_t]
    L.baz_music_debug_host_table_image.restype = ctypes.c_size_t
    L.baz_music_debug_host_table_image.argtypes = [_u32, _u32, _u32, _f32p, ctypes.c_int, _vp, ctypes.c_size_t]
    return L


def _table_f32(table, res, m):
    t = np.ascontiguousarray(np.asarray(table, dtype=np.complex64))
    if t.shape != (res, m):
        raise ValueError("array_response must be resolution x m = %dx%d, got %s" % (res, m, t.shape))
    return t


class Context:
    """One baz_music_ctx: the device-side state of one baz_music_doa block instance."""

    def __init__(self, m, n, nsamples, resolution, table, device_id=-1, lab=None):
        # lab=None: the lab form of the library where BAZ_MUSIC_LAB_LIB is set (lab / quick: tests/lab harnesses whose switches
        # the release form does not read), else the release form
        if lab is None:
            lab = bool(os.environ.get("BAZ_MUSIC_LAB_LIB"))
        L = self._L = lib(lab)
        self.m, self.n, self.nsamples, self.res = int(m), int(n), int(nsamples), int(resolution)
        t = _table_f32(table, self.res, self.m) if (self.res > 0 and self.m > 0) else \
            np.zeros((1, 1), np.complex64)
        h = _vp()
        r = L.baz_music_create(ctypes.byref(h), self.m, self.n, self.nsamples, self.res,
                               t.view(np.float32).ctypes.data_as(_f32p), int(device_id))
        if r != OK:
            raise MusicError(r, "baz_music_create")
        self._h = h

    def _chk(self, r, where):
        if r < 0:
            raise MusicError(r, where, self._L.baz_music_last_hip_error(self._h).decode())
        return r

    def close(self):
        if getattr(self, "_h", None):
            self._L.baz_music_destroy(self._h)
            self._h = None

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_table(self, table):
        t = _table_f32(table, self.res, self.m)
        self._chk(self._L.baz_music_set_table(self._h, t.view(np.float32).ctypes.data_as(_f32p)),
                  "baz_music_set_table")

    def debug_sort_state(self):
        """{"sorted_calls", "unsorted_calls", "fired", "walked", "last_sorted"} of the gated scan's sorting policy (synchronises)."""
        v = (ctypes.c_uint64 * 5)()
        self._chk(self._L.baz_music_debug_sort_state(self._h, v), "baz_music_debug_sort_state")
        return {"sorted_calls": int(v[0]), "unsorted_calls": int(v[1]), "fired": int(v[2]), "walked": int(v[3]), "last_sorted": bool(v[4])}

    def last_retune_ms(self):
        """(wall milliseconds of the last set_table, milliseconds of it spent waiting for / holding the lock shared with process)."""
        a, b = ctypes.c_double(0.0), ctypes.c_double(0.0)
        self._chk(self._L.baz_music_last_retune_ms(self._h, ctypes.byref(a), ctypes.byref(b)), "baz_music_last_retune_ms")
        return float(a.value), float(b.value)

    def debug_table_image(self, which):
        """Image `which` of the table in force, copied back from the device (uint8 array), or None when the configuration has none."""
        n = int(self._L.baz_music_debug_table_image(self._h, int(which), None, 0))
        if n == 0:
            return None
        out = np.zeros(n, np.uint8)
        got = int(self._L.baz_music_debug_table_image(self._h, int(which), _vp(out.ctypes.data), n))
        if got != n:
            raise MusicError(E_HIP, "baz_music_debug_table_image")
        return out

    def process(self, items, want_lvl=True, want_spectrum=True, out=None):
        """items: (batch, nsamples) complex64 host array -> (ang, lvl|None, spectrum|None).
        out: optional (ang, lvl|None, spectrum|None) float32 C-contiguous arrays to write into (a scheduler's
        persistent buffers; page-locked ones are handed to the DMA engines without staging)."""
        x = np.ascontiguousarray(np.asarray(items, dtype=np.complex64))
        if x.ndim == 1:
            x = x[None, :]
        if x.shape[1] != self.nsamples:
            raise ValueError("items must be (batch, %d) complex64" % self.nsamples)
        B = x.shape[0]
        if out is not None:
            ang, lvl, spec = out
            want_lvl, want_spectrum = lvl is not None, spec is not None
            for a, cols in ((ang, self.n), (lvl, self.n), (spec, self.res)):
                if a is not None and (a.dtype != np.float32 or not a.flags["C_CONTIGUOUS"] or a.shape != (B, cols)):
                    raise ValueError("out arrays must be C-contiguous float32 of shape (batch, n|n|resolution)")
        else:
            ang = np.zeros((B, self.n), np.float32)
            lvl = np.zeros((B, self.n), np.float32) if want_lvl else None
            spec = np.zeros((B, self.res), np.float32) if want_spectrum else None
        r = self._L.baz_music_process(
            self._h, x.view(np.float32).ctypes.data_as(_f32p), B, ang.ctypes.data_as(_f32p),
            lvl.ctypes.data_as(_f32p) if want_lvl else None,
            spec.ctypes.data_as(_f32p) if want_spectrum else None)
        self._chk(r, "baz_music_process")
        return ang, lvl, spec

    # ---- device-resident path (pointers are plain integers, e.g. torch.Tensor.data_ptr()) ----
    def process_device(self, d_in, batch, d_ang, d_lvl=None, d_spec=None, stream=None):
        """Enqueues one batch.  stream=None: on the context's stream, unordered against other streams (the caller
        orders: set_stream(), sync(), or a device-wide synchronize).  stream=<hipStream_t as int, 0 = the legacy
        default stream, e.g. torch.cuda.current_stream().cuda_stream>: with stream semantics relative to it."""
        if stream is None:
            r = self._L.baz_music_process_device(self._h, _vp(d_in), int(batch), _vp(d_ang),
                                               _vp(d_lvl) if d_lvl else None, _vp(d_spec) if d_spec else None)
        else:
            r = self._L.baz_music_process_device_on(self._h, _vp(stream) if stream else None, _vp(d_in), int(batch),
                                                  _vp(d_ang), _vp(d_lvl) if d_lvl else None,
                                                  _vp(d_spec) if d_spec else None)
        self._chk(r, "baz_music_process_device")

    def refined_values(self):
        """(item, bin) VALUES of the last process call that were recomputed in the reference's literal form (near-null
        bins, extreme SNR) -- not items: one item can contribute up to `resolution` of them.  On the wide path counted by the
        matrix-core scan only (17 <= m <= 64, n <= 8), else 0."""
        return int(self._L.baz_music_refined_values(self._h))

    refined_items = refined_values      # the round-1 name of the same statistic (kept for callers; the unit is values)

    def debug_coarse_fired(self):
        """Lab statistic (context created under BAZ_MUSIC_COARSE_STATS=1): exact tile evaluations since the last read."""
        return int(self._L.baz_music_debug_coarse_fired(self._h))

    def debug_coarse_margin(self, d_in, batch):
        """Worst observed |coarse - exact| / allowance of the coarse-gated scan over every (item, bin) of the batch
        (baz_music_debug_coarse_margin; the gate is sound below 1)."""
        w = ctypes.c_float(0.0)
        self._chk(self._L.baz_music_debug_coarse_margin(self._h, _vp(d_in), int(batch), ctypes.byref(w)),
                  "baz_music_debug_coarse_margin")
        return float(w.value)

    def uses_i8_scan(self):
        """True when this context's scan runs on the int8 matrix core (6 <= m <= 16, n <= 4, not BAZ_MUSIC_EXACT=1)."""
        return bool(self._L.baz_music_uses_i8_scan(self._h))

    def debug_i8_margin(self, d_in, batch):
        """(worst |d5 - d| / E5, worst |d7 - d| / allowance, worst |d4 - d| / E4) of the int8 scan's five-, seven- and
        four-digit forms over every (item, bin) of the batch, against the fp64 form (the bounds hold below 1)."""
        w = (ctypes.c_float * 3)(0.0, 0.0, 0.0)
        self._chk(self._L.baz_music_debug_i8_margin(self._h, _vp(d_in), int(batch), w), "baz_music_debug_i8_margin")
        return float(w[0]), float(w[1]), float(w[2])

    def debug_i8_stats(self):
        """(wave tiles -- 16 items x 16 bins -- that ran the refined form, wave tiles walked) of the int8 scan since the last read; resets."""
        a, b = ctypes.c_uint64(0), ctypes.c_uint64(0)
        self._chk(self._L.baz_music_debug_i8_stats(self._h, ctypes.byref(a), ctypes.byref(b)), "baz_music_debug_i8_stats")
        return int(a.value), int(b.value)

    # ---- page-locking of caller buffers that live across calls (a scheduler's stream buffers) ----
    def host_register(self, array):
        """Page-locks a numpy array's memory for this context (the array must outlive the registration: call
        host_unregister_all() or close() before dropping it).  Returns the library's code: 0 = locked (or already was)."""
        return int(self._L.baz_music_host_register(self._h, _vp(array.ctypes.data), array.nbytes))

    def set_host_pinning(self, enable):
        """process() page-locks the ranges of every call the first time it sees them (persistent buffers only)."""
        self._chk(self._L.baz_music_set_host_pinning(self._h, 1 if enable else 0), "baz_music_set_host_pinning")

    def host_unregister_all(self):
        self._chk(self._L.baz_music_host_unregister_all(self._h), "baz_music_host_unregister_all")

    def host_pinned_bytes(self):
        return int(self._L.baz_music_host_pinned_bytes(self._h))

    def set_peak_mode(self, mode):
        """0: the reference's n strongest bins (default); 1: n strongest local maxima (opt-in extension)."""
        self._chk(self._L.baz_music_set_peak_mode(self._h, int(mode)), "baz_music_set_peak_mode")

    def set_stream(self, hip_stream):
        self._chk(self._L.baz_music_set_stream(self._h, _vp(hip_stream) if hip_stream else None),
                  "baz_music_set_stream")

    def sync(self):
        self._chk(self._L.baz_music_sync(self._h), "baz_music_sync")

    def reserve(self, max_batch):
        self._chk(self._L.baz_music_reserve(self._h, int(max_batch)), "baz_music_reserve")

    def profile(self, enable):
        """enable: False/0 off, True/1 every stage, 2 only the scan stage (cheapest)."""
        self._chk(self._L.baz_music_profile(self._h, int(enable)), "baz_music_profile")

    def stage_ms(self, stage):
        ms = ctypes.c_double(0.0)
        cnt = ctypes.c_uint64(0)
        self._chk(self._L.baz_music_stage_ms(self._h, stage, ctypes.byref(ms), ctypes.byref(cnt)),
                  "baz_music_stage_ms")
        return ms.value, cnt.value

    def stage_name(self, stage):
        return self._L.baz_music_stage_name(self._h, stage).decode()

    def debug_cov(self, d_in, batch, d_R):
        self._chk(self._L.baz_music_debug_cov(self._h, _vp(d_in), int(batch), _vp(d_R)), "baz_music_debug_cov")

    def debug_evd(self, d_R, batch, d_Q):
        self._chk(self._L.baz_music_debug_evd(self._h, _vp(d_R), int(batch), _vp(d_Q)), "baz_music_debug_evd")

    def debug_q(self, d_in, batch, d_Q):
        self._chk(self._L.baz_music_debug_q(self._h, _vp(d_in), int(batch), _vp(d_Q)), "baz_music_debug_q")

    def bytes_per_item(self, with_spectrum=True):
        return int(self._L.baz_music_bytes_per_item(self._h, 1 if with_spectrum else 0))


def debug_i8_image(m, resolution, table):
    """HOST-ONLY: (image bytes as a uint8 array -- [step][tile][block][digit 0..4][lane][16] followed by the same with
    [digit 5, 6] --, params dict) that the library builds for the int8 scan from `table`, or (None, None) when the table has
    no image.  Needs no device."""
    t = _table_f32(table, int(resolution), int(m))
    tp = t.view(np.float32).ctypes.data_as(_f32p)
    n = lib().baz_music_debug_i8_image(int(m), int(resolution), tp, None, 0, None)
    if n == 0:
        return None, None
    img = np.zeros(n, np.uint8)
    par = np.zeros(16, np.float64)
    lib().baz_music_debug_i8_image(int(m), int(resolution), tp, img.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8)), n,
                                   par.ctypes.data_as(ctypes.POINTER(ctypes.c_double)))
    return img, {"wt": par[:7].copy(), "sq": par[7], "t_acc": par[8], "e_bound": par[9], "e_refined": par[10],
                 "ns": int(par[11]), "nd": int(par[12]), "e4_bound": par[13], "t4": par[14]}


TABLE_IMAGES = {0: "FB", 1: "TB", 2: "coarse", 3: "i8", 4: "a2p", 5: "TA", 6: "a2", 7: "params", 8: "i8 packed (m <= 4)"}


def debug_host_table_image(m, n, resolution, table, which, lab=None):
    """HOST-ONLY: image `which` of `table` as the round-4 host routines build it (uint8 array), or None.  Needs no device.
    lab=None: the lab form of the library where BAZ_MUSIC_LAB_LIB is set (like Context); image 8 -- the level-packed int8 operands of
    2 .. 4 antennas -- exists in the lab form only."""
    if lab is None:
        lab = bool(os.environ.get("BAZ_MUSIC_LAB_LIB"))
    t = _table_f32(table, int(resolution), int(m))
    tp = t.view(np.float32).ctypes.data_as(_f32p)
    L = lib(lab)
    nb = int(L.baz_music_debug_host_table_image(int(m), int(n), int(resolution), tp, int(which), None, 0))
    if nb == 0:
        return None
    out = np.zeros(nb, np.uint8)
    L.baz_music_debug_host_table_image(int(m), int(n), int(resolution), tp, int(which), _vp(out.ctypes.data), nb)
    return out


def guard_active(lab=True):
    """True when the (lab) library allocates with guard zones: lab build and BAZ_MUSIC_GUARD=1 in the environment at load time."""
    return bool(lib(lab).baz_music_debug_guard_active())


def guard_check(lab=True):
    """Guard zones found overwritten since the process started (lab library under BAZ_MUSIC_GUARD=1; synchronises the device); 0 otherwise."""
    return int(lib(lab).baz_music_debug_guard_check())


def q_stride(batch):
    return int(lib().baz_music_q_stride(int(batch)))


def device_count():
    """Usable gfx950 devices (0 on a CPU-only box)."""
    return int(lib().baz_music_device_count())


def version():
    return lib().baz_music_version().decode()
