"""CPU tests of the drop-in boundary: the C-ABI library loads, exports exactly what
include/baz_music_hip.h declares, validates arguments, and fails loudly without a GPU
(no compute is attempted here)."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from gr_baz_amd import capi


def header_functions():
    src = open(os.path.join(ROOT, "include", "baz_music_hip.h")).read()
    return sorted(set(re.findall(r"BAZ_MUSIC_API\s+[\w\s\*]+?\b(baz_music_\w+)\s*\(", src)))


def test_library_is_built_in_tree():
    assert os.path.exists(capi.LIB_PATH), "run `python -m gr_baz_amd.build` (or __graft_entry__.build())"
    assert os.path.dirname(capi.LIB_PATH).startswith(ROOT)


def test_every_declared_symbol_is_exported():
    declared = header_functions()
    assert declared == sorted(capi.SYMBOLS), "capi.SYMBOLS is out of sync with include/baz_music_hip.h"
    raw = ctypes.CDLL(capi.LIB_PATH)
    for name in declared:
        assert getattr(raw, name) is not None


def test_header_cites_reference_lines():
    src = open(os.path.join(ROOT, "include", "baz_music_hip.h")).read()
    for cite in ("lib/baz_music_doa.cc:72-161", "lib/baz_music_doa.cc:35-53", "lib/baz_music_doa.cc:60-70",
                 "lib/baz_music_doa.h:36,48,59"):
        assert cite in src


def test_strerror_and_version():
    L = capi.lib()
    assert L.baz_music_strerror(0) == b"ok"
    for code in (-1, -2, -3, -4, -5):
        assert len(L.baz_music_strerror(code)) > 0
    assert b"gfx950" in L.baz_music_version()
    assert capi.q_stride(1) == 64 and capi.q_stride(64) == 64 and capi.q_stride(65) == 128


@pytest.mark.parametrize("m,n,N,res", [
    (0, 1, 8, 4),     # m > 0                         lib/baz_music_doa.cc:45
    (4, 0, 8, 4),     # n > 0
    (4, 4, 8, 4),     # n == m underflows .cc:93 (assert only demands m >= n, .cc:46)
    (4, 5, 8, 4),     # m >= n                        .cc:46
    (4, 2, 0, 4),     # nsamples > 0                  .cc:47
    (4, 2, 10, 4),    # nsamples % m == 0             .cc:47
    (4, 2, 8, 0),     # resolution > 0                .cc:48
])
def test_create_validates_like_the_reference_asserts(m, n, N, res):
    """Argument validation precedes any device access, so it is testable without a GPU."""
    h = ctypes.c_void_p()
    tab = np.zeros((max(res, 1) * max(m, 1) * 2,), np.float32)
    r = capi.lib().baz_music_create(ctypes.byref(h), m, n, N, res,
                                    tab.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), -1)
    assert r == capi.E_INVALID and not h.value


def test_create_rejects_null_and_unsupported():
    L = capi.lib()
    h = ctypes.c_void_p()
    assert L.baz_music_create(ctypes.byref(h), 4, 2, 8, 4, None, -1) == capi.E_INVALID
    tab = np.zeros(65 * 2 * 4, np.float32)
    r = L.baz_music_create(ctypes.byref(h), 65, 2, 130, 4, tab.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), -1)
    assert r == capi.E_UNSUPPORTED     # m > BAZ_MUSIC_MAX_M (64): no silent CPU fallback
    assert L.baz_music_process(None, None, 0, None, None, None) == capi.E_INVALID
    assert L.baz_music_set_table(None, None) == capi.E_INVALID
    L.baz_music_destroy(None)          # harmless
    # the page-locking calls without a context
    assert L.baz_music_host_register(None, None, 0) == capi.E_INVALID
    assert L.baz_music_set_host_pinning(None, 1) == capi.E_INVALID
    assert L.baz_music_host_unregister_all(None) == capi.E_INVALID
    assert L.baz_music_host_pinned_bytes(None) == 0


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(capi.MusicError) as ei:
        capi.Context(4, 2, 1024, 360, np.zeros((360, 4), np.complex64))
    assert ei.value.code in (capi.E_NODEVICE, capi.E_HIP)


def test_product_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under gr_baz_amd/ may import, link or load it."""
    bad = []
    for top in ("gr_baz_amd", "scripts", "include"):
        for dp, _, fns in os.walk(os.path.join(ROOT, top)):
            for fn in fns:
                if fn.endswith((".py", ".cc", ".h", ".hip", ".cpp", ".sh")):
                    txt = open(os.path.join(dp, fn), errors="ignore").read()
                    if re.search(r"^\s*(from|import)\s+oracle\b|lib(music|agc|resamp)_ref|(music|agc|resamp)_ref\.|oracle/_ref|libbaz_\w+_ref",
                                 txt, re.M):
                        bad.append(os.path.join(dp, fn))
    assert not bad, bad
    # bench.py uses the oracle in exactly two functions -- cpu_baseline (the CPU leg) and verify_against_oracle (the checker of what a
    # timed region left in the output buffers) -- and never inside a timed region: every call of the checker comes AFTER the timing
    # loop of the function it is in
    import ast
    src = open(os.path.join(ROOT, "bench.py")).read()
    tree = ast.parse(src)
    users = set()
    for fn in [n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef)]:
        for node in ast.walk(fn):
            if isinstance(node, ast.ImportFrom) and (node.module or "").split(".")[0] == "oracle":
                users.add(fn.name)
            if isinstance(node, ast.Import) and any(a.name.split(".")[0] == "oracle" for a in node.names):
                users.add(fn.name)
    assert users == {"cpu_baseline", "verify_against_oracle"}, users
    assert not [n for n in tree.body if isinstance(n, (ast.Import, ast.ImportFrom)) and "oracle" in ast.dump(n)]      # nothing at module level

    def call_lines(fn, name):
        return [n.lineno for n in ast.walk(fn) if isinstance(n, ast.Call) and getattr(n.func, "id", getattr(n.func, "attr", None)) == name]
    fns = {n.name: n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef)}
    main_timed_end = max(call_lines(fns["main"], "max_over_ranks"))            # the last statement pair of a timed round
    assert call_lines(fns["main"], "verify_against_oracle") and min(call_lines(fns["main"], "verify_against_oracle")) > main_timed_end
    assert min(call_lines(fns["main"], "cpu_baseline")) > main_timed_end
    for leg in ("extra_music", "extra_cfg5"):
        assert min(call_lines(fns[leg], "verify_against_oracle")) > max(call_lines(fns[leg], "timed_loop")), leg


def test_headers_are_plain_c():
    """include/*.h are the FFI surface: they must compile as C (no C++ or HIP types in the signatures)."""
    import subprocess
    for h in sorted(os.listdir(os.path.join(ROOT, "include"))):
        if h.endswith(".h"):
            subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-x", "c",
                                   os.path.join(ROOT, "include", h)])


def _build_c_consumer(tmp_path):
    import subprocess
    exe = str(tmp_path / "music_c_smoke")
    csrc = os.path.join(ROOT, "gr_baz_amd", "csrc")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c_abi", "music_c_smoke.c"), "-L", csrc, "-lbaz_music_hip", "-lm",
                           "-Wl,-rpath," + csrc, "-o", exe])
    return exe


def test_plain_c_consumer_links_and_fails_loudly_without_a_gpu(tmp_path):
    """The boundary is usable from plain C (what a cgo / JNI / ctypes host binds): compile + link with gcc only."""
    import subprocess
    import torch
    exe = _build_c_consumer(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu-marked run")
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 2 and "gfx950" in r.stderr


@pytest.mark.gpu
def test_plain_c_consumer_finds_the_emitter(tmp_path, gpu_device):
    import subprocess
    exe = _build_c_consumer(tmp_path)
    r = subprocess.run([exe, "137.0"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("item")]
    assert len(lines) == 40
    for l in lines:
        f = l.split()
        assert abs(float(f[3]) - 137.0) <= 1.0 and f[-1] == "1"


# the tuning knobs INTEGRATION.md 5 documents; none of them can make a result wrong
RELEASE_KNOBS = {
    "BAZ_MUSIC_EXACT",           # 1: every value on the fp64 matrix core (no int8 form) -- A/B
    "BAZ_MUSIC_COARSE",          # 0: the full fp64 scan also without the spectrum port -- A/B, bit-identical outputs
    "BAZ_MUSIC_CHUNK_MIB", "BAZ_MUSIC_PIN_LIMIT_MIB", "BAZ_MUSIC_ZERO_COPY", "BAZ_MUSIC_SINGLE_MIB",     # host-fed path
}


def _env_names(path):
    data = open(path, "rb").read()
    return set(x.decode() for x in re.findall(rb"BAZ_MUSIC_[A-Z0-9_]+", data))


def test_release_library_reads_only_the_documented_knobs():
    """A drop-in block must not change its results because of a stray environment variable (the reference has no hidden
    modes, lib/baz_music_doa.cc:35-53).  Ablations, older kernels, geometry overrides and dumps are compiled only into
    libbaz_music_hip_lab.so (-DBAZ_MUSIC_LAB); the release library contains the names of the documented knobs and no other
    BAZ_MUSIC_* string."""
    from gr_baz_amd import build as native
    rel = _env_names(native.HIP_LIB)
    assert rel == RELEASE_KNOBS, sorted(rel ^ RELEASE_KNOBS)
    lab = _env_names(native.HIP_LAB_LIB)
    assert RELEASE_KNOBS < lab and {"BAZ_MUSIC_SCAN_VARIANT", "BAZ_MUSIC_NO_REFINE", "BAZ_MUSIC_COARSE_LAB", "BAZ_MUSIC_NSPLIT"} <= lab
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for k in RELEASE_KNOBS:
        assert k in doc, "%s is not documented in INTEGRATION.md" % k


def test_lab_library_exports_the_same_abi():
    L = capi.lib(lab=True)
    for sym in capi.SYMBOLS:
        assert hasattr(L, sym), sym


def test_collecting_the_suite_does_not_load_the_native_libraries():
    """pytest imports every test module before the first test runs.  A module that imports gr_baz_amd.baz (or calls
    capi.lib()) at import time loads /opt/rocm's HIP runtime BEFORE torch brings its own copy of the same soname, and the
    process then runs torch on a runtime it was not built for (seen once as 'no usable gfx950 device' in the first GPU
    test).  Native code is loaded lazily, inside tests."""
    import subprocess
    import sys
    code = ("import sys, importlib, glob, os\n"
            "sys.path.insert(0, 'tests')\n"
            "for f in sorted(glob.glob('tests/test_*.py')):\n"
            "    importlib.import_module(os.path.basename(f)[:-3])\n"
            "print(sorted(set(os.path.basename(l.split()[-1]) for l in open('/proc/self/maps') if 'libbaz_' in l or '_baz_music' in l "
            "or 'libgnuradio_baz' in l)))\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT)
    assert r.returncode == 0, r.stderr
    assert r.stdout.strip().splitlines()[-1] == "[]", r.stdout
