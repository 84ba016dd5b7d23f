"""Builds the in-tree native artefacts of gr_baz_amd (gfx950 only, no JIT cache):

  csrc/libbaz_music_hip.so   HIP kernels + the C-ABI of include/baz_music_hip.h   (hipcc)
  csrc/libbaz_music_hip_lab.so   the same with -DBAZ_MUSIC_LAB: lab switches compiled in (tests/lab, A/B tests only)
  csrc/libbaz_agc_hip.so     AGC kernels + the C-ABI of include/baz_agc_hip.h      (hipcc)
  csrc/libbaz_resamp_hip.so  fractional resampler kernel + the C-ABI of include/baz_resamp_hip.h (hipcc)
  host/libgnuradio_baz_music.so   the gr::sync_block host block on the GNU Radio API shim (g++)
  host/_baz_music*.so        pybind11 module exposing baz.music_doa (SWIG stand-in)

`python -m gr_baz_amd.build` rebuilds everything; build_all() is what __graft_entry__.build() calls.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
HOST = os.path.join(HERE, "host")
INCLUDE = os.path.join(ROOT, "include")

HIP_LIB = os.path.join(CSRC, "libbaz_music_hip.so")
HIP_LAB_LIB = os.path.join(CSRC, "libbaz_music_hip_lab.so")   # -DBAZ_MUSIC_LAB (tests/lab, A/B tests)
AGC_LIB = os.path.join(CSRC, "libbaz_agc_hip.so")
RESAMP_LIB = os.path.join(CSRC, "libbaz_resamp_hip.so")
HOST_LIB = os.path.join(HOST, "libgnuradio_baz_music.so")

# -amdgpu-mfma-vgpr-form: MFMA accumulators in VGPRs (gfx950 has a unified file): no v_accvgpr_read per result
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden",
               "-mllvm", "-amdgpu-mfma-vgpr-form"]


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources if os.path.exists(s))


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (ROCm toolchain required to build the gfx950 kernels)")


def build_hip(force=False, verbose=False):
    """The three HIP libraries, plus the LAB form of the MUSIC library (-DBAZ_MUSIC_LAB: the same sources with the
    ablation / geometry / older-kernel switches that the release form does not read; loaded only by tests/lab and the
    A/B tests through capi.Context(..., lab=True)).  Out-of-date targets compile side by side."""
    music_srcs = [os.path.join(CSRC, "baz_music_hip.hip"), os.path.join(CSRC, "music_kernels.hip.h"),
                  os.path.join(CSRC, "music_wide_kernels.hip.h"), os.path.join(CSRC, "scan_coarse_kernels.hip.h"),
                  os.path.join(CSRC, "scan_i8_kernels.hip.h"), os.path.join(CSRC, "scan_i8p_kernels.hip.h"),
                  os.path.join(CSRC, "table_kernels.hip.h"), os.path.join(CSRC, "sort_kernels.hip.h"), os.path.join(INCLUDE, "baz_music_hip.h")]
    agc_srcs = [os.path.join(CSRC, "baz_agc_hip.hip"), os.path.join(CSRC, "agc_kernels.hip.h"),
                os.path.join(INCLUDE, "baz_agc_hip.h")]
    rs_srcs = [os.path.join(CSRC, "baz_resamp_hip.hip"), os.path.join(CSRC, "resamp_kernels.hip.h"),
               os.path.join(INCLUDE, "baz_resamp_hip.h")]
    jobs = []
    for target, srcs, extra in ((HIP_LIB, music_srcs, []), (HIP_LAB_LIB, music_srcs, ["-DBAZ_MUSIC_LAB"]),
                                (AGC_LIB, agc_srcs, []), (RESAMP_LIB, rs_srcs, [])):
        if force or _newer(target, srcs):
            cmd = [_hipcc()] + HIPCC_FLAGS + extra + ["-I", INCLUDE, "-o", target + ".tmp", srcs[0]]
            if verbose:
                print(" ".join(cmd), flush=True)
            jobs.append((target, cmd, subprocess.Popen(cmd, cwd=CSRC)))
    failed = []
    for target, cmd, proc in jobs:
        if proc.wait() != 0:
            failed.append(" ".join(cmd))
        else:
            os.replace(target + ".tmp", target)
    if failed:
        raise subprocess.CalledProcessError(1, failed[0])
    return HIP_LIB


QUICK_LIB = os.path.join(CSRC, "libbaz_music_hip_quick.so")   # lab iterations only: -DBAZ_MUSIC_LAB -DBAZ_MUSIC_QUICK (m = 4, 8, 16)


def build_quick(verbose=False):
    """Lab iterations on one kernel: the LAB form of the MUSIC library restricted to m = 4, 8, 16 (a fraction of the compile
    time).  Loaded by capi.lib(lab=True) when BAZ_MUSIC_LAB_LIB=quick; never built by build_all(), never shipped."""
    cmd = [_hipcc()] + HIPCC_FLAGS + ["-DBAZ_MUSIC_LAB", "-DBAZ_MUSIC_QUICK", "-I", INCLUDE, "-o", QUICK_LIB + ".tmp",
                                      os.path.join(CSRC, "baz_music_hip.hip")]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd, cwd=CSRC)
    os.replace(QUICK_LIB + ".tmp", QUICK_LIB)
    return QUICK_LIB


def pybind_module_path():
    ext = sysconfig.get_config_var("EXT_SUFFIX") or ".so"
    return os.path.join(HOST, "_baz_music" + ext)


def build_host(force=False, verbose=False):
    """C++ host block (gr::sync_block surface) + pybind11 module; both link libbaz_music_hip.so."""
    block_srcs = [os.path.join(HOST, "baz_music_doa.cc"), os.path.join(HOST, "baz_music_doa.h"),
                  os.path.join(INCLUDE, "baz_music_hip.h"), os.path.join(HOST, "baz_agc_cc.cc"),
                  os.path.join(HOST, "baz_agc_cc.h"), os.path.join(INCLUDE, "baz_agc_hip.h"),
                  os.path.join(HOST, "baz_fractional_resampler_cc.cc"), os.path.join(HOST, "baz_fractional_resampler_cc.h"),
                  os.path.join(INCLUDE, "baz_resamp_hip.h")]
    if not os.path.exists(block_srcs[0]):
        return None
    shim_inc = os.path.join(HOST, "gr_shim")
    common = ["-O2", "-std=c++14", "-fPIC", "-I", INCLUDE, "-I", HOST, "-I", shim_inc]
    link = ["-L", CSRC, "-lbaz_music_hip", "-lbaz_agc_hip", "-lbaz_resamp_hip", "-Wl,-rpath,$ORIGIN/../csrc"]
    if force or _newer(HOST_LIB, block_srcs + [HIP_LIB, AGC_LIB, RESAMP_LIB]):
        cmd = ["g++"] + common + ["-shared", "-o", HOST_LIB, block_srcs[0], block_srcs[3], block_srcs[6]] + link
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd, cwd=HOST)
    mod = pybind_module_path()
    mod_src = os.path.join(HOST, "baz_pybind.cc")
    if os.path.exists(mod_src) and (force or _newer(mod, [mod_src, HOST_LIB])):
        import pybind11
        cmd = ["g++"] + common + ["-I", pybind11.get_include(), "-I", sysconfig.get_paths()["include"],
                                   "-shared", "-fvisibility=hidden", "-o", mod, mod_src,
                                   "-L", HOST, "-lgnuradio_baz_music", "-Wl,-rpath,$ORIGIN"] + link
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd, cwd=HOST)
    return HOST_LIB


def build_all(force=False, verbose=False):
    build_hip(force, verbose)
    build_host(force, verbose)


if __name__ == "__main__":
    if "--quick" in sys.argv:
        print("built:", build_quick(verbose=True))
    else:
        build_all(force="--force" in sys.argv, verbose=True)
        print("built:", HIP_LIB)
