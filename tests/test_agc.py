"""baz_agc_cc: oracle pins on CPU, HIP parity on the GPU.  Tolerance: the parallel scan
re-associates an fp64 recurrence (~1e-15), the outputs are float32 -> 1e-5 relative like the main path;
measured differences are at most 1 ulp_f32."""
import ctypes
import glob
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN_DIR, ROOT
from oracle import agc_ref as ar


def agc_golden():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "agc_*.npz")))


def load(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


def close(a, b, rtol=1e-5):
    a = np.asarray(a); b = np.asarray(b)
    if np.iscomplexobj(a):
        return close(a.real, b.real, rtol) and close(a.imag, b.imag, rtol)
    fin = np.isfinite(b)
    if not np.array_equal(np.isfinite(a), fin):
        return False
    # components of out near a zero crossing are small: bound by the magnitude scale of the pair
    return bool(np.all(np.abs(a[fin] - b[fin]) <= rtol * np.maximum(np.abs(b[fin]), 1e-30) + 1e-37))


def run_calls(blk, x, calls):
    outs, envs, muls = [], [], []
    pos = 0
    for c in calls:
        o, e, m = blk.work(x[pos:pos + c])
        outs.append(o); envs.append(e); muls.append(m)
        pos += c
    return np.concatenate(outs), np.concatenate(envs), np.concatenate(muls)


@pytest.mark.parametrize("name", agc_golden())
def test_c_oracle_matches_reference_source_vectors(name):
    g = load(name)
    out, env, mul = run_calls(ar.Agc(float(g["rate"]), float(g["reference"])), g["x"], g["calls"])
    assert np.array_equal(out, g["out"]) and np.array_equal(env, g["env"]) and np.array_equal(mul, g["mul"])


@pytest.mark.skipif(not ar.have_ref(), reason="oracle/_ref not built")
def test_reference_source_build_is_stateful_like_the_restatement():
    x = (np.arange(1, 3001) * (0.01 + 0.01j)).astype(np.complex64)
    a = ar.Agc(1e-3, 1.0)
    b = ar.Agc(1e-3, 1.0, use_reference_source=True)
    for lo, hi in ((0, 1), (1, 2000), (2000, 3000)):
        ra, rb = a.work(x[lo:hi]), b.work(x[lo:hi])
        assert all(np.array_equal(p, q) for p, q in zip(ra, rb))
    o, e, m = ar.Agc(1e-3, 1.0).work(x[:5])
    assert e[0] == np.float32(abs(x[0]))          # count == 0: env = |x0|  (lib/baz_agc_cc.cc:79-80)
    assert m[0] == np.float32(1.0 / np.float64(np.float32(abs(np.complex128(x[0])))) ) or abs(m[0] * e[0] - 1) < 1e-6


def test_abi_symbols_and_loud_failure():
    from gr_baz_amd import agc
    src = open(os.path.join(ROOT, "include", "baz_agc_hip.h")).read()
    declared = sorted(set(re.findall(r"BAZ_AGC_API\s+[\w\s\*]+?\b(baz_agc_\w+)\s*\(", src)))
    assert declared == sorted(agc.SYMBOLS)
    raw = ctypes.CDLL(agc.LIB_PATH)
    for s in declared:
        assert getattr(raw, s) is not None
    assert "lib/baz_agc_cc.cc:64-102" in src
    L = agc.lib()
    h = ctypes.c_void_p()
    assert L.baz_agc_create(ctypes.byref(h), 0, 1e-4, 1.0, 1.0, 0.0, -1) == -1     # nstreams == 0
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(agc.AgcError):
            agc.Agc()


# --------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("name", agc_golden())
def test_hip_agc_matches_golden(name, gpu_device):
    from gr_baz_amd import agc
    g = load(name)
    with agc.Agc(float(g["rate"]), float(g["reference"])) as blk:
        out, env, mul = run_calls(blk, g["x"], g["calls"])
        assert blk.count == int(g["calls"].sum())
    assert close(env, g["env"]) and close(mul, g["mul"]) and close(out, g["out"])
    # The per-sample arithmetic is the reference's operation for operation; only a tile's entry state comes out of the
    # re-associated scan (a few ulp_f64 off): float32 outputs are the sequential loop's bit for bit, except a 1-ulp flip
    # in < 1e-5 of the samples (measured 8e-7 at rate 1e-4, 6e-8 at 1e-3: profiles/r02_agc_exactness.txt)
    for got, ref in ((np.ascontiguousarray(out).view(np.float32), np.ascontiguousarray(g["out"]).view(np.float32)),
                     (env, g["env"]), (mul, g["mul"])):
        fin = np.isfinite(ref)
        u = np.abs(np.ascontiguousarray(got).view(np.int32).astype(np.int64) - np.ascontiguousarray(ref).view(np.int32).astype(np.int64))[fin]
        assert u.size == 0 or (u.max() <= 1 and np.count_nonzero(u) <= max(1, 1e-5 * u.size))


@pytest.mark.gpu
def test_hip_agc_multi_stream_device_path_and_tail_alignment(gpu_device):
    import torch
    from gr_baz_amd import agc
    S, n, stride = 5, 10007, 10240           # odd length: scalar tail; stride keeps stream starts 16-B aligned
    rng = np.random.default_rng(5)
    x = ((rng.standard_normal((S, n)) + 1j * rng.standard_normal((S, n))) * np.linspace(0.2, 4.0, n)).astype(np.complex64)
    xin = np.zeros((S, stride), np.complex64); xin[:, :n] = x
    with agc.Agc(3e-3, 1.5, nstreams=S) as blk:
        d_in = torch.from_numpy(xin.view(np.float32)).to(gpu_device)
        d_out = torch.zeros_like(d_in)
        d_env = torch.zeros(S, stride, dtype=torch.float32, device=gpu_device)
        d_mul = torch.zeros_like(d_env)
        torch.cuda.synchronize()                       # the engine runs on its own stream: fills first
        for lo, hi in ((0, 4097), (4097, n)):          # second call starts at an odd offset (unaligned float2 pairs)
            blk.process_device(d_in.data_ptr() + lo * 8, hi - lo, stride, d_out.data_ptr() + lo * 8,
                               d_env.data_ptr() + lo * 4, d_mul.data_ptr() + lo * 4)
        blk.sync()
        out = d_out.cpu().numpy().view(np.complex64)[:, :n]
        env = d_env.cpu().numpy()[:, :n]
        mul = d_mul.cpu().numpy()[:, :n]
    for s in range(S):
        o, e, m = ar.Agc(3e-3, 1.5).work(x[s])
        assert close(env[s], e) and close(mul[s], m) and close(out[s], o)


@pytest.mark.gpu
def test_agc_host_block_and_python_surface(gpu_device):
    """baz.agc_cc -> C++ host block -> C-ABI: ports of lib/baz_agc_cc.cc:52-54, optional outputs (:68-69)."""
    from gr_baz_amd import baz
    g = load("agc_stateful_rate1e-2")
    blk = baz.agc_cc(float(g["rate"]), float(g["reference"]))
    assert blk.name() == "gr_agc_cc" and blk.input_item_sizes() == [8] and blk.output_item_sizes() == [8, 4]
    assert blk.output_streams() == (1, 3)
    # requests for large work() calls that cost no sample of a finite capture: look-back, not an output multiple
    assert blk.output_multiple() == 1 and blk.history() == 16385 and blk.min_output_buffer() == 2 * 16384
    outs, envs, muls = [], [], []
    pos = 0
    for c in g["calls"]:
        produced, o, e, m = blk.work(g["x"][pos:pos + c], 3)
        assert produced == c
        outs.append(o); envs.append(e); muls.append(m)
        pos += c
    assert close(np.concatenate(outs), g["out"]) and close(np.concatenate(envs), g["env"]) and close(np.concatenate(muls), g["mul"])
    p1, o1, e1, m1 = baz.agc_cc().work(g["x"][:100], 1)
    assert p1 == 100 and e1 is None and m1 is None


@pytest.mark.gpu
def test_agc_zero_first_sample_and_reset(gpu_device):
    """|x0| = 0 makes env = 0 and gain = inf in the reference (reference/0, lib/baz_agc_cc.cc:89): same here."""
    from gr_baz_amd import agc
    x = np.zeros(300, np.complex64); x[5:] = (1 + 1j)
    o, e, m = ar.Agc(0.1, 1.0).work(x)
    with agc.Agc(0.1, 1.0) as blk:
        o2, e2, m2 = blk.work(x)
        assert close(e2, e) and close(m2, m)
        assert np.array_equal(np.isnan(o2.real), np.isnan(o.real))
        blk.reset()
        assert blk.count == 0
        o3, e3, m3 = blk.work(x[5:])
    assert close(e3, ar.Agc(0.1, 1.0).work(x[5:])[1])


@pytest.mark.gpu
def test_fast_sqrt_and_division_are_the_rounded_ones(gpu_device):
    """Full tiles of ordinary values run lean sqrt / division sequences (agc_kernels.hip.h): the Newton cores of the rounded
    library functions without their range scaling.  Bit for bit the same as __dsqrt_rn / __ddiv_rn on |x|^2 of random float
    samples, on envelope-like divisors, and log-uniformly over the whole range the fast path accepts."""
    import torch
    from gr_baz_amd import agc
    rng = np.random.default_rng(11)
    n = 1 << 22
    xs = (rng.standard_normal((n, 2)) * 10.0 ** rng.uniform(-18, 18, size=(n, 1))).astype(np.float32).astype(np.float64)
    a1 = xs[:, 0] * xs[:, 0] + xs[:, 1] * xs[:, 1]                                  # |x|^2 as the kernel forms it
    b1 = np.sqrt(a1) * rng.uniform(0.5, 2.0, size=n)                                 # envelope-like
    a2 = np.ldexp(rng.uniform(1.0, 2.0, size=n), rng.integers(-399, 399, size=n))    # the whole accepted range
    b2 = np.ldexp(rng.uniform(1.0, 2.0, size=n), rng.integers(-399, 399, size=n))
    a3 = np.full(n, 1.0); b3 = np.ldexp(rng.uniform(1.0, 2.0, size=n), rng.integers(-60, 60, size=n))   # reference / env
    a4 = np.nextafter(np.ldexp(1.0, rng.integers(-300, 300, size=n)), rng.choice([0.0, np.inf], size=n))  # around powers of two
    with agc.Agc(1e-4, 1.0) as blk:
        for a, b in ((a1, b1), (a2, b2), (a3, b3), (a4, b2), (a2, a4)):
            da = torch.from_numpy(np.ascontiguousarray(a)).to(gpu_device)
            db = torch.from_numpy(np.ascontiguousarray(b)).to(gpu_device)
            torch.cuda.synchronize()
            assert blk.debug_selfcheck(da.data_ptr(), db.data_ptr(), n) == (0, 0)


@pytest.mark.gpu
def test_fast_and_general_tile_paths_agree(gpu_device, monkeypatch):
    """BAZ_AGC_FAST=0 keeps every tile on the general path.  The two builds differ only in how a tile's ENTRY state is
    re-associated (DPP scan with host-computed powers against the (A, S) pair scan): float32 outputs identical except a
    1-ulp flip in < 1e-5 of the samples, like either against the sequential loop; zeros in the stream (general path for
    those tiles) and a ragged tail included."""
    import torch
    from gr_baz_amd import agc
    S, n = 3, 256 * 40 + 77
    rng = np.random.default_rng(3)
    x = ((rng.standard_normal((S, n)) + 1j * rng.standard_normal((S, n))) * np.linspace(0.1, 5.0, n)).astype(np.complex64)
    x[1, 1000:1300] = 0                                   # a zero run: |x|^2 = 0 is not "ordinary"
    outs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("BAZ_AGC_FAST", mode)
        with agc.Agc(1e-3, 2.0, nstreams=S) as blk:
            d_in = torch.from_numpy(x.view(np.float32)).to(gpu_device)
            d_out = torch.zeros_like(d_in)
            d_env = torch.zeros(S, n, dtype=torch.float32, device=gpu_device)
            d_mul = torch.zeros_like(d_env)
            d_items = torch.zeros(n, S, 2, dtype=torch.float32, device=gpu_device)
            torch.cuda.synchronize()
            blk.process_device(d_in.data_ptr(), n, n, d_out.data_ptr(), d_env.data_ptr(), d_mul.data_ptr())
            blk.sync()
            blk.reset()
            blk.process_device_interleaved(d_in.data_ptr(), n, n, d_items.data_ptr())
            blk.sync()
            outs[mode] = [t.cpu().numpy() for t in (d_out, d_env, d_mul, d_items)]
    for got, ref in zip(outs["1"], outs["0"]):
        fin = np.isfinite(ref)
        u = np.abs(got.view(np.int32).astype(np.int64) - ref.view(np.int32).astype(np.int64))[fin]
        assert u.max() <= 1 and np.count_nonzero(u) <= max(1, 1e-5 * u.size)
    assert np.array_equal(outs["1"][3].reshape(n, S, 2).transpose(1, 0, 2).reshape(S, -1), outs["1"][0].reshape(S, -1))   # both output forms agree
    for s in range(S):
        o, e, m = ar.Agc(1e-3, 2.0).work(x[s])
        assert close(outs["1"][1][s], e) and close(outs["1"][0][s].view(np.complex64), o)
