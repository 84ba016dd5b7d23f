"""gr::baz::fractional_resampler_cc: oracle pins on CPU, HIP parity on the GPU.

PARITY UNPINNED with respect to a real gnuradio-filter (its MMSE tap table is not vendored in gr-baz and not
available offline; oracle/resamp_ref.c regenerates it from the published criterion).  What IS pinned: the reference's
own general_work()/setter source, compiled in place (oracle/_ref), agrees bit for bit with the C restatement, and the
HIP path agrees with both.  Tolerance for the float32 outputs: 1e-5 of the signal scale (measured: identical)."""
import ctypes
import glob
import os
import re

import numpy as np
import pytest

from conftest import GOLDEN_DIR, ROOT
from oracle import resamp_ref as rr

# Rows of gnuradio-filter's interpolator_taps.h AS RECOLLECTED -- there is no copy of that header in this image, and two
# recollections of the outermost tap of rows 1 and 3 exist (round 2 wrote -1.98047e-04 / -5.94874e-04, round 3
# -1.98993e-04 / -5.92100e-04, the values of the closed form; the reviewers' memory is the former).  Neither is a fixture:
# they are kept as CANDIDATES, and nothing below asserts equality with either.  The only pin that counts is the real table
# (tests/golden/mmse_taps_gr37.npz, produced on a GNU Radio host by scripts/dump_gr_mmse_taps.py; absent here).
CANDIDATE_ROWS = {
    1: [np.array([-1.54700e-04, 8.53777e-04, -2.76968e-03, 7.89295e-03, 9.98534e-01, -5.41054e-03, 1.24642e-03, -1.98047e-04]),
        np.array([-1.54700e-04, 8.53777e-04, -2.76968e-03, 7.89295e-03, 9.98534e-01, -5.41054e-03, 1.24642e-03, -1.98993e-04])],
    3: [np.array([-4.64053e-04, 2.56486e-03, -8.34364e-03, 2.39714e-02, 9.95074e-01, -1.59305e-02, 3.69852e-03, -5.94874e-04]),
        np.array([-4.64053e-04, 2.56486e-03, -8.34364e-03, 2.39714e-02, 9.95074e-01, -1.59305e-02, 3.69852e-03, -5.92100e-04])],
}
REAL_TABLE = os.path.join(GOLDEN_DIR, "mmse_taps_gr37.npz")


def resamp_golden():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "resamp_*.npz")))


def load(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


def apply_event(blk, kind, a, b):
    kind = int(kind)
    if kind == 1: blk.set_mu(float(a))
    elif kind == 2: blk.set_resamp_ratio(float(a))
    elif kind == 3: blk.adjust(float(a))
    elif kind == 4: blk.set_resamp_ratio_rational(int(a), int(b))


def run_fixture(blk, g):
    outs, cons, mus = [], [], []
    pos = 0
    for i, c in enumerate(g["calls"]):
        for (ci, kind, a, b) in g["events"]:
            if int(ci) == i:
                apply_event(blk, kind, a, b)
        o, k = blk.work(g["x"][pos:], int(c))
        assert o.shape[-1] == int(c)
        outs.append(o); cons.append(k); mus.append(blk.mu())
        pos += k
    return np.concatenate(outs, axis=-1), np.asarray(cons), np.asarray(mus)


def make(cls, g, **kw):
    return cls(float(g["phase"]), float(g["ratio"]), int(g["num"]), int(g["denom"]), **kw)


# ------------------------------------------------------------------ CPU: the oracle
def test_tap_table_structure_and_closed_form():
    """The DEFAULT table is the closed-form MMSE design (oracle/resamp_ref.c) rounded to six significant digits and then to
    float -- PARITY UNPINNED against gnuradio-filter's own header.  Asserted here: its structure, that it is what the closed
    form gives, and that every recollected candidate row lies within 3e-6 of it (the size of the disagreement between the
    candidates themselves: 9.5e-7 and 2.8e-6) -- which bounds what a wrong default could cost (outputs within ~1e-6 of the
    signal scale), and proves nothing more."""
    t = rr.taps()
    assert t.shape == (129, 8)
    for i, cands in CANDIDATE_ROWS.items():
        for row in cands:
            assert np.max(np.abs(t[i].astype(np.float64) - row)) <= 3e-6, (i, row)
    # every entry is a six-digit decimal: printing it the generator's way and reading it back changes nothing
    assert np.array_equal(np.array([[float("%.5e" % v) for v in r] for r in t.astype(np.float64)]).astype(np.float32), t)
    assert np.array_equal(t[::-1, ::-1], t)                            # taps(1 - mu) = reversed taps(mu)
    assert np.array_equal(t[0], [0, 0, 0, 0, 1, 0, 0, 0]) and np.array_equal(t[128], [0, 0, 0, 1, 0, 0, 0, 0])
    assert np.all(np.abs(t.sum(axis=1) - 1.0) < 4e-4)                  # DC gain of a band-limited design
    # interpolating a slow complex exponential reproduces it: the table is a fractional delay of 3 + mu samples
    n = np.arange(8)
    for i in (0, 17, 64, 100, 128):
        f = 0.05
        got = np.sum(np.exp(2j * np.pi * f * n) * t[i][::-1])
        assert abs(got - np.exp(2j * np.pi * f * (3 + i / 128.0))) < 2e-4


@pytest.mark.skipif(not os.path.exists(REAL_TABLE), reason="no gnuradio-filter table fixture (scripts/dump_gr_mmse_taps.py makes it on a GNU Radio host)")
def test_default_table_equals_gnuradio_filters_bit_for_bit():
    """The day someone runs scripts/dump_gr_mmse_taps.py on a GNU Radio host and commits its output, the resampler becomes
    pinned -- or this fails loudly and the default table must be replaced by the fixture's."""
    z = np.load(REAL_TABLE, allow_pickle=False)
    real = z["taps"]
    assert real.shape == (129, 8) and real.dtype == np.float32
    assert np.array_equal(rr.taps().view(np.uint32), real.view(np.uint32)), \
        "default MMSE table differs from gnuradio-filter's in %d entries (worst %.3e)" % (
            int((rr.taps() != real).sum()), float(np.abs(rr.taps().astype(np.float64) - real).max()))


@pytest.mark.parametrize("name", resamp_golden())
def test_c_oracle_matches_reference_source_vectors(name):
    g = load(name)
    out, cons, mus = run_fixture(make(rr.Resampler, g), g)
    assert np.array_equal(out.view(np.uint32), g["out"].view(np.uint32))
    assert np.array_equal(cons, g["consumed"]) and np.array_equal(mus, g["mu_after"])


@pytest.mark.skipif(not rr.have_ref(), reason="oracle/_ref not built")
def test_reference_source_build_equals_the_restatement_incl_two_input_branch():
    rng = np.random.default_rng(5)
    x = (rng.standard_normal(30000) + 1j * rng.standard_normal(30000)).astype(np.complex64)
    a, b = rr.Resampler(0.25, 1.1), rr.RefResampler(0.25, 1.1)
    ratios = (1.0 + 0.2 * np.sin(np.arange(30000) / 500.0)).astype(np.float32)
    pos = 0
    for n in (100, 3000, 1):
        (oa, ca), (ob, cb) = a.work(x[pos:], n, rr=ratios[pos:]), b.work(x[pos:], n, rr=ratios[pos:])
        assert np.array_equal(oa.view(np.uint32), ob.view(np.uint32)) and ca == cb and a.mu() == b.mu()
        pos += ca
    assert a.forecast(1000) == b.forecast(1000)
    with pytest.raises(ValueError):
        rr.Resampler(0.0, 0.0)
    with pytest.raises(ValueError):
        rr.RefResampler(1.5, 1.0)


def test_abi_library_exports_every_declared_symbol():
    from gr_baz_amd import resamp
    hdr = open(os.path.join(ROOT, "include", "baz_resamp_hip.h")).read()
    declared = sorted(set(re.findall(r"\b(baz_resamp_[a-z0-9_]+)\s*\(", hdr)))
    assert declared == sorted(resamp.SYMBOLS)
    L = ctypes.CDLL(resamp.LIB_PATH)
    for s in declared:
        assert hasattr(L, s), s
    assert resamp.lib().baz_resamp_strerror(-4).decode() == "unsupported configuration"


# ------------------------------------------------------------------ GPU: parity through the C-ABI
def scale_close(a, b, scale, rtol=1e-5):
    return bool(np.all(np.abs(a.real - b.real) <= rtol * scale) and np.all(np.abs(a.imag - b.imag) <= rtol * scale))


@pytest.mark.gpu
@pytest.mark.parametrize("name", resamp_golden())
def test_hip_matches_golden(name, gpu_device):
    from gr_baz_amd import resamp
    g = load(name)
    with make(resamp.Resampler, g) as blk:
        out, cons, mus = run_fixture(blk, g)
        exact = blk.phase_exact()
    assert np.array_equal(cons, g["consumed"])
    assert np.allclose(mus, g["mu_after"], rtol=0, atol=1e-15)
    assert scale_close(out, g["out"], float(np.abs(g["x"]).max()))
    if exact:   # double-typed ratios: the closed-form phase IS the x87 sequence; same float accumulation order
        assert np.array_equal(out.view(np.uint32), g["out"].view(np.uint32))


def test_host_block_reads_the_tap_table_out_of_gnuradio_filters_interpolator():
    """Review r2, missing 2: on a host that has gnuradio-filter the block must interpolate with THAT library's table
    (/root/reference/lib/baz_fractional_resampler_cc.cc:87 constructs gr::filter::mmse_fir_interpolator_cc, :172,203 call
    interpolate()).  The host block reads the table out of that class -- 8 unit impulses x 129 phases through
    interpolate() -- and installs it with baz_resamp_set_taps.  Against the stand-in interpolator of this image (whose
    table is the engine's closed form) the read-out must return that table BIT FOR BIT: any slip in the phase selection
    (mu = imu / 128), the tap reversal or the impulse position would show."""
    from gr_baz_amd import baz, resamp
    t = resamp.default_taps()
    got = baz._native.recover_mmse_taps()
    assert got.shape == (129, 8) and got.dtype == np.float32
    assert np.array_equal(got.view(np.uint32), t.view(np.uint32))
    assert np.array_equal(t.view(np.uint32), rr.taps().view(np.uint32))   # ... and it is the oracle's table, bit for bit
    assert "baz_resamp_set_taps" in resamp.SYMBOLS and "baz_resamp_default_taps" in resamp.SYMBOLS
    assert resamp.lib().baz_resamp_set_taps(None, None) == -1


@pytest.mark.gpu
def test_hip_set_taps_installs_the_hosts_table(gpu_device):
    """baz_resamp_set_taps: after it the engine's outputs are sum_k in[ii + k] * taps[imu][7 - k] with the NEW table, in
    the interpolator's float operation order (multiply, then add, k ascending) -- checked bit for bit against that loop in
    numpy float32 with a table that is not the default one; a table with a NaN is refused and changes nothing."""
    from gr_baz_amd import resamp
    rng = np.random.default_rng(17)
    x = (rng.standard_normal(3000) + 1j * rng.standard_normal(3000)).astype(np.complex64)
    table = (resamp.default_taps().astype(np.float64) * (1.0 + 1e-3 * rng.standard_normal((129, 8)))).astype(np.float32)
    phase, ratio, n = 0.25, 1.3, 2000
    with resamp.Resampler(phase, ratio) as blk:
        base, _ = blk.work(x, n)
        blk.set_mu(phase)
        bad = table.copy()
        bad[5, 3] = np.nan
        with pytest.raises(resamp.ResampError):
            blk.set_taps(bad)
        again, _ = blk.work(x, n)
        assert np.array_equal(again.view(np.uint32), base.view(np.uint32))
        blk.set_taps(table)
        assert np.array_equal(blk.taps().view(np.uint32), table.view(np.uint32))
        blk.set_mu(phase)
        out, consumed = blk.work(x, n)
    # the reference's phase walk (.cc:183-193) in exact rational arithmetic, then the interpolator's loop in float32
    from fractions import Fraction
    mu, inc, ii = Fraction(phase), Fraction(ratio), 0
    want = np.zeros(n, np.complex64)
    for o in range(n):
        imu = int(np.rint(np.float32(float(mu)) * np.float32(128)))
        re = im = np.float32(0)
        for k in range(8):
            w = table[imu, 7 - k]
            re = np.float32(re + np.float32(x[ii + k].real * w))
            im = np.float32(im + np.float32(x[ii + k].imag * w))
        want[o] = re + 1j * im
        s_ = mu + inc
        ii += int(s_)
        mu = s_ - int(s_)
    assert out.shape[0] == n and np.array_equal(out.view(np.uint32), want.view(np.uint32))
    assert not np.array_equal(out.view(np.uint32), base.view(np.uint32))


@pytest.mark.gpu
def test_hip_tap_table_equals_the_oracle_table(gpu_device):
    from gr_baz_amd import resamp
    with resamp.Resampler(0.0, 1.0) as blk:
        t = blk.taps()
    assert np.array_equal(t.view(np.uint32), rr.taps().view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("phase,ratio,num,denom", [(0.0, 1.0, 0, 0), (1.0, 2.5, 0, 0), (0.37, 0.0123, 0, 0),
                                                   (0.0, 3.999999, 0, 0), (0.2, 0.0, 160, 147), (0.0, 17.3, 0, 0)])
def test_hip_multistream_chunked_equals_oracle(phase, ratio, num, denom, gpu_device):
    """16 antenna streams in one context, irregular call sizes, one oracle instance per stream."""
    from gr_baz_amd import resamp
    rng = np.random.default_rng(11)
    S, L = 16, 40000
    x = (rng.standard_normal((S, L)) + 1j * rng.standard_normal((S, L))).astype(np.complex64)
    eff = num / denom if denom else ratio
    calls = [int(c) for c in (1, 255, 256, 257, 5000, 3, 12000) if c * eff + 16 < L / 3]
    with resamp.Resampler(phase, ratio, num, denom, nstreams=S) as blk:
        oracles = [rr.Resampler(phase, ratio, num, denom) for _ in range(S)]
        pos = 0
        for c in calls:
            out, k = blk.work(x[:, pos:], c)
            assert out.shape == (S, c)
            for s in range(S):
                o, ks = oracles[s].work(x[s, pos:], c)
                assert ks == k
                if blk.phase_exact():
                    assert np.array_equal(out[s].view(np.uint32), o.view(np.uint32))
                else:   # 64-bit quotient ratio: imu may differ where mu*128 sits on a rounding boundary (none expected here)
                    assert scale_close(out[s], o, float(np.abs(x).max()))
            pos += k
            assert abs(blk.mu() - oracles[0].mu()) < 1e-15


@pytest.mark.gpu
def test_hip_short_input_produces_what_fits_and_device_path(gpu_device):
    import torch
    from gr_baz_amd import resamp
    rng = np.random.default_rng(3)
    x = (rng.standard_normal(1000) + 1j * rng.standard_normal(1000)).astype(np.complex64)
    with resamp.Resampler(0.0, 1.5) as blk:
        out, k = blk.work(x[:100], 500)          # only floor((100 - 8) / 1.5) + 1 = 62 outputs fit
        o, ko = rr.Resampler(0.0, 1.5).work(x, out.shape[0])
        assert out.shape[0] == 62 and k == ko and np.array_equal(out.view(np.uint32), o.view(np.uint32))
        assert blk.work(x[:7], 10)[0].shape[0] == 0
    with resamp.Resampler(0.5, 0.8, nstreams=2) as blk:
        xd = torch.from_numpy(np.stack([x, x[::-1].copy()]).view(np.float32)).to(gpu_device)
        od = torch.zeros(2, 2 * 1100, dtype=torch.float32, device=gpu_device)
        torch.cuda.synchronize()                       # the engine runs on its own stream: fills first
        n, k = blk.process_device(xd.data_ptr(), 1000, 1000, od.data_ptr(), 1100, 1100)
        blk.sync()
        got = od.cpu().numpy().view(np.complex64)[:, :n]
        for s, xs in enumerate((x, x[::-1].copy())):
            o, ko = rr.Resampler(0.5, 0.8).work(xs, n)
            assert ko == k and np.array_equal(got[s].view(np.uint32), o.view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("S,phase,lo,hi,calls", [(1, 0.0, 0.9, 1.6, (500, 37, 1200)), (3, 0.37, 0.25, 0.75, (2000, 999)),
                                                 (2, 0.5, 2.0, 9.5, (300, 300)), (1, 0.1, 1.0, 1.0, (64,))])
def test_hip_two_input_branch_matches_the_oracle(S, phase, lo, hi, calls, gpu_device):
    """Second input = the ratio per input sample (.cc:205-217): same outputs, consume counts, mu and ratio state as the
    sequential loop, bit for bit, over stateful call sequences (float ratios add exactly in 64.64 fixed point)."""
    from gr_baz_amd import resamp
    rng = np.random.default_rng(77)
    L = int(sum(calls) * hi) + 64
    x = (rng.standard_normal((S, L)) + 1j * rng.standard_normal((S, L))).astype(np.complex64)
    ratio = rng.uniform(lo, hi, L).astype(np.float32)
    with resamp.Resampler(phase, 1.0, nstreams=S) as blk:
        oracles = [rr.Resampler(phase, 1.0) for _ in range(S)]
        pos = 0
        for c in calls:
            out, k = blk.work(x[:, pos:], c, rr=ratio[pos:])
            assert out.shape == (S, c)
            for s in range(S):
                o, ks = oracles[s].work(x[s, pos:], c, rr=ratio[pos:])
                assert ks == k and np.array_equal(out[s].view(np.uint32), o.view(np.uint32))
            pos += k
            assert blk.mu() == oracles[0].mu() and blk.resamp_ratio() == oracles[0].resamp_ratio()
        assert blk.phase_exact()


@pytest.mark.gpu
def test_hip_two_input_branch_stops_at_the_end_of_the_input_and_at_unusable_ratios(gpu_device):
    import torch
    from gr_baz_amd import resamp
    rng = np.random.default_rng(5)
    x = (rng.standard_normal(4000) + 1j * rng.standard_normal(4000)).astype(np.complex64)
    ratio = np.full(4000, 1.5, np.float32)
    with resamp.Resampler(0.0, 1.0) as blk:
        out, k = blk.work(x[:100], 500, rr=ratio[:100])       # only floor((100 - 8) / 1.5) + 1 = 62 outputs fit
        o, ko = rr.Resampler(0.0, 1.0).work(x, 62, rr=ratio)
        assert out.shape[0] == 62 and k == ko and np.array_equal(out.view(np.uint32), o.view(np.uint32))
    bad = ratio.copy()
    bad[30] = np.nan                                          # index 30 is read after output 20 (ii = 30)
    with resamp.Resampler(0.0, 1.0) as blk:
        out, k = blk.work(x, 500, rr=bad)
        o, _ = rr.Resampler(0.0, 1.0).work(x, 21, rr=ratio)
        assert out.shape[0] == 21 and k == 30 and np.array_equal(out.view(np.uint32), o.view(np.uint32))
        # the scheduler would now call again with the window advanced by 30: it starts ON the unusable sample.  That is an
        # error, not a one-output / zero-consumed loop
        with pytest.raises(resamp.ResampError):
            blk.work(x[30:], 500, rr=bad[30:])
    # a control stream that BEGINS with an unusable sample: the first call delivers what the reference would (its one output,
    # nothing consumed); only the call after it -- same window, nothing consumed -- is the loop that must be an error
    first_bad = ratio.copy()
    first_bad[0] = np.nan
    with resamp.Resampler(0.0, 1.0) as blk:
        out, k = blk.work(x, 500, rr=first_bad)
        assert out.shape[0] <= 1 and k == 0
        with pytest.raises(resamp.ResampError):
            blk.work(x, 500, rr=first_bad)
    tiny = np.full(4000, 2.0 ** -20, np.float32)              # valid tiny ratios (< 2^-11) keep running, like the reference
    with resamp.Resampler(0.0, 1.0) as blk:
        out, k = blk.work(x, 300, rr=tiny)
        o, ko = rr.Resampler(0.0, 1.0).work(x, 300, rr=tiny)
        assert out.shape[0] == 300 and k == ko == 0 and np.array_equal(out.view(np.uint32), o.view(np.uint32))
    with resamp.Resampler(0.25, 1.0, nstreams=2) as blk:      # device buffers
        xd = torch.from_numpy(np.stack([x, x[::-1].copy()]).view(np.float32)).to(gpu_device)
        rd = torch.from_numpy(ratio).to(gpu_device)
        od = torch.zeros(2, 2 * 1000, dtype=torch.float32, device=gpu_device)
        torch.cuda.synchronize()
        n, k = blk.process2_device(xd.data_ptr(), 4000, 4000, rd.data_ptr(), od.data_ptr(), 1000, 1000)
        got = od.cpu().numpy().view(np.complex64)[:, :n]
        for s, xs in enumerate((x, x[::-1].copy())):
            o, ko = rr.Resampler(0.25, 1.0).work(xs, n, rr=ratio)
            assert n == 1000 and ko == k and np.array_equal(got[s].view(np.uint32), o.view(np.uint32))


@pytest.mark.gpu
def test_hip_rejects_what_the_reference_rejects(gpu_device):
    from gr_baz_amd import resamp
    for args in ((0.0, 0.0), (0.0, -1.0), (-0.1, 1.0), (1.1, 1.0)):
        with pytest.raises(resamp.ResampError):
            resamp.Resampler(*args)
    with pytest.raises(resamp.ResampError):
        resamp.Resampler(0.0, 1e-5)               # below 2^-11: not representable on the 64.64 grid


# ------------------------------------------------------------------ host block (gr::block surface, pybind = SWIG stand-in)
def test_host_block_surface_and_constructor_rules():
    from gr_baz_amd import baz
    assert callable(baz.fractional_resampler_cc)          # swig/baz_swig.i:966
    with pytest.raises((IndexError, ValueError, RuntimeError)):
        baz.fractional_resampler_cc(0.0, 0.0)             # std::out_of_range (.cc:94-95) wins over "no device"
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="gfx950"):
            baz.fractional_resampler_cc(0.0, 1.25)        # no CPU fallback


@pytest.mark.gpu
def test_host_block_general_work_like_the_scheduler(gpu_device, capfd):
    from gr_baz_amd import baz
    g = load("resamp_setters")
    blk = baz.fractional_resampler_cc(float(g["phase"]), float(g["ratio"]))
    assert blk.name() == "fractional_resampler_cc" and blk.input_item_sizes() == [8, 4] and blk.output_item_sizes() == [8]   # .cc:84-85
    ref = rr.Resampler(float(g["phase"]), float(g["ratio"]))
    assert blk.forecast(1000) == ref.forecast(1000) and abs(blk.relative_rate() - 1.0 / float(g["ratio"])) < 1e-12
    pos = 0
    for i, c in enumerate(g["calls"]):
        for (ci, kind, a, b) in g["events"]:
            if int(ci) == i:
                k = int(kind)
                if k == 1: blk.set_mu(float(a))
                elif k == 2: blk.set_resamp_ratio(float(a))
                elif k == 3: blk.handle_adjust(float(a))
                elif k == 4: blk.set_resamp_ratio(int(a), int(b))
        # (the window is not cut to forecast(): a pending ratio increase needs more input than the old ratio forecast;
        # the reference then reads past ninput_items, this block returns fewer items -- checked below)
        produced, out, consumed = blk.general_work(g["x"][pos:], int(c))
        lo = int(np.sum(g["calls"][:i]))
        assert produced == int(c) and consumed == int(g["consumed"][i])
        assert np.array_equal(out.view(np.uint32), g["out"][lo:lo + int(c)].view(np.uint32))
        pos += consumed
    assert abs(blk.mu() - float(g["mu_after"][-1])) < 1e-15 and abs(blk.relative_rate() - 1.0 / 1.5) < 1e-12
    assert "Ratio" in capfd.readouterr().err      # the constructor banner of .cc:92
    blk = baz.fractional_resampler_cc(0.0, 1.0)
    blk.set_resamp_ratio(2.0)
    need = blk.forecast(1000)                      # 1008: still the old ratio, like the reference's forecast
    produced, out, consumed = blk.general_work(g["x"][:need], 1000)
    assert produced == (need - 8) // 2 + 1 and consumed == 2 * produced   # never reads past the window it was given


@pytest.mark.gpu
def test_host_block_second_input_and_msg_port(gpu_device, capfd):
    """The block's full port surface (.cc:84, 101-102): a second float input with the per-sample ratio, and the PMT
    "msg" port whose pair / number / anything-else cases follow handle_msg (.cc:109-139)."""
    from gr_baz_amd import baz
    rng = np.random.default_rng(9)
    x = (rng.standard_normal(6000) + 1j * rng.standard_normal(6000)).astype(np.complex64)
    blk = baz.fractional_resampler_cc(0.2, 1.0)
    assert blk.input_item_sizes() == [8, 4] and blk.input_streams() == (1, 2) and blk.has_msg_port("msg")
    ratio = rng.uniform(0.8, 1.4, 6000).astype(np.float32)
    produced, out, consumed = blk.general_work2(x, ratio, 1500)
    o, k = rr.Resampler(0.2, 1.0).work(x, 1500, rr=ratio)
    assert produced == 1500 and consumed == k and np.array_equal(out.view(np.uint32), o.view(np.uint32))
    assert abs(blk.relative_rate() - 1.0 / blk.resamp_ratio()) < 1e-12         # .cc:215
    # msg port: ppb pair -> set_resamp_ratio((i + frac) / 1e9); number -> adjustment of d * ratio; both deferred to the
    # next general_work like the direct setters (compared with an oracle driven through the same events)
    blk = baz.fractional_resampler_cc(0.0, 1.25)
    ref = rr.RefResampler(0.0, 1.25) if rr.have_ref() else None
    res = rr.Resampler(0.0, 1.25)
    pos = 0
    for step, (kind, a, b) in enumerate((("none", 0, 0), ("ppb", 1100000000, 0.5), ("num", 0.375, 0), ("none", 0, 0))):
        if kind == "ppb":
            blk.post_msg_ppb(a, b)
            res.set_resamp_ratio(float((np.longdouble(a) + np.longdouble(b)) / np.longdouble(1e9)))
            if ref: ref.post_ppb(a, b)
        elif kind == "num":
            blk.post_msg_double(a)
            res.adjust(a)
            if ref: ref.adjust(a)
        produced, out, consumed = blk.general_work(x[pos:], 700)
        o, k = res.work(x[pos:], 700)
        assert produced == 700 and consumed == k and np.array_equal(out.view(np.uint32), o.view(np.uint32))
        if ref:
            o2, k2 = ref.work(x[pos:], 700)
            assert k2 == k and np.array_equal(o2.view(np.uint32), o.view(np.uint32))
        pos += consumed
    capfd.readouterr()
    blk.post_msg_symbol("not-a-number")             # .cc:136-138: caught, reported, ignored
    assert "Failed to handle PMT" in capfd.readouterr().err
    produced, out, consumed = blk.general_work(x[pos:], 100)
    o, k = res.work(x[pos:], 100)
    assert consumed == k and np.array_equal(out.view(np.uint32), o.view(np.uint32))


@pytest.mark.gpu
def test_hip_host_path_with_row_stride_larger_than_the_window(gpu_device):
    """Stream-major buffers whose row stride exceeds the samples offered (a scheduler hands out a window of a larger
    buffer): in_stride > ninput and out_stride > noutput through baz_resamp_process."""
    from gr_baz_amd import resamp
    rng = np.random.default_rng(21)
    S, L, OS = 3, 5000, 4000
    x = (rng.standard_normal((S, L)) + 1j * rng.standard_normal((S, L))).astype(np.complex64)
    out = np.full((S, OS), np.complex64(-7 - 7j), np.complex64)
    with resamp.Resampler(0.25, 1.3, nstreams=S) as blk:
        consumed = ctypes.c_uint64(0)
        f32p = ctypes.POINTER(ctypes.c_float)
        n = resamp.lib().baz_resamp_process(blk._h, x.view(np.float32).ctypes.data_as(f32p), L, 2000,
                                            out.view(np.float32).ctypes.data_as(f32p), OS, 1000, ctypes.byref(consumed))
        assert n == 1000
    for s in range(S):
        o, k = rr.Resampler(0.25, 1.3).work(x[s, :2000], 1000)
        assert k == consumed.value and np.array_equal(out[s, :1000].view(np.uint32), o.view(np.uint32))
        assert np.all(out[s, 1000:] == np.complex64(-7 - 7j))        # nothing written past the outputs produced
