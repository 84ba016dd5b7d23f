"""The drop-in surface above the C-ABI: baz.music_doa (SWIG stand-in), the C++ gr::sync_block host
block and music_doa_helper.  CPU part: construction-time behaviour that needs no device.  GPU part:
work() driven like the GNU Radio scheduler drives it, against the golden vectors."""
import numpy as np
import pytest

from conftest import load_golden
from helpers import assert_doa_match, assert_spectrum_close
from oracle import music_oracle as mo


def _baz():
    from gr_baz_amd import baz
    return baz


def test_python_surface_names_match_the_reference():
    baz = _baz()
    # swig/baz_swig.i:562-571 and python/music_doa_helper.py:29,32,48,100
    assert callable(baz.music_doa)
    assert hasattr(baz.baz_music_doa_sptr, "set_array_response")
    h = baz.music_doa_helper
    for name in ("unit_vect", "calculate_antenna_array_response", "music_doa_helper"):
        assert hasattr(h, name)
    assert hasattr(h.music_doa_helper, "set_frequency")
    import inspect
    sig = inspect.signature(h.music_doa_helper.__init__)
    assert list(sig.parameters)[1:] == ["m", "n", "nsamples", "angular_resolution", "frequency",
                                        "array_spacing", "antenna_array", "output_spectrum"]
    assert sig.parameters["output_spectrum"].default is False


def test_helper_table_is_bit_identical_to_the_reference_loop():
    """The vectorised table must round to the same complex64 values as the reference's double loop
    (restated in oracle.music_oracle.calculate_antenna_array_response)."""
    h = _baz().music_doa_helper
    for arr, res, l in (([[0, 0], [0.5, 0], [0.5, 0.5], [0, 0.5]], 3600, 1.0),
                        ([[0, 0], [1, 0], [2, 0], [3, 0]], 360, 2.0),
                        (mo.scaled_array(mo.array_geometry(8), 0.5), 36000, 1.0)):
        a = np.array(h.calculate_antenna_array_response(arr, res, l))
        b = np.array(mo.calculate_antenna_array_response(arr, res, l))
        assert np.array_equal(a, b)
    assert np.array_equal(h.unit_vect(0.3), mo.unit_vect(0.3))


def test_helper_table_fast_path_and_its_fallback(monkeypatch):
    """One numpy.inner over all (step, antenna) pairs forms the phases (a retune at config 3: 0.4 - 1.1 s -> 0.03 - 0.09 s); a sample is recomputed the
    reference's way and a single differing bit sends the whole table through the per-element loop: either way the complex128 table equals the loop's."""
    h = _baz().music_doa_helper
    rng = np.random.default_rng(77)
    cases = [((rng.random((m, 2)) * 3.0).tolist(), res, float(rng.uniform(0.3, 2.0))) for m, res in ((4, 3600), (8, 9000), (16, 360), (5, 1000), (3, 361), (2, 1), (4, 2))]
    want = [np.array(mo.calculate_antenna_array_response(arr, res, l)) for arr, res, l in cases]
    calls = []
    loop = h._phase_by_element
    monkeypatch.setattr(h, "_phase_by_element", lambda a, r, l, steps=None: (calls.append(steps is None), loop(a, r, l, steps))[1])
    for (arr, res, l), w in zip(cases, want):
        got = np.array(h.calculate_antenna_array_response(arr, res, l))
        assert np.array_equal(got.view(np.float64), w.view(np.float64))
    fell_back = any(calls)       # legitimate on a BLAS whose blocked product rounds differently (the table is right either way); not seen so far
    # a BLAS that rounds the blocked product differently: the sample check notices, the loop takes over
    real_inner = np.inner
    monkeypatch.setattr(h.numpy, "inner", lambda a, b: real_inner(a, b) * (1.0 + 2.0 ** -52) if np.ndim(a) == 2 else real_inner(a, b))
    calls.clear()
    for (arr, res, l), w in zip(cases[:3], want[:3]):
        got = np.array(h.calculate_antenna_array_response(arr, res, l))
        assert np.array_equal(got.view(np.float64), w.view(np.float64))
    assert calls.count(True) == 3
    # a BLAS that rounds only SOME rows of the matrix product differently (a block tail, a thread partition) -- rows the 64-row sample
    # does not visit -- must not get through either: every element is compared with a restatement of the rounding the sample shows
    monkeypatch.setattr(h.numpy, "inner", real_inner)
    arr, res, l = cases[1]
    sample = set([0, res - 1] + [(k * 2654435761) % res for k in range(1, 63)])
    odd_rows = [r for r in range(res) if r not in sample][100:103]

    def partly_off(a, b):
        v = real_inner(a, b)
        if np.ndim(a) == 2:
            v = v.copy()
            v[odd_rows, 0] *= (1.0 + 2.0 ** -52)
        return v
    monkeypatch.setattr(h.numpy, "inner", partly_off)
    calls.clear()
    got = np.array(h.calculate_antenna_array_response(arr, res, l))
    assert np.array_equal(got.view(np.float64), want[1].view(np.float64))
    assert calls.count(True) == (0 if fell_back else 1) or calls.count(True) == 1
    # ... and the switch that forces the reference's loop
    monkeypatch.setattr(h.numpy, "inner", real_inner)
    monkeypatch.setenv("BAZ_MUSIC_HELPER_PER_ELEMENT", "1")
    calls.clear()
    got = np.array(h.calculate_antenna_array_response(*cases[0]))
    assert np.array_equal(got.view(np.float64), want[0].view(np.float64)) and calls == [True]
    if fell_back:
        pytest.skip("this BLAS rounds numpy.inner over a matrix differently from the per-element call: the helper used its per-element loop")


@pytest.mark.parametrize("args", [
    dict(m=0, n=1, nsamples=8, res=4), dict(m=4, n=0, nsamples=8, res=4), dict(m=4, n=4, nsamples=8, res=4),
    dict(m=4, n=2, nsamples=10, res=4), dict(m=4, n=2, nsamples=0, res=4), dict(m=4, n=2, nsamples=8, res=0),
])
def test_make_rejects_what_the_reference_asserts(args):
    """lib/baz_music_doa.cc:45-50 are assert()s (no-ops in Release); make() raises instead.  GNU Radio
    surfaces C++ exceptions from make() as python exceptions, pybind11 maps std::invalid_argument
    to ValueError."""
    table = [[0j] * max(args["m"], 1)] * max(args["res"], 1)
    with pytest.raises(ValueError):
        _baz().music_doa(args["m"], args["n"], args["nsamples"], table, args["res"])


def test_make_rejects_misshaped_table():
    import torch
    if not torch.cuda.is_available():
        # table shape is checked in the constructor, after the (cheap) scalar checks
        with pytest.raises((ValueError, RuntimeError)):
            _baz().music_doa(4, 2, 8, [[0j] * 4] * 3, 4)
    else:
        with pytest.raises(ValueError):
            _baz().music_doa(4, 2, 8, [[0j] * 4] * 3, 4)
        with pytest.raises(ValueError):
            _baz().music_doa(4, 2, 8, [[0j] * 3] * 4, 4)


def test_helper_checks_nsamples_like_the_reference():
    with pytest.raises(Exception, match="nsamples must be multiple of m"):     # music_doa_helper.py:58-59
        _baz().music_doa_helper.music_doa_helper(4, 2, 1022, 360, 1e9, 0.15, [[0, 0], [1, 0], [1, 1], [0, 1]])


def test_no_device_is_a_loud_runtime_error():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="gfx950"):
        _baz().music_doa(4, 2, 8, [[0j] * 4] * 4, 4)


# --------------------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("name", ["cfg1_m4_n2_N256_r360", "grc_default_ula", "m5_n3_N1000_r720", "cfg3_m8_n2_N4096_r36000",
                                  "cfg5_m16_n2_N4096_r3600", "wide_m17_n2_N816_r360", "wide_m33_n32_N2112_r90"])
def test_host_block_work_matches_golden(name, gpu_device, capfd):
    g = load_golden(name)
    blk = _baz().music_doa(g["m"], g["n"], g["nsamples"], [list(map(complex, r)) for r in g["table"]], g["res"])
    err = capfd.readouterr().err
    assert "MUSIC DOA: M: %d, N: %d, # samples: %d, angular resolution: %d" % (g["m"], g["n"], g["nsamples"], g["res"]) in err
    assert blk.name() == "music_doa"
    assert blk.input_item_sizes() == [8 * g["nsamples"]]                        # lib/baz_music_doa.cc:37
    assert blk.output_item_sizes() == [4 * g["n"], 4 * g["n"], 4 * g["res"]]    # :38
    assert blk.output_streams() == (1, 3)
    produced, ang, lvl, spec = blk.work(g["items"], 3)
    assert produced == g["items"].shape[0]       # all noutput_items handled in one call
    assert_spectrum_close(spec, g["spectrum"])
    assert_doa_match(ang, lvl, g["ang"], g["lvl"], g["res"], g["strength64"])
    # one-item calls (how the reference is driven, .cc:160) give the same stream
    p1, a1, l1, s1 = blk.work(g["items"][0], 3)
    assert p1 == 1 and np.array_equal(a1[0], ang[0]) and np.array_equal(s1[0], spec[0])
    # only port 0 wired: the reference would dereference a NULL lvl (.cc:147-154)
    p0, a0, l0, s0 = blk.work(g["items"], 1)
    assert p0 == produced and l0 is None and s0 is None and np.array_equal(a0, ang)


@pytest.mark.gpu
def test_scheduler_hints_are_requests_not_caps(gpu_device, monkeypatch):
    """the block asks for large work() calls WITHOUT a minimum call size or a cap: output multiple 1 (every
    item of a finite stream is processed, like the reference: lib/baz_music_doa.cc:160 consumes what it is given), look-back
    items it never reads (history H + 1: the runtime sizes the input buffer for 2 (H + 2) items) and output buffers of 2 H."""
    tab = [[1 + 0j] * 4] * 8
    for k in ("BAZ_MUSIC_OUTPUT_MULTIPLE", "BAZ_MUSIC_MIN_OUTPUT_BUFFER", "BAZ_MUSIC_MAX_NOUTPUT", "BAZ_MUSIC_INPUT_LOOKBACK"):
        monkeypatch.delenv(k, raising=False)
    blk = _baz().music_doa(4, 2, 16, tab, 8)
    assert blk.output_multiple() == 1 and blk.history() == 2049 and blk.min_output_buffer() == 4096 and blk.max_noutput_items() == 0
    monkeypatch.setenv("BAZ_MUSIC_OUTPUT_MULTIPLE", "256")       # the round-2 way stays available (a minimum call size)
    monkeypatch.setenv("BAZ_MUSIC_INPUT_LOOKBACK", "0")
    monkeypatch.setenv("BAZ_MUSIC_MAX_NOUTPUT", "4096")
    blk = _baz().music_doa(4, 2, 16, tab, 8)
    assert blk.output_multiple() == 256 and blk.history() == 1 and blk.min_output_buffer() == 2048 and blk.max_noutput_items() == 4096
    monkeypatch.setenv("BAZ_MUSIC_OUTPUT_MULTIPLE", "1")
    monkeypatch.setenv("BAZ_MUSIC_MIN_OUTPUT_BUFFER", "0")
    monkeypatch.delenv("BAZ_MUSIC_MAX_NOUTPUT")
    blk = _baz().music_doa(4, 2, 16, tab, 8)          # the reference's behaviour: no hints at all
    assert blk.output_multiple() == 1 and blk.history() == 1 and blk.min_output_buffer() == -1 and blk.max_noutput_items() == 0


@pytest.mark.gpu
def test_finite_stream_is_processed_to_its_last_item(gpu_device, monkeypatch):
    """with an output multiple N a finite source loses its last < N items and a capture shorter than N gives
    nothing.  The default hints (look-back, multiple 1) lose nothing: 1, 7 and 2,500 items through the scheduler model
    come out complete and equal to plain work() calls; the round-2 hint (multiple 64) drops the tail, as documented."""
    from conftest import load_golden
    g = load_golden("cfg1_m4_n2_N256_r360")
    from gr_baz_amd import baz
    for k in ("BAZ_MUSIC_OUTPUT_MULTIPLE", "BAZ_MUSIC_MIN_OUTPUT_BUFFER", "BAZ_MUSIC_MAX_NOUTPUT", "BAZ_MUSIC_INPUT_LOOKBACK"):
        monkeypatch.delenv(k, raising=False)
    table = [list(map(complex, r)) for r in g["table"]]
    blk = baz.music_doa(g["m"], g["n"], g["nsamples"], table, g["res"])
    base = g["items"]
    for count in (1, 7, 2500):
        items = np.concatenate([base] * (count // base.shape[0] + 1))[:count]
        st, ang, lvl, spec = blk.run_flowgraph(items, 3, True, False)
        assert st["items"] == count and st["dropped_at_end"] == 0 and st["last_return"] > 0
        p, a1, l1, s1 = blk.work(items, 3)
        assert p == count and np.array_equal(a1, ang) and np.array_equal(l1, lvl) and np.array_equal(s1, spec)
        if count == 2500:
            assert max(st["call_sizes"]) == 2048 and st["in_bufsize"] >= 2 * 2050       # calls of H items, input sized by the look-back
    monkeypatch.setenv("BAZ_MUSIC_OUTPUT_MULTIPLE", "64")
    monkeypatch.setenv("BAZ_MUSIC_INPUT_LOOKBACK", "0")
    blk = baz.music_doa(g["m"], g["n"], g["nsamples"], table, g["res"])
    items = np.concatenate([base] * (100 // base.shape[0] + 1))[:100]
    st, ang, lvl, spec = blk.run_flowgraph(items, 3, True, False)
    assert st["items"] == 64 and st["dropped_at_end"] == 36


@pytest.mark.gpu
def test_helper_end_to_end_and_retune(gpu_device, capfd):
    """music_doa_helper -> baz.music_doa -> host block -> C-ABI -> HIP, incl. set_frequency
    (grc/baz_music_doa.xml:7-8)."""
    h = _baz().music_doa_helper
    arr = mo.array_geometry(4)
    hb = h.music_doa_helper(m=4, n=2, nsamples=256, angular_resolution=360, frequency=mo.FREQUENCY,
                            array_spacing=mo.SPACING, antenna_array=arr, output_spectrum=True)
    out = capfd.readouterr()
    assert "MUSIC DOA Helper: M: 4, N: 2, # samples: 256" in out.out
    items = mo.synth_items(20, 4, 256, arr, mo.FREQUENCY, mo.SPACING, seed=5)
    ang, lvl, spec = hb.work(items)
    ao, lo, so, st = mo.music_doa_work_batch(items, mo.steering_table_c64(arr, 360, mo.FREQUENCY, mo.SPACING), 4, 2)
    assert_spectrum_close(spec, so)
    assert_doa_match(ang, lvl, ao, lo, 360, st)
    hb.set_frequency(0.75 * mo.FREQUENCY)
    assert "Updating array response" in capfd.readouterr().err
    ang2, lvl2, spec2 = hb.work(items)
    _, _, so2, _ = mo.music_doa_work_batch(items, mo.steering_table_c64(arr, 360, 0.75 * mo.FREQUENCY, mo.SPACING), 4, 2)
    assert_spectrum_close(spec2, so2)
    hb2 = h.music_doa_helper(4, 2, 256, 360, mo.FREQUENCY, mo.SPACING, arr)          # 2-output form
    a2, l2 = hb2.work(items)
    assert np.array_equal(a2, ang)
