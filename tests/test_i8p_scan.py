"""The level-packed int8 scan for 2 .. 4 antennas (gr_baz_amd/csrc/scan_i8p_kernels.hip.h; /root/reference/lib/baz_music_doa.cc:101-121).
LAB BUILD ONLY: measured in round 5 and not shipped (a negative result with its evidence, profiles/r05_i8p_negative.txt).

Same integer forms and bounds as the 6 .. 16-antenna scan (tests/test_i8_scan.py), with the digit pairs of a level laid side by
side along K (4 MFMAs per tile for levels 0 .. 3) and scan_mfma_kernel's row classes.  Pinned here: the packed operands against a
numpy restatement (CPU), and on the GPU the scan against the fp64 scan of the same build and the oracle over row-class shapes
(res % 64 = 0 / 16 / 8 / odd), 0 .. 120 dB, coherent and incoherent batches, every bound on every (item, bin) on the hardware,
poisoned items, an item's bits independent of its wave-mates and of the bin-range split, arbitrary tables and a table swap."""
import numpy as np
import pytest

from helpers import assert_doa_match, assert_spectrum_close
from oracle import music_oracle as mo
from test_i8_scan import _assert_same_choice, _both, _capi, _run, _scene


@pytest.fixture(autouse=True)
def _lab_build_with_the_packed_scan(monkeypatch):
    """The packed scan is NOT shipped (profiles/r05_i8p_negative.txt: no faster than the fp64 scan, whose time is the spectrum
    stores'; incoherent batches 2.4 x slower): it exists in the lab build of the library only, behind BAZ_MUSIC_I8P=1."""
    monkeypatch.setenv("BAZ_MUSIC_LAB_LIB", "lab")
    monkeypatch.setenv("BAZ_MUSIC_I8P", "1")


# ---------------------------------------------------------------------------------------------------------- CPU: the operands
@pytest.mark.parametrize("m,res", [(4, 130), (3, 77), (2, 64)])
def test_packed_operands_equal_the_numpy_restatement(m, res):
    capi = _capi()
    t = mo.steering_table_c64(mo.array_geometry(m), res, mo.FREQUENCY, mo.SPACING) * np.float32(1.7)
    t = t.astype(np.complex64)
    img = capi.debug_host_table_image(m, 1, res, t, 8)
    assert img is not None
    steps = (res + 63) // 64
    units = (steps + 2) * 256
    assert img.size == 2 * units * 16
    B = img.reshape(2, steps + 2, 4, 4, 16, 16).view(np.int8)          # [operand][step + 1][tile][slot][column][term]
    # F and its integer image, as the library defines them (music_kernels.hip.h 4., scan_i8_kernels.hip.h)
    a = t.astype(np.complex128)
    F = np.zeros((res, m * m))
    for r in range(m):
        for c in range(m):
            if r == c:
                F[:, r * m + c] = a[:, r].real ** 2 + a[:, r].imag ** 2          # (not |a|^2 through hypot: one rounding, like the library)
            elif r < c:
                F[:, r * m + c] = (np.conj(a[:, r]) * a[:, c]).real
            else:
                F[:, r * m + c] = (np.conj(a[:, c]) * a[:, r]).imag
    fscale = 2.0 ** np.frexp(np.abs(F).max() / 1.0009765625)[1]          # the power of two with max|F| / Fscale in (0.5 QMAX, QMAX]
    Fi = np.rint(F * (2.0 ** 54 / fscale)).astype(np.int64)
    rng = np.random.default_rng(1)
    for b in list(rng.integers(0, res, size=40)) + [0, res - 1]:
        st, w = divmod(int(b), 64)
        c, tt = divmod(w, 4)
        for e in range(m * m):
            digs = [int(B[0, st + 1, tt, s, c, e]) for s in range(4)] + [int(B[1, st + 1, tt, s, c, e]) for s in range(3)]
            assert all(-128 <= d <= 127 for d in digs[1:]) and abs(digs[0]) <= 65
            assert sum(d * 256 ** (6 - s) for s, d in enumerate(digs)) == int(Fi[b, e]), (b, e)
        assert not B[1, st + 1, tt, 3, c].any()                          # slot 3 of B' is empty
    assert not B[:, 0].any() and not B[:, steps + 1].any()               # the padded steps


# ---------------------------------------------------------------------------------------------------------- GPU
#          m  n  nsamples res  batch     res % 64 -> row classes
SHAPES = [(4, 2, 1024, 3600, 150),    # cfg2: 16 -> 4 classes
          (4, 2, 256, 360, 300),      # cfg1: 40 -> 8 classes
          (4, 1, 64, 4096, 70),       # 0 -> 1 class
          (4, 3, 512, 1000, 130),     # lists of 4 keys; 40 -> 8 classes
          (4, 2, 64, 77, 200),        # res % 4 != 0: dword stores, one class
          (3, 2, 300, 357, 90),       # odd everything
          (3, 1, 96, 720, 129),
          (2, 1, 64, 1444, 65)]


@pytest.mark.gpu
@pytest.mark.parametrize("incoherent", [False, True])
@pytest.mark.parametrize("snr", [0.0, 20.0, 40.0, 80.0, 120.0])
@pytest.mark.parametrize("m,n,nsamples,res,batch", SHAPES)
def test_packed_int8_scan_against_the_fp64_scan_and_the_oracle(m, n, nsamples, res, batch, snr, incoherent, gpu_device, monkeypatch):
    table, items = _scene(m, n, nsamples, res, batch, snr, 7000 + int(snr) + 17 * m + n, incoherent)
    (a_i, l_i, s_i, st, r_i), (a_x, l_x, s_x, _, r_x) = _both(monkeypatch, m, n, nsamples, res, table, items, gpu_device)
    assert (s_i >= 0).all() and (a_i >= 0).all() and (l_i >= 0).all()                     # every value written
    # 7.5e-7 on d by construction + 1.3e-7 of the float32 combination; both scans then convert and take the reciprocal the same way
    worst = assert_spectrum_close(s_i, s_x, rtol=1.2e-6, what="packed int8 vs fp64 scan")
    _assert_same_choice(a_i, a_x, s_x, res)
    bins = np.round(a_i.astype(np.float64) * res / 360.0).astype(np.int64) % res
    assert np.array_equal(l_i.view(np.uint32), np.take_along_axis(s_i, bins, axis=1).view(np.uint32)), "lvl != spectrum[bin] (.cc:153)"
    refined_tiles, tiles = st
    assert tiles > 0
    if snr >= 80.0:
        assert r_i == r_x                                  # the literal form recomputes the same near-null values in both
    if snr <= 40.0:
        ao, lo, so, s64 = mo.music_doa_work_batch(items, table, m, n)
        w2 = assert_spectrum_close(s_i, so, what="packed int8 scan vs oracle")
        assert w2 <= 1.2e-6, w2
        assert_doa_match(a_i, l_i, ao, lo, res, s64)
    print("m=%d n=%d res=%d snr=%g %s: worst vs fp64 scan %.3g, refined tiles %d of %d" %
          (m, n, res, snr, "incoherent" if incoherent else "coherent", worst, refined_tiles, tiles))


@pytest.mark.gpu
def test_the_headline_shape_runs_the_packed_scan_and_mostly_its_first_tier(gpu_device, monkeypatch):
    m, n, N, res, batch = 4, 2, 1024, 3600, 512
    arr = mo.array_geometry(m)
    table = mo.steering_table_c64(arr, res, mo.FREQUENCY, mo.SPACING)
    items = mo.synth_items(batch, m, N, arr, mo.FREQUENCY, mo.SPACING, snr_db=20.0, seed=1002)
    monkeypatch.setenv("BAZ_MUSIC_EXACT", "0")
    with _capi().Context(m, n, N, res, table) as ctx:
        assert ctx.uses_i8_scan()
        a, l, s = _run(ctx, items, gpu_device)
        assert "scan_i8p_kernel" in ctx.stage_name(2)
        refined_tiles, tiles = ctx.debug_i8_stats()
    ao, lo, so, s64 = mo.music_doa_work_batch(items, table, m, n)
    assert assert_spectrum_close(s, so) <= 1.2e-6
    assert_doa_match(a, l, ao, lo, res, s64)
    assert tiles > 0 and refined_tiles < 0.05 * tiles, (refined_tiles, tiles)


@pytest.mark.gpu
@pytest.mark.parametrize("m,n,K,res", [(4, 2, 256, 720), (4, 3, 64, 1000), (3, 1, 100, 357), (2, 1, 64, 1444)])
def test_error_bounds_hold_on_the_hardware(m, n, K, res, gpu_device):
    """Every form on EVERY (item, bin) with the packed operands: worst |d4 - d| / E4, |d5 - d| / E5, |d7 - d| / allowance below 1."""
    import torch
    for snr, inc in ((20.0, False), (0.0, True), (60.0, True)):
        table, items = _scene(m, n, m * K, res, 200, snr, 900 + m + int(snr), inc)
        x = torch.from_numpy(np.ascontiguousarray(items).view(np.float32)).to(gpu_device)
        with _capi().Context(m, n, m * K, res, table) as ctx:
            w5, w7, w4 = ctx.debug_i8_margin(x.data_ptr(), items.shape[0])
        assert 0.0 < w5 < 0.5, "m=%d snr=%g: worst five-digit error / bound = %.3g" % (m, snr, w5)
        assert 0.0 <= w7 < 0.5, "m=%d snr=%g: worst seven-digit error / allowance = %.3g" % (m, snr, w7)
        assert 0.0 < w4 < 0.5, "m=%d snr=%g: worst four-digit error / bound = %.3g" % (m, snr, w4)
        print("m=%d n=%d snr=%g %s: worst |d4 - d| / E4 = %.3g, |d5 - d| / E5 = %.3g, |d7 - d| / allowance = %.3g"
              % (m, n, snr, "incoherent" if inc else "coherent", w4, w5, w7))


@pytest.mark.gpu
def test_poisoned_items_and_ragged_batches(gpu_device, monkeypatch):
    m, n, N, res = 4, 2, 256, 777
    table, items = _scene(m, n, N, res, 211, 20.0, 5, True)
    items = items.copy()
    items[3] = 0
    items[17, 5] = np.nan
    items[64, 100] = np.inf
    items[65] *= np.float32(1e18)
    items[130] *= np.float32(1e-18)
    (a_i, l_i, s_i, _, _), (a_x, l_x, s_x, _, _) = _both(monkeypatch, m, n, N, res, table, items, gpu_device)
    for b in (17, 64):
        assert np.array_equal(s_i[b].view(np.uint32), s_x[b].view(np.uint32)) and np.array_equal(a_i[b], a_x[b])
        assert np.array_equal(l_i[b].view(np.uint32), l_x[b].view(np.uint32))
    assert (a_i[17] == 0).all() and (l_i[17] == 0).all()
    ok = np.ones(211, bool)
    ok[[17, 64]] = False
    assert_spectrum_close(s_i[ok], s_x[ok], rtol=1.2e-6)
    _assert_same_choice(a_i[ok], a_x[ok], s_x[ok], res)


@pytest.mark.gpu
@pytest.mark.parametrize("m,n,N,res", [(4, 2, 256, 3600), (4, 2, 64, 1000), (3, 2, 96, 357)])
def test_an_items_bits_do_not_depend_on_its_wave_mates_or_the_range_split(m, n, N, res, gpu_device, monkeypatch):
    """Which form a value takes is decided per VALUE, the lists are those of the exact values: an item gives the same bits alone,
    in reversed order, among items of other scenes, in another row class, and under another split of the bins into ranges."""
    table, items = _scene(m, n, N, res, 77, 30.0, 21 + m, True)
    monkeypatch.setenv("BAZ_MUSIC_EXACT", "0")
    with _capi().Context(m, n, N, res, table) as ctx:
        full = _run(ctx, items, gpu_device)
        rev = _run(ctx, items[::-1].copy(), gpu_device)
        one = [_run(ctx, items[i:i + 1], gpu_device) for i in (0, 40, 76)]
        part = _run(ctx, items[5:38], gpu_device)
    for x, y in zip(full, rev):
        assert np.array_equal(x[::-1].view(np.uint32), y.view(np.uint32))
    for k, i in enumerate((0, 40, 76)):
        for x, y in zip(full, one[k]):
            assert np.array_equal(x[i:i + 1].view(np.uint32), y.view(np.uint32))
    for x, y in zip(full, part):
        assert np.array_equal(x[5:38].view(np.uint32), y.view(np.uint32))
    for split in ("1", "3", "7"):
        monkeypatch.setenv("BAZ_MUSIC_NSPLIT", split)
        with _capi().Context(m, n, N, res, table, lab=True) as ctx:
            other = _run(ctx, items, gpu_device)
        for x, y in zip(full, other):
            assert np.array_equal(x.view(np.uint32), y.view(np.uint32)), "outputs depend on the range split (%s)" % split


@pytest.mark.gpu
def test_arbitrary_tables_and_a_table_swap(gpu_device, monkeypatch):
    rng = np.random.default_rng(4)
    m, n, N, res = 4, 2, 256, 500
    _, items = _scene(m, n, N, res, 150, 15.0, 8, True)
    mag = 10.0 ** rng.uniform(-3, 3, size=(res, m))
    table = (mag * np.exp(2j * np.pi * rng.uniform(size=(res, m)))).astype(np.complex64)
    for tb in (table, (table * np.float32(3e-12)).astype(np.complex64), (table * np.float32(7e11)).astype(np.complex64)):
        (a_i, l_i, s_i, _, _), (a_x, l_x, s_x, _, _) = _both(monkeypatch, m, n, N, res, tb, items, gpu_device)
        assert_spectrum_close(s_i, s_x, rtol=1.2e-6)
        _assert_same_choice(a_i, a_x, s_x, res)
    monkeypatch.setenv("BAZ_MUSIC_EXACT", "0")
    steer = mo.steering_table_c64(mo.array_geometry(m), res, mo.FREQUENCY, mo.SPACING)
    with _capi().Context(m, n, N, res, steer) as ctx:
        a_s, l_s, s_s = _run(ctx, items, gpu_device)
        ctx.set_table(table)
        a_t, l_t, s_t = _run(ctx, items, gpu_device)
        ctx.set_table(steer)
        a_b, l_b, s_b = _run(ctx, items, gpu_device)
    assert np.array_equal(s_b.view(np.uint32), s_s.view(np.uint32)) and np.array_equal(a_b, a_s)
    with _capi().Context(m, n, N, res, table) as ctx:
        a_r, l_r, s_r = _run(ctx, items, gpu_device)
    assert np.array_equal(s_t.view(np.uint32), s_r.view(np.uint32)) and np.array_equal(a_t, a_r)


@pytest.mark.gpu
@pytest.mark.parametrize("m,n,N,res,batch", [(4, 2, 1024, 3600, 150), (3, 2, 300, 357, 90)])
def test_ang_lvl_across_the_two_wirings_agree_to_two_eps(m, n, N, res, batch, gpu_device):
    """With the spectrum port the packed int8 scan produces lvl (d good to 8.8e-7), without it the coarse-gated scan produces it
    from exact fp64 tile values: lvl within 1.8e-6 of each other, DoA bins identical except between bins that tie to 2 eps."""
    table, items = _scene(m, n, N, res, batch, 20.0, 31 + m)
    with _capi().Context(m, n, N, res, table) as ctx:
        assert ctx.uses_i8_scan()
        a_w, l_w, s_w = _run(ctx, items, gpu_device, want_spec=True)
        assert "scan_i8p_kernel" in ctx.stage_name(2)
        a_o, l_o, _ = _run(ctx, items, gpu_device, want_spec=False)
        assert "scan_coarse_kernel" in ctx.stage_name(2)
    same = a_w == a_o
    assert np.max(np.abs(l_w[same].astype(np.float64) - l_o[same]) / l_o[same]) <= 1.8e-6
    _assert_same_choice(a_w, a_o, s_w, res)
    ao, lo, so, s64 = mo.music_doa_work_batch(items, table, m, n)
    assert_doa_match(a_w, l_w, ao, lo, res, s64)
    assert_doa_match(a_o, l_o, ao, lo, res, s64)
