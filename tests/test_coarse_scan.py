"""The coarse-gated scan (gr_baz_amd/csrc/scan_coarse_kernels.hip.h): what runs when the spectrum port is NOT wired
(music_doa_helper's default output_spectrum=False, /root/reference/python/music_doa_helper.py:49,61-64) and m <= 8.
Only the top-n list is observable then (lib/baz_music_doa.cc:97-99,129-155); the kernel skips bin tiles that provably
cannot hold a member of it.  Three things are pinned here, all on the GPU through the C-ABI:
  1. ang / lvl are BIT-IDENTICAL to the full fp64 scan of the same build (BAZ_MUSIC_COARSE=0), on coherent and incoherent
     scenes, SNR 0 ... 90 dB, every (m, n) the kernel takes, ragged batch sizes, forced bin-range splits, poisoned items;
  2. they equal the CPU oracle's (the parity rule of the full scan);
  3. the error bound the gate rests on holds on this hardware with room to spare (baz_music_debug_coarse_margin)."""
import numpy as np
import pytest

from helpers import assert_doa_match
from oracle import music_oracle as mo

pytestmark = pytest.mark.gpu


def _capi():
    from gr_baz_amd import capi
    return capi


def _run_nospec(ctx, items, gpu_device):
    import torch
    B = items.shape[0]
    x = torch.from_numpy(np.ascontiguousarray(items).view(np.float32)).to(gpu_device)
    ang = torch.full((B, ctx.n), -1.0, dtype=torch.float32, device=gpu_device)
    lvl = torch.full((B, ctx.n), -1.0, dtype=torch.float32, device=gpu_device)
    ctx.process_device(x.data_ptr(), B, ang.data_ptr(), lvl.data_ptr(), None, stream=torch.cuda.current_stream().cuda_stream)
    return ang.cpu().numpy(), lvl.cpu().numpy()


def _scene(m, n, nsamples, res, batch, snr_db, seed, incoherent):
    """Items of one stream (every item sees the same emitters) or of `batch` different scenes (angles drawn per item)."""
    arr = mo.array_geometry(m) if m >= 3 else [[0.0, 0.0], [1.0, 0.0]]
    table = mo.steering_table_c64(arr, res, mo.FREQUENCY, mo.SPACING)
    rng = np.random.default_rng(seed)
    if not incoherent:
        angles = tuple(rng.uniform(0.0, 360.0, size=n))
        return table, mo.synth_items(batch, m, nsamples, arr, mo.FREQUENCY, mo.SPACING, angles_deg=angles, snr_db=snr_db, seed=seed)
    items = np.concatenate([mo.synth_items(1, m, nsamples, arr, mo.FREQUENCY, mo.SPACING,
                                           angles_deg=tuple(rng.uniform(0.0, 360.0, size=n)), snr_db=snr_db, seed=seed + 7 * i)
                            for i in range(batch)], axis=0)
    return table, items


def _both(monkeypatch, m, n, nsamples, res, table, items, gpu_device, env=None):
    """env: lab switches (geometry overrides) -- those exist only in the lab build of the library (capi.Context(lab=True))."""
    out = {}
    for coarse in ("1", "0"):
        monkeypatch.setenv("BAZ_MUSIC_COARSE", coarse)
        for k, v in (env or {}).items():
            monkeypatch.setenv(k, v)
        with _capi().Context(m, n, nsamples, res, table, lab=bool(env)) as ctx:
            out[coarse] = _run_nospec(ctx, items, gpu_device) + (ctx.refined_values(),)
    return out["1"], out["0"]


@pytest.mark.parametrize("incoherent", [False, True])
@pytest.mark.parametrize("snr", [0.0, 10.0, 20.0, 40.0, 70.0, 90.0])
@pytest.mark.parametrize("m,n,nsamples,res,batch", [(4, 2, 1024, 3600, 333), (4, 1, 256, 360, 257), (4, 3, 256, 1000, 130),
                                                    (3, 2, 96, 721, 200), (3, 1, 96, 64, 70), (2, 1, 64, 90, 65),
                                                    # 5 .. 8 antennas: 2 .. 4 operand groups per tile, fp64 operands from L2
                                                    (8, 2, 1024, 3600, 150), (8, 3, 512, 1000, 70), (7, 2, 280, 721, 130), (5, 1, 160, 200, 40),
                                                    (6, 2, 384, 500, 200), (5, 4, 320, 360, 100), (5, 2, 320, 64, 65)])
def test_coarse_gated_scan_equals_the_full_scan(m, n, nsamples, res, batch, snr, incoherent, gpu_device, monkeypatch):
    table, items = _scene(m, n, nsamples, res, batch, snr, 9000 + int(snr) + 13 * m + n, incoherent)
    (a1, l1, r1), (a0, l0, r0) = _both(monkeypatch, m, n, nsamples, res, table, items, gpu_device)
    assert np.array_equal(a1, a0), "DoA bins of the gated scan differ from the full scan"
    assert np.array_equal(l1.view(np.uint32), l0.view(np.uint32)), "levels are not bit-identical"
    assert (a1 >= 0).all() and (l1 >= 0).all()              # every slot written
    if snr >= 70.0 and n < m - 0:                           # near-null values reach the literal form in both builds
        assert r0 >= r1 >= 0
    # and against the CPU oracle (the full scan's parity rule)
    ao, lo, so, st = mo.music_doa_work_batch(items, table, m, n)
    if snr <= 40.0:
        assert_doa_match(a1, l1, ao, lo, res, st)


@pytest.mark.parametrize("nsplit", ["1", "3", "16"])
def test_gated_scan_is_independent_of_the_bin_range_split(nsplit, gpu_device, monkeypatch):
    """Small batches are cut into ranges of table phases (one candidate list per range, folded by the merge kernel):
    the split must not show in the outputs."""
    table, items = _scene(4, 2, 1024, 3600, 100, 20.0, 77, True)
    (a1, l1, _), (a0, l0, _) = _both(monkeypatch, 4, 2, 1024, 3600, table, items, gpu_device, env={"BAZ_MUSIC_NSPLIT": nsplit})
    assert np.array_equal(a1, a0) and np.array_equal(l1.view(np.uint32), l0.view(np.uint32))


@pytest.mark.parametrize("rg", ["2", "4"])
def test_gated_scan_row_group_variants_and_poisoned_items(rg, gpu_device, monkeypatch):
    """Both register layouts (2 or 4 row groups per wave); items that are all zero, NaN, inf or huge sit between ordinary
    ones without disturbing them: a non-finite covariance gives (0, 0) pairs in both builds (.cc:95: the lists' initial
    content; NaN never inserts, .cc:131)."""
    table, items = _scene(4, 2, 1024, 3600, 300, 20.0, 5, True)
    items = items.copy()
    items[3] = 0
    items[17, 5] = np.nan
    items[64, 100] = np.inf
    items[65] *= np.float32(1e18)
    items[130] *= np.float32(1e-18)
    (a1, l1, _), (a0, l0, _) = _both(monkeypatch, 4, 2, 1024, 3600, table, items, gpu_device, env={"BAZ_MUSIC_COARSE_RG": rg})
    assert np.array_equal(a1, a0) and np.array_equal(l1.view(np.uint32), l0.view(np.uint32))
    assert (a1[17] == 0).all() and (l1[17] == 0).all() and (a1[64] == 0).all()


def test_gated_scan_with_an_arbitrary_table(gpu_device, monkeypatch):
    """set_array_response takes ANY res x m complex table (.cc:60-70), not only unit-modulus steering vectors: random
    magnitudes over six decades, a scaled copy, and a table swap in a live context (the coarse image is rebuilt)."""
    rng = np.random.default_rng(4)
    m, n, N, res = 4, 2, 256, 500
    _, items = _scene(m, n, N, res, 150, 15.0, 8, True)
    mag = 10.0 ** rng.uniform(-3, 3, size=(res, m))
    table = (mag * np.exp(2j * np.pi * rng.uniform(size=(res, m)))).astype(np.complex64)
    for tb in (table, (table * np.float32(3e-12)).astype(np.complex64), (table * np.float32(7e11)).astype(np.complex64)):
        (a1, l1, _), (a0, l0, _) = _both(monkeypatch, m, n, N, res, tb, items, gpu_device)
        assert np.array_equal(a1, a0) and np.array_equal(l1.view(np.uint32), l0.view(np.uint32))
    monkeypatch.setenv("BAZ_MUSIC_COARSE", "1")
    steer = mo.steering_table_c64(mo.array_geometry(m), res, mo.FREQUENCY, mo.SPACING)
    with _capi().Context(m, n, N, res, steer) as ctx:
        a_s, l_s = _run_nospec(ctx, items, gpu_device)
        ctx.set_table(table)
        a_t, l_t = _run_nospec(ctx, items, gpu_device)
    monkeypatch.setenv("BAZ_MUSIC_COARSE", "0")
    with _capi().Context(m, n, N, res, table) as ctx:
        a_f, l_f = _run_nospec(ctx, items, gpu_device)
    assert np.array_equal(a_t, a_f) and np.array_equal(l_t.view(np.uint32), l_f.view(np.uint32)) and not np.array_equal(a_s, a_t)


@pytest.mark.parametrize("snr", [0.0, 20.0, 60.0])
@pytest.mark.parametrize("m,n,nsamples,res", [(4, 2, 1024, 3600), (4, 3, 256, 1000), (3, 1, 96, 721), (2, 1, 64, 90),
                                              (8, 2, 1024, 3600), (7, 3, 448, 1000), (6, 2, 384, 721), (5, 1, 320, 360)])
def test_coarse_error_bound_holds_with_room_to_spare(m, n, nsamples, res, snr, gpu_device):
    """|coarse / SC - exact| <= NG 2^-16 (S + |exact|), NG = ceil(m^2 / 16), is what makes skipping a tile safe.  The debug tap evaluates both forms
    on EVERY (item, bin) of the batch and returns the worst error / allowance: < 1 is sound; the derivation leaves a factor
    > 2, the f16 matrix core accumulates more accurately than the worst case assumed, so < 0.5 is asserted."""
    import torch
    table, items = _scene(m, n, nsamples, res, 2048, snr, 31 + int(snr), True)
    x = torch.from_numpy(np.ascontiguousarray(items).view(np.float32)).to(gpu_device)
    torch.cuda.synchronize()
    with _capi().Context(m, n, nsamples, res, table) as ctx:
        worst = ctx.debug_coarse_margin(x.data_ptr(), items.shape[0])
    assert 0.0 < worst < 0.5, worst


def test_the_gate_actually_skips_work(gpu_device, monkeypatch):
    """Not a parity property but the point of the kernel: on one stream's items (a coherent scene) the gated scan must be
    several times faster than the full scan it replaces (measured 0.50 ms -> see profiles/r03_*): guards against a gate
    that silently fires on every tile."""
    import torch
    from gr_baz_amd import synth
    c = mo.make_config("cfg2", 8, seed=1)
    x = synth.synth_stream(torch, gpu_device, 65536, 4, 1024, mo.array_geometry(4), mo.FREQUENCY, mo.SPACING, seed=1002)
    ang = torch.zeros(65536, 2, dtype=torch.float32, device=gpu_device)
    lvl = torch.zeros_like(ang)
    ms = {}
    for coarse in ("1", "0"):
        monkeypatch.setenv("BAZ_MUSIC_COARSE", coarse)
        with _capi().Context(4, 2, 1024, 3600, c["table"]) as ctx:
            ctx.reserve(65536)
            for _ in range(20):
                ctx.process_device(x.data_ptr(), 65536, ang.data_ptr(), lvl.data_ptr(), None)
            ctx.sync()
            ctx.profile(1)
            for _ in range(20):
                ctx.process_device(x.data_ptr(), 65536, ang.data_ptr(), lvl.data_ptr(), None)
            ctx.sync()
            t, k = ctx.stage_ms(_capi().STAGE_SCAN)
            ms[coarse] = t / k
    assert ms["1"] < 0.8 * ms["0"], ms


def test_the_gate_skips_work_at_eight_antennas(gpu_device, monkeypatch):
    """BASELINE configs[2]'s shape (m8, N4096, res36000) in the helper's default wiring: the gated scan against the full one."""
    import torch
    from gr_baz_amd import synth
    m, N, res, B = 8, 4096, 36000, 4096
    arr = mo.array_geometry(m)
    table = mo.steering_table_c64(arr, res, mo.FREQUENCY, mo.SPACING)
    x = synth.synth_stream(torch, gpu_device, B, m, N, arr, mo.FREQUENCY, mo.SPACING, seed=1003)
    out, ms = {}, {}
    for coarse in ("1", "0"):
        monkeypatch.setenv("BAZ_MUSIC_COARSE", coarse)
        ang = torch.zeros(B, 2, dtype=torch.float32, device=gpu_device)
        lvl = torch.zeros_like(ang)
        with _capi().Context(m, 2, N, res, table) as ctx:
            ctx.reserve(B)
            for _ in range(3):
                ctx.process_device(x.data_ptr(), B, ang.data_ptr(), lvl.data_ptr(), None)
            ctx.sync()
            ctx.profile(1)
            for _ in range(5):
                ctx.process_device(x.data_ptr(), B, ang.data_ptr(), lvl.data_ptr(), None)
            ctx.sync()
            t, k = ctx.stage_ms(_capi().STAGE_SCAN)
            ms[coarse] = t / k
        out[coarse] = (ang.cpu().numpy(), lvl.cpu().numpy())
    assert np.array_equal(out["1"][0], out["0"][0]) and np.array_equal(out["1"][1].view(np.uint32), out["0"][1].view(np.uint32))
    assert ms["1"] < 0.6 * ms["0"], ms
