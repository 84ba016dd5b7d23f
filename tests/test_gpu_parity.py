"""GPU parity tests (run with -m gpu on the MI355X box): the HIP path, called through the C-ABI,
against (1) the golden vectors produced by the reference's own work() source, (2) the CPU oracle on
seeded inputs, (3) size-independent properties at BASELINE.json's full sizes.

Tolerance: north_star's 1e-5 relative on float32 spectra / levels; DoA bins identical except at
reference-side ties (< 2e-5 relative)."""
import numpy as np
import pytest

from conftest import golden_names, load_golden
from helpers import assert_doa_match, assert_spectrum_close
from oracle import music_oracle as mo
from oracle import music_ref as mr

pytestmark = pytest.mark.gpu


def _capi():
    from gr_baz_amd import capi
    return capi


def _torch():
    import torch
    return torch


def device_run(ctx, items, gpu_device, want_lvl=True, want_spec=True):
    torch = _torch()
    B = items.shape[0]
    x = torch.from_numpy(np.ascontiguousarray(items).view(np.float32)).to(gpu_device)
    ang = torch.full((B, ctx.n), -1.0, dtype=torch.float32, device=gpu_device)
    lvl = torch.full((B, ctx.n), -1.0, dtype=torch.float32, device=gpu_device) if want_lvl else None
    spec = torch.full((B, ctx.res), -1.0, dtype=torch.float32, device=gpu_device) if want_spec else None
    # stream semantics against torch's stream: the fills above precede the batch, the reads below follow it
    ctx.process_device(x.data_ptr(), B, ang.data_ptr(), lvl.data_ptr() if want_lvl else None,
                       spec.data_ptr() if want_spec else None, stream=torch.cuda.current_stream().cuda_stream)
    return (ang.cpu().numpy(), lvl.cpu().numpy() if want_lvl else None,
            spec.cpu().numpy() if want_spec else None)


# ------------------------------------------------------------------ golden vectors
@pytest.mark.parametrize("name", golden_names())
def test_hip_matches_golden_device_path(name, gpu_device):
    g = load_golden(name)
    with _capi().Context(g["m"], g["n"], g["nsamples"], g["res"], g["table"]) as ctx:
        ang, lvl, spec = device_run(ctx, g["items"], gpu_device)
    worst = assert_spectrum_close(spec, g["spectrum"])
    assert_doa_match(ang, lvl, g["ang"], g["lvl"], g["res"], g["strength64"])
    assert worst < 1e-6      # measured ~1.2e-7: fp64 accumulation + 2-ulp f32 reciprocal


@pytest.mark.parametrize("name", golden_names())
def test_hip_matches_golden_host_path(name, gpu_device):
    g = load_golden(name)
    with _capi().Context(g["m"], g["n"], g["nsamples"], g["res"], g["table"]) as ctx:
        ang, lvl, spec = ctx.process(g["items"])
    assert_spectrum_close(spec, g["spectrum"])
    assert_doa_match(ang, lvl, g["ang"], g["lvl"], g["res"], g["strength64"])


# ------------------------------------------------------------------ seeded inputs vs the oracle
@pytest.mark.parametrize("snr", [10.0, 20.0, 40.0])
@pytest.mark.parametrize("cfg,batch", [("cfg1", 300), ("cfg2", 200), ("cfg3", 24)])
def test_hip_matches_oracle_on_seeded_batches(cfg, batch, snr, gpu_device):
    c = mo.make_config(cfg, batch, snr_db=snr, seed=4242 + int(snr))
    ao, lo, so, st = mo.music_doa_work_batch(c["items"], c["table"], c["m"], c["n"])
    with _capi().Context(c["m"], c["n"], c["nsamples"], c["res"], c["table"]) as ctx:
        ang, lvl, spec = device_run(ctx, c["items"], gpu_device)
    assert_spectrum_close(spec, so)
    assert_doa_match(ang, lvl, ao, lo, c["res"], st)


def test_stage_taps_covariance_and_projector(gpu_device):
    """The two intermediate stages against fp64 numpy: R = x x^H / K (.cc:85) and the projector of
    the (m-n) smallest eigenvectors (.cc:88-93)."""
    torch = _torch()
    capi = _capi()
    for cfg, B in (("cfg1", 130), ("cfg2", 70), ("cfg3", 37)):      # cfg2: the dwordx4 / 4x4x4 covariance (K % 256 == 0)
        c = mo.make_config(cfg, B, snr_db=25.0, seed=99)
        m, n, N = c["m"], c["n"], c["nsamples"]
        with capi.Context(m, n, N, c["res"], c["table"]) as ctx:
            x = torch.from_numpy(c["items"].view(np.float32)).to(gpu_device)
            R = torch.zeros(B, m * m, 2, dtype=torch.float64, device=gpu_device)
            ctx.debug_cov(x.data_ptr(), B, R.data_ptr())
            Q = torch.zeros(m * m, capi.q_stride(B), dtype=torch.float64, device=gpu_device)
            ctx.debug_evd(R.data_ptr(), B, Q.data_ptr())
            Qp = torch.zeros_like(Q)                       # the same two stages the way process_device() runs them
            torch.cuda.synchronize()                       # (the taps run on the context's own stream)
            ctx.debug_q(x.data_ptr(), B, Qp.data_ptr())
            ctx.sync()
        Rg = R.cpu().numpy()
        Rg = (Rg[..., 0] + 1j * Rg[..., 1]).reshape(B, m, m)
        xs = c["items"].astype(np.complex128).reshape(B, N // m, m).transpose(0, 2, 1)
        Rn = xs @ xs.conj().transpose(0, 2, 1) / (N // m)
        assert np.abs(Rg - Rn).max() <= 1e-14 * np.abs(Rn).max()
        assert np.array_equal(Rg, Rg.conj().transpose(0, 2, 1))          # exactly Hermitian
        w, V = np.linalg.eigh(Rn)
        G = V[:, :, :m - n]
        P = G @ G.conj().transpose(0, 2, 1)
        assert np.abs((Qp - Q).cpu().numpy()[:, :B]).max() < 1e-12
        Qg = Qp.cpu().numpy()[:, :B].T.reshape(B, m, m)
        for i in range(m):
            assert np.abs(Qg[:, i, i] - P[:, i, i].real).max() < 1e-12
            for j in range(i + 1, m):
                assert np.abs(Qg[:, i, j] - 2 * P[:, i, j].real).max() < 1e-12
                assert np.abs(Qg[:, j, i] + 2 * P[:, i, j].imag).max() < 1e-12


@pytest.mark.parametrize("m,n,N,res,batch", [
    (4, 1, 512, 360, 1),        # a single item (the reference's one-item work())
    (4, 3, 64, 33, 70),         # n = m-1, res % 4 != 0 (scalar spectrum stores), batch % 16 != 0
    (2, 1, 6, 5, 17),           # tiny everything, K = 3 (MFMA tail path)
    (3, 2, 33, 1000, 65),       # odd m: K padding of the 9-term form
    (5, 1, 40, 121, 100),
    (6, 5, 96, 250, 33),        # NMAX = 8 list
    (7, 3, 7 * 37, 77, 48),
    (8, 7, 8 * 20, 1024, 20),   # n = 7 on the 8-antenna path
    (8, 1, 8 * 1000, 90, 5),    # long integration, short table
    (9, 4, 9 * 50, 333, 21),    # two-tile covariance with padding rows (2m = 18)
    (13, 12, 13 * 40, 128, 9),  # n = m-1 = 12: 16-entry list
    (16, 2, 16 * 256, 900, 40), # config-5 antenna count
    (16, 15, 16 * 64, 64, 6),
])
def test_hip_matches_oracle_on_odd_shapes(m, n, N, res, batch, gpu_device):
    arr = mo.array_geometry(m) if m != 2 else [[0.0, 0.0], [1.0, 0.0]]
    table = mo.steering_table_c64(arr, res, mo.FREQUENCY, mo.SPACING)
    angles = tuple(np.linspace(17.0, 311.0, n))
    items = mo.synth_items(batch, m, N, arr, mo.FREQUENCY, mo.SPACING, angles_deg=angles, snr_db=20.0,
                           seed=100 * m + n)
    ao, lo, so, st = mo.music_doa_work_batch(items, table, m, n)
    with _capi().Context(m, n, N, res, table) as ctx:
        ang, lvl, spec = device_run(ctx, items, gpu_device)
        a2, l2, s2 = ctx.process(items)
    assert_spectrum_close(spec, so)
    assert_doa_match(ang, lvl, ao, lo, res, st)
    assert np.array_equal(a2, ang) and np.array_equal(l2, lvl) and np.array_equal(s2, spec)


@pytest.mark.parametrize("m,n,N,res,batch", [
    (17, 2, 17 * 40, 360, 9),       # the first antenna count past the specialised kernels; odd m (phantom index)
    (20, 19, 20 * 64, 91, 5),       # n = m-1: one noise vector, 19 list entries
    (31, 4, 31 * 48, 1000, 4),
    (32, 2, 32 * 128, 3600, 6),
    (48, 7, 48 * 96, 500, 3),
    (64, 2, 64 * 64, 720, 3),       # BAZ_MUSIC_MAX_M: 137 KB of LDS per item
    (37, 2, 37 * 50, 360, 5),       # 33 .. 64 antennas: covariance by pairs of 16-antenna blocks (3 blocks, the last one 5 wide; K % 4 != 0)
    (49, 3, 49 * 33, 200, 3),       # 4 blocks, the last one a single antenna; K = 32 + 1
])
def test_wide_arrays_match_the_oracle(m, n, N, res, batch, gpu_device):
    """17..64 antennas (music_wide_kernels.hip.h): device path, host path and stage tap against the oracle."""
    arr = mo.array_geometry(m)
    table = mo.steering_table_c64(arr, res, mo.FREQUENCY, mo.SPACING)
    angles = tuple(np.linspace(17.0, 311.0, n))
    items = mo.synth_items(batch, m, N, arr, mo.FREQUENCY, mo.SPACING, angles_deg=angles, snr_db=20.0,
                           seed=100 * m + n)
    ao, lo, so, st = mo.music_doa_work_batch(items, table, m, n)
    torch = _torch()
    with _capi().Context(m, n, N, res, table) as ctx:
        ang, lvl, spec = device_run(ctx, items, gpu_device)
        a1, _, _ = device_run(ctx, items, gpu_device, want_lvl=False, want_spec=False)
        a2, l2, s2 = ctx.process(items)
        x = torch.from_numpy(np.ascontiguousarray(items).view(np.float32)).to(gpu_device)
        R = torch.zeros(batch, m * m, 2, dtype=torch.float64, device=gpu_device)
        ctx.debug_cov(x.data_ptr(), batch, R.data_ptr())
        ctx.sync()
        assert ctx.refined_values() >= 0          # (counted by the matrix-core scan only: n <= 8)
    assert_spectrum_close(spec, so)
    assert_doa_match(ang, lvl, ao, lo, res, st)
    assert np.array_equal(a1, ang)
    assert np.array_equal(a2, ang) and np.array_equal(l2, lvl) and np.array_equal(s2, spec)
    xs = items.astype(np.complex128).reshape(batch, N // m, m).transpose(0, 2, 1)
    Rn = (xs @ xs.conj().transpose(0, 2, 1)) / float(N // m)
    Rg = R.cpu().numpy()
    Rg = (Rg[..., 0] + 1j * Rg[..., 1]).reshape(batch, m, m)
    assert np.abs(Rg - Rn).max() <= 1e-13 * np.abs(Rn).max()
    assert np.array_equal(Rg, Rg.conj().transpose(0, 2, 1))          # exactly Hermitian


@pytest.mark.parametrize("m,n,K,res", [(24, 2, 40, 180), (40, 3, 48, 120), (64, 1, 64, 90), (32, 4, 64, 180),
                                       (20, 5, 64, 90), (48, 7, 64, 120), (64, 8, 96, 90)])      # five to eight emitters: the same iteration
def test_wide_arrays_subspace_iteration_and_hand_back(m, n, K, res, gpu_device, monkeypatch):
    """the wide path's sub_wide_kernel against its Jacobi (BAZ_MUSIC_SUB_EVD=0) on a batch that mixes easy, slow and
    rank-deficient items; bits independent of the batch around an item"""
    N = m * K
    arr = mo.array_geometry(m)
    table = mo.steering_table_c64(arr, res, mo.FREQUENCY, mo.SPACING)
    ang_n = tuple(np.linspace(25.0, 290.0, n))
    parts = [mo.synth_items(3, m, N, arr, mo.FREQUENCY, mo.SPACING, angles_deg=ang_n, snr_db=snr, seed=700 + 7 * i + m)
             for i, snr in enumerate((30.0, 10.0, -10.0))]
    parts.append(mo.synth_items(2, m, N, arr, mo.FREQUENCY, mo.SPACING, angles_deg=ang_n[:n - 1] or (), snr_db=20.0, seed=750 + m))
    parts.append(np.zeros((1, N), np.complex64))
    items = np.concatenate(parts)
    outs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("BAZ_MUSIC_SUB_EVD", mode)
        with _capi().Context(m, n, N, res, table, lab=True) as ctx:
            outs[mode] = (device_run(ctx, items, gpu_device), device_run(ctx, items[::-1].copy(), gpu_device),
                          device_run(ctx, items[4:5], gpu_device))
    full, rev, one = outs["1"]
    for x, y in zip(full, rev):
        assert np.array_equal(x[::-1], y)
    for x, y in zip(full, one):
        assert np.array_equal(x[4:5], y)
    s1, s0 = full[2].astype(np.float64), outs["0"][0][2].astype(np.float64)
    assert np.all(np.abs(s1 - s0) <= 2e-6 * np.abs(s0))            # same subspace either way (float32 spectra: 1-2 ulp)
    ao, lo, so, st = mo.music_doa_work_batch(items[:-1], table, m, n)
    assert_spectrum_close(full[2][:-1], so)
    assert_doa_match(full[0][:-1], full[1][:-1], ao, lo, res, st)
    assert np.array_equal(full[2][-1], outs["0"][0][2][-1])          # zero item: handed back -> the Jacobi's answer


@pytest.mark.parametrize("m,n,K,res,snr", [(20, 2, 64, 720, 60.0), (32, 1, 48, 360, 100.0), (24, 3, 80, 500, 80.0), (18, 2, 64, 360, 20.0)])
def test_wide_arrays_short_form_and_literal_form_agree(m, n, K, res, snr, gpu_device, monkeypatch):
    """scan_wide_kernel evaluates ||a||^2 - ||S^H a||^2 away from the nulls and the reference's literal form near them:
    up to 100 dB SNR (nulls 1e-10 of ||a||^2 deep) the spectrum must match the literal-only build and the oracle"""
    N = m * K
    arr = mo.array_geometry(m)
    table = mo.steering_table_c64(arr, res, mo.FREQUENCY, mo.SPACING)
    bins = np.round(np.linspace(0.09, 0.84, n) * res)                       # emitters exactly on bins: the nulls are SNR-deep
    items = mo.synth_items(8, m, N, arr, mo.FREQUENCY, mo.SPACING, angles_deg=tuple(bins * 360.0 / res), snr_db=snr,
                           seed=31 * m + n)
    outs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("BAZ_MUSIC_WIDE_LITERAL", mode)
        with _capi().Context(m, n, N, res, table, lab=True) as ctx:
            outs[mode] = device_run(ctx, items, gpu_device)
    s_short, s_lit = outs["0"][2].astype(np.float64), outs["1"][2].astype(np.float64)
    assert np.all(np.abs(s_short - s_lit) <= 3e-6 * s_lit)
    assert np.array_equal(outs["0"][0], outs["1"][0]) or np.abs(s_short - s_lit).max() > 0      # same bins unless a tie moved
    ao, lo, so, st = mo.music_doa_work_batch(items, table, m, n)
    assert_spectrum_close(outs["0"][2], so)
    assert_doa_match(outs["0"][0], outs["0"][1], ao, lo, res, st)
    assert so.max() / np.median(so) > (1e5 if snr >= 60 else 10)        # the peaks really are that sharp


@pytest.mark.parametrize("m,n,K,res,batch,snr", [
    (17, 2, 40, 360, 9, 20.0),        # odd m: 34 real coordinates, 9 k-steps (the last half empty)
    (24, 1, 37, 91, 21, 20.0),        # one emitter (two idle output rows); res neither a multiple of 64 nor of 4; K % 4 != 0
    (32, 2, 64, 3600, 70, 20.0),      # the headline wide shape; 70 items: a ragged last workgroup and wave
    (29, 2, 50, 1000, 1200, 10.0),    # enough items that the bins of an item are NOT split over workgroups; K = 32 + 18
    (32, 2, 48, 640, 5, 80.0),        # sharp nulls: the literal form runs inside the kernel
    (33, 1, 20, 360, 9, 20.0),        # 33 .. 64 antennas: four staged phases per step (17 k-steps: the third phase holds one)
    (48, 2, 24, 724, 40, 20.0),       # 24 k-steps = three full phases
    (64, 2, 32, 3600, 70, 20.0),      # the widest array the kernels take: 32 k-steps
    (64, 2, 16, 640, 5, 80.0),        # ... with sharp nulls
    (31, 4, 48, 1000, 9, 20.0),       # three and four emitters: a tile = 2 items x 8 outputs, four keys per list
    (40, 3, 40, 360, 13, 20.0),       # (six of the eight output rows used; an odd item count: the last wave holds one item)
    (64, 4, 32, 724, 5, 20.0),
    (24, 3, 64, 640, 6, 80.0),        # ... with sharp nulls
    (48, 7, 64, 500, 6, 20.0),        # five to eight emitters: one item per tile, eight keys per list
    (64, 8, 96, 724, 5, 20.0),
    (20, 5, 64, 360, 9, 80.0),        # ... with sharp nulls
])
def test_wide_arrays_matrix_core_scan(m, n, K, res, batch, snr, gpu_device, monkeypatch):
    """scan_wide_mfma_kernel (17 <= m <= 64, n <= 8) against scan_wide_kernel + topn_wide_kernel (BAZ_MUSIC_WIDE_MFMA=0) and
    the oracle; an item's bits do not depend on the batch around it (hence not on how its bins were split)"""
    N = m * K
    arr = mo.array_geometry(m)
    table = mo.steering_table_c64(arr, res, mo.FREQUENCY, mo.SPACING)
    if snr >= 60:
        bins = np.round(np.linspace(0.13, 0.71, n) * res)
        angles = tuple(bins * 360.0 / res)
    else:
        angles = tuple(np.linspace(41.0, 263.0, n))
    base = mo.synth_items(min(batch, 24), m, N, arr, mo.FREQUENCY, mo.SPACING, angles_deg=angles, snr_db=snr, seed=9 * m + n)
    items = np.concatenate([base] * ((batch + len(base) - 1) // len(base)))[:batch]
    outs = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("BAZ_MUSIC_WIDE_MFMA", mode)
        with _capi().Context(m, n, N, res, table, lab=True) as ctx:
            outs[mode] = device_run(ctx, items, gpu_device)
            if mode == "1":
                assert ctx.stage_name(2).endswith("scan_wide_mfma_kernel")
                refined = ctx.refined_values()
                nospec = device_run(ctx, items, gpu_device, want_spec=False)
                one = device_run(ctx, items[3:4], gpu_device)
            else:
                assert ctx.stage_name(2).endswith("scan_wide_kernel")
    (a1, l1, s1), (a0, l0, s0) = outs["1"], outs["0"]
    assert np.all(np.abs(s1.astype(np.float64) - s0) <= 3e-6 * s0)
    same = np.all(a1 == a0, axis=1)
    assert same.mean() > 0.9                                     # (near-ties between adjacent bins may swap)
    assert np.array_equal(l1, np.take_along_axis(s1, np.rint(a1 * res / 360.0).astype(int), axis=1))   # lvl == spectrum[bin]
    assert np.array_equal(nospec[0], a1)
    assert np.all(np.abs(nospec[1].astype(np.float64) - l1) <= 3e-7 * l1)     # (no spectrum row to read lvl back from)
    for x, y in zip(outs["1"], one):
        assert np.array_equal(x[3:4], y)
    if snr >= 60:
        assert refined > 0
    nb = min(batch, 24)
    ao, lo, so, st = mo.music_doa_work_batch(items[:nb], table, m, n)
    assert_spectrum_close(s1[:nb], so)
    assert_doa_match(a1[:nb], l1[:nb], ao, lo, res, st)


def test_wide_contexts_of_different_sizes_coexist(gpu_device):
    """the run-time-m kernels use more than 64 KiB of dynamic LDS at m = 64, a per-function attribute: a small wide
    context created AFTER a large one must not shrink it"""
    cfgs = []
    for m, n, K, res in ((64, 2, 64, 90), (20, 2, 32, 90)):
        arr = mo.array_geometry(m)
        table = mo.steering_table_c64(arr, res, mo.FREQUENCY, mo.SPACING)
        items = mo.synth_items(3, m, m * K, arr, mo.FREQUENCY, mo.SPACING, snr_db=20.0, seed=m)
        cfgs.append((m, n, m * K, res, table, items))
    big = _capi().Context(*cfgs[0][:5])
    small = _capi().Context(*cfgs[1][:5])
    try:
        for ctx, c in ((big, cfgs[0]), (small, cfgs[1]), (big, cfgs[0])):
            ang, lvl, spec = device_run(ctx, c[5], gpu_device)
            ao, lo, so, st = mo.music_doa_work_batch(c[5], c[4], c[0], c[1])
            assert_spectrum_close(spec, so)
            assert_doa_match(ang, lvl, ao, lo, c[3], st)
    finally:
        small.close(); big.close()


def test_wide_arrays_top_n_and_non_finite_items(gpu_device):
    """the wide path's own top-n: exact ties keep the earlier bin, an item with a NaN sample yields (0, 0) pairs and
    a NaN spectrum, its neighbours are untouched"""
    m, n, N, res = 18, 3, 18 * 32, 120
    arr = mo.array_geometry(m)
    table = mo.steering_table_c64(arr, res, mo.FREQUENCY, mo.SPACING)
    table[60:] = table[:60]                              # every strength occurs twice: bins b and b + 60
    items = mo.synth_items(4, m, N, arr, mo.FREQUENCY, mo.SPACING, angles_deg=(33.0, 100.0, 170.0), snr_db=20.0, seed=77)
    items[2, 5] = np.nan
    with _capi().Context(m, n, N, res, table) as ctx:
        ang, lvl, spec = device_run(ctx, items, gpu_device)
    assert np.all(np.isnan(spec[2])) and np.all(ang[2] == 0.0) and np.all(lvl[2] == 0.0)
    for it in (0, 1, 3):
        assert np.array_equal(spec[it, :60], spec[it, 60:])
        bins = np.rint(ang[it] * res / 360.0).astype(int)
        assert bins[1] == bins[0] + 60 and bins[2] < 60      # strongest bin, its twin (the LATER bin second), then the next value
        assert np.array_equal(lvl[it], spec[it][bins])
    clean = np.delete(items, 2, axis=0)
    ao, lo, so, st = mo.music_doa_work_batch(clean, table, m, n)
    assert_spectrum_close(np.delete(spec, 2, axis=0), so)


def test_optional_ports(gpu_device):
    """output_items.size() in {1,2,3} (lib/baz_music_doa.cc:97-99,147-154; lvl NULL must not crash)."""
    c = mo.make_config("cfg1", 50, seed=3)
    with _capi().Context(c["m"], c["n"], c["nsamples"], c["res"], c["table"]) as ctx:
        a3, l3, s3 = device_run(ctx, c["items"], gpu_device)
        a2, l2, s2 = device_run(ctx, c["items"], gpu_device, want_spec=False)
        a1, l1, s1 = device_run(ctx, c["items"], gpu_device, want_lvl=False, want_spec=False)
        h1 = ctx.process(c["items"], want_lvl=False, want_spectrum=False)
    assert s2 is None and l1 is None and s1 is None and h1[1] is None and h1[2] is None
    assert np.array_equal(a3, a2) and np.array_equal(a3, a1) and np.array_equal(l3, l2)
    assert np.array_equal(h1[0], a3)


def test_set_table_takes_effect_from_the_next_item(gpu_device):
    """set_array_response (lib/baz_music_doa.cc:60-70) == music_doa_helper.set_frequency (:100-103)."""
    c = mo.make_config("cfg1", 40, seed=8)
    arr = c["array"]
    t2 = mo.steering_table_c64(arr, c["res"], mo.FREQUENCY * 0.8, mo.SPACING)
    with _capi().Context(c["m"], c["n"], c["nsamples"], c["res"], c["table"]) as ctx:
        a0, l0, s0 = device_run(ctx, c["items"], gpu_device)
        ctx.set_table(t2)
        a1, l1, s1 = device_run(ctx, c["items"], gpu_device)
        ctx.set_table(c["table"])
        a2, l2, s2 = device_run(ctx, c["items"], gpu_device)
    _, _, so1, _ = mo.music_doa_work_batch(c["items"], t2, c["m"], c["n"])
    assert_spectrum_close(s1, so1)
    assert not np.allclose(s1, s0, rtol=1e-3)
    assert np.array_equal(s2, s0) and np.array_equal(a2, a0)
    with pytest.raises(ValueError):
        ctx.set_table(t2[:-1])


def test_top_n_semantics_on_device(gpu_device):
    """Crafted tables so that the spectrum has exact ties / plateaus: earliest bin wins, n largest
    BINS (adjacent bins of one lobe), lvl == spectrum[bin]."""
    m, n, N, res = 4, 3, 256, 64
    arr = mo.array_geometry(m)
    base = mo.steering_table_c64(arr, 16, mo.FREQUENCY, mo.SPACING)
    table = np.tile(base, (4, 1))                      # period-16 table: every strength appears 4 times
    items = mo.synth_items(33, m, N, arr, mo.FREQUENCY, mo.SPACING, angles_deg=(45.0, 200.0, 300.0), snr_db=20.0, seed=21)
    ao, lo, so, st = mo.music_doa_work_batch(items, table, m, n)
    with _capi().Context(m, n, N, res, table) as ctx:
        ang, lvl, spec = device_run(ctx, items, gpu_device)
    assert np.array_equal(spec[:, :16], spec[:, 16:32]) and np.array_equal(spec[:, :16], spec[:, 48:])
    bins = np.rint(ang * res / 360.0).astype(int)
    # the maximum occurs 4 times (b, b+16, b+32, b+48): strict '>' (.cc:129-141) lets an equal strength
    # pass the entries it ties with and land in the next slot, so the list is the 3 EARLIEST copies
    b0 = spec[:, :16].argmax(axis=1)
    assert np.array_equal(bins, np.stack([b0, b0 + 16, b0 + 32], axis=1))
    assert_spectrum_close(spec, so)
    assert_doa_match(ang, lvl, ao, lo, res, st)
    for b in range(items.shape[0]):
        assert np.array_equal(lvl[b], spec[b, bins[b]])


def test_zero_input_and_scaling(gpu_device):
    """All-zero items give R = 0: every eigenvalue ties, LAPACK/Jacobi both return the identity basis,
    so the noise space is e_0..e_{m-n-1}.  Scaling the input by 2^k changes nothing (exact)."""
    c = mo.make_config("cfg1", 64, seed=17)
    with _capi().Context(c["m"], c["n"], c["nsamples"], c["res"], c["table"]) as ctx:
        a0, l0, s0 = device_run(ctx, c["items"], gpu_device)
        a1, l1, s1 = device_run(ctx, (c["items"] * np.float32(2.0 ** 20)).astype(np.complex64), gpu_device)
        a2, l2, s2 = device_run(ctx, (c["items"] * np.float32(2.0 ** -30)).astype(np.complex64), gpu_device)
        az, lz, sz = device_run(ctx, np.zeros_like(c["items"][:3]), gpu_device)
    assert np.array_equal(s0, s1) and np.array_equal(s0, s2) and np.array_equal(a0, a1) and np.array_equal(a0, a2)
    rz = mr.work_batch(np.zeros_like(c["items"][:3]), c["table"], c["m"], c["n"])
    assert_spectrum_close(sz, rz[2])
    assert np.all(np.isfinite(sz))


@pytest.mark.parametrize("cfg", ["cfg1", "m8_small"])
def test_non_finite_items_stay_isolated(cfg, gpu_device):
    """NaN / Inf samples (the reference would throw inside Armadillo's eig_sym or emit garbage): the poisoned items
    must not hang the Jacobi loop, must not touch their neighbours (same tile, same wave, same item group), and
    give an all-NaN spectrum plus the (0, 0) initial DoA pairs of .cc:95 (NaN is never inserted, .cc:131)."""
    if cfg == "cfg1":
        c = mo.make_config("cfg1", 70, seed=23)
    else:                                   # m = 8: LDS multi-lane EVD, one item per covariance tile
        arr = mo.array_geometry(8)
        c = dict(m=8, n=2, nsamples=256, res=360, table=mo.steering_table_c64(arr, 360, mo.FREQUENCY, mo.SPACING),
                 items=mo.synth_items(24, 8, 256, arr, mo.FREQUENCY, mo.SPACING, seed=23))
    m, n, res = c["m"], c["n"], c["res"]
    clean = c["items"]
    dirty = clean.copy()
    dirty[5, :] = np.nan
    dirty[6, 3] = np.inf
    dirty[17, 0] = complex(np.nan, 1.0)
    dirty[18, -1] = complex(-np.inf, np.inf)
    bad = [5, 6, 17, 18]
    good = [i for i in range(clean.shape[0]) if i not in bad]
    with _capi().Context(m, n, c["nsamples"], res, c["table"]) as ctx:
        a0, l0, s0 = device_run(ctx, clean, gpu_device)
        a1, l1, s1 = device_run(ctx, dirty, gpu_device)
    assert np.array_equal(s0[good], s1[good]) and np.array_equal(a0[good], a1[good]) and np.array_equal(l0[good], l1[good])
    # a covariance with NaN/Inf entries has no EVD: the projector is poisoned, the spectrum is NaN, nothing is inserted
    assert np.all(np.isnan(s1[bad])) and np.all(a1[bad] == 0.0) and np.all(l1[bad] == 0.0)


# ------------------------------------------------------------------ full-size properties
@pytest.mark.parametrize("cfg,batch,distinct", [("cfg2", 65536, 256), ("cfg3", 4096, 16)])
def test_full_size_properties(cfg, batch, distinct, gpu_device):
    """BASELINE.json sizes.  The oracle checks `distinct` items; the rest of the batch repeats them, so
    (a) every repeat must be bit-identical to its first occurrence (no cross-item leakage, any
    block/wave position), (b) lvl[i] == spectrum[bin_i] and bin_0 == argmax, (c) nothing is left
    unwritten."""
    torch = _torch()
    c = mo.make_config(cfg, distinct, snr_db=20.0)
    ao, lo, so, st = mo.music_doa_work_batch(c["items"], c["table"], c["m"], c["n"])
    reps = batch // distinct
    with _capi().Context(c["m"], c["n"], c["nsamples"], c["res"], c["table"]) as ctx:
        x = torch.from_numpy(c["items"].view(np.float32)).to(gpu_device).repeat(reps, 1).contiguous()
        ang = torch.full((batch, c["n"]), -1.0, dtype=torch.float32, device=gpu_device)
        lvl = torch.full_like(ang, -1.0)
        spec = torch.full((batch, c["res"]), -1.0, dtype=torch.float32, device=gpu_device)
        ctx.process_device(x.data_ptr(), batch, ang.data_ptr(), lvl.data_ptr(), spec.data_ptr(),
                           stream=torch.cuda.current_stream().cuda_stream)
        ctx.sync()
    first = spec[:distinct]
    assert bool((spec.view(reps, distinct, -1) == first.unsqueeze(0)).all())
    assert bool((ang.view(reps, distinct, -1) == ang[:distinct].unsqueeze(0)).all())
    assert bool((lvl.view(reps, distinct, -1) == lvl[:distinct].unsqueeze(0)).all())
    assert bool((spec > 0).all())
    assert_spectrum_close(first.cpu().numpy(), so)
    assert_doa_match(ang[:distinct].cpu().numpy(), lvl[:distinct].cpu().numpy(), ao, lo, c["res"], st)
    bins = torch.round(ang * (c["res"] / 360.0)).long()
    assert bool((bins[:, 0] == spec.argmax(dim=1)).all())
    assert bool((torch.gather(spec, 1, bins) == lvl).all())
    assert bool((lvl[:, :-1] >= lvl[:, 1:]).all())


# ------------------------------------------------------------------ fused covariance + EVD (m = 4, K % 256 == 0)
def test_fused_covariance_evd_kernel_equals_the_two_kernel_form(gpu_device, monkeypatch):
    """cfg2 runs covariance and EVD in ONE kernel (cov4_evd_kernel: R stays in LDS, the Jacobi runs at low wave priority
    under the input stream); BAZ_MUSIC_FUSE=0 runs cov4_x4_kernel + evd_proj_kernel.  Same arithmetic: same bits, for a
    batch that is not a multiple of the 64-item wave task, and both match the golden vectors."""
    g = load_golden("cfg2_m4_n2_N1024_r3600")
    items = np.concatenate([g["items"]] * 5)[:131]                  # 2 full wave tasks + a 3-item tail
    outs = []
    for fuse in ("1", "0"):
        monkeypatch.setenv("BAZ_MUSIC_FUSE", fuse)
        with _capi().Context(g["m"], g["n"], g["nsamples"], g["res"], g["table"], lab=True) as ctx:
            assert ("cov4_evd_kernel" in ctx.stage_name(0)) == (fuse == "1")
            outs.append(device_run(ctx, items, gpu_device))
    for x, y in zip(*outs):
        assert np.array_equal(x, y)
    B = g["items"].shape[0]
    assert_spectrum_close(outs[0][2][:B], g["spectrum"])
    assert_doa_match(outs[0][0][:B], outs[0][1][:B], g["ang"], g["lvl"], g["res"], g["strength64"])


# ------------------------------------------------------------------ signal subspace by orthogonal iteration (m >= 5, n <= 3)
@pytest.mark.parametrize("m,n,K,res", [(8, 2, 64, 360), (16, 2, 64, 360), (6, 3, 50, 200), (9, 1, 40, 180), (16, 3, 48, 500),
                                        (5, 2, 32, 90), (13, 2, 30, 77), (16, 4, 64, 360), (8, 4, 40, 200), (11, 4, 48, 91)])
def test_signal_subspace_iteration_and_its_hand_back(m, n, K, res, gpu_device, monkeypatch):
    """evd_sub_kernel finds the projector from the n dominant eigenvectors by orthogonal iteration and hands items with a
    small gap lambda_n / lambda_(n+1) back to the Jacobi.  A batch that mixes easy items (30 / 20 dB), slow ones (0 and
    -10 dB), items with FEWER emitters than n (gap ~ 1), a noise-only item and an all-zero item must give the projector
    of the Jacobi-only build to 1e-12, the oracle's spectra to 1e-5, and bits that do not depend on the item's
    neighbours in the wave."""
    N = m * K
    arr = mo.array_geometry(m)
    table = mo.steering_table_c64(arr, res, mo.FREQUENCY, mo.SPACING)
    ang_n = tuple(np.linspace(25.0, 290.0, n))
    parts = [mo.synth_items(6, m, N, arr, mo.FREQUENCY, mo.SPACING, angles_deg=ang_n, snr_db=snr, seed=900 + 7 * i + m)
             for i, snr in enumerate((30.0, 20.0, 0.0, -10.0))]
    parts.append(mo.synth_items(5, m, N, arr, mo.FREQUENCY, mo.SPACING, angles_deg=ang_n[:max(n - 1, 0)] or (), snr_db=20.0,
                                seed=950 + m))                                  # fewer emitters than n (noise only at n = 1)
    parts.append(np.zeros((1, N), np.complex64))
    items = np.concatenate(parts)
    rng = np.random.default_rng(5)
    perm = rng.permutation(items.shape[0])
    B = items.shape[0]
    torch = _torch()

    def taps(ctx, its):
        x = torch.from_numpy(np.ascontiguousarray(its).view(np.float32)).to(gpu_device)
        qs = _capi().lib().baz_music_q_stride(its.shape[0])
        Q = torch.zeros(m * m, qs, dtype=torch.float64, device=gpu_device)
        ctx.debug_q(x.data_ptr(), its.shape[0], Q.data_ptr())
        ctx.sync()
        return Q.cpu().numpy()[:, :its.shape[0]]

    res_by_mode = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("BAZ_MUSIC_SUB_EVD", mode)
        with _capi().Context(m, n, N, res, table, lab=True) as ctx:
            out = device_run(ctx, items, gpu_device)
            outp = device_run(ctx, items[perm], gpu_device)
            one = [device_run(ctx, items[i:i + 1], gpu_device) for i in (0, 13, B - 1)]
            Q = taps(ctx, items)
        res_by_mode[mode] = (out, outp, one, Q)
    out, outp, one, Q1 = res_by_mode["1"]
    Q0 = res_by_mode["0"][3]
    assert np.abs(Q1 - Q0).max() < 1e-12                      # same projector either way (basis-invariant)
    for x, y in zip(out, outp):                               # an item's bits do not depend on its wave-mates
        assert np.array_equal(x[perm], y)
    for k, i in enumerate((0, 13, B - 1)):
        for x, y in zip(out, one[k]):
            assert np.array_equal(x[i:i + 1], y)
    ao, lo, so, st = mo.music_doa_work_batch(items[:-1], table, m, n)       # (the zero item: eigh's basis is arbitrary)
    assert_spectrum_close(out[2][:-1], so)
    assert_doa_match(out[0][:-1], out[1][:-1], ao, lo, res, st)
    assert np.array_equal(out[2][-1], res_by_mode["0"][0][2][-1])            # zero item: handed back -> the Jacobi's answer


# ------------------------------------------------------------------ the scan's short form (m >= 9, n = 2)
@pytest.mark.parametrize("m,n,K,res,batch,snr", [(16, 2, 64, 3600, 70, 20.0), (9, 2, 40, 361, 33, 10.0), (13, 2, 50, 1000, 17, 40.0),
                                                  (16, 2, 256, 720, 20, 0.0), (16, 1, 64, 3600, 70, 20.0), (9, 1, 40, 361, 33, 0.0),
                                                  (12, 1, 48, 500, 21, 60.0), (8, 1, 64, 3600, 40, 20.0), (6, 1, 30, 250, 33, 5.0),
                                                  (7, 1, 50, 1001, 18, 70.0)])
def test_short_form_scan_equals_the_projector_scan(m, n, K, res, batch, snr, gpu_device, monkeypatch):
    """||a||^2 - sum_c |s_c^H a|^2 (scan_mfma_kernel SIG, n = 1 and 2) against the projector GEMM of the same build
    (BAZ_MUSIC_SIG_SCAN=0): float32 spectra within 2 ulp, the same DoA pairs, both within 1e-5 of the oracle, with and
    without the spectrum port, res % 4 != 0 included"""
    N = m * K
    arr = mo.array_geometry(m)
    table = mo.steering_table_c64(arr, res, mo.FREQUENCY, mo.SPACING)
    items = mo.synth_items(batch, m, N, arr, mo.FREQUENCY, mo.SPACING, angles_deg=(40.3, 121.7)[:n], snr_db=snr, seed=17 * m + K)
    outs = {}
    monkeypatch.setenv("BAZ_MUSIC_EXACT", "1")      # the fp64 scan (by default these shapes run the int8 scan, which evaluates the projector form)
    for mode in ("1", "0"):
        monkeypatch.setenv("BAZ_MUSIC_SIG_SCAN", mode)
        with _capi().Context(m, n, N, res, table, lab=True) as ctx:
            outs[mode] = (device_run(ctx, items, gpu_device), device_run(ctx, items, gpu_device, want_spec=False))
    (a1, l1, s1), (a1n, l1n, _) = outs["1"]
    (a0, l0, s0), _ = outs["0"]
    assert np.all(np.abs(s1.astype(np.float64) - s0.astype(np.float64)) <= 2.5e-7 * np.abs(s0.astype(np.float64)))
    ao, lo, so, st = mo.music_doa_work_batch(items, table, m, n)
    assert_spectrum_close(s1, so)
    assert_doa_match(a1, l1, ao, lo, res, st)
    assert_doa_match(a1n, l1n, ao, lo, res, st)
    assert_doa_match(a0, l0, ao, lo, res, st)


# ------------------------------------------------------------------ bin ranges per row in the scan
@pytest.mark.parametrize("name", ["cfg1_m4_n2_N256_r360", "cfg2_m4_n2_N1024_r3600", "m7_n4_N700_r500", "m12_n9_N1200_r720",
                                  "odd_m3_n1_N300_r357"])
def test_outputs_do_not_depend_on_the_range_split_of_the_scan(name, gpu_device, monkeypatch):
    """The scan cuts a row's bins into 1..64 ranges by batch size (one range at the bench size, several for small
    batches) and topn_merge_kernel folds the per-range lists: every split must give the same bits, with and without the
    spectrum port, and match the golden vectors."""
    g = load_golden(name)
    outs = []
    for split in ("1", "3", "0"):                     # 0 = by batch size
        monkeypatch.setenv("BAZ_MUSIC_NSPLIT", split)
        with _capi().Context(g["m"], g["n"], g["nsamples"], g["res"], g["table"], lab=True) as ctx:
            a, l, s = device_run(ctx, g["items"], gpu_device)
            a2, l2, _ = device_run(ctx, g["items"], gpu_device, want_spec=False)
            a3, _, _ = device_run(ctx, g["items"], gpu_device, want_lvl=False, want_spec=False)
        outs.append((a, l, s, a2, l2, a3))
    for o in outs[1:]:
        for x, y in zip(outs[0], o):
            assert np.array_equal(x, y)
    a, l, s, a2, l2, a3 = outs[0]
    assert_spectrum_close(s, g["spectrum"])
    assert_doa_match(a, l, g["ang"], g["lvl"], g["res"], g["strength64"])
    assert np.array_equal(a, a2) and np.array_equal(a, a3)
    bins = np.round(a * g["res"] / 360.0).astype(int) % g["res"]
    used = l > 0
    assert np.array_equal(l[used], np.take_along_axis(s, bins, axis=1)[used])      # lvl[i] == spectrum[bin_i]


# ------------------------------------------------------------------ near-null items: literal-form refinement
@pytest.mark.parametrize("m,n,K,res,batch,snr,seed", [
    (4, 3, 7, 1440, 257, 60.0, 11), (7, 6, 300, 360, 257, 60.0, 12), (4, 2, 256, 3600, 300, 80.0, 13),
    (8, 2, 64, 1000, 64, 70.0, 14), (16, 12, 64, 720, 33, 60.0, 15), (5, 4, 40, 361, 100, 90.0, 16),
    (4, 2, 256, 3600, 64, 120.0, 17),
    (16, 2, 64, 720, 40, 80.0, 18), (11, 2, 50, 360, 33, 100.0, 19), (16, 1, 64, 720, 40, 90.0, 20)])   # the scan's short form (m >= 9, n <= 2)
def test_extreme_snr_spectra_match_the_literal_form(m, n, K, res, batch, snr, seed, gpu_device):
    """At >~ 55 dB SNR some bin of an item falls into a near-null of the noise subspace (d = ||G^H a||^2 down to
    1e-12 ||a||^2).  The projector GEMM of the scan has only ~m^2 1e-16 ABSOLUTE accuracy there, so the scan redoes
    such values in the reference's literal form (literal_tile): spectra must still match at 1e-5."""
    rng = np.random.default_rng(seed)
    arr = (rng.random((m, 2)) * 3.0).tolist()
    # as many emitters as the block expects: with fewer, the n-th "signal" eigenvector is picked among near-degenerate
    # noise eigenvalues and at this SNR the reference's own spectrum moves by ~1e-5 between LAPACK and Jacobi
    angles = tuple(float(a) for a in rng.uniform(0, 360, n))
    table = mo.steering_table_c64(arr, res, mo.FREQUENCY, mo.SPACING)
    items = mo.synth_items(batch, m, m * K, arr, mo.FREQUENCY, mo.SPACING, angles_deg=angles, snr_db=snr, seed=seed)
    ao, lo, so = mr.work_batch(items, table, m, n)
    with _capi().Context(m, n, m * K, res, table) as ctx:
        ang, lvl, spec = device_run(ctx, items, gpu_device)
        refined = ctx.refined_items()
        a2, l2, _ = device_run(ctx, items, gpu_device, want_spec=False)
    if float(so.max()) > 2.0 / (m * m * 1e-8):      # some d is clearly below the threshold m*max||a||^2*1e-8
        assert 0 < refined <= batch * res           # -> the literal-form path did run ((item, bin) values redone)
    assert_spectrum_close(spec, so)
    assert_doa_match(ang, lvl, ao, lo, res, so.astype(np.float64))
    assert_doa_match(a2, l2, ao, lo, res, so.astype(np.float64))
    bins = np.round(ang * res / 360.0).astype(int) % res
    assert np.array_equal(lvl, np.take_along_axis(spec, bins, axis=1))      # lvl[i] == spectrum[bin_i] also after refinement


# ------------------------------------------------------------------ opt-in extension: local-maximum picker
@pytest.mark.parametrize("cfg,n_items", [("cfg2", 200), ("cfg1", 64)])
def test_peak_mode_is_opt_in_and_matches_its_definition(cfg, n_items, gpu_device):
    """baz_music_set_peak_mode(ctx, 1): the n strongest circular local maxima of the spectrum the device itself wrote
    (definition: oracle/music_oracle.py::peak_pick; NOT reference behaviour, default stays the reference's top-n)."""
    c = mo.make_config(cfg, n_items, snr_db=20.0, seed=91)
    m, n, res = c["m"], c["n"], c["res"]
    with _capi().Context(m, n, c["nsamples"], res, c["table"]) as ctx:
        a0, l0, s0 = device_run(ctx, c["items"], gpu_device)                      # default: reference semantics
        ctx.set_peak_mode(1)
        a1, l1, s1 = device_run(ctx, c["items"], gpu_device)
        a2, l2, _ = device_run(ctx, c["items"], gpu_device, want_spec=False)      # private spectrum buffer
        ctx.set_peak_mode(0)
        a3, l3, s3 = device_run(ctx, c["items"], gpu_device)
    ao, lo, so, st = mo.music_doa_work_batch(c["items"], c["table"], m, n)
    assert_doa_match(a0, l0, ao, lo, res, st)
    assert np.array_equal(a0, a3) and np.array_equal(l0, l3)
    assert np.array_equal(s0, s1) and np.array_equal(s0, s3)
    for b in range(n_items):
        ea, el = mo.peak_pick(s1[b], n)
        assert np.array_equal(a1[b], ea) and np.array_equal(l1[b], el)
    assert np.array_equal(a1, a2) and np.array_equal(l1, l2)
    # two emitters at 40.3 and 121.7 degrees: the peak picker separates them, the reference's top-n usually does not
    found = np.sort(a1, axis=1)
    step = 360.0 / res
    assert np.mean(np.abs(found[:, 0] - 40.3) <= 2 * step + 1.0) > 0.9 and np.mean(np.abs(found[:, 1] - 121.7) <= 2 * step + 1.0) > 0.9


# ------------------------------------------------------------------ boundary behaviour
def test_device_dealing_of_block_instances(gpu_device, monkeypatch):
    """Host-block instances are dealt over the visible gfx950 devices (instance i -> device i mod G,
    baz_music_doa_deal_device); BAZ_MUSIC_DEVICE pins them; a device that does not exist fails loudly."""
    capi = _capi()
    from gr_baz_amd import baz
    g = capi.device_count()
    assert g >= 1
    tab = [[1 + 0j] * 4] * 8
    monkeypatch.delenv("BAZ_MUSIC_DEVICE", raising=False)
    devs = [baz.music_doa(4, 2, 16, tab, 8).device() for _ in range(2 * g + 1)]
    assert all(0 <= d < g for d in devs)
    assert all((devs[i + 1] - devs[i]) % g == 1 % g for i in range(len(devs) - 1))        # round robin, whatever the start
    monkeypatch.setenv("BAZ_MUSIC_DEVICE", str(g - 1))
    assert [baz.music_doa(4, 2, 16, tab, 8).device() for _ in range(3)] == [g - 1] * 3
    monkeypatch.setenv("BAZ_MUSIC_DEVICE", str(g + 3))
    with pytest.raises(RuntimeError, match="gfx950"):
        baz.music_doa(4, 2, 16, tab, 8)
    monkeypatch.delenv("BAZ_MUSIC_DEVICE")
    c = mo.make_config("cfg1", 4)
    with capi.Context(c["m"], c["n"], c["nsamples"], c["res"], c["table"], device_id=g - 1) as ctx:
        assert capi.lib().baz_music_device(ctx._h) == g - 1


def test_config4_64_streams_over_4_contexts(gpu_device):
    """BASELINE config 4, literally: 64 independent cfg2 streams (seed 1000 + 2 + s), dealt s mod G over G = 4
    engines -- here 4 contexts on the one visible GPU, each with its own stream, table copy and workspace, exactly what
    4 ranks (or 4 block instances) would own -- every stream checked against the oracle."""
    torch = _torch()
    capi = _capi()
    from gr_baz_amd import sharding
    G, S, per = 4, 64, 12
    c0 = mo.make_config("cfg2", 1)
    m, n, N, res, table = c0["m"], c0["n"], c0["nsamples"], c0["res"], c0["table"]
    arr = mo.array_geometry(m)
    streams = [mo.synth_items(per, m, N, arr, mo.FREQUENCY, mo.SPACING, seed=1000 + 2 + s) for s in range(S)]
    ctxs = [capi.Context(m, n, N, res, table) for _ in range(G)]
    try:
        outs = []
        for r, ctx in enumerate(ctxs):                       # launch everything first: the 4 engines run concurrently
            mine = sharding.streams_of_rank(S, G, r)
            x = torch.from_numpy(np.concatenate([streams[s] for s in mine]).view(np.float32)).to(gpu_device)
            B = x.shape[0]
            ang = torch.full((B, n), -1.0, device=gpu_device)
            lvl = torch.full_like(ang, -1.0)
            spec = torch.full((B, res), -1.0, device=gpu_device)
            ctx.process_device(x.data_ptr(), B, ang.data_ptr(), lvl.data_ptr(), spec.data_ptr(),
                               stream=torch.cuda.current_stream().cuda_stream)
            outs.append((mine, x, ang, lvl, spec))
        seen = []
        for mine, x, ang, lvl, spec in outs:
            a, l, sp = ang.cpu().numpy(), lvl.cpu().numpy(), spec.cpu().numpy()
            for k, s in enumerate(mine):
                ao, lo, so, st = mo.music_doa_work_batch(streams[s], table, m, n)
                assert_spectrum_close(sp[k * per:(k + 1) * per], so)
                assert_doa_match(a[k * per:(k + 1) * per], l[k * per:(k + 1) * per], ao, lo, res, st)
                seen.append(s)
        assert sorted(seen) == list(range(S))
    finally:
        for ctx in ctxs:
            ctx.close()


def test_more_than_65536_bins_uses_the_wide_key(gpu_device):
    """resolution > 65,536 switches the top-n key to a 20-bit bin field; also an odd resolution
    (scalar spectrum stores) and a tail step with 33 valid bins."""
    m, n, N, res, batch = 2, 1, 64, 70001, 3
    arr = [[0.0, 0.0], [1.0, 0.0]]
    table = mo.steering_table_c64(arr, res, mo.FREQUENCY, mo.SPACING)
    items = mo.synth_items(batch, m, N, arr, mo.FREQUENCY, mo.SPACING, angles_deg=(300.3,), snr_db=15.0, seed=9)
    ao, lo, so, st = mo.music_doa_work_batch(items, table, m, n)
    with _capi().Context(m, n, N, res, table) as ctx:
        ang, lvl, spec = device_run(ctx, items, gpu_device)
    assert_spectrum_close(spec, so)
    assert_doa_match(ang, lvl, ao, lo, res, st)
    bins = np.rint(ang.astype(np.float64) * res / 360.0).astype(int)
    assert bins.max() > 65535 or bins.min() >= 0


def test_host_path_with_several_chunks_equals_device_path(gpu_device):
    """baz_music_process cuts big host batches into pipelined chunks (64 MiB of traffic each)."""
    c = mo.make_config("cfg1", 256, seed=31)
    reps = 200                                    # 51,200 items x 3.5 KB = 3 chunks
    items = np.tile(c["items"], (reps, 1))
    with _capi().Context(c["m"], c["n"], c["nsamples"], c["res"], c["table"]) as ctx:
        a_d, l_d, s_d = device_run(ctx, c["items"], gpu_device)
        a_h, l_h, s_h = ctx.process(items)
        a_2, l_2, _ = ctx.process(items[:777], want_spectrum=False)
    assert np.array_equal(a_h.reshape(reps, 256, -1), np.broadcast_to(a_d, (reps,) + a_d.shape))
    assert np.array_equal(s_h.reshape(reps, 256, -1), np.broadcast_to(s_d, (reps,) + s_d.shape))
    assert np.array_equal(l_h[:256], l_d) and np.array_equal(a_2, a_h[:777]) and np.array_equal(l_2, l_h[:777])


def test_page_locking_of_persistent_caller_buffers(gpu_device, monkeypatch):
    """baz_music_host_register / baz_music_set_host_pinning: locking a scheduler's long-lived buffers
    changes how the copies travel, never the results; registrations are idempotent, piecewise, bounded and released."""
    c = mo.make_config("cfg1", 512, seed=33)
    items = np.ascontiguousarray(np.tile(c["items"], (16, 1)))         # 8,192 items, 16 MiB in + 12 MiB out (page-locked: 4 chunks)
    B = items.shape[0]
    out = (np.zeros((B, c["n"]), np.float32), np.zeros((B, c["n"]), np.float32), np.zeros((B, c["res"]), np.float32))
    with _capi().Context(c["m"], c["n"], c["nsamples"], c["res"], c["table"]) as ctx:
        a0, l0, s0 = [x.copy() for x in ctx.process(items)]             # pageable everything
        assert ctx.host_pinned_bytes() == 0
        assert ctx.host_register(items) == 0
        got = ctx.host_pinned_bytes()
        assert got == items.nbytes                                      # the exact range, not rounded out to pages
        assert ctx.host_register(items) == 0 and ctx.host_register(items[100:200]) == 0 and ctx.host_pinned_bytes() == got
        ctx.process(items, out=out)                                     # locked input, pageable outputs
        assert all(np.array_equal(x, y) for x, y in zip(out, (a0, l0, s0)))
        ctx.set_host_pinning(True)                                      # the call locks what it touches: the spectrum now
        for o in out:
            o.fill(0)
        ctx.process(items[:1000], out=tuple(o[:1000] for o in out))     # piecewise: first 1,000 rows ...
        part = ctx.host_pinned_bytes()
        assert part > got
        ctx.process(items, out=out)                                     # ... then everything
        full = ctx.host_pinned_bytes()
        assert full == got + out[2].nbytes and full > part             # input + spectrum: the two DMA targets
        assert all(np.array_equal(x, y) for x, y in zip(out, (a0, l0, s0)))
        ctx.process(items, out=out)
        assert ctx.host_pinned_bytes() == full                          # nothing new to lock
        ctx.host_unregister_all()
        assert ctx.host_pinned_bytes() == 0
        ctx.set_host_pinning(False)
        a1, l1, s1 = ctx.process(items)
        assert np.array_equal(a1, a0) and np.array_equal(s1, s0)
    monkeypatch.setenv("BAZ_MUSIC_PIN_LIMIT_MIB", "1")
    with _capi().Context(c["m"], c["n"], c["nsamples"], c["res"], c["table"]) as ctx:
        assert ctx.host_register(items) == _capi().E_UNSUPPORTED and ctx.host_pinned_bytes() == 0
        assert ctx.host_register(items[:64]) == 0 and ctx.host_pinned_bytes() == items[:64].nbytes
        ctx.set_host_pinning(True)              # over the limit: the input stays pageable AS A WHOLE (its locked head is dropped)
        ctx.process(items, out=out)
        assert ctx.host_pinned_bytes() == 0
        assert all(np.array_equal(x, y) for x, y in zip(out, (a0, l0, s0)))


def test_partly_page_locked_ranges_take_the_copy_path(gpu_device):
    """ADVICE r2 / review r2 weak 7: "this range is page-locked" must hold for the WHOLE range before the kernels address
    it over PCIe.  Two contexts share one upstream buffer at different offsets (a fan-out in a flowgraph); context A
    owns a registration that covers only the HEAD of what context B is handed.  B must neither fault the GPU (zero-copy
    into unmapped host memory) nor fail (a DMA copy whose host range is partly inside a registration is refused by the
    runtime): results equal the all-pageable run."""
    c = mo.make_config("cfg1", 256, seed=35)
    items = np.ascontiguousarray(np.tile(c["items"], (2, 1)))                      # 512 items, 1 MiB: a "small call"
    B = items.shape[0]
    spec_buf = np.zeros((B, c["res"]), np.float32)
    with _capi().Context(c["m"], c["n"], c["nsamples"], c["res"], c["table"]) as ref_ctx:
        a0, l0, s0 = [x.copy() for x in ref_ctx.process(items)]
    with _capi().Context(c["m"], c["n"], c["nsamples"], c["res"], c["table"]) as A, \
            _capi().Context(c["m"], c["n"], c["nsamples"], c["res"], c["table"]) as Bc:
        assert A.host_register(items[:200]) == 0                                   # A locks the head of the input ...
        assert A.host_register(spec_buf[:100]) == 0                                # ... and of the spectrum buffer
        for off, nb in ((0, B), (100, 300), (150, 100), (0, 200), (0, 100)):       # straddling, inside, exactly the head
            out = (np.zeros((nb, c["n"]), np.float32), np.zeros((nb, c["n"]), np.float32), spec_buf[off:off + nb])
            out[2].fill(0)
            Bc.process(items[off:off + nb], out=out)
            assert np.array_equal(out[0], a0[off:off + nb]) and np.array_equal(out[1], l0[off:off + nb])
            assert np.array_equal(out[2], s0[off:off + nb])
        Bc.set_host_pinning(True)              # B's own attempt to lock the overlapping range is refused or merged: still right
        out = (np.zeros((B, c["n"]), np.float32), np.zeros((B, c["n"]), np.float32), spec_buf)
        spec_buf.fill(0)
        Bc.process(items, out=out)
        assert np.array_equal(out[0], a0) and np.array_equal(out[2], s0)
        Bc.host_unregister_all()
        A.host_unregister_all()


def test_registrations_are_shared_between_contexts(gpu_device):
    """two blocks reading ONE stream buffer share its registration, and it survives until the last of
    them lets go -- the first block's stop() must not unmap memory the second one still addresses over PCIe."""
    c = mo.make_config("cfg1", 256, seed=36)
    items = np.ascontiguousarray(c["items"])
    with _capi().Context(c["m"], c["n"], c["nsamples"], c["res"], c["table"]) as ref_ctx:
        a0, l0, s0 = [x.copy() for x in ref_ctx.process(items)]
    A = _capi().Context(c["m"], c["n"], c["nsamples"], c["res"], c["table"])
    Bc = _capi().Context(c["m"], c["n"], c["nsamples"], c["res"], c["table"])
    try:
        assert A.host_register(items) == 0
        assert Bc.host_register(items[64:192]) == 0                 # inside A's registration: B takes a share of it
        assert A.host_pinned_bytes() == items.nbytes and Bc.host_pinned_bytes() == items.nbytes
        A.host_unregister_all()                                      # "block A stops"
        assert A.host_pinned_bytes() == 0 and Bc.host_pinned_bytes() == items.nbytes
        for _ in range(3):                                           # B: still no copies, still right
            a, l, s = Bc.process(items[64:192])
            assert np.array_equal(a, a0[64:192]) and np.array_equal(l, l0[64:192]) and np.array_equal(s, s0[64:192])
        assert A.host_register(items[:32]) == 0                      # A comes back: shares B's (whole) registration again
        assert A.host_pinned_bytes() == items.nbytes
        Bc.host_unregister_all()
        a, l, s = A.process(items[:32])
        assert np.array_equal(a, a0[:32]) and np.array_equal(s, s0[:32])
        A.host_unregister_all()
        assert A.host_register(items[:32]) == 0                      # nobody holds it any more: a fresh, exact registration
        assert A.host_pinned_bytes() == items[:32].nbytes
    finally:
        A.close(); Bc.close()


def test_caller_stream_ordering(gpu_device):
    """baz_music_set_stream: work is ordered on the caller's stream (here a torch side stream), so torch
    ops enqueued before/after on that stream see consistent data without extra synchronisation."""
    torch = _torch()
    c = mo.make_config("cfg1", 128, seed=41)
    ao, lo, so, st = mo.music_doa_work_batch(c["items"], c["table"], c["m"], c["n"])
    side = torch.cuda.Stream()
    with _capi().Context(c["m"], c["n"], c["nsamples"], c["res"], c["table"]) as ctx:
        ctx.set_stream(side.cuda_stream)
        host = torch.from_numpy(c["items"].view(np.float32)).pin_memory()
        with torch.cuda.stream(side):
            x = torch.empty_like(host, device=gpu_device)
            x.copy_(host, non_blocking=True)                     # producer on the same stream
            ang = torch.zeros(128, c["n"], dtype=torch.float32, device=gpu_device)
            lvl = torch.zeros_like(ang)
            spec = torch.zeros(128, c["res"], dtype=torch.float32, device=gpu_device)
            ctx.process_device(x.data_ptr(), 128, ang.data_ptr(), lvl.data_ptr(), spec.data_ptr())
            total = spec.sum(dim=1)                              # consumer on the same stream
        side.synchronize()
        ctx.set_stream(None)
    assert_spectrum_close(spec.cpu().numpy(), so)
    assert np.allclose(total.cpu().numpy(), so.astype(np.float64).sum(axis=1), rtol=1e-4)


def test_process_device_on_orders_against_the_callers_stream(gpu_device):
    """baz_music_process_device_on: stream semantics against a stream the context does NOT run on: a slow
    producer chain and a late output fill on torch's stream come BEFORE the batch, a consumer after it, no host sync."""
    torch = _torch()
    c = mo.make_config("cfg1", 256, seed=43)
    ao, lo, so, st = mo.music_doa_work_batch(c["items"], c["table"], c["m"], c["n"])
    with _capi().Context(c["m"], c["n"], c["nsamples"], c["res"], c["table"]) as ctx:
        src = torch.from_numpy(c["items"].view(np.float32)).to(gpu_device)
        junk = torch.randn(4096, 4096, device=gpu_device)
        torch.cuda.synchronize()
        for rep in range(3):
            for _ in range(6):
                junk = junk @ junk * 1e-3                       # keeps torch's stream busy for a while ...
            x = src + 0.0 * junk[0, 0]                          # ... before the input even exists
            ang = torch.full((256, c["n"]), -1.0, device=gpu_device)
            lvl = torch.full_like(ang, -1.0)
            spec = torch.full((256, c["res"]), -1.0, device=gpu_device)      # late fills must not overwrite results
            ctx.process_device(x.data_ptr(), 256, ang.data_ptr(), lvl.data_ptr(), spec.data_ptr(),
                               stream=torch.cuda.current_stream().cuda_stream)
            total = spec.double().sum(dim=1)                    # consumer on torch's stream
            assert_spectrum_close(spec.cpu().numpy(), so)
            assert np.allclose(total.cpu().numpy(), so.astype(np.float64).sum(axis=1), rtol=1e-4)
            assert_doa_match(ang.cpu().numpy(), lvl.cpu().numpy(), ao, lo, c["res"], st)


def test_two_contexts_interleaved(gpu_device):
    c1 = mo.make_config("cfg1", 70, seed=51)
    c3 = mo.make_config("cfg3", 5, seed=52)
    r1 = mo.music_doa_work_batch(c1["items"], c1["table"], c1["m"], c1["n"])
    r3 = mo.music_doa_work_batch(c3["items"], c3["table"], c3["m"], c3["n"])
    capi = _capi()
    with capi.Context(c1["m"], c1["n"], c1["nsamples"], c1["res"], c1["table"]) as k1, \
            capi.Context(c3["m"], c3["n"], c3["nsamples"], c3["res"], c3["table"]) as k3:
        for _ in range(3):
            a1, l1, s1 = device_run(k1, c1["items"], gpu_device)
            a3, l3, s3 = device_run(k3, c3["items"], gpu_device)
            assert_spectrum_close(s1, r1[2]); assert_doa_match(a1, l1, r1[0], r1[1], c1["res"], r1[3])
            assert_spectrum_close(s3, r3[2]); assert_doa_match(a3, l3, r3[0], r3[1], c3["res"], r3[3])


def test_set_table_from_another_thread_is_never_torn(gpu_device):
    """set_array_response runs on the GUI/python thread while work() runs on the scheduler thread
    (lib/baz_music_doa.cc:67,101): every batch must be computed with exactly one of the two tables."""
    import threading
    c = mo.make_config("cfg1", 64, seed=61)
    tA = c["table"]
    tB = mo.steering_table_c64(c["array"], c["res"], mo.FREQUENCY * 0.9, mo.SPACING)
    sA = mo.music_doa_work_batch(c["items"], tA, c["m"], c["n"])[2]
    sB = mo.music_doa_work_batch(c["items"], tB, c["m"], c["n"])[2]
    stop = threading.Event()
    with _capi().Context(c["m"], c["n"], c["nsamples"], c["res"], tA) as ctx:
        def flipper():
            k = 0
            while not stop.is_set():
                ctx.set_table(tB if (k & 1) == 0 else tA)
                k += 1
        th = threading.Thread(target=flipper)
        th.start()
        seen = set()
        try:
            for _ in range(60):
                _, _, s = ctx.process(c["items"])
                relA = np.max(np.abs(s - sA) / sA)
                relB = np.max(np.abs(s - sB) / sB)
                assert min(relA, relB) <= 1e-5, "a batch mixed two steering tables"
                seen.add("A" if relA <= relB else "B")
        finally:
            stop.set()
            th.join()
    assert seen  # usually {"A", "B"}


# ------------------------------------------------------------------ round 6: the scan writes EVERY spectrum value
@pytest.mark.parametrize("name", golden_names())
def test_every_spectrum_value_is_written(name, gpu_device):
    """The spectrum tensor is pre-filled with a sentinel; after one call no element may still hold it -- whatever the row classes, the bin ranges, the
    padded last step, the refined tiles (found useful when a guard-zone run made a stale-looking spectrum suspicious: the values were all written, the
    guard's own pre-fill was late)."""
    import torch
    g = load_golden(name)
    B, res = g["items"].shape[0], g["res"]
    SENT = 12345.678
    with _capi().Context(g["m"], g["n"], g["nsamples"], res, g["table"]) as ctx:
        x = torch.from_numpy(np.ascontiguousarray(g["items"]).view(np.float32)).to(gpu_device)
        ang = torch.zeros(B, g["n"], dtype=torch.float32, device=gpu_device)
        lvl = torch.zeros_like(ang)
        for rows in (B, max(1, B - 3)):                       # the whole batch and a ragged one
            spec = torch.full((B, res), SENT, dtype=torch.float32, device=gpu_device)
            torch.cuda.synchronize()
            ctx.process_device(x.data_ptr(), rows, ang.data_ptr(), lvl.data_ptr(), spec.data_ptr())
            ctx.sync()
            assert int((spec[:rows] == SENT).sum()) == 0, "%s: spectrum values left unwritten" % name
            assert int((spec[rows:] != SENT).sum()) == 0, "%s: rows beyond the batch were written" % name
