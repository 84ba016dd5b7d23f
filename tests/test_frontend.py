"""BASELINE config 5: the wideband front-end (fractional resampler -> AGC) fused ahead of MUSIC-DoA, 16 antennas.
Device-resident chain through the three C-ABIs, checked against the oracle chain (one oracle resampler and one
oracle AGC per antenna, numpy interleave, MUSIC oracle).  Tolerances as for the single blocks: resampler bit-exact,
AGC / MUSIC 1e-5 relative."""
import numpy as np
import pytest

from helpers import assert_doa_match, assert_spectrum_close
from oracle import agc_ref as ar
from oracle import music_oracle as mo
from oracle import music_ref as mr
from oracle import resamp_ref as rr


def planar_array_streams(m, T, seed):
    """m per-antenna streams of one array capture: two emitters + noise (the per-item generator, de-interleaved)."""
    arr = mo.array_geometry(m)
    K = 64
    items = mo.synth_items((T + K - 1) // K, m, m * K, arr, mo.FREQUENCY, mo.SPACING, seed=seed)
    cols = items.reshape(-1, K, m)                     # [item][column][antenna]
    return np.ascontiguousarray(cols.reshape(-1, m).T[:, :T]), arr     # [antenna][time]


@pytest.mark.gpu
@pytest.mark.parametrize("S,calls", [(16, (1000, 255, 257, 4096, 3)), (4, (300, 5000)), (5, (777,)), (1, (512, 100))])
def test_agc_interleaved_equals_per_stream_oracle(S, calls, gpu_device):
    import torch
    from gr_baz_amd import agc
    x, _ = planar_array_streams(max(S, 2), int(sum(calls)), 5)
    x = np.ascontiguousarray(x[:S] * np.linspace(0.2, 3.0, S)[:, None]).astype(np.complex64)
    oracles = [ar.Agc(1e-2, 0.7) for _ in range(S)]
    with agc.Agc(1e-2, 0.7, nstreams=S) as blk, agc.Agc(1e-2, 0.7, nstreams=S) as planar:
        xd = torch.from_numpy(x.view(np.float32)).to(gpu_device)
        pos = 0
        for c in calls:
            items = torch.full((c * S * 2,), -7.0, dtype=torch.float32, device=gpu_device)
            pl = torch.zeros(S, 2 * x.shape[1], dtype=torch.float32, device=gpu_device)   # in and out share `stride`
            torch.cuda.synchronize()                   # the engines run on their own streams: fills first
            blk.process_device_interleaved(xd.data_ptr() + pos * 8, c, x.shape[1], items.data_ptr())
            planar.process_device(xd.data_ptr() + pos * 8, c, x.shape[1], pl.data_ptr())
            blk.sync(); planar.sync()
            got = items.cpu().numpy().view(np.complex64).reshape(c, S)
            plan = pl.cpu().numpy().view(np.complex64)[:, :c]
            for s in range(S):
                o, _, _ = oracles[s].work(x[s, pos:pos + c])
                scale = np.abs(o).max()
                assert np.all(np.abs(got[:, s] - o) <= 1e-5 * scale)
                assert np.all(np.abs(got[:, s] - plan[s]) <= 2.5e-7 * scale)      # the two scan shapes: <= 1-2 ulp apart
            pos += c
        assert blk.count == sum(calls)


@pytest.mark.gpu
def test_config5_frontend_fused_ahead_of_music(gpu_device):
    import torch
    from gr_baz_amd import agc, capi, resamp
    m, n, K, res = 16, 2, 256, 3600
    N = m * K
    nitems, ratio = 6, 1.25
    T_out = nitems * K
    raw, arr = planar_array_streams(m, int(T_out * ratio) + 64, 31)
    raw = (raw * np.linspace(0.5, 2.0, m)[:, None]).astype(np.complex64)      # unequal channel gains: the AGC's job
    table = mo.steering_table_c64(arr, res, mo.FREQUENCY, mo.SPACING)

    # ---- oracle chain, one block instance per antenna
    o_items = np.zeros((T_out, m), np.complex64)
    for a in range(m):
        y, consumed = rr.Resampler(0.0, ratio).work(raw[a], T_out)
        z, _, _ = ar.Agc(1e-3, 1.0).work(y)
        o_items[:, a] = z
    o_items = o_items.reshape(nitems, N)
    ao, lo, so = mr.work_batch(o_items, table, m, n)
    st = so.astype(np.float64)

    # ---- device chain: resampler (planar) -> AGC (interleaving) -> MUSIC, nothing leaves HBM in between
    L = raw.shape[1]
    d_raw = torch.from_numpy(raw.view(np.float32)).to(gpu_device)
    d_rs = torch.zeros(m, 2 * T_out, dtype=torch.float32, device=gpu_device)
    d_items = torch.zeros(nitems, 2 * N, dtype=torch.float32, device=gpu_device)
    ang = torch.zeros(nitems, n, dtype=torch.float32, device=gpu_device)
    lvl = torch.zeros_like(ang)
    spec = torch.zeros(nitems, res, dtype=torch.float32, device=gpu_device)
    torch.cuda.synchronize()                           # fills (torch's stream) before the engines' stream starts
    with resamp.Resampler(0.0, ratio, nstreams=m) as R, agc.Agc(1e-3, 1.0, nstreams=m) as A, \
            capi.Context(m, n, N, res, table) as M:
        stream = torch.cuda.Stream(device=gpu_device)            # one stream for the three engines: ordered, no host syncs
        for eng in (R, A, M):
            eng.set_stream(stream.cuda_stream)
        produced, consumed = R.process_device(d_raw.data_ptr(), L, L, d_rs.data_ptr(), T_out, T_out)
        assert produced == T_out and consumed == int(T_out * ratio)
        A.process_device_interleaved(d_rs.data_ptr(), T_out, T_out, d_items.data_ptr())
        M.process_device(d_items.data_ptr(), nitems, ang.data_ptr(), lvl.data_ptr(), spec.data_ptr())
        M.sync()
        for eng in (R, A, M):
            eng.set_stream(None)
    got_items = d_items.cpu().numpy().view(np.complex64)
    # front-end stage: the items entering MUSIC are the oracle chain's float32 values -- the resampler bit for bit,
    # the AGC too except where the scan's entry state (a few ulp_f64 from the sequential loop's) flips a rounding:
    # never more than 1 ulp_f32, in < 1e-5 of the samples (measured: 0 here, 8e-7 of them at rate 1e-4 over 2M
    # samples per stream, profiles/r02_agc_exactness.txt)
    u = np.abs(got_items.view(np.int32).astype(np.int64) - o_items.view(np.int32).astype(np.int64))
    assert u.max() <= 1 and np.count_nonzero(u) <= 1e-5 * u.size
    # MUSIC stage on exactly the items the device produced
    a2, l2, s2 = mr.work_batch(got_items, table, m, n)
    assert_spectrum_close(spec.cpu().numpy(), s2)
    assert_doa_match(ang.cpu().numpy(), lvl.cpu().numpy(), a2, l2, res, s2.astype(np.float64))
    # END TO END against the all-oracle chain (one oracle resampler + AGC per antenna, oracle MUSIC) at north_star's 1e-5
    assert_spectrum_close(spec.cpu().numpy(), so)
    assert_doa_match(ang.cpu().numpy(), lvl.cpu().numpy(), ao, lo, res, st)
