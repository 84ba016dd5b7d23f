"""Items ordered by their nulls in front of the gated scan (gr_baz_amd/csrc/sort_kernels.hip.h; the scan without port 2,
/root/reference/lib/baz_music_doa.cc:97-99,129-155).  The order is an index list the scan walks; ang / lvl must be bit-identical with and without
it, on incoherent batches of every gated shape, ragged batch sizes and poisoned items; the adaptive policy must switch it on for an incoherent
batch and leave a coherent stream alone.
LAB BUILD ONLY (BAZ_MUSIC_SORT): measured in round 5 and not shipped -- the sort costs what the better order saves (profiles/r05_sort_negative.txt)."""
import numpy as np
import pytest

from helpers import assert_doa_match
from oracle import music_oracle as mo

pytestmark = pytest.mark.gpu


def _capi():
    from gr_baz_amd import capi
    return capi


def _run(ctx, items, gpu_device, reps=1):
    import torch
    B = items.shape[0]
    x = torch.from_numpy(np.ascontiguousarray(items).view(np.float32)).to(gpu_device)
    ang = torch.full((B, ctx.n), -1.0, dtype=torch.float32, device=gpu_device)
    lvl = torch.full((B, ctx.n), -1.0, dtype=torch.float32, device=gpu_device)
    for _ in range(reps):
        ctx.process_device(x.data_ptr(), B, ang.data_ptr(), lvl.data_ptr(), None, stream=torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()         # (like work(): the policy reads the statistic of calls that have finished)
    return ang.cpu().numpy(), lvl.cpu().numpy()


def _incoherent(m, n, nsamples, batch, seed, snr=20.0):
    arr = mo.array_geometry(m)
    rng = np.random.default_rng(seed)
    base = 64                                         # 64 scenes, tiled and shuffled: every row group sees 16 different ones
    scenes = np.concatenate([mo.synth_items(1, m, nsamples, arr, mo.FREQUENCY, mo.SPACING, angles_deg=tuple(rng.uniform(0.0, 360.0, size=n)),
                                            snr_db=snr, seed=seed + 7 * i) for i in range(base)], axis=0)
    idx = rng.integers(0, base, size=batch)
    return arr, scenes[idx], idx


@pytest.mark.parametrize("m,n,nsamples,res,batch", [(4, 2, 1024, 3600, 8192), (4, 3, 256, 1000, 5000), (4, 1, 64, 360, 4097), (3, 2, 96, 357, 4500),
                                                     (2, 1, 64, 1444, 4200), (5, 2, 320, 720, 4096), (8, 2, 512, 3600, 6000)])
def test_sorted_and_unsorted_scans_give_the_same_bits(m, n, nsamples, res, batch, gpu_device, monkeypatch):
    arr, items, idx = _incoherent(m, n, nsamples, batch, 100 + m + n)
    items = items.copy()
    items[17, 3] = np.nan                             # a poisoned item travels through the order like any other
    items[batch - 1] = 0
    table = mo.steering_table_c64(arr, res, mo.FREQUENCY, mo.SPACING)
    out = {}
    for mode in ("0", "1"):
        # "0": the adaptive policy's FIRST call, which is never sorted (it has no statistic yet) -- the lab instantiation that keeps the fire
        # statistic; BAZ_MUSIC_SORT=0 itself is the product's kernel, which keeps none
        monkeypatch.setenv("BAZ_MUSIC_SORT", "-1" if mode == "0" else "1")
        with _capi().Context(m, n, nsamples, res, table, lab=True) as ctx:
            out[mode] = _run(ctx, items, gpu_device)
            st = ctx.debug_sort_state()
            want = mode == "1" and m <= 4 and n <= 2   # (from 5 antennas on -- and with lists of 4 keys -- the scan is never sorted)
            assert (st["sorted_calls"] > 0) == want and st["last_sorted"] == want
            out[mode + "rate"] = st["fired"] / max(st["walked"], 1)
    assert np.array_equal(out["0"][0], out["1"][0]), "ang differs between the sorted and the unsorted scan"
    assert np.array_equal(out["0"][1].view(np.uint32), out["1"][1].view(np.uint32)), "lvl differs between the sorted and the unsorted scan"
    # and against the oracle on the 64 distinct scenes
    first = np.array([np.flatnonzero(idx == k)[0] for k in range(64) if (idx == k).any() and np.flatnonzero(idx == k)[0] not in (17, batch - 1)])
    ao, lo, so, s64 = mo.music_doa_work_batch(items[first], table, m, n)
    assert_doa_match(out["1"][0][first], out["1"][1][first], ao, lo, res, s64)
    print("m=%d n=%d res=%d %d items: exact evaluations per (row group, tile): unsorted %.3f, sorted %.3f" % (m, n, res, batch, out["0rate"], out["1rate"]))
    assert out["1rate"] <= out["0rate"] * 1.05 + 0.01


def test_the_policy_sorts_incoherent_batches_and_leaves_coherent_streams_alone(gpu_device, monkeypatch):
    monkeypatch.setenv("BAZ_MUSIC_SORT", "-1")           # the adaptive policy (lab builds)
    m, n, N, res, B = 4, 2, 1024, 3600, 65536
    arr, inc, _ = _incoherent(m, n, N, B, 7)
    table = mo.steering_table_c64(arr, res, mo.FREQUENCY, mo.SPACING)
    coh = np.tile(mo.synth_items(256, m, N, arr, mo.FREQUENCY, mo.SPACING, snr_db=20.0, seed=3), (B // 256, 1))
    with _capi().Context(m, n, N, res, table, lab=True) as ctx:
        a0, l0 = _run(ctx, coh, gpu_device, reps=40)
        st = ctx.debug_sort_state()
        assert st["sorted_calls"] <= 4, st                    # (at most a trial: no order improves a coherent batch)
        a1, l1 = _run(ctx, inc, gpu_device, reps=700)         # past the retry pause of a failed trial: the incoherent batch switches the sort on
        st1 = ctx.debug_sort_state()
        assert st1["sorted_calls"] - st["sorted_calls"] >= 100, (st, st1)
        a2, l2 = _run(ctx, coh, gpu_device, reps=300)         # ... and a probe call on the coherent stream switches it off again
        st2 = ctx.debug_sort_state()
        assert st2["sorted_calls"] - st1["sorted_calls"] < 150, (st1, st2)
        assert np.array_equal(a0, a2) and np.array_equal(l0.view(np.uint32), l2.view(np.uint32))
    monkeypatch.setenv("BAZ_MUSIC_SORT", "0")
    with _capi().Context(m, n, N, res, table, lab=True) as ctx:
        ar, lr = _run(ctx, inc, gpu_device)
    assert np.array_equal(a1, ar) and np.array_equal(l1.view(np.uint32), lr.view(np.uint32))
