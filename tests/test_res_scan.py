"""The table-resident scan (scan_res_kernel in gr_baz_amd/csrc/music_kernels.hip.h): m = 4, spectrum port wired, large
batches.  It runs scan_mfma_kernel's instructions on the same operands (lib/baz_music_doa.cc:101-141 in projector form),
with the table slice held in LDS instead of staged per step, so spectrum, ang and lvl must be BIT-IDENTICAL to the staged
kernel of the same build (BAZ_MUSIC_RES_SCAN=0) -- and equal the CPU oracle's under the parity rule.  Small batches are
pushed through it with BAZ_MUSIC_RES_SCAN=2 (the product takes it from 65,536 items on)."""
import numpy as np
import pytest

from helpers import assert_doa_match, assert_spectrum_close
from oracle import music_oracle as mo

pytestmark = pytest.mark.gpu


def _run(ctx, items, res, gpu_device):
    import torch
    B = items.shape[0]
    x = torch.from_numpy(np.ascontiguousarray(items).view(np.float32)).to(gpu_device)
    ang = torch.full((B, ctx.n), -1.0, dtype=torch.float32, device=gpu_device)
    lvl = torch.full((B, ctx.n), -1.0, dtype=torch.float32, device=gpu_device)
    spec = torch.full((B, res), -1.0, dtype=torch.float32, device=gpu_device)
    ctx.process_device(x.data_ptr(), B, ang.data_ptr(), lvl.data_ptr(), spec.data_ptr(), stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return ang.cpu().numpy(), lvl.cpu().numpy(), spec.cpu().numpy()


def _scene(n, nsamples, res, batch, snr_db, seed, incoherent):
    m = 4
    arr = mo.array_geometry(m)
    table = mo.steering_table_c64(arr, res, mo.FREQUENCY, mo.SPACING)
    rng = np.random.default_rng(seed)
    if not incoherent:
        angles = tuple(rng.uniform(0.0, 360.0, size=n))
        return table, mo.synth_items(batch, m, nsamples, arr, mo.FREQUENCY, mo.SPACING, angles_deg=angles, snr_db=snr_db, seed=seed)
    items = np.concatenate([mo.synth_items(1, m, nsamples, arr, mo.FREQUENCY, mo.SPACING,
                                           angles_deg=tuple(rng.uniform(0.0, 360.0, size=n)), snr_db=snr_db, seed=seed + 7 * i)
                            for i in range(batch)], axis=0)
    return table, items


def _both(monkeypatch, n, nsamples, res, table, items, gpu_device, env=None):
    from gr_baz_amd import capi
    out = {}
    for mode in ("2", "0"):
        monkeypatch.setenv("BAZ_MUSIC_RES_SCAN", mode)
        for k, v in (env or {}).items():
            monkeypatch.setenv(k, v)
        with capi.Context(4, n, nsamples, res, table) as ctx:
            out[mode] = _run(ctx, items, res, gpu_device) + (ctx.refined_values(),)
    return out["2"], out["0"]


# res 3600: 4 row classes, 4 ranges; 360: 8 classes, one range; 1000: 8 classes; 4096: no classes, 4 ranges; 721 / 90:
# resolution not a multiple of 4 (scalar stores, no classes); 36000: more ranges than the kernel takes (falls back)
@pytest.mark.parametrize("incoherent", [False, True])
@pytest.mark.parametrize("snr", [10.0, 20.0, 40.0, 80.0])
@pytest.mark.parametrize("n,nsamples,res,batch", [(2, 1024, 3600, 333), (1, 256, 360, 257), (2, 256, 1000, 130), (2, 64, 4096, 70),
                                                  (2, 96, 721, 200), (1, 64, 90, 65), (2, 64, 1104, 17), (2, 64, 36000, 40)])
def test_resident_scan_equals_the_staged_scan(n, nsamples, res, batch, snr, incoherent, gpu_device, monkeypatch):
    table, items = _scene(n, nsamples, res, batch, snr, 9100 + int(snr) + n + res % 97, incoherent)
    (a1, l1, s1, r1), (a0, l0, s0, r0) = _both(monkeypatch, n, nsamples, res, table, items, gpu_device)
    assert np.array_equal(s1.view(np.uint32), s0.view(np.uint32)), "spectrum floats differ"
    assert np.array_equal(a1, a0) and np.array_equal(l1.view(np.uint32), l0.view(np.uint32))
    assert r1 == r0                                       # the same values went through the literal form
    ao, lo, so, st = mo.music_doa_work_batch(items, table, 4, n)
    assert_spectrum_close(s1, so)
    if snr <= 40.0:
        assert_doa_match(a1, l1, ao, lo, res, st)


@pytest.mark.parametrize("nsplit", ["5", "8"])
def test_resident_scan_with_more_ranges_than_needed(nsplit, gpu_device, monkeypatch):
    table, items = _scene(2, 256, 3600, 150, 20.0, 9300, True)
    (a1, l1, s1, _), (a0, l0, s0, _) = _both(monkeypatch, 2, 256, 3600, table, items, gpu_device, env={"BAZ_MUSIC_NSPLIT": nsplit})
    assert np.array_equal(s1.view(np.uint32), s0.view(np.uint32))
    assert np.array_equal(a1, a0) and np.array_equal(l1.view(np.uint32), l0.view(np.uint32))


def test_resident_scan_is_what_large_batches_run(gpu_device, monkeypatch):
    """65,536 + 5 items of cfg2 (ragged last groups of every class) on the default settings take the resident kernel; the
    same items through the staged kernel give the same bits, poisoned items (NaN, zeros) included."""
    import torch
    from gr_baz_amd import capi
    c = mo.make_config("cfg2", 512)
    B = 65536 + 5
    items = np.tile(c["items"], ((B + 511) // 512, 1))[:B].copy()
    items[7] = 0.0
    items[B - 2, 3] = np.nan
    outs = []
    for mode in ("1", "0"):
        monkeypatch.setenv("BAZ_MUSIC_RES_SCAN", mode)
        with capi.Context(c["m"], c["n"], c["nsamples"], c["res"], c["table"]) as ctx:
            a, l, s = _run(ctx, items, c["res"], gpu_device)
            outs.append((a, l, s, ctx.stage_name(2)))
    (a1, l1, s1, k1), (a0, l0, s0, k0) = outs
    assert "scan_res_kernel" in k1 and "scan_mfma_kernel" in k0
    assert np.array_equal(s1.view(np.uint32), s0.view(np.uint32))
    assert np.array_equal(a1.view(np.uint32), a0.view(np.uint32)) and np.array_equal(l1.view(np.uint32), l0.view(np.uint32))
    ao, lo, so, st = mo.music_doa_work_batch(c["items"][:64], c["table"], c["m"], c["n"])
    assert_spectrum_close(s1[:64][np.arange(64) != 7], so[np.arange(64) != 7])
