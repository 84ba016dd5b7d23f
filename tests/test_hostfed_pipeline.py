"""The host-fed path's chunked form on PAGE-LOCKED caller memory (baz_music_process, gr_baz_amd/csrc/baz_music_hip.hip; what the host block's work() --
/root/reference/lib/baz_music_doa.cc:72-161 per item -- does with a large call): since round 6 the whole call is enqueued without the host waiting for
anything, four device slots whose reuse is ordered on the device by events, ang / lvl handed over after one synchronise.  Pinned here: with MANY more chunks
than slots (BAZ_MUSIC_CHUNK_MIB=1) the outputs equal the single zero-copy launch sequence's bit for bit and the oracle's within 1e-5 -- all port
combinations, ragged call sizes, the gated scan, the int8 scan, the run-time-m path -- and calls of different sizes on one context do not disturb each other."""
import numpy as np
import pytest

from oracle import music_oracle as mo


def _pinned(torch, a):
    return torch.from_numpy(np.ascontiguousarray(a)).pin_memory().numpy()


SHAPES = [
    # name, m, n, nsamples, res, batch
    ("cfg2", 4, 2, 1024, 3600, 1000),
    ("cfg2_ragged", 4, 2, 1024, 3600, 777),
    ("m8_int8_scan", 8, 2, 512, 3000, 600),
    ("m16_short_form", 16, 2, 1024, 1200, 300),
    ("m3_odd_res", 3, 1, 96, 1001, 2500),
    ("wide24", 24, 2, 768, 720, 300),
]


@pytest.mark.gpu
@pytest.mark.parametrize("shape", SHAPES, ids=[s[0] for s in SHAPES])
@pytest.mark.parametrize("ports", ["ang_lvl_spec", "ang_spec", "ang_lvl", "ang"])
def test_deep_pipeline_equals_the_single_sequence(gpu_device, monkeypatch, shape, ports):
    import torch
    from gr_baz_amd import capi
    from oracle import music_ref as mr
    _, m, n, N, res, B = shape
    want_lvl, want_spec = "lvl" in ports, "spec" in ports
    arr = mo.array_geometry(m)
    table = mo.steering_table_c64(arr, res, mo.FREQUENCY, mo.SPACING)
    items = mo.synth_items(B, m, N, arr, mo.FREQUENCY, mo.SPACING, snr_db=25.0, seed=4100 + m)
    x = _pinned(torch, items.view(np.float32)).view(np.complex64)
    mk = lambda cols: torch.full((B, cols), -7.0, dtype=torch.float32).pin_memory().numpy()
    outs = {}
    for mode in ("single", "deep"):
        monkeypatch.delenv("BAZ_MUSIC_CHUNK_MIB", raising=False)
        monkeypatch.setenv("BAZ_MUSIC_SINGLE_MIB", "1024")
        if mode == "deep":
            monkeypatch.setenv("BAZ_MUSIC_CHUNK_MIB", "1")             # 1 MiB per chunk: 5 .. 40 chunks here, many more than the four slots
        o = (mk(n), mk(n) if want_lvl else None, mk(res) if want_spec else None)
        with capi.Context(m, n, N, res, table) as ctx:
            ctx.process(x, out=o)
            first = tuple(None if a is None else a.copy() for a in o)
            ctx.process(x[: B // 3], out=tuple(None if a is None else a[: B // 3] for a in o))      # a smaller call on the same context, then the big one again
            ctx.process(x, out=o)
        for a, b in zip(first, o):
            assert a is None or np.array_equal(a, b), "%s: the second call differs from the first" % mode
        outs[mode] = first
    for a, b, what in zip(outs["single"], outs["deep"], ("ang", "lvl", "spectrum")):
        assert a is None or np.array_equal(a.view(np.uint32), b.view(np.uint32)), "%s differs between the chunked and the single form" % what
    ao, lo, so = mr.work_batch(items, table, m, n)
    ga, gl, gs = outs["deep"]
    if want_spec:
        assert np.max(np.abs(gs.astype(np.float64) - so) / so) <= 1e-5
    if want_lvl:
        assert np.max(np.abs(gl.astype(np.float64) - lo) / lo) <= 1e-5
    same = np.all(ga == ao, axis=1)
    assert same.mean() >= 0.98                                         # (ties between neighbouring bins whose strengths agree to 2e-5 may swap: helpers.assert_doa_match)
    for r in np.flatnonzero(~same):
        gb = np.rint(ga[r].astype(np.float64) * res / 360.0).astype(np.int64) % res
        rb = np.rint(ao[r].astype(np.float64) * res / 360.0).astype(np.int64) % res
        assert np.all(np.abs(so[r][gb] - so[r][rb]) <= 2e-5 * so[r][rb])


@pytest.mark.gpu
def test_deep_pipeline_at_the_library_s_own_chunking(gpu_device, monkeypatch):
    """No knob set: an 8,192-item config-2 call with port 2 from page-locked memory (185 MB) is cut by the library itself (six chunks, four slots)."""
    import torch
    from gr_baz_amd import capi
    for k in ("BAZ_MUSIC_CHUNK_MIB", "BAZ_MUSIC_SINGLE_MIB"):
        monkeypatch.delenv(k, raising=False)
    c = mo.make_config("cfg2", 256, snr_db=20.0, seed=5)
    B = 8192
    items = np.tile(c["items"], (B // 256, 1))
    x = _pinned(torch, items.view(np.float32)).view(np.complex64)
    o = tuple(torch.zeros((B, k), dtype=torch.float32).pin_memory().numpy() for k in (2, 2, 3600))
    with capi.Context(4, 2, 1024, 3600, c["table"]) as ctx:
        ctx.process(x, out=o)
        a, l, s = ctx.process(items[:256])                             # pageable, one chunk: the same items
    for blk in range(0, B, 256):
        assert np.array_equal(o[0][blk:blk + 256], a) and np.array_equal(o[1][blk:blk + 256], l)
    assert np.array_equal(o[2][:256], s) and np.array_equal(o[2][B - 256:], s)
