"""Default wiring (port 2 not wired), large batches: the batch is cut into parts and the covariance + EVD of part p + 1 runs beside the gated
scan of part p on a second stream (gr_baz_amd/csrc/baz_music_hip.hip, process_split_locked; /root/reference/lib/baz_music_doa.cc:72-161 is
what every item still gets).  Pinned here: ang / lvl of the split pipeline are BIT-IDENTICAL to the unsplit launch sequence (BAZ_MUSIC_SPLIT=0)
for every number of parts, for ragged batch sizes, coherent and incoherent batches, 20 and 60 dB; they agree with the CPU oracle; the caller's
stream is ordered behind the second stream's work (outputs read right after a stream synchronise); a retune between split calls in flight
never tears a batch; the policy itself (CPU)."""
import numpy as np
import pytest

from oracle import music_oracle as mo


def _capi():
    from gr_baz_amd import capi
    return capi


def _scene(torch, dev, batch, incoherent, snr_db, seed):
    from gr_baz_amd import synth
    arr = synth.array_geometry(4)
    if incoherent:
        return synth.synth_scenes(torch, dev, batch, 4, 1024, arr, mo.FREQUENCY, mo.SPACING, 2, snr_db=snr_db, seed=seed)
    return synth.synth_stream(torch, dev, batch, 4, 1024, arr, mo.FREQUENCY, mo.SPACING, snr_db=snr_db, seed=seed).reshape(batch, -1)


def _run(monkeypatch, split, x, batch, want_lvl=True):
    import torch
    capi = _capi()
    if split is None:
        monkeypatch.delenv("BAZ_MUSIC_SPLIT", raising=False)
    else:
        monkeypatch.setenv("BAZ_MUSIC_SPLIT", str(split))
    c = mo.make_config("cfg2", 1)
    ang = torch.full((batch, 2), -1.0, dtype=torch.float32, device=x.device)
    lvl = torch.full((batch, 2), -1.0, dtype=torch.float32, device=x.device) if want_lvl else None
    torch.cuda.synchronize()
    stream = torch.cuda.Stream(device=x.device)
    with capi.Context(4, 2, 1024, 3600, c["table"]) as ctx:
        ctx.set_stream(stream.cuda_stream)
        ctx.reserve(batch)
        for _ in range(2):                     # twice: the second call's covariance must wait for the first call's scans (shared workspace)
            ctx.process_device(x.data_ptr(), batch, ang.data_ptr(), lvl.data_ptr() if want_lvl else None, None)
        stream.synchronize()                   # ONLY the caller's stream: the second stream's work must be ordered before it
        a = ang.cpu().numpy()
        l = lvl.cpu().numpy() if want_lvl else None
        ctx.set_stream(None)
    return a, l


@pytest.mark.gpu
@pytest.mark.parametrize("batch,incoherent,snr", [(40000, False, 20.0), (40000, True, 20.0), (36869, True, 60.0), (8192, False, 20.0),
                                                   (262144, False, 20.0), (131072 + 77, True, 20.0)])
def test_split_pipeline_is_bit_identical_to_the_single_sequence(gpu_device, monkeypatch, batch, incoherent, snr):
    import torch
    x = _scene(torch, gpu_device, batch, incoherent, snr, seed=900 + batch % 97)
    a0, l0 = _run(monkeypatch, 0, x, batch)
    assert (a0 >= 0).all() and (l0 > 0).all()
    for split in ((None, 2, 3, 8) if batch <= 50000 else (None, 4)):
        a, l = _run(monkeypatch, split, x, batch)
        assert np.array_equal(a, a0), "ang differs with BAZ_MUSIC_SPLIT=%s" % split
        assert np.array_equal(l.view(np.uint32), l0.view(np.uint32)), "lvl differs with BAZ_MUSIC_SPLIT=%s" % split
    # ... and against the oracle (a sample: the split and the unsplit outputs are the same bits)
    from oracle import music_ref as mr
    idx = np.unique(np.linspace(0, batch - 1, 48).astype(np.int64))
    items = x[torch.from_numpy(idx).to(gpu_device)].cpu().numpy().view(np.complex64).reshape(len(idx), 1024)
    ao, lo, so = mr.work_batch(np.ascontiguousarray(items), mo.make_config("cfg2", 1)["table"], 4, 2)
    assert np.max(np.abs(l0[idx].astype(np.float64) - lo) / lo) <= 1e-5
    for r in np.flatnonzero(np.any(a0[idx] != ao, axis=1)):          # the reference's own tie rule: bins whose strengths agree to 2e-5
        gb = np.rint(a0[idx][r].astype(np.float64) * 10.0).astype(np.int64) % 3600
        rb = np.rint(ao[r].astype(np.float64) * 10.0).astype(np.int64) % 3600
        assert np.all(np.abs(so[r][gb] - so[r][rb]) <= 2e-5 * so[r][rb])


@pytest.mark.gpu
def test_split_pipeline_without_the_lvl_port(gpu_device, monkeypatch):
    import torch
    x = _scene(torch, gpu_device, 40000, True, 20.0, seed=31)
    a0, _ = _run(monkeypatch, 0, x, 40000, want_lvl=False)
    a4, _ = _run(monkeypatch, 4, x, 40000, want_lvl=False)
    assert np.array_equal(a0, a4)


@pytest.mark.gpu
def test_retune_between_split_calls_in_flight_never_tears(gpu_device, monkeypatch):
    """Split calls queued without waiting, the table exchanged twice in between: the scans of a call run on the second stream, and the
    retired table set must not be rebuilt while one of them still reads it (the swap event is recorded behind the call's join)."""
    import torch
    from oracle import music_ref as mr
    capi = _capi()
    monkeypatch.setenv("BAZ_MUSIC_SPLIT", "4")
    c = mo.make_config("cfg2", 64, snr_db=20.0, seed=77)
    tabs = {"A": c["table"], "B": mo.steering_table_c64(c["array"], 3600, mo.FREQUENCY * 0.9, mo.SPACING),
            "C": mo.steering_table_c64(c["array"], 3600, mo.FREQUENCY * 1.1, mo.SPACING)}
    want = {k: mr.work_batch(np.ascontiguousarray(c["items"][:8]), t, 4, 2) for k, t in tabs.items()}
    B = 65536
    x = torch.from_numpy(np.ascontiguousarray(np.tile(c["items"], (B // 64, 1))).view(np.float32)).to(gpu_device)
    outs = [(torch.zeros(B, 2, dtype=torch.float32, device=gpu_device), torch.zeros(B, 2, dtype=torch.float32, device=gpu_device)) for _ in range(6)]
    torch.cuda.synchronize()
    with capi.Context(4, 2, 1024, 3600, tabs["A"]) as ctx:
        ctx.reserve(B)
        for rep in range(3):
            order = []
            for k, (a, l) in enumerate(outs):
                ctx.process_device(x.data_ptr(), B, a.data_ptr(), l.data_ptr(), None)
                order.append("ABC"[min(k // 2, 2)])
                if k == 1:
                    ctx.set_table(tabs["B"])
                if k == 3:
                    ctx.set_table(tabs["C"])
            ctx.sync()
            for lab, (a, l) in zip(order, outs):
                ao, lo, _ = want[lab]
                for rows in (slice(0, 8), slice(B - 64, B - 56)):          # the first part and the last
                    assert np.array_equal(a[rows].cpu().numpy(), ao), "round %d: a batch expected under table %s" % (rep, lab)
                    assert np.max(np.abs(l[rows].cpu().numpy().astype(np.float64) - lo) / lo) <= 1e-5
            ctx.set_table(tabs["A"])


def test_split_needs_the_gated_scan_and_a_large_batch():
    """CPU: what the header documents -- BAZ_MUSIC_SPLIT is one of the knobs the release library reads."""
    import os
    from conftest import ROOT
    src = open(os.path.join(ROOT, "include", "baz_music_hip.h")).read()
    assert "BAZ_MUSIC_SPLIT" in src
    lib = open(os.path.join(ROOT, "gr_baz_amd", "csrc", "libbaz_music_hip.so"), "rb").read()
    assert b"BAZ_MUSIC_SPLIT" in lib
