"""Ad-hoc first-light script (GPU box): stage-by-stage parity vs the numpy oracle + quick timing."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from gr_baz_amd import capi
from oracle import music_oracle as mo

dev = torch.device("cuda:0")
print(torch.cuda.get_device_name(0), capi.version(), flush=True)

def stage_check(name, B, snr=20.0):
    c = mo.make_config(name, B, snr_db=snr)
    m, n, N, res = c["m"], c["n"], c["nsamples"], c["res"]
    ctx = capi.Context(m, n, N, res, c["table"])
    x = torch.from_numpy(c["items"].view(np.float32)).to(dev)
    R = torch.zeros(B, m * m, 2, dtype=torch.float64, device=dev)
    ctx.debug_cov(x.data_ptr(), B, R.data_ptr()); ctx.sync()
    Rg = R.cpu().numpy(); Rg = (Rg[..., 0] + 1j * Rg[..., 1]).reshape(B, m, m)
    xs = c["items"].astype(np.complex128).reshape(B, N // m, m).transpose(0, 2, 1)
    Rn = xs @ xs.conj().transpose(0, 2, 1) / (N // m)
    print(name, "cov max abs err", np.abs(Rg - Rn).max(), "scale", np.abs(Rn).max(), flush=True)
    qs = capi.q_stride(B)
    Q = torch.zeros(m * m, qs, dtype=torch.float64, device=dev)
    ctx.debug_evd(R.data_ptr(), B, Q.data_ptr()); ctx.sync()
    Qg = Q.cpu().numpy()[:, :B].T.reshape(B, m, m)
    w, V = np.linalg.eigh(Rn); G = V[:, :, :m - n]; P = G @ G.conj().transpose(0, 2, 1)
    Qn = np.zeros_like(Qg)
    for i in range(m):
        Qn[:, i, i] = P[:, i, i].real
        for j in range(i + 1, m):
            Qn[:, i, j] = 2 * P[:, i, j].real
            Qn[:, j, i] = -2 * P[:, i, j].imag
    print(name, "proj max abs err", np.abs(Qg - Qn).max(), flush=True)
    ang = torch.zeros(B, n, dtype=torch.float32, device=dev); lvl = torch.zeros_like(ang)
    spec = torch.zeros(B, res, dtype=torch.float32, device=dev)
    ctx.process_device(x.data_ptr(), B, ang.data_ptr(), lvl.data_ptr(), spec.data_ptr()); ctx.sync()
    ao, lo, so, st = mo.music_doa_work_batch(c["items"], c["table"], m, n)
    sg = spec.cpu().numpy().astype(np.float64)
    rel = np.abs(sg - st) / st
    print(name, "spectrum max rel err", rel.max(), "ang equal", (ang.cpu().numpy() == ao).all(),
          "lvl max rel", (np.abs(lvl.cpu().numpy() - lo) / lo).max(), flush=True)
    if not (ang.cpu().numpy() == ao).all():
        bad = np.where((ang.cpu().numpy() != ao).any(axis=1))[0][:5]
        for b in bad: print("  item", b, ang.cpu().numpy()[b], ao[b], lvl.cpu().numpy()[b], lo[b])
    # host path
    a2, l2, s2 = ctx.process(c["items"])
    print(name, "host path equal device path:", (a2 == ang.cpu().numpy()).all(), (s2 == spec.cpu().numpy()).all(), flush=True)
    ctx.close()

for snr in (10.0, 20.0, 40.0):
    stage_check("cfg1", 100, snr)
    stage_check("cfg2", 130, snr)
stage_check("cfg3", 70)

def timing(name, B, iters=10, spec_on=True):
    c = mo.make_config(name, 256)
    m, n, N, res = c["m"], c["n"], c["nsamples"], c["res"]
    ctx = capi.Context(m, n, N, res, c["table"])
    base = torch.from_numpy(c["items"].view(np.float32)).to(dev)
    x = base.repeat((B + 255) // 256, 1)[:B].contiguous()
    ang = torch.zeros(B, n, dtype=torch.float32, device=dev); lvl = torch.zeros_like(ang)
    spec = torch.zeros(B, res, dtype=torch.float32, device=dev) if spec_on else None
    ctx.reserve(B)
    sp = spec.data_ptr() if spec_on else None
    for _ in range(3): ctx.process_device(x.data_ptr(), B, ang.data_ptr(), lvl.data_ptr(), sp)
    ctx.sync()
    ctx.profile(True)
    t0 = time.perf_counter()
    for _ in range(iters): ctx.process_device(x.data_ptr(), B, ang.data_ptr(), lvl.data_ptr(), sp)
    ctx.sync(); t1 = time.perf_counter()
    dt = (t1 - t0) / iters
    bpi = ctx.bytes_per_item(spec_on)
    print("%s B=%d spec=%s: %.3f ms/step  %.3e items/s  %.2f TB/s algorithmic (%.1f%% of 8 TB/s)" % (
        name, B, spec_on, dt * 1e3, B / dt, B / dt * bpi / 1e12, B / dt * bpi / 8e12 * 100), flush=True)
    for s in range(4):
        ms, cnt = ctx.stage_ms(s)
        print("   stage %d %-32s %.3f ms/launch (%d launches)" % (s, ctx.stage_name(s), ms / max(cnt, 1), cnt), flush=True)
    ctx.close()

timing("cfg2", 65536)
timing("cfg2", 65536, spec_on=False)
timing("cfg2", 262144)
timing("cfg1", 65536)
timing("cfg3", 4096, iters=3)
