"""Lab (GPU box): how far the tiled-scan AGC is from the reference's sequential loop, in float32 output bits, and what
that does to the config-5 chain end to end (resampler -> AGC -> MUSIC vs the all-oracle chain).
argv: [log2 samples per stream = 21]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gr_baz_amd import agc, capi, resamp
from oracle import agc_ref as ar
from oracle import music_oracle as mo
from oracle import music_ref as mr
from oracle import resamp_ref as rr

dev = torch.device("cuda:0")
LOG2 = int(sys.argv[1]) if len(sys.argv) > 1 else 21


def ulps(a, b):
    ia = a.view(np.int32).astype(np.int64)
    ib = b.view(np.int32).astype(np.int64)
    return np.abs(ia - ib)


S, n = 4, 1 << LOG2
rng = np.random.default_rng(7)
x = ((rng.standard_normal((S, n)) + 1j * rng.standard_normal((S, n))) * (1.0 + 0.5 * np.sin(np.arange(n) / 5000.0))).astype(np.complex64)
for rate in (1e-4, 1e-3, 1e-2, 0.5):
    with agc.Agc(rate, 1.0, nstreams=S) as blk:
        d_in = torch.from_numpy(x.view(np.float32)).to(dev)
        d_out = torch.zeros_like(d_in)
        d_env = torch.zeros(S, n, dtype=torch.float32, device=dev)
        d_mul = torch.zeros_like(d_env)
        torch.cuda.synchronize()
        blk.process_device(d_in.data_ptr(), n, n, d_out.data_ptr(), d_env.data_ptr(), d_mul.data_ptr())
        blk.sync()
        out = d_out.cpu().numpy()
        env = d_env.cpu().numpy()
        mul = d_mul.cpu().numpy()
    tot = dict(out=0, env=0, mul=0)
    worst = dict(out=0, env=0, mul=0)
    for s in range(S):
        o, e, m = ar.Agc(rate, 1.0).work(x[s])
        for k, got, ref in (("out", out[s], o.view(np.float32)), ("env", env[s], e), ("mul", mul[s], m)):
            u = ulps(np.ascontiguousarray(got), np.ascontiguousarray(ref))
            tot[k] += int((u != 0).sum())
            worst[k] = max(worst[k], int(u.max()))
    print("AGC rate %-6g: %d streams x %d samples: float32 values differing from the sequential loop: out %d of %d (max %d ulp), "
          "env %d of %d (max %d ulp), gain %d of %d (max %d ulp)" % (rate, S, n, tot["out"], 2 * S * n, worst["out"], tot["env"], S * n,
                                                                     worst["env"], tot["mul"], S * n, worst["mul"]), flush=True)

# ---- config-5 chain
m, nn, K, res = 16, 2, 256, 3600
N = m * K
for nitems, seed in ((64, 31), (256, 32)):
    ratio = 1.25
    T_out = nitems * K
    arr = mo.array_geometry(m)
    items = mo.synth_items((int(T_out * ratio) + 64 + 63) // 64, m, m * 64, arr, mo.FREQUENCY, mo.SPACING, seed=seed)
    raw = np.ascontiguousarray(items.reshape(-1, 64, m).reshape(-1, m).T[:, :int(T_out * ratio) + 64])
    raw = (raw * np.linspace(0.5, 2.0, m)[:, None]).astype(np.complex64)
    table = mo.steering_table_c64(arr, res, mo.FREQUENCY, mo.SPACING)
    o_items = np.zeros((T_out, m), np.complex64)
    for a in range(m):
        y, consumed = rr.Resampler(0.0, ratio).work(raw[a], T_out)
        z, _, _ = ar.Agc(1e-3, 1.0).work(y)
        o_items[:, a] = z
    o_items = o_items.reshape(nitems, N)
    ao, lo, so = mr.work_batch(o_items, table, m, nn)
    L = raw.shape[1]
    d_raw = torch.from_numpy(raw.view(np.float32)).to(dev)
    d_rs = torch.zeros(m, 2 * T_out, dtype=torch.float32, device=dev)
    d_items = torch.zeros(nitems, 2 * N, dtype=torch.float32, device=dev)
    ang = torch.zeros(nitems, nn, dtype=torch.float32, device=dev)
    lvl = torch.zeros_like(ang)
    spec = torch.zeros(nitems, res, dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    with resamp.Resampler(0.0, ratio, nstreams=m) as R, agc.Agc(1e-3, 1.0, nstreams=m) as A, capi.Context(m, nn, N, res, table) as M:
        st = torch.cuda.Stream(device=dev)
        for e in (R, A, M):
            e.set_stream(st.cuda_stream)
        R.process_device(d_raw.data_ptr(), L, L, d_rs.data_ptr(), T_out, T_out)
        A.process_device_interleaved(d_rs.data_ptr(), T_out, T_out, d_items.data_ptr())
        M.process_device(d_items.data_ptr(), nitems, ang.data_ptr(), lvl.data_ptr(), spec.data_ptr())
        M.sync()
        for e in (R, A, M):
            e.set_stream(None)
    got = d_items.cpu().numpy()
    u = ulps(got.reshape(-1), o_items.view(np.float32).reshape(-1))
    sp = spec.cpu().numpy().astype(np.float64)
    rel = np.abs(sp - so) / so
    bins_same = np.array_equal(ang.cpu().numpy(), ao)
    print("cfg5 chain %d items: front-end floats differing from the oracle chain: %d of %d (max %d ulp); spectrum vs all-oracle chain "
          "max rel err %.3g; DoA bins identical: %s" % (nitems, int((u != 0).sum()), u.size, int(u.max()), rel.max(), bins_same), flush=True)
