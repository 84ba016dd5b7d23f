"""Lab (GPU box): the gated top-n of the scan (round 2) against the ungated network of round 1
(BAZ_MUSIC_SCAN_VARIANT=2), on (a) coherent streams -- every item of a stream sees the same two emitters, the
bench's and a real flowgraph's situation -- and (b) incoherent batches: every ITEM has its own random emitter
angles, the worst case for a wave-uniform gate (16 items x 16 lanes vote together).
Checks that both variants give bit-identical ang / lvl / spectrum, then prints per-stage times.
argv: [batch=262144]"""
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from gr_baz_amd import capi, synth
from gr_baz_amd.baz.music_doa_helper import calculate_antenna_array_response

M, NE, N, RES = 4, 2, 1024, 3600
B = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
dev = torch.device("cuda:0")
arr = synth.array_geometry(M)
lam = 1.0
table = np.array(calculate_antenna_array_response([[0.5 * x, 0.5 * y] for x, y in arr], RES, lam)).astype(np.complex64)


def incoherent(batch, snr_db=20.0, seed=5):
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    K = N // M
    p = torch.tensor(arr, dtype=torch.float64, device=dev) * 0.5
    x = torch.zeros(batch, K, M, dtype=torch.complex64, device=dev)
    for e in range(NE):
        th = torch.rand(batch, generator=g, device=dev, dtype=torch.float64) * (2 * math.pi)
        ph = -2 * math.pi * (p[None, :, 0] * torch.cos(th)[:, None] + p[None, :, 1] * torch.sin(th)[:, None]) / lam
        a = torch.polar(torch.ones_like(ph), ph).to(torch.complex64)          # (batch, M)
        s = torch.view_as_complex(torch.randn(batch, K, 2, generator=g, device=dev)) * (1 / math.sqrt(2))
        x += s[:, :, None] * a[:, None, :]
    sigma = 10.0 ** (-snr_db / 20.0) / math.sqrt(2.0)
    x += torch.view_as_complex(torch.randn(batch, K, M, 2, generator=g, device=dev)) * sigma
    return torch.view_as_real(x.reshape(batch, K * M)).reshape(batch, 2 * N).contiguous()


def coherent(batch):
    per = batch // 8
    return torch.cat([synth.synth_stream(torch, dev, per, M, N, arr, synth.C_LIGHT, 0.5, seed=1002 + s)
                      for s in range(8)], dim=0)


VARIANTS = {0: ("shipped: gated, row classes, sc0 sc1 nt", {}),
            1: ("round-1 row order (no row classes)", {"BAZ_MUSIC_NO_ROWCLASS": "1"}),
            2: ("ungated top-n network", {"BAZ_MUSIC_SCAN_VARIANT": "2"}),
            3: ("plain cached stores", {"BAZ_MUSIC_SCAN_VARIANT": "3"}),
            4: ("nt stores", {"BAZ_MUSIC_SCAN_VARIANT": "4"}),
            5: ("sc0 sc1 stores", {"BAZ_MUSIC_SCAN_VARIANT": "5"}),
            6: ("no literal refinement", {"BAZ_MUSIC_NO_REFINE": "1"}),
            7: ("round-1 covariance kernel (dword loads, 16x16x4)", {"BAZ_MUSIC_COV_OLD": "1"}),
            9: ("covariance grid 1 block per CU", {"BAZ_MUSIC_COV_BLOCKS_PER_CU": "1"}),
            10: ("covariance grid 2 blocks per CU", {"BAZ_MUSIC_COV_BLOCKS_PER_CU": "2"}),
            11: ("covariance grid 6 blocks per CU", {"BAZ_MUSIC_COV_BLOCKS_PER_CU": "6"}),
            15: ("cov4_x4 + evd_proj as two kernels (no fusion)", {"BAZ_MUSIC_FUSE": "0"})}
# (the range split is read once per process: BAZ_MUSIC_NSPLIT=1 python tests/lab/r02_scan_gate.py ... to compare)
ORDER = [int(v) for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else "0,1,2,3,4,5,6,7".split(","))]


def run(x, label):
    out = {}
    for v in ORDER:
        for k in ("BAZ_MUSIC_SCAN_VARIANT", "BAZ_MUSIC_NO_ROWCLASS", "BAZ_MUSIC_NO_REFINE", "BAZ_MUSIC_COV_OLD", "BAZ_MUSIC_COV_BLOCKS_PER_CU", "BAZ_MUSIC_FUSE"):
            os.environ.pop(k, None)
        os.environ.update(VARIANTS[v][1])
        ctx = capi.Context(M, NE, N, RES, table)
        ctx.reserve(B)
        ang = torch.zeros(B, NE, dtype=torch.float32, device=dev)
        lvl = torch.zeros_like(ang)
        spec = torch.zeros(B, RES, dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        step = lambda: ctx.process_device(x.data_ptr(), B, ang.data_ptr(), lvl.data_ptr(), spec.data_ptr())
        for _ in range(30):
            step()
        ctx.sync()
        ctx.profile(True)
        for _ in range(10):
            step()
        ctx.sync()
        ms = [ctx.stage_ms(s)[0] / 10 for s in range(4)]
        ctx.profile(False)
        t0 = time.perf_counter()
        for _ in range(50):
            step()
        ctx.sync()
        wall = (time.perf_counter() - t0) / 50 * 1e3
        out[v] = (ang.clone(), lvl.clone(), spec[:4096].clone(), ms, wall)
        # no-spectrum wiring
        for _ in range(5):
            ctx.process_device(x.data_ptr(), B, ang.data_ptr(), lvl.data_ptr(), 0)
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(30):
            ctx.process_device(x.data_ptr(), B, ang.data_ptr(), lvl.data_ptr(), 0)
        ctx.sync()
        wall_ns = (time.perf_counter() - t0) / 30 * 1e3
        out[v] += (ang.clone(), wall_ns)
        print("%s variant %d (%s): cov %.3f evd %.3f scan %.3f merge %.3f ms | wall %.3f ms/step = %.3e items/s | no-spectrum wall %.3f ms = %.3e items/s"
              % (label, v, VARIANTS[v][0], ms[0], ms[1], ms[2], ms[3], wall, B / wall * 1e3, wall_ns, B / wall_ns * 1e3), flush=True)
        ctx.close()
    same = True
    for v in ORDER[1:]:
        eq = all(bool(torch.equal(out[ORDER[0]][i], out[v][i])) for i in (0, 1, 2, 5))
        print("%s: variant %d == variant %d bit for bit (ang, lvl, spectrum[:4096], ang without spectrum): %s"
              % (label, v, ORDER[0], eq), flush=True)
        same &= eq
    print("%s: ang with spectrum == ang without: %s" % (label, bool(torch.equal(out[ORDER[0]][0], out[ORDER[0]][5]))), flush=True)
    return same


ok = run(coherent(B), "coherent streams (bench data)")
ok &= run(incoherent(B), "incoherent items (random angles per item)")
sys.exit(0 if ok else 1)
