"""Lab (GPU box): default wiring (no spectrum port), config 2, 262,144 items per call: the serial launch sequence against
BAZ_MUSIC_OVERLAP=k (scan of sub-batch i on a second stream beside covariance + EVD of sub-batch i + 1)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gr_baz_amd import capi, synth
from gr_baz_amd.baz.music_doa_helper import calculate_antenna_array_response

M, NE, N, RES = 4, 2, 1024, 3600
B = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
dev = torch.device("cuda:0")
arr = synth.array_geometry(M)
table = np.array(calculate_antenna_array_response([[0.5 * x, 0.5 * y] for x, y in arr], RES, 1.0)).astype(np.complex64)
x = torch.cat([synth.synth_stream(torch, dev, B // 8, M, N, arr, synth.C_LIGHT, 0.5, seed=1002 + s) for s in range(8)], dim=0)
ref = None
for k in (0, 2, 3, 4, 6, 8):
    os.environ["BAZ_MUSIC_OVERLAP"] = str(k)
    ang = torch.zeros(B, NE, dtype=torch.float32, device=dev)
    lvl = torch.zeros_like(ang)
    with capi.Context(M, NE, N, RES, table) as ctx:
        ctx.reserve(B)
        step = lambda: ctx.process_device(x.data_ptr(), B, ang.data_ptr(), lvl.data_ptr(), None)
        for _ in range(30): step()
        ctx.sync()
        ws = []
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(40): step()
            ctx.sync()
            ws.append((time.perf_counter() - t0) / 40 * 1e3)
        got = (ang.clone(), lvl.clone())
        refined = ctx.refined_values()
    if ref is None: ref = got
    same = bool(torch.equal(got[0], ref[0]) and torch.equal(got[1].view(torch.int32), ref[1].view(torch.int32)))
    print("overlap %d: ms/step min %.4f median %.4f -> %.3e items/s = %.1f %% of the HBM-read roofline | identical to the serial form: %s | refined %d"
          % (k, min(ws), sorted(ws)[2], B / sorted(ws)[2] * 1e3, B / sorted(ws)[2] * 1e3 * 8192 / 8e12 * 100, same, refined), flush=True)
