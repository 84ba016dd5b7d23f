"""Lab (GPU box): default wiring (no spectrum port), config 2: the serial launch sequence (BAZ_MUSIC_ROLES=0) against
covariance + EVD of sub-batch i + 1 and the gated scan of sub-batch i as the two roles of one launch (default).
argv: [items=262144]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gr_baz_amd import capi, synth
from gr_baz_amd.baz.music_doa_helper import calculate_antenna_array_response

M, NE, N, RES = 4, 2, 1024, 3600
dev = torch.device("cuda:0")
arr = synth.array_geometry(M)
table = np.array(calculate_antenna_array_response([[0.5 * x, 0.5 * y] for x, y in arr], RES, 1.0)).astype(np.complex64)
for B in ([int(sys.argv[1])] if len(sys.argv) > 1 else [262144, 131072, 200000, 524288]):
    x = torch.cat([synth.synth_stream(torch, dev, -(-B // 8), M, N, arr, synth.C_LIGHT, 0.5, seed=1002 + s) for s in range(8)], dim=0)[:B].contiguous()
    ref = None
    for roles in ("0", "1"):
        os.environ["BAZ_MUSIC_ROLES"] = roles
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
        print("%7d items, roles %s: ms/step min %.4f median %.4f -> %.3e items/s = %.1f %% of the HBM-read roofline | identical to the serial form: %s | refined %d"
              % (B, roles, min(ws), sorted(ws)[2], B / sorted(ws)[2] * 1e3, B / sorted(ws)[2] * 1e3 * 8192 / 8e12 * 100, same, refined), flush=True)
