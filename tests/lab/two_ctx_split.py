"""Lab (GPU box, lab build): cfg2 WITH port 2, steps dealt alternately over two contexts on their own streams, so that the covariance of one step can run
beside the scan of the one before.  Round 3 found no gain with the fused cov4_evd_kernel (192 registers: nothing fits beside four 128-register scan waves
per SIMD).  Here: the covariance as its own 80-register kernel (BAZ_MUSIC_FUSE=0: cov4_x4_kernel + evd_proj_kernel) and the scan capped at three
workgroups per CU by an LDS request it never touches (BAZ_MUSIC_SCAN_LDS_PAD), which leaves a slot per SIMD for a covariance wave.
argv: [items=262144]"""
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
table = np.array(calculate_antenna_array_response([[0.5 * x, 0.5 * y] for x, y in arr], RES, 1.0)).astype(np.complex64)
x = torch.cat([synth.synth_stream(torch, dev, B // 8, M, N, arr, synth.C_LIGHT, 0.5, seed=1002 + s) for s in range(8)], dim=0)
outs = [(torch.zeros(B, NE, dtype=torch.float32, device=dev), torch.zeros(B, NE, dtype=torch.float32, device=dev),
         torch.zeros(B, RES, dtype=torch.float32, device=dev)) for _ in range(2)]
ref = None
for name, nctx, fuse, pad in (("one context, fused covariance + EVD (the product)", 1, 1, 0),
                              ("two contexts, fused", 2, 1, 0),
                              ("one context, covariance and EVD as two kernels", 1, 0, 0),
                              ("two contexts, two kernels", 2, 0, 0),
                              ("two contexts, two kernels, scan asks for 41 KiB of LDS (3 per CU)", 2, 0, 41 * 1024 - 16384),
                              ("two contexts, fused, scan asks for 41 KiB (3 per CU)", 2, 1, 41 * 1024 - 16384),
                              ("one context, two kernels, scan 3 per CU", 1, 0, 41 * 1024 - 16384),
                              ("two contexts, two kernels, scan asks for 54 KiB (2 per CU)", 2, 0, 54 * 1024 - 16384)):
    os.environ["BAZ_MUSIC_FUSE"] = str(fuse)
    os.environ["BAZ_MUSIC_SCAN_LDS_PAD"] = str(pad)
    ctxs = [capi.Context(M, NE, N, RES, table, lab=True) for _ in range(nctx)]
    for cx in ctxs:
        cx.reserve(B)
    torch.cuda.synchronize()
    k = [0]

    def step():
        i = k[0] % nctx
        k[0] += 1
        a, l, s = outs[i]
        ctxs[i].process_device(x.data_ptr(), B, a.data_ptr(), l.data_ptr(), s.data_ptr())

    def sync():
        for cx in ctxs:
            cx.sync()

    for _ in range(30):
        step()
    sync()
    ws = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(40):
            step()
        sync()
        ws.append((time.perf_counter() - t0) / 40 * 1e3)
    got = (outs[0][0].clone(), outs[0][1].clone(), outs[0][2][::997].clone())
    if ref is None:
        ref = got
    same = all(bool(torch.equal(g.view(torch.int32), r.view(torch.int32))) for g, r in zip(got, ref))
    print("%-72s ms/step min %.4f median %.4f -> %.3e items/s | identical: %s" % (name, min(ws), sorted(ws)[2], B / sorted(ws)[2] * 1e3, same), flush=True)
    for cx in ctxs:
        cx.close()
