"""Lab (GPU box; prepared at the end of round 2, NOT YET RUN): the helper's default wiring (no spectrum port) with the batch
dealt over NCTX contexts, each on its own stream.  Without stores the scan is bound by fp64 matrix issue and the
covariance by the HBM read stream, so kernels of different contexts should overlap: the question is
how much of 0.87 ms (one context, 262,144 items) is left.  Also runs the wired-spectrum case for contrast (there both
kernels are HBM-bound and round 1 found no gain).  argv: [nctx list=1,2,4] [items=262144]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from gr_baz_amd import capi, synth
from gr_baz_amd.baz.music_doa_helper import calculate_antenna_array_response

M, NE, N, RES = 4, 2, 1024, 3600
ncs = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1,2,4").split(",")]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
dev = torch.device("cuda:0")
arr = synth.array_geometry(M)
table = np.array(calculate_antenna_array_response([[0.5 * x, 0.5 * y] for x, y in arr], RES, 1.0)).astype(np.complex64)
x = torch.cat([synth.synth_stream(torch, dev, B // 8, M, N, arr, synth.C_LIGHT, 0.5, seed=1002 + s) for s in range(8)], dim=0)
ang = torch.zeros(B, NE, dtype=torch.float32, device=dev)
lvl = torch.zeros_like(ang)
spec = torch.zeros(B, RES, dtype=torch.float32, device=dev)
ref = None
for with_spec in (False, True):
    for nc in ncs:
        per = B // nc
        ctxs = [capi.Context(M, NE, N, RES, table) for _ in range(nc)]
        for cx in ctxs:
            cx.reserve(per)
        torch.cuda.synchronize()

        def step():
            for i, cx in enumerate(ctxs):
                o = i * per
                cx.process_device(x[o:].data_ptr(), per, ang[o:].data_ptr(), lvl[o:].data_ptr(), spec[o:].data_ptr() if with_spec else 0)

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
        got = ang.clone()
        if ref is None:
            ref = got
        print("%s  nctx %d (%6d items each): ms/step min %.4f median %.4f -> %.3e items/s   ang identical to the first run: %s"
              % ("spectrum wired " if with_spec else "no spectrum    ", nc, per, min(ws), sorted(ws)[2], B / sorted(ws)[2] * 1e3,
                 bool(torch.equal(got, ref))), flush=True)
        for cx in ctxs:
            cx.close()
