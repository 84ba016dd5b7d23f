"""Experiment: the GPU's 8 streams handled by NCTX contexts (own HIP stream each) issued round-robin, vs one context.
argv: nctx list (e.g. 1,2,4), items per GPU."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from gr_baz_amd import capi
from oracle import music_oracle as mo

dev = torch.device("cuda:0")
ncs = [int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1,2,4").split(",")]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
c = mo.make_config("cfg2", 512)
m, n, N, res = c["m"], c["n"], c["nsamples"], c["res"]
base = torch.from_numpy(c["items"].view(np.float32)).to(dev)
x = base.repeat((B + 511) // 512, 1)[:B].contiguous()
ang = torch.zeros(B, n, dtype=torch.float32, device=dev); lvl = torch.zeros_like(ang)
spec = torch.zeros(B, res, dtype=torch.float32, device=dev)
for nc in ncs:
    per = B // nc
    ctxs = [capi.Context(m, n, N, res, c["table"]) for _ in range(nc)]
    for cx in ctxs: cx.reserve(per)
    def step():
        for i, cx in enumerate(ctxs):
            o = i * per
            cx.process_device(x[o:].data_ptr(), per, ang[o:].data_ptr(), lvl[o:].data_ptr(), spec[o:].data_ptr())
    def sync():
        for cx in ctxs: cx.sync()
    for _ in range(3): step()
    sync()
    ws = []
    for r in range(3):
        t0 = time.perf_counter()
        for _ in range(20): step()
        sync(); ws.append((time.perf_counter() - t0) / 20 * 1e3)
    print("nctx %d (%d items each): wall ms/step %s -> %.3e items/s" % (nc, per, " ".join("%.4f" % w for w in ws), B / min(ws) * 1e3), flush=True)
    for cx in ctxs: cx.close()
