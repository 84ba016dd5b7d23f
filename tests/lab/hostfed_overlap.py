"""Lab (GPU box): host-fed calls of baz_music_process on PAGE-LOCKED buffers, cfg2 with port 2, by call size: one launch sequence
(BAZ_MUSIC_HOSTFED_OVERLAP=0) against the overlapped form (cov4_evd_kernel on a small grid counting finished groups, the groups' scans on a
second stream behind a spinning gate), by covariance grid and group size.  Outputs must be identical.  PCIe-inclusive: not the metric.
usage: hostfed_overlap.py [blocks:rounds ...]   (0 = one launch sequence)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from gr_baz_amd import capi
from oracle import music_oracle as mo

forms = sys.argv[1:] or ["0", "4:1", "8:1", "16:1", "4:2", "8:2"]
c = mo.make_config("cfg2", 512)
m, n, N, res = c["m"], c["n"], c["nsamples"], c["res"]
BMAX = 4096
items = np.tile(c["items"], (BMAX // 512, 1))
x = torch.from_numpy(items.view(np.float32)).pin_memory().numpy().view(np.complex64)
mk = lambda shape: torch.zeros(shape, dtype=torch.float32).pin_memory().numpy()
out = (mk((BMAX, n)), mk((BMAX, n)), mk((BMAX, res)))
ref = {}
for B in (512, 1024, 2048, 2900, 4096):
    line = "cfg2 with port 2 %5d-item calls:" % B
    for f in forms:
        if f == "0":
            os.environ["BAZ_MUSIC_HOSTFED_OVERLAP"] = "0"
        else:
            os.environ["BAZ_MUSIC_HOSTFED_OVERLAP"] = "1"
            os.environ["BAZ_MUSIC_OVERLAP_MIN_ITEMS"] = "256"
            os.environ["BAZ_MUSIC_OVERLAP_COV_BLOCKS"], os.environ["BAZ_MUSIC_OVERLAP_GROUP_ROUNDS"] = f.split(":")
        with capi.Context(m, n, N, res, c["table"]) as ctx:
            o = (out[0][:B], out[1][:B], out[2][:B])
            for a in o:
                a[:] = 0
            for _ in range(3):
                ctx.process(x[:B], out=o)
            reps = max(5, 16384 // B)
            t0 = time.perf_counter()
            for _ in range(reps):
                ctx.process(x[:B], out=o)
            dt = (time.perf_counter() - t0) / reps
            got = (o[0].copy(), o[1].copy(), o[2].copy())
            if B not in ref:
                ref[B] = got
            same = all(np.array_equal(g.view(np.int32), r.view(np.int32)) for g, r in zip(got, ref[B]))
        line += "  [%4s] %.3f ms %.2fe6/s%s" % (f, dt * 1e3, B / dt / 1e6, "" if same else " DIFFERENT")
    print(line, flush=True)
