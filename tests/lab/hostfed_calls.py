"""Lab (GPU box, lab build): host-fed calls of baz_music_process on PAGE-LOCKED buffers (what work() does under a scheduler whose stream buffers
the block has locked), cfg2, by call size.  Columns: 0 = the library's zero-copy launch sequence; -64 = the same with round 4's 64-item covariance
tasks (BAZ_MUSIC_COVEVD_TASK_ITEMS, lab build).  (Positive values selected the sub-chunk size of the rejected hybrid form -- input by DMA per sub-chunk,
outputs zero-copy -- whose code is not kept: profiles/r05_hostfed_calls.txt has its numbers; they now run the library's form.)  PCIe-inclusive: not the metric.
usage: hostfed_calls.py [0 | -64 ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from gr_baz_amd import capi
from oracle import music_oracle as mo

chunks = [int(v) for v in sys.argv[1:]] or [0, -64]
c = mo.make_config("cfg2", 512)
m, n, N, res = c["m"], c["n"], c["nsamples"], c["res"]
BMAX = 8192
items = np.tile(c["items"], (BMAX // 512, 1))
x = torch.from_numpy(items.view(np.float32)).pin_memory().numpy().view(np.complex64)
mk = lambda shape: torch.zeros(shape, dtype=torch.float32).pin_memory().numpy()
ref = {}
for spec_on in (True, False):
    out = (mk((BMAX, n)), mk((BMAX, n)), mk((BMAX, res)) if spec_on else None)
    for B in (256, 512, 1024, 2048, 4096, 8192):
        line = "cfg2 %-12s %5d-item calls:" % ("with port 2" if spec_on else "ang/lvl only", B)
        for ch in chunks:
            os.environ.pop("BAZ_MUSIC_COVEVD_TASK_ITEMS", None)
            if ch < 0:                       # -64: one zero-copy launch sequence with round 4's 64-item covariance tasks
                os.environ["BAZ_MUSIC_COVEVD_TASK_ITEMS"] = str(-ch)
            with capi.Context(m, n, N, res, c["table"], lab=True) as ctx:
                o = (out[0][:B], out[1][:B], out[2][:B] if spec_on else None)
                for _ in range(3):
                    ctx.process(x[:B], out=o)
                reps = max(5, 16384 // B)
                t0 = time.perf_counter()
                for _ in range(reps):
                    ctx.process(x[:B], out=o)
                dt = (time.perf_counter() - t0) / reps
                key = (spec_on, B)
                got = (o[0].copy(), o[1].copy(), o[2][:64].copy() if spec_on else None)
                if key not in ref:
                    ref[key] = got
                same = all(g is None or np.array_equal(g, r) for g, r in zip(got, ref[key]))
            line += "  [%3d] %.3f ms %.2fe6/s%s" % (ch, dt * 1e3, B / dt / 1e6, "" if same else " DIFFERENT")
        print(line, flush=True)
