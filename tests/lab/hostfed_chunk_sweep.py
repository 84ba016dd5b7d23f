"""Lab (GPU box): host-fed calls of baz_music_process on PAGE-LOCKED buffers, cfg2 with port 2 (and without), by call size and by the size of
the pipelined chunks (round 6: four slots, the whole call enqueued without the host waiting; BAZ_MUSIC_CHUNK_MIB forces the chunk size,
BAZ_MUSIC_SINGLE_MIB=1024 forces one zero-copy launch sequence).  Column "zc" = zero-copy, "def" = the library's own choice, numbers = MiB per chunk.
Outputs of every column are compared with the zero-copy column's (bit for bit).  PCIe-inclusive: never the metric.
usage: hostfed_chunk_sweep.py [MiB ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from gr_baz_amd import capi
from oracle import music_oracle as mo

sizes = [int(v) for v in sys.argv[1:]] or [4, 6, 8, 12, 16, 24]
c = mo.make_config("cfg2", 512)
m, n, N, res = c["m"], c["n"], c["nsamples"], c["res"]
BMAX = 16384
items = np.tile(c["items"], (BMAX // 512, 1))
x = torch.from_numpy(items.view(np.float32)).pin_memory().numpy().view(np.complex64)
mk = lambda shape: torch.zeros(shape, dtype=torch.float32).pin_memory().numpy()
for spec_on in (True, False):
    out = (mk((BMAX, n)), mk((BMAX, n)), mk((BMAX, res)) if spec_on else None)
    for B in (1024, 2048, 4096, 8192, 16384):
        line = "cfg2 %-12s %5d-item calls:" % ("with port 2" if spec_on else "ang/lvl only", B)
        ref = None
        for col in ["zc", "def"] + sizes:
            for k in ("BAZ_MUSIC_CHUNK_MIB", "BAZ_MUSIC_SINGLE_MIB"):
                os.environ.pop(k, None)
            if col == "zc":
                os.environ["BAZ_MUSIC_SINGLE_MIB"] = "1024"
            elif col != "def":
                os.environ["BAZ_MUSIC_CHUNK_MIB"] = str(col)
            with capi.Context(m, n, N, res, c["table"]) as ctx:
                o = (out[0][:B], out[1][:B], out[2][:B] if spec_on else None)
                for a in o:
                    if a is not None:
                        a[:] = 0
                for _ in range(3):
                    ctx.process(x[:B], out=o)
                reps = max(5, 32768 // B)
                t0 = time.perf_counter()
                for _ in range(reps):
                    ctx.process(x[:B], out=o)
                dt = (time.perf_counter() - t0) / reps
                got = (o[0].copy(), o[1].copy(), o[2].copy() if spec_on else None)
                if ref is None:
                    ref = got
                same = all(g is None or np.array_equal(g, r) for g, r in zip(got, ref))
            line += "  [%3s] %.3f ms %.2fe6/s%s" % (col, dt * 1e3, B / dt / 1e6, "" if same else " DIFFERENT")
        print(line, flush=True)
for k in ("BAZ_MUSIC_CHUNK_MIB", "BAZ_MUSIC_SINGLE_MIB"):
    os.environ.pop(k, None)
