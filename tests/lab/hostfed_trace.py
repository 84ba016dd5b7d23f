"""Lab (GPU box, under rocprofv3 --hip-trace --memory-copy-trace --kernel-trace --stats): a handful of host-fed calls cut into small pipelined chunks,
to see where a chunk's fixed cost goes.  usage: hostfed_trace.py <items per call> <MiB per chunk> <calls>"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
B, mib, calls = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
os.environ["BAZ_MUSIC_CHUNK_MIB"] = str(mib)
from gr_baz_amd import capi
from oracle import music_oracle as mo
c = mo.make_config("cfg2", 512)
items = np.tile(c["items"], (B // 512, 1))
x = torch.from_numpy(items.view(np.float32)).pin_memory().numpy().view(np.complex64)
mk = lambda shape: torch.zeros(shape, dtype=torch.float32).pin_memory().numpy()
o = (mk((B, 2)), mk((B, 2)), mk((B, 3600)))
with capi.Context(4, 2, 1024, 3600, c["table"]) as ctx:
    for _ in range(3):
        ctx.process(x, out=o)
    t0 = time.perf_counter()
    for _ in range(calls):
        ctx.process(x, out=o)
    print("%d-item calls, %d MiB chunks: %.3f ms per call" % (B, mib, (time.perf_counter() - t0) / calls * 1e3))
