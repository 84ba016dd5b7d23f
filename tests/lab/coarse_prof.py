"""Lab: one configuration of the coarse-gated scan for rocprofv3 (config 2 without the spectrum port, coherent streams)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from gr_baz_amd import capi, synth
from gr_baz_amd.baz.music_doa_helper import calculate_antenna_array_response
M, NE, N, RES, B = 4, 2, 1024, 3600, 262144
dev = torch.device("cuda:0")
arr = synth.array_geometry(M)
table = np.array(calculate_antenna_array_response([[0.5 * x, 0.5 * y] for x, y in arr], RES, 1.0)).astype(np.complex64)
x = torch.cat([synth.synth_stream(torch, dev, B // 8, M, N, arr, synth.C_LIGHT, 0.5, seed=1002 + s) for s in range(8)], dim=0)
ang = torch.zeros(B, NE, dtype=torch.float32, device=dev)
lvl = torch.zeros_like(ang)
with capi.Context(M, NE, N, RES, table) as ctx:
    ctx.reserve(B)
    for _ in range(int(sys.argv[1]) if len(sys.argv) > 1 else 10):
        ctx.process_device(x.data_ptr(), B, ang.data_ptr(), lvl.data_ptr(), None)
    ctx.sync()
