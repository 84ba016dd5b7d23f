"""Small fixed workload for rocprofv3 counter passes on the int8 scan: argv = m res batch iters [lab-env assignments ...]
(BASELINE configs[2]: 8 36000 16384; configs[4]'s MUSIC stage: 16 3600 16384), spectrum port wired, coherent 20-dB streams."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from gr_baz_amd import capi, synth
from gr_baz_amd.baz.music_doa_helper import calculate_antenna_array_response

M, RES, B, iters = (int(v) for v in sys.argv[1:5])
lab = False
for kv in sys.argv[5:]:
    k, v = kv.split("=")
    os.environ[k] = v
    lab = True
dev = torch.device("cuda:0")
N = 4096
arr = synth.array_geometry(M)
table = np.array(calculate_antenna_array_response([[0.5 * x, 0.5 * y] for x, y in arr], RES, 1.0)).astype(np.complex64)
x = torch.cat([synth.synth_stream(torch, dev, B // 8, M, N, arr, synth.C_LIGHT, 0.5, snr_db=20.0, seed=1003 + s) for s in range(8)], dim=0)
ang = torch.zeros(B, 2, dtype=torch.float32, device=dev)
lvl = torch.zeros_like(ang)
spec = torch.zeros(B, RES, dtype=torch.float32, device=dev)
with capi.Context(M, 2, N, RES, table, lab=lab) as ctx:
    ctx.reserve(B)
    for _ in range(iters):
        ctx.process_device(x.data_ptr(), B, ang.data_ptr(), lvl.data_ptr(), spec.data_ptr())
    ctx.sync()
print("done", flush=True)
