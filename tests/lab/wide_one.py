"""Lab: one wide shape under rocprofv3 (per-kernel times).  argv: m n K res batch [snr]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gr_baz_amd import capi
from oracle import music_oracle as mo
m, n, K, res, B = (int(v) for v in sys.argv[1:6])
snr = float(sys.argv[6]) if len(sys.argv) > 6 else 20.0
dev = torch.device("cuda:0")
N = m * K
arr = mo.array_geometry(m)
table = mo.steering_table_c64(arr, res, mo.FREQUENCY, mo.SPACING)
angles = (40.3, 121.7) if n == 2 else tuple(np.linspace(40.3, 300.0, n))
items = mo.synth_items(64, m, N, arr, mo.FREQUENCY, mo.SPACING, angles_deg=angles, snr_db=snr, seed=5)
x = torch.from_numpy(np.ascontiguousarray(items).view(np.float32)).to(dev).repeat(B // 64, 1)
ang = torch.zeros(B, n, device=dev); lvl = torch.zeros(B, n, device=dev); spec = torch.zeros(B, res, device=dev)
with capi.Context(m, n, N, res, table) as ctx:
    for _ in range(3): ctx.process_device(x.data_ptr(), B, ang.data_ptr(), lvl.data_ptr(), spec.data_ptr())
    ctx.sync()
print("done", ang[0].tolist())
