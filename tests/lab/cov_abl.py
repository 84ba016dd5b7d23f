"""Lab: what bounds cov4_x4_kernel?  Ablations (BAZ_MUSIC_COV_ABL: 1 no LDS transpose, 2 no MFMA, 3 neither) x grid."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
os.environ["BAZ_MUSIC_FUSE"] = "0"
from gr_baz_amd import capi, synth
from gr_baz_amd.baz.music_doa_helper import calculate_antenna_array_response
M, NE, N, RES, B = 4, 2, 1024, 3600, 262144
dev = torch.device("cuda:0")
arr = synth.array_geometry(M)
table = np.array(calculate_antenna_array_response([[0.5 * x, 0.5 * y] for x, y in arr], RES, 1.0)).astype(np.complex64)
x = torch.cat([synth.synth_stream(torch, dev, B // 8, M, N, arr, synth.C_LIGHT, 0.5, seed=1002 + s) for s in range(8)], dim=0)
R = torch.zeros(B, 16, 2, dtype=torch.float64, device=dev)
torch.cuda.synchronize()
import subprocess
abl = os.environ.get("BAZ_MUSIC_COV_ABL", "0")
for per_cu in (1, 2, 4):
    os.environ["BAZ_MUSIC_COV_BLOCKS_PER_CU"] = str(per_cu)
    with capi.Context(M, NE, N, RES, table) as ctx:
        for _ in range(20): ctx.debug_cov(x.data_ptr(), B, R.data_ptr())
        ctx.sync(); t0 = time.perf_counter()
        for _ in range(50): ctx.debug_cov(x.data_ptr(), B, R.data_ptr())
        ctx.sync(); ms = (time.perf_counter() - t0) / 50 * 1e3
    print("abl %s blocks/CU %d: %.3f ms -> %.2f TB/s" % (abl, per_cu, ms, 2.147 / ms), flush=True)
