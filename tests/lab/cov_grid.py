"""Lab: covariance kernel time against its grid size (blocks per CU), cfg2, 262,144 items, two input allocations."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gr_baz_amd import capi, synth
from gr_baz_amd.baz.music_doa_helper import calculate_antenna_array_response
M, NE, N, RES, B = 4, 2, 1024, 3600, 262144
dev = torch.device("cuda:0")
arr = synth.array_geometry(M)
table = np.array(calculate_antenna_array_response([[0.5 * x, 0.5 * y] for x, y in arr], RES, 1.0)).astype(np.complex64)
xs = [torch.cat([synth.synth_stream(torch, dev, B // 8, M, N, arr, synth.C_LIGHT, 0.5, seed=1002 + s) for s in range(8)], dim=0) for _ in range(2)]
R = torch.zeros(B, 16, 2, dtype=torch.float64, device=dev)
torch.cuda.synchronize()
for per_cu in (0, 1, 2, 3, 4, 6, 8):
    if per_cu: os.environ["BAZ_MUSIC_COV_BLOCKS_PER_CU"] = str(per_cu)
    else: os.environ.pop("BAZ_MUSIC_COV_BLOCKS_PER_CU", None)
    with capi.Context(M, NE, N, RES, table) as ctx:
        res = []
        for x in xs:
            for _ in range(20): ctx.debug_cov(x.data_ptr(), B, R.data_ptr())
            ctx.sync(); t0 = time.perf_counter()
            for _ in range(50): ctx.debug_cov(x.data_ptr(), B, R.data_ptr())
            ctx.sync(); res.append((time.perf_counter() - t0) / 50 * 1e3)
    print("blocks per CU %s: cov %.3f / %.3f ms (two input buffers) -> %.2f / %.2f TB/s" % (per_cu or "occupancy", res[0], res[1], 2.147 / res[0], 2.147 / res[1]), flush=True)
