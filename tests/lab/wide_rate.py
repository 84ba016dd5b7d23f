"""Lab: throughput of the run-time-m path (17..64 antennas), device-resident."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gr_baz_amd import capi
from oracle import music_oracle as mo
dev = torch.device("cuda:0")
for m, n, K, res, B in ((17, 2, 256, 3600, 4096), (24, 2, 128, 3600, 4096), (32, 1, 128, 3600, 4096), (32, 2, 128, 3600, 4096), (32, 4, 128, 3600, 4096), (32, 2, 128, 3600, 16384), (48, 3, 64, 3600, 2048), (64, 2, 64, 3600, 2048), (64, 4, 64, 3600, 2048), (64, 8, 256, 720, 2048)):
    N = m * K
    arr = mo.array_geometry(m)
    table = mo.steering_table_c64(arr, res, mo.FREQUENCY, mo.SPACING)
    angles = (40.3, 121.7) if n == 2 else tuple(np.linspace(40.3, 300.0, n))          # as many emitters as the block is told to expect
    items = mo.synth_items(64, m, N, arr, mo.FREQUENCY, mo.SPACING, angles_deg=angles, snr_db=20.0, seed=5)
    x = torch.from_numpy(np.ascontiguousarray(items).view(np.float32)).to(dev).repeat(B // 64, 1)
    ang = torch.zeros(B, n, device=dev); lvl = torch.zeros(B, n, device=dev); spec = torch.zeros(B, res, device=dev)
    with capi.Context(m, n, N, res, table) as ctx:
        for _ in range(2): ctx.process_device(x.data_ptr(), B, ang.data_ptr(), lvl.data_ptr(), spec.data_ptr())
        ctx.sync(); t0 = time.perf_counter()
        reps = 5
        for _ in range(reps): ctx.process_device(x.data_ptr(), B, ang.data_ptr(), lvl.data_ptr(), spec.data_ptr())
        ctx.sync(); dt = (time.perf_counter() - t0) / reps
        ctx.profile(True)
        ctx.process_device(x.data_ptr(), B, ang.data_ptr(), lvl.data_ptr(), spec.data_ptr())
        ctx.sync()
        st = {ctx.stage_name(k).split("::")[1]: round(ctx.stage_ms(k)[0] / max(1, ctx.stage_ms(k)[1]), 3) for k in range(4)}
    print("m=%d n=%d K=%d res=%d batch=%d: %.2f ms/step -> %.3g items/s  %s" % (m, n, K, res, B, dt * 1e3, B / dt, st), flush=True)
