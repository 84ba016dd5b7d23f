"""Lab: 16 antennas, MUSIC only (device-resident), n = 1 and 2, short-form scan on / off."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import music_oracle as mo
dev = torch.device("cuda:0")
m, K, res, B = 16, 256, 3600, 16384
for n in (1, 2):
    for sig in ("1", "0"):
        os.environ["BAZ_MUSIC_SIG_SCAN"] = sig
        from gr_baz_amd import capi
        arr = mo.array_geometry(m)
        table = mo.steering_table_c64(arr, res, mo.FREQUENCY, mo.SPACING)
        items = mo.synth_items(64, m, m * K, arr, mo.FREQUENCY, mo.SPACING, angles_deg=(40.3, 121.7)[:n], snr_db=20.0, seed=5)
        x = torch.from_numpy(np.ascontiguousarray(items).view(np.float32)).to(dev).repeat(B // 64, 1)
        ang = torch.zeros(B, n, device=dev); lvl = torch.zeros(B, n, device=dev); spec = torch.zeros(B, res, device=dev)
        with capi.Context(m, n, m * K, res, table) as ctx:
            for _ in range(3): ctx.process_device(x.data_ptr(), B, ang.data_ptr(), lvl.data_ptr(), spec.data_ptr())
            ctx.sync(); t0 = time.perf_counter()
            for _ in range(10): ctx.process_device(x.data_ptr(), B, ang.data_ptr(), lvl.data_ptr(), spec.data_ptr())
            ctx.sync(); dt = (time.perf_counter() - t0) / 10
            ctx.profile(True)
            ctx.process_device(x.data_ptr(), B, ang.data_ptr(), lvl.data_ptr(), spec.data_ptr()); ctx.sync()
            st = {ctx.stage_name(k).split("::")[1][:24]: round(ctx.stage_ms(k)[0] / max(1, ctx.stage_ms(k)[1]), 3) for k in range(4)}
        print("m=16 n=%d short form %s: %.3f ms/step -> %.3g items/s %s" % (n, sig, dt * 1e3, B / dt, st), flush=True)
