"""How many (item, bin) values take the literal-form refinement path of the scan, and what it costs, by SNR
(cfg2 / cfg3 shapes)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gr_baz_amd import capi
from oracle import music_oracle as mo
dev = torch.device("cuda:0")
for cfg, B in (("cfg2", 262144), ("cfg3", 4096)):
    for snr in (20.0, 30.0, 40.0, 50.0, 60.0, 80.0):
        c = mo.make_config(cfg, 512, snr_db=snr)
        m, n, N, res = c["m"], c["n"], c["nsamples"], c["res"]
        x = torch.from_numpy(c["items"].view(np.float32)).to(dev).repeat(B // 512, 1).contiguous()
        ang = torch.zeros(B, n, dtype=torch.float32, device=dev); lvl = torch.zeros_like(ang)
        spec = torch.zeros(B, res, dtype=torch.float32, device=dev)
        with capi.Context(m, n, N, res, c["table"]) as ctx:
            for _ in range(3): ctx.process_device(x.data_ptr(), B, ang.data_ptr(), lvl.data_ptr(), spec.data_ptr())
            ctx.sync(); t0 = time.perf_counter()
            for _ in range(10): ctx.process_device(x.data_ptr(), B, ang.data_ptr(), lvl.data_ptr(), spec.data_ptr())
            ctx.sync(); dt = (time.perf_counter() - t0) / 10
            r = ctx.refined_items()
        print("%s snr %3.0f dB: %9d of %d values redone in literal form (%.4f %%), %.3f ms/step -> %.3e items/s, spectrum max %.3g"
              % (cfg, snr, r, B * res, 100.0 * r / (B * res), dt * 1e3, B / dt, float(spec.max())), flush=True)
