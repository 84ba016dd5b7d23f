"""cfg2 without the spectrum port (ports 0/1 only): per-stage and wall throughput. argv: batch"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gr_baz_amd import capi
from oracle import music_oracle as mo
B = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
dev = torch.device("cuda:0")
c = mo.make_config("cfg2", 512)
m, n, N, res = c["m"], c["n"], c["nsamples"], c["res"]
x = torch.from_numpy(c["items"].view(np.float32)).to(dev).repeat(B // 512, 1).contiguous()
ang = torch.zeros(B, n, dtype=torch.float32, device=dev); lvl = torch.zeros_like(ang)
spec = torch.zeros(B, res, dtype=torch.float32, device=dev)
ctx = capi.Context(m, n, N, res, c["table"]); ctx.reserve(B)
for with_spec in (True, False):
    sp = spec.data_ptr() if with_spec else None
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        for _ in range(5): ctx.process_device(x.data_ptr(), B, ang.data_ptr(), lvl.data_ptr(), sp)
        ctx.sync()
    a_ref = ang.clone() if with_spec else a_ref
    t0 = time.perf_counter()
    for _ in range(20): ctx.process_device(x.data_ptr(), B, ang.data_ptr(), lvl.data_ptr(), sp)
    ctx.sync(); dt = (time.perf_counter() - t0) / 20
    ctx.profile(1)
    for _ in range(3): ctx.process_device(x.data_ptr(), B, ang.data_ptr(), lvl.data_ptr(), sp)
    ctx.sync(); st = [ctx.stage_ms(s)[0] / 3 for s in range(4)]; ctx.profile(0)
    print("cfg2 B=%d spectrum port %s: %.3f ms/step -> %.3e items/s  [cov %.3f evd %.3f scan %.3f merge %.3f]  same DoA as with port: %s"
          % (B, with_spec, dt * 1e3, B / dt, *st, bool((ang == a_ref).all())), flush=True)
ctx.close()
