"""Small fixed workload for rocprofv3 runs: cfg, batch, iters from argv."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gr_baz_amd import capi
from oracle import music_oracle as mo
name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda:0")
c = mo.make_config(name, 512)
m, n, N, res = c["m"], c["n"], c["nsamples"], c["res"]
base = torch.from_numpy(c["items"].view(np.float32)).to(dev)
x = base.repeat((B + 511) // 512, 1)[:B].contiguous()
ang = torch.zeros(B, n, dtype=torch.float32, device=dev); lvl = torch.zeros_like(ang)
spec = torch.zeros(B, res, dtype=torch.float32, device=dev)
ctx = capi.Context(m, n, N, res, c["table"]); ctx.reserve(B)
for _ in range(iters):
    ctx.process_device(x.data_ptr(), B, ang.data_ptr(), lvl.data_ptr(), spec.data_ptr())
ctx.sync()
print("done", flush=True)
ctx.close()
