"""Host-fed throughput of baz_music_process (what the GNU Radio block's work() calls), cfg2.
PCIe-inclusive: NOT the headline metric."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gr_baz_amd import capi
from oracle import music_oracle as mo
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
c = mo.make_config("cfg2", 512)
m, n, N, res = c["m"], c["n"], c["nsamples"], c["res"]
items = np.tile(c["items"], (B // 512, 1))
ctx = capi.Context(m, n, N, res, c["table"])
for spec_on in (True, False):
    for pinned in (False, True):
        if pinned:
            tin = torch.from_numpy(items.view(np.float32)).pin_memory()
            x = tin.numpy().view(np.complex64)
        else:
            x = items
        ctx.process(x[:4096], want_spectrum=spec_on)   # warm-up / allocate
        reps = 3
        t0 = time.perf_counter()
        for _ in range(reps):
            ang, lvl, spec = ctx.process(x, want_spectrum=spec_on)
        dt = (time.perf_counter() - t0) / reps
        bytes_item = N * 8 + 8 * n + (4 * res if spec_on else 0)
        print("host-fed cfg2 B=%d spectrum=%s input %s: %.1f ms -> %.3e items/s, %.1f GB/s over PCIe (outputs always pageable numpy)"
              % (B, spec_on, "pinned" if pinned else "pageable", dt * 1e3, B / dt, B / dt * bytes_item / 1e9), flush=True)
ctx.close()
