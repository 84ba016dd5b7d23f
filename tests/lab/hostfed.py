"""Host-fed throughput of baz_music_process (what the GNU Radio block's work() calls), cfg2.
PCIe-inclusive: NOT the headline metric.  Output buffers are allocated once and reused, like a
scheduler's; argv: batch [chunk MiB list for BAZ_MUSIC_CHUNK_MIB, e.g. 4,16,64]."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gr_baz_amd import capi
from oracle import music_oracle as mo
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
threads = [v for v in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["default"])]
c = mo.make_config("cfg2", 512)
m, n, N, res = c["m"], c["n"], c["nsamples"], c["res"]
items = np.tile(c["items"], (B // 512, 1))
ref_ang = None
for th in threads:
    if th == "default":
        os.environ.pop("BAZ_MUSIC_CHUNK_MIB", None)
    else:
        os.environ["BAZ_MUSIC_CHUNK_MIB"] = th
    ctx = capi.Context(m, n, N, res, c["table"])
    for spec_on in (True, False):
        for pinned in (False, True):
            if pinned:
                x = torch.from_numpy(items.view(np.float32)).pin_memory().numpy().view(np.complex64)
                mk = lambda shape: torch.zeros(shape, dtype=torch.float32).pin_memory().numpy()
            else:
                x = items
                mk = lambda shape: np.zeros(shape, np.float32)
            out = (mk((B, n)), mk((B, n)), mk((B, res)) if spec_on else None)
            ctx.process(x[:4096], want_spectrum=spec_on)   # warm-up / allocate
            ctx.process(x, out=out)
            reps = 3
            t0 = time.perf_counter()
            for _ in range(reps):
                ctx.process(x, out=out)
            dt = (time.perf_counter() - t0) / reps
            if ref_ang is None:
                ref_ang = out[0].copy(); ref_spec = out[2][:1024].copy()
            ok = bool(np.array_equal(out[0], ref_ang)) and (not spec_on or bool(np.array_equal(out[2][:1024], ref_spec)))
            bytes_item = N * 8 + 8 * n + (4 * res if spec_on else 0)
            print("host-fed cfg2 B=%d chunk_MiB=%s spectrum=%s buffers %s: %.1f ms -> %.3e items/s, %.1f GB/s over PCIe  same-results %s"
                  % (B, th, spec_on, "pinned" if pinned else "pageable", dt * 1e3, B / dt, B / dt * bytes_item / 1e9, ok), flush=True)
    ctx.close()
