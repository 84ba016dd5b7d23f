"""Per-stage timing on the GPU box + parity spot check (argv: label list, batch, cfg).  The label used to
select BAZ_MUSIC_SCAN_VARIANT builds while the scan kernel was being tuned; it is now just a tag."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from gr_baz_amd import capi
from oracle import music_oracle as mo

dev = torch.device("cuda:0")
variants = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,1,2,3,4,5".split(","))]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
name = sys.argv[3] if len(sys.argv) > 3 else "cfg2"
c = mo.make_config(name, 512)
m, n, N, res = c["m"], c["n"], c["nsamples"], c["res"]
ao, lo, so, st = mo.music_doa_work_batch(c["items"][:192], c["table"], m, n)
base = torch.from_numpy(c["items"].view(np.float32)).to(dev)
x = base.repeat((B + 511) // 512, 1)[:B].contiguous()
ang = torch.zeros(B, n, dtype=torch.float32, device=dev); lvl = torch.zeros_like(ang)
spec = torch.zeros(B, res, dtype=torch.float32, device=dev)
rounds = 3
results = {v: [] for v in variants}
ctxs = {}
for v in variants:
    os.environ["BAZ_MUSIC_SCAN_VARIANT"] = str(v)
    ctx = capi.Context(m, n, N, res, c["table"]); ctx.reserve(B); ctxs[v] = ctx
    spec.zero_(); ang.zero_(); lvl.zero_()
    ctx.process_device(x.data_ptr(), B, ang.data_ptr(), lvl.data_ptr(), spec.data_ptr()); ctx.sync()
    sg = spec[:192].cpu().numpy().astype(np.float64)
    rel = (np.abs(sg - st) / st).max()
    ok_all = bool((spec[:512] == spec[512:1024]).all()) if B >= 1024 else True
    tail_ok = bool((spec[B - 512:] == spec[:512]).all()) if B % 512 == 0 and B >= 1024 else True
    print("variant %d: parity spectrum max rel %.3g ang_equal %s lvl rel %.3g  repeat-consistent %s %s" % (
        v, rel, bool((ang[:192].cpu().numpy() == ao).all()), (np.abs(lvl[:192].cpu().numpy() - lo) / lo).max(), ok_all, tail_ok), flush=True)
for r in range(rounds):   # interleaved rounds
    for v in variants:
        ctx = ctxs[v]
        ctx.profile(True)
        for _ in range(5): ctx.process_device(x.data_ptr(), B, ang.data_ptr(), lvl.data_ptr(), spec.data_ptr())
        ctx.sync()
        ms = [ctx.stage_ms(s)[0] / 5 for s in range(4)]
        ctx.profile(False)
        results[v].append(ms)
wall = {v: [] for v in variants}
for r in range(rounds):   # un-instrumented wall clock, 20 steps back to back
    for v in variants:
        ctx = ctxs[v]
        for _ in range(3): ctx.process_device(x.data_ptr(), B, ang.data_ptr(), lvl.data_ptr(), spec.data_ptr())
        ctx.sync(); t0 = time.perf_counter()
        for _ in range(20): ctx.process_device(x.data_ptr(), B, ang.data_ptr(), lvl.data_ptr(), spec.data_ptr())
        ctx.sync(); wall[v].append((time.perf_counter() - t0) / 20 * 1e3)
for v in variants:
    print("variant %d: wall ms/step %s" % (v, " ".join("%.4f" % w for w in wall[v])), flush=True)
for v in variants:
    a = np.array(results[v])
    med = np.median(a, axis=0)
    print("variant %d: cov %.3f evd %.3f scan %.3f ms (median of %d rounds; scan min %.3f) -> scan %.3e items/s, pipeline %.3e items/s" % (
        v, med[0], med[1], med[2], rounds, a[:, 2].min(), B / med[2] * 1e3, B / med.sum() * 1e3), flush=True)
