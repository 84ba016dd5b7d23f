"""The table-resident scan (scan_res_kernel, BAZ_MUSIC_RES_SCAN=1) against the staged one (=0) on the bench's cfg2 inputs:
bit identity of ang / lvl / spectrum, per-stage and wall times.  argv: batch [scene = coherent|incoherent] [snr_db]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
from gr_baz_amd import capi, synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
scene = sys.argv[2] if len(sys.argv) > 2 else "coherent"
snr = float(sys.argv[3]) if len(sys.argv) > 3 else 20.0
m, n, N, res = 4, 2, 1024, 3600
dev = torch.device("cuda:0")
arr, table = bench.helper_table(np, synth, m, res)
if scene == "incoherent":
    x = synth.synth_scenes(torch, dev, B, m, N, arr, bench.FREQUENCY, bench.SPACING, n, snr_db=snr, seed=1007)
else:
    x = torch.cat([synth.synth_stream(torch, dev, B // 8, m, N, arr, bench.FREQUENCY, bench.SPACING, snr_db=snr, seed=1003 + s)
                   for s in range(8)], dim=0)
out = {}
for mode in (0, 1, 0, 1):
    os.environ["BAZ_MUSIC_RES_SCAN"] = str(mode)
    ang = torch.zeros(B, n, dtype=torch.float32, device=dev); lvl = torch.zeros_like(ang)
    spec = torch.zeros(B, res, dtype=torch.float32, device=dev)
    ctx = capi.Context(m, n, N, res, table); ctx.reserve(B)
    step = lambda: ctx.process_device(x.data_ptr(), B, ang.data_ptr(), lvl.data_ptr(), spec.data_ptr())
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        for _ in range(5): step()
        ctx.sync()
    ts = []
    for _ in range(8):
        t0 = time.perf_counter()
        for _ in range(20): step()
        ctx.sync(); ts.append((time.perf_counter() - t0) / 20 * 1e3)
    ctx.profile(1)
    for _ in range(5): step()
    ctx.sync(); st = [ctx.stage_ms(s)[0] / max(ctx.stage_ms(s)[1], 1) for s in range(4)]; ctx.profile(0)
    step(); ctx.sync(); refined = ctx.refined_values()
    cur = (ang.clone(), lvl.clone(), spec)
    same = "-"
    if 0 in out and mode == 1:
        a0, l0, s0 = out[0]
        same = "ang %s lvl %s spectrum %s" % (bool((a0 == cur[0]).all()), bool((l0 == cur[1]).all()), bool(torch.equal(s0, cur[2])))
    if mode == 0 and 0 not in out:
        out[0] = (cur[0], cur[1], spec.clone())
    print("%s %g dB B=%d res_scan %d: %.4f ms/step (min %.4f) -> %.3e items/s | cov+evd %.3f scan %.3f (%.2f TB/s of spectrum) merge %.3f | refined %d | identical to the staged form: %s"
          % (scene, snr, B, mode, float(np.median(ts)), min(ts), B / float(np.median(ts)) * 1e3, st[0] + st[1], st[2],
             B * res * 4 / (st[2] * 1e-3) / 1e12, st[3], refined, same), flush=True)
    ctx.close(); del spec
