"""Lab: the staged scan's ablations on the bench's cfg2 inputs (BAZ_MUSIC_SCAN_VARIANT): where does its launch time go?"""
import sys, os, time, subprocess
if len(sys.argv) > 1 and sys.argv[1] == "one":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import numpy as np, torch, bench
    from gr_baz_amd import capi, synth
    B, m, n, N, res = 262144, 4, 2, 1024, 3600
    dev = torch.device("cuda:0")
    arr, table = bench.helper_table(np, synth, m, res)
    x = torch.cat([synth.synth_stream(torch, dev, B // 8, m, N, arr, bench.FREQUENCY, bench.SPACING, snr_db=20.0, seed=1003 + s) for s in range(8)], dim=0)
    ang = torch.zeros(B, n, dtype=torch.float32, device=dev); lvl = torch.zeros_like(ang)
    spec = torch.zeros(B, res, dtype=torch.float32, device=dev)
    ctx = capi.Context(m, n, N, res, table); ctx.reserve(B)
    step = lambda: ctx.process_device(x.data_ptr(), B, ang.data_ptr(), lvl.data_ptr(), spec.data_ptr())
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        for _ in range(5): step()
        ctx.sync()
    ctx.profile(1)
    for _ in range(10): step()
    ctx.sync(); st = [ctx.stage_ms(s)[0] / max(ctx.stage_ms(s)[1], 1) for s in range(4)]
    fp = [int(t.view(torch.int32).to(torch.int64).sum()) & (2**64 - 1) for t in (spec, ang, lvl)]
    print("variant %s: cov+evd %.3f scan %.3f merge %.3f ms | fingerprint of spectrum / ang / lvl bits %x %x %x"
          % (os.environ.get("BAZ_MUSIC_SCAN_VARIANT", "0"), st[0] + st[1], st[2], st[3], fp[0], fp[1], fp[2]), flush=True)
    ctx.close()
else:
    names = {"1": "shipped (staged)", "2": "ungated top-n network", "6": "everything but the spectrum stores", "7": "stores + staging + barriers only",
             "8": "MFMA + float conversions only (no top-n, no stores)"}
    for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else ("1", "6", "8", "7", "2", "1")):
        env = dict(os.environ, BAZ_MUSIC_SCAN_VARIANT=v, BAZ_MUSIC_RES_SCAN="0")
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "one"], env=env, capture_output=True, text=True)
        print(names[v].ljust(52), (r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1], flush=True)
