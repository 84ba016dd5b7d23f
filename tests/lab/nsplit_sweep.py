"""Lab: per-stage times of one MUSIC shape by forced number of bin ranges per row (BAZ_MUSIC_NSPLIT; 0 = the rule of
scan_geometry()).  argv: m nsamples res batch spectrum(0/1) ranges,comma,separated"""
import os, sys, subprocess
if sys.argv[1] == "one":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import numpy as np, torch, bench
    from gr_baz_amd import capi, synth
    m, N, res, B, sp = (int(v) for v in sys.argv[2:7])
    dev = torch.device("cuda:0")
    r = bench.extra_music(torch, np, capi, synth, dev, torch.cuda.Stream(device=dev), m, N, res, B, bool(sp), 0.3)
    print("ranges %s: %.4f ms/step -> %.3e items/s | %s" % (os.environ.get("BAZ_MUSIC_NSPLIT", "0"), r["ms_per_step"], r["snapshots_per_s"],
          " ".join("%s %.3f" % (k, v) for k, v in r["stage_ms_per_launch"].items())), flush=True)
else:
    for ns in sys.argv[6].split(","):
        env = dict(os.environ, BAZ_MUSIC_NSPLIT=ns)
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "one"] + sys.argv[1:6], env=env, capture_output=True, text=True)
        print((r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1], flush=True)
