"""Lab (GPU box): the int8-matrix-core scan against the fp64 scan of the same build (BAZ_MUSIC_EXACT=1), spectrum port
wired.  Shapes: BASELINE configs[2] (m8, N4096, res36000; 16,384 items), configs[4]'s MUSIC stage (m16, N4096, res3600; 16,384
items) and a few in between; coherent streams and per-item scenes.  Prints per-stage ms, items/s, the share of wave steps the
integer scan recomputed in the fp64 form and the worst relative difference of the two spectra over a sample of the batch.

usage: python tests/lab/i8_rate.py [quick]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from gr_baz_amd import capi, synth
from gr_baz_amd.baz.music_doa_helper import calculate_antenna_array_response

dev = torch.device("cuda:0")
quick = "quick" in sys.argv[1:]
CASES = [(8, 2, 4096, 36000, 16384, "coherent", 20.0), (16, 2, 4096, 3600, 16384, "coherent", 20.0)]
if "ab" in sys.argv[1:]:          # same-box A/B of two builds of the library (BAZ_MUSIC_LAB_LIB=quick / lab): four shapes
    CASES += [(8, 2, 4096, 36000, 16384, "incoherent", 20.0), (8, 2, 1024, 3600, 65536, "coherent", 20.0)]
elif not quick:
    CASES += [(8, 2, 4096, 36000, 16384, "incoherent", 20.0), (8, 2, 4096, 36000, 16384, "coherent", 60.0),
              (16, 2, 4096, 3600, 16384, "incoherent", 20.0), (8, 2, 1024, 3600, 65536, "coherent", 20.0),
              (6, 2, 6 * 128, 3600, 65536, "coherent", 20.0), (12, 2, 12 * 128, 3600, 16384, "coherent", 20.0)]
for M, NE, N, RES, B, kind, snr in CASES:
    arr = synth.array_geometry(M)
    table = np.array(calculate_antenna_array_response([[0.5 * x, 0.5 * y] for x, y in arr], RES, 1.0)).astype(np.complex64)
    if kind == "coherent":
        x = torch.cat([synth.synth_stream(torch, dev, B // 8, M, N, arr, synth.C_LIGHT, 0.5, snr_db=snr, seed=1003 + s) for s in range(8)], dim=0)
    else:
        x = synth.synth_scenes(torch, dev, B, M, N, arr, synth.C_LIGHT, 0.5, 2, snr_db=snr, seed=78)
    ang = torch.zeros(B, NE, dtype=torch.float32, device=dev)
    lvl = torch.zeros_like(ang)
    spec = torch.zeros(B, RES, dtype=torch.float32, device=dev)
    ref = None
    for label, exact in (("fp64 scan", "1"), ("int8 scan", "0")):
        os.environ["BAZ_MUSIC_EXACT"] = exact
        with capi.Context(M, NE, N, RES, table) as ctx:
            ctx.reserve(B)
            step = lambda: ctx.process_device(x.data_ptr(), B, ang.data_ptr(), lvl.data_ptr(), spec.data_ptr())
            for _ in range(5):
                step()
            ctx.sync()
            if exact == "0":
                ctx.debug_i8_stats()                      # (reset)
            t0 = time.perf_counter()
            for _ in range(10):
                step()
            ctx.sync()
            ms = (time.perf_counter() - t0) / 10 * 1e3
            fell = ctx.debug_i8_stats() if exact == "0" else (0, 0)
            ctx.profile(1)
            for _ in range(5):
                step()
            ctx.sync()
            st = [ctx.stage_ms(s) for s in range(capi.NUM_STAGES)]
            ctx.profile(False)
            got = (ang.clone(), spec[:256].clone())
        if ref is None:
            ref = got
        rel = float(((got[1].double() - ref[1].double()).abs() / ref[1].double()).max())
        same_bins = float((got[0] == ref[0]).float().mean())
        print("m%d n%d N%d res%d %6d items %-10s %2.0f dB %-10s: %.3f ms/step -> %.3e items/s | cov %.3f evd %.3f scan %.3f merge %.3f | refined tiles %s | vs fp64 scan: worst %.2e, same bins %.4f"
              % (M, NE, N, RES, B, kind, snr, label, ms, B / ms * 1e3, *[s[0] / max(1, s[1]) for s in st],
                 ("%.2f %%" % (100.0 * fell[0] / max(1, fell[1]))) if fell[1] else "-", rel, same_bins), flush=True)
    del x, spec
    torch.cuda.empty_cache()
