"""Lab (GPU box, lab build): the gated scan (cfg2 without port 2, 262,144 items) with its items sorted by their nulls (BAZ_MUSIC_SORT=1), unsorted (0)
and under the adaptive policy (-1): coherent streams, an incoherent batch, both at 60 dB.  ms per step, scan ms, exact evaluations per (row group, tile)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from gr_baz_amd import capi, synth
from gr_baz_amd.baz.music_doa_helper import calculate_antenna_array_response

dev = torch.device("cuda:0")
CASES = [(4, 2, 1024, 3600, 262144), (8, 2, 4096, 36000, 16384)] if "all" in sys.argv else [(4, 2, 1024, 3600, 262144)]
for M, NE, N, RES, B in CASES:
    arr = synth.array_geometry(M)
    table = np.array(calculate_antenna_array_response([[0.5 * x, 0.5 * y] for x, y in arr], RES, 1.0)).astype(np.complex64)
    for scene, snr in (("coherent", 20.0), ("incoherent", 20.0), ("incoherent", 60.0)):
        if scene == "incoherent":
            x = synth.synth_scenes(torch, dev, B, M, N, arr, synth.C_LIGHT, 0.5, NE, snr_db=snr, seed=1007)
        else:
            x = torch.cat([synth.synth_stream(torch, dev, B // 8, M, N, arr, synth.C_LIGHT, 0.5, snr_db=snr, seed=1002 + s) for s in range(8)], dim=0)
        ang = torch.zeros(B, NE, dtype=torch.float32, device=dev)
        lvl = torch.zeros_like(ang)
        ref = None
        for label, mode in (("unsorted", "0"), ("sorted", "1"), ("adaptive", "-1")):
            os.environ["BAZ_MUSIC_SORT"] = mode
            with capi.Context(M, NE, N, RES, table, lab=True) as ctx:
                ctx.reserve(B)
                step = lambda: ctx.process_device(x.data_ptr(), B, ang.data_ptr(), lvl.data_ptr(), None)
                for _ in range(20):
                    step()
                ctx.sync()
                t0 = time.perf_counter()
                for _ in range(60):
                    step()
                ctx.sync()
                wall = (time.perf_counter() - t0) / 60 * 1e3
                ctx.profile(1)
                for _ in range(10):
                    step()
                ctx.sync()
                st = [ctx.stage_ms(s) for s in range(4)]
                ctx.profile(False)
                ss = ctx.debug_sort_state()
                got = (ang.clone(), lvl.clone())
            same = ""
            if ref is None:
                ref = got
            else:
                same = " | identical to the unsorted run: %s" % all(bool((a.view(torch.int32) == b.view(torch.int32)).all()) for a, b in zip(got, ref))
            print("m%d res%d %-10s %2.0f dB %-9s step %.3f ms = %.3e items/s | cov+evd %.3f scan(+sort) %.3f | exact per pair %.4f | sorted calls %d of %d%s"
                  % (M, RES, scene, snr, label, wall, B / wall * 1e3, st[0][0] / st[0][1], st[2][0] / st[2][1], ss["fired"] / max(ss["walked"], 1),
                     ss["sorted_calls"], ss["sorted_calls"] + ss["unsorted_calls"], same), flush=True)
        del x
        torch.cuda.empty_cache()
