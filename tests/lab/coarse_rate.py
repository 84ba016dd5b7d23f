"""Lab (GPU box): the coarse-gated scan against the full fp64 scan, config 2 WITHOUT the spectrum port, 262,144 items:
coherent streams (bench.py's inputs) and an incoherent batch (angles drawn per item), 2 / 4 row groups per wave,
SNR 20 / 60 dB.  Prints per-stage ms, items/s, identical-output checks and the error-bound margin.
argv: [items=262144]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from gr_baz_amd import capi, synth
from gr_baz_amd.baz.music_doa_helper import calculate_antenna_array_response

M, NE, N, RES = 4, 2, 1024, 3600
B = int(sys.argv[1]) if len(sys.argv) > 1 else 262144
dev = torch.device("cuda:0")
arr = synth.array_geometry(M)
table = np.array(calculate_antenna_array_response([[0.5 * x, 0.5 * y] for x, y in arr], RES, 1.0)).astype(np.complex64)


def inputs(kind, snr):
    if kind == "coherent":
        return torch.cat([synth.synth_stream(torch, dev, B // 8, M, N, arr, synth.C_LIGHT, 0.5, snr_db=snr, seed=1002 + s) for s in range(8)], dim=0)
    return synth.synth_scenes(torch, dev, B, M, N, arr, synth.C_LIGHT, 0.5, NE, snr_db=snr, seed=77)


ang = torch.zeros(B, NE, dtype=torch.float32, device=dev)
lvl = torch.zeros_like(ang)
for kind in ("coherent", "incoherent"):
    for snr in ((20.0, 60.0) if len(sys.argv) < 3 else (20.0,)):
        x = inputs(kind, snr)
        ref = None
        for label, env in (("full fp64 scan", {"BAZ_MUSIC_COARSE": "0"}), ("gated, 4 row groups", {"BAZ_MUSIC_COARSE": "1", "BAZ_MUSIC_COARSE_RG": "4"}),
                           ("gated, 2 row groups", {"BAZ_MUSIC_COARSE": "1", "BAZ_MUSIC_COARSE_RG": "2"}),
                           ("coarse passes only (lab)", {"BAZ_MUSIC_COARSE": "1", "BAZ_MUSIC_COARSE_RG": "4", "BAZ_MUSIC_COARSE_LAB": "1"}),
                           ("... without staging X (lab)", {"BAZ_MUSIC_COARSE": "1", "BAZ_MUSIC_COARSE_RG": "4", "BAZ_MUSIC_COARSE_LAB": "2"})):
            os.environ["BAZ_MUSIC_COARSE_LAB"] = "0"
            os.environ["BAZ_MUSIC_COARSE_STATS"] = "1"
            os.environ.update(env)
            with capi.Context(M, NE, N, RES, table) as ctx:
                ctx.reserve(B)
                step = lambda: ctx.process_device(x.data_ptr(), B, ang.data_ptr(), lvl.data_ptr(), None)
                for _ in range(30):
                    step()
                ctx.sync()
                ws = []
                for _ in range(5):
                    t0 = time.perf_counter()
                    for _ in range(20):
                        step()
                    ctx.sync()
                    ws.append((time.perf_counter() - t0) / 20 * 1e3)
                ctx.profile(1)
                for _ in range(10):
                    step()
                ctx.sync()
                st = [ctx.stage_ms(s) for s in range(capi.NUM_STAGES)]
                ctx.profile(False)
                got = (ang.clone(), lvl.clone())
                refined = ctx.refined_values()
                ctx.debug_coarse_fired()
                step()
                fired = ctx.debug_coarse_fired()
                margin = ctx.debug_coarse_margin(x.data_ptr(), min(B, 65536)) if env["BAZ_MUSIC_COARSE"] == "1" else None
            if ref is None:
                ref = got
            same = bool(torch.equal(got[0], ref[0]) and torch.equal(got[1].view(torch.int32), ref[1].view(torch.int32)))
            ms = sorted(ws)[2]
            print("%-10s %2.0f dB  %-20s step %.4f ms (min %.4f) = %.3e items/s = %.1f %% of the HBM-read roofline | cov+evd %.4f scan %.4f merge %.4f | "
                  "bit-identical to the full scan: %s | refined values %d | exact (row group, tile) evaluations %d = %.1f %% of all%s"
                  % (kind, snr, label, ms, min(ws), B / ms * 1e3, B / ms * 1e3 * 8192 / 8e12 * 100, st[0][0] / st[0][1] + (st[1][0] / st[1][1] if st[1][1] else 0),
                     st[2][0] / st[2][1], st[3][0] / st[3][1], same, refined, fired, 100.0 * fired / (B / 16 * 225), "" if margin is None else " | error/allowance worst %.3f" % margin), flush=True)
        del x
