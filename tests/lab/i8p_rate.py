"""Lab (GPU box, lab build): the level-packed int8 scan at the headline shape (cfg2: m4, N1024, res3600, 262,144 items, spectrum port
wired) against the fp64 scan of the same build, per-stage hipEvent times; BAZ_MUSIC_I8_ABL 1 = no spectrum stores, 2 = no tile
arithmetic (staging, barriers, stores), 3 = neither (timing only, wrong results).  Usage: i8p_rate.py [scene ...] (coherent incoherent snr60)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from gr_baz_amd import capi, synth
from gr_baz_amd.baz.music_doa_helper import calculate_antenna_array_response

dev = torch.device("cuda:0")
M, NE, N, RES, B = 4, 2, 1024, 3600, 262144
arr = synth.array_geometry(M)
table = np.array(calculate_antenna_array_response([[0.5 * x, 0.5 * y] for x, y in arr], RES, 1.0)).astype(np.complex64)
scenes = sys.argv[1:] or ["coherent"]
for scene in scenes:
    snr = 60.0 if "snr60" in scene else 20.0
    if scene.startswith("incoherent"):
        x = synth.synth_scenes(torch, dev, B, M, N, arr, synth.C_LIGHT, 0.5, NE, snr_db=snr, seed=1007)
    else:
        x = torch.cat([synth.synth_stream(torch, dev, B // 8, M, N, arr, synth.C_LIGHT, 0.5, snr_db=snr, seed=1002 + s) for s in range(8)], dim=0)
    ang = torch.zeros(B, NE, dtype=torch.float32, device=dev)
    lvl = torch.zeros_like(ang)
    spec = torch.zeros(B, RES, dtype=torch.float32, device=dev)
    ref = None
    for label, env in (("fp64 scan (BAZ_MUSIC_I8P=0)", {"BAZ_MUSIC_I8P": "0"}), ("packed int8 scan", {"BAZ_MUSIC_I8P": "1"}),
                       ("packed int8, no stores", {"BAZ_MUSIC_I8P": "1", "BAZ_MUSIC_I8_ABL": "1"}),
                       ("packed int8, stores + staging only", {"BAZ_MUSIC_I8P": "1", "BAZ_MUSIC_I8_ABL": "2"}),
                       ("packed int8, staging + barriers only", {"BAZ_MUSIC_I8P": "1", "BAZ_MUSIC_I8_ABL": "3"}),
                       ("packed int8, first pass alone", {"BAZ_MUSIC_I8P": "1", "BAZ_MUSIC_I8_ABL": "4"}),
                       ("packed int8, first pass alone, no stores", {"BAZ_MUSIC_I8P": "1", "BAZ_MUSIC_I8_ABL": "5"})):
        for k in ("BAZ_MUSIC_I8P", "BAZ_MUSIC_I8_ABL"):
            os.environ.pop(k, None)
        os.environ.update(env)
        with capi.Context(M, NE, N, RES, table, lab=True) as ctx:
            ctx.reserve(B)
            step = lambda: ctx.process_device(x.data_ptr(), B, ang.data_ptr(), lvl.data_ptr(), spec.data_ptr())
            for _ in range(20):
                step()
            ctx.sync()
            import time
            t0 = time.perf_counter()
            for _ in range(40):
                step()
            ctx.sync()
            wall = (time.perf_counter() - t0) / 40 * 1e3
            ctx.profile(1)
            for _ in range(10):
                step()
            ctx.sync()
            st = [ctx.stage_ms(s) for s in range(4)]
            ctx.profile(False)
            name = ctx.stage_name(capi.STAGE_SCAN)
            stats = ctx.debug_i8_stats() if "BAZ_MUSIC_I8_ABL" not in env and env["BAZ_MUSIC_I8P"] == "1" else None
            if stats:
                import ctypes
                v = (ctypes.c_uint64 * 4)()
                ctx._L.baz_music_debug_i8_times(ctx._h, v)
                stats = stats + (int(v[1]),)
            s_now = spec[::4099].clone()
            a_now = ang[::4099].clone()
        note = ""
        if "BAZ_MUSIC_I8_ABL" not in env:
            if ref is None:
                ref = (s_now, a_now)
            else:
                rel = ((s_now - ref[0]).abs() / ref[0]).max().item()
                note = " | vs fp64 scan: worst rel %.3g, bins equal %s" % (rel, bool((a_now == ref[1]).all()))
        print("%s %-38s step %.3f ms | cov+evd %.3f scan %.3f merge %.4f | %s%s%s" % (
            scene, label, wall, st[0][0] / st[0][1], st[2][0] / st[2][1], st[3][0] / st[3][1], name,
            " | refined tiles %d of %d, flagged %d" % stats if stats else "", note), flush=True)
    del x, spec
    torch.cuda.empty_cache()
