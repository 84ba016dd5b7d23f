"""Lab (GPU box, lab build of the library): where the int8 scan's time goes.  BASELINE configs[2] (m8, N4096, res36000; 16,384
items) and configs[4]'s MUSIC stage (m16, N4096, res3600; 16,384 items), spectrum port wired, with parts of the bulk loop
compiled out (BAZ_MUSIC_I8_ABL: 1 no spectrum stores, 4 no MFMAs, 8 no staging loads / waits / barriers; timing only, results are wrong)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from gr_baz_amd import capi, synth
from gr_baz_amd.baz.music_doa_helper import calculate_antenna_array_response

dev = torch.device("cuda:0")
NAMES = {0: "product", 1: "no stores", 4: "no MFMAs", 5: "no MFMAs, no stores", 9: "first phase staged only, no stores: the tiles' arithmetic alone",
         32: "staging + barriers + stores, no arithmetic", 33: "staging + barriers alone", 256: "steps left to right (round 4's walk)",
         257: "steps left to right, no stores"}
ABLS = [int(v) for v in sys.argv[1:]] or [0, 1, 256, 257, 32]
for M, NE, N, RES, B in ((8, 2, 4096, 36000, 16384), (16, 2, 4096, 3600, 16384)):
    arr = synth.array_geometry(M)
    table = np.array(calculate_antenna_array_response([[0.5 * x, 0.5 * y] for x, y in arr], RES, 1.0)).astype(np.complex64)
    x = torch.cat([synth.synth_stream(torch, dev, B // 8, M, N, arr, synth.C_LIGHT, 0.5, snr_db=20.0, seed=1003 + s) for s in range(8)], dim=0)
    ang = torch.zeros(B, NE, dtype=torch.float32, device=dev)
    lvl = torch.zeros_like(ang)
    spec = torch.zeros(B, RES, dtype=torch.float32, device=dev)
    for abl in ABLS:
        os.environ["BAZ_MUSIC_I8_ABL"] = str(abl)
        with capi.Context(M, NE, N, RES, table, lab=True) as ctx:
            ctx.reserve(B)
            step = lambda: ctx.process_device(x.data_ptr(), B, ang.data_ptr(), lvl.data_ptr(), spec.data_ptr())
            for _ in range(5):
                step()
            ctx.sync()
            ctx.profile(2)
            for _ in range(10):
                step()
            ctx.sync()
            t, k = ctx.stage_ms(capi.STAGE_SCAN)
            ctx.profile(False)
            extra = ""
            if abl & 8192:
                import ctypes
                v = (ctypes.c_uint64 * 4)()
                ctx._L.baz_music_debug_i8_times(ctx._h, v)
                tot = max(1, v[3])
                extra = " | per wave: wait %.1f %%, stores %.1f %%, barrier %.1f %% of the step loop (%.0f cycles per step and wave)" % (
                    100.0 * v[0] / tot, 100.0 * v[1] / tot, 100.0 * v[2] / tot, v[3] / (15.0 * (B / 16.0) * ((RES + 63) // 64)))
        print("m%d res%d %d items: scan %.3f ms  (%s)%s" % (M, RES, B, t / k, NAMES.get(abl, str(abl)), extra), flush=True)
    del x, spec
    torch.cuda.empty_cache()
