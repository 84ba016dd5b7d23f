"""Lab (GPU box, lab build): scan_mfma_kernel's staging, cfg2 with port 2, 262,144 items: the register staging of the product (variant 0) against
a rotating LDS-DMA loader (BAZ_MUSIC_SCAN_VARIANT=9), both at 3 waves per SIMD (11, 10).  Outputs must be identical."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from gr_baz_amd import capi, synth
from gr_baz_amd.baz.music_doa_helper import calculate_antenna_array_response

dev = torch.device("cuda:0")
M, NE, N, RES, B = 4, 2, 1024, 3600, 262144
arr = synth.array_geometry(M)
table = np.array(calculate_antenna_array_response([[0.5 * x, 0.5 * y] for x, y in arr], RES, 1.0)).astype(np.complex64)
NAMES = {6: "everything but the stores", 7: "stores + staging only", 13: "stores + staging only, compact order", 12: "blocks in the compact order", 0: "the product (class by class)", 9: "rotating LDS-DMA loader", 10: "rotating loader, 3 waves/SIMD", 11: "register staging, 3 waves/SIMD"}
for scene, snr in (("coherent", 20.0), ("incoherent", 20.0)):
    if scene == "incoherent":
        x = synth.synth_scenes(torch, dev, B, M, N, arr, synth.C_LIGHT, 0.5, NE, snr_db=snr, seed=1007)
    else:
        x = torch.cat([synth.synth_stream(torch, dev, B // 8, M, N, arr, synth.C_LIGHT, 0.5, snr_db=snr, seed=1002 + s) for s in range(8)], dim=0)
    ang = torch.zeros(B, NE, dtype=torch.float32, device=dev)
    lvl = torch.zeros_like(ang)
    spec = torch.zeros(B, RES, dtype=torch.float32, device=dev)
    ref = None
    for rep in range(2):
        for var in [int(v) for v in os.environ.get('AB_VARIANTS', '0,9,11,10').split(',')]:
            os.environ["BAZ_MUSIC_SCAN_VARIANT"] = str(var)
            with capi.Context(M, NE, N, RES, table, lab=True) as ctx:
                ctx.reserve(B)
                step = lambda: ctx.process_device(x.data_ptr(), B, ang.data_ptr(), lvl.data_ptr(), spec.data_ptr())
                for _ in range(20):
                    step()
                ctx.sync()
                t0 = time.perf_counter()
                for _ in range(40):
                    step()
                ctx.sync()
                wall = (time.perf_counter() - t0) / 40 * 1e3
                ctx.profile(2)
                for _ in range(10):
                    step()
                ctx.sync()
                sm, sn = ctx.stage_ms(capi.STAGE_SCAN)
                ctx.profile(False)
                got = (ang.clone(), lvl.clone(), spec[::977].clone())
            same = ""
            if ref is None:
                ref = got
            else:
                same = " | identical to the first run: %s" % all(bool((a.view(torch.int32) == b.view(torch.int32)).all()) for a, b in zip(got, ref))
            print("%-10s %2.0f dB %-36s step %.3f ms scan %.3f ms%s" % (scene, snr, NAMES[var], wall, sm / sn, same), flush=True)
    del x, spec
    torch.cuda.empty_cache()
