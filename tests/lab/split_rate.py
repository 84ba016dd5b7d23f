"""Lab (GPU box): default wiring (port 2 not wired), cfg2: ms per step of the single launch sequence (BAZ_MUSIC_SPLIT=0) against the split
pipeline -- covariance + EVD of part p + 1 beside the gated scan of part p -- by number of parts and by the covariance's grid while a scan
runs beside it (BAZ_MUSIC_SPLIT_COV_BLOCKS_PER_CU, lab build).  Device-resident inputs; ang compared with the unsplit run's (bit for bit).
usage: split_rate.py [batch ...]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from gr_baz_amd import capi, synth
from oracle import music_oracle as mo

batches = [int(v) for v in sys.argv[1:]] or [262144, 65536, 32768]
dev = torch.device("cuda:0")
arr = synth.array_geometry(4)
table = mo.make_config("cfg2", 1)["table"]
stream = torch.cuda.Stream(device=dev)
for B in batches:
    for scene, snr in (("coherent", 20.0), ("incoherent", 20.0), ("incoherent", 60.0)):
        if scene == "incoherent":
            x = synth.synth_scenes(torch, dev, B, 4, 1024, arr, mo.FREQUENCY, mo.SPACING, 2, snr_db=snr, seed=1007)
        else:
            x = torch.cat([synth.synth_stream(torch, dev, B // 8, 4, 1024, arr, mo.FREQUENCY, mo.SPACING, snr_db=snr, seed=1003 + s) for s in range(8)], dim=0)
        ang = torch.zeros(B, 2, dtype=torch.float32, device=dev)
        lvl = torch.zeros_like(ang)
        torch.cuda.synchronize()
        ref = None
        line = "%7d items %-10s %2.0f dB:" % (B, scene, snr)
        for split, per_cu in ((0, 0), (2, 1), (4, 1), (8, 1), (4, 2), (2, 2)):
            os.environ["BAZ_MUSIC_SPLIT"] = str(split)
            os.environ.pop("BAZ_MUSIC_SPLIT_COV_BLOCKS_PER_CU", None)
            if per_cu:
                os.environ["BAZ_MUSIC_SPLIT_COV_BLOCKS_PER_CU"] = str(per_cu)
            with capi.Context(4, 2, 1024, 3600, table, lab=True) as ctx:
                ctx.set_stream(stream.cuda_stream)
                ctx.reserve(B)
                step = lambda: ctx.process_device(x.data_ptr(), B, ang.data_ptr(), lvl.data_ptr(), None)
                for _ in range(20):
                    step()
                stream.synchronize()
                n, t0 = 0, time.perf_counter()
                while time.perf_counter() - t0 < 0.4:
                    for _ in range(10):
                        step()
                    stream.synchronize()
                    n += 10
                ms = (time.perf_counter() - t0) / n * 1e3
                got = ang.cpu().numpy()
                if ref is None:
                    ref = got
                same = np.array_equal(got, ref)
                ctx.set_stream(None)
            line += "  [%d parts, %s/CU] %.4f ms %.3ge8/s%s" % (split, per_cu or "-", ms, B / ms * 1e3 / 1e8, "" if same else " DIFFERENT")
        print(line, flush=True)
os.environ.pop("BAZ_MUSIC_SPLIT", None)
os.environ.pop("BAZ_MUSIC_SPLIT_COV_BLOCKS_PER_CU", None)
