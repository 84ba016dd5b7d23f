"""Lab (GPU box): page-locked host-fed config-2 calls with the spectrum port wired, one launch sequence per call
(BAZ_MUSIC_DUPLEX=0) against the call cut for both link directions (=1): items/s by items per call.  PCIe-inclusive:
NOT the headline metric.  argv: [call sizes = 256,384,512,1024,2048]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import music_oracle as mo
from gr_baz_amd import capi
sizes = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [256, 384, 512, 1024, 2048]
c = mo.make_config("cfg2", 256)
for nb in sizes:
    items = np.ascontiguousarray(np.tile(c["items"], ((nb + 255) // 256, 1))[:nb])
    ref = None
    for mode, dmin in (("0", None), ("1", None), ("1", "128"), ("0", None), ("1", None)):
        os.environ["BAZ_MUSIC_DUPLEX"] = mode
        os.environ.pop("BAZ_MUSIC_DUPLEX_MIN", None)
        if dmin: os.environ["BAZ_MUSIC_DUPLEX_MIN"] = dmin
        with capi.Context(c["m"], c["n"], c["nsamples"], c["res"], c["table"]) as ctx:
            out = (np.zeros((nb, c["n"]), np.float32), np.zeros((nb, c["n"]), np.float32), np.zeros((nb, c["res"]), np.float32))
            assert ctx.host_register(items) == 0 and ctx.host_register(out[2]) == 0
            for _ in range(5): ctx.process(items, out=out)
            t0 = time.perf_counter(); k = 0
            while time.perf_counter() - t0 < 0.4:
                ctx.process(items, out=out); k += 1
            dt = (time.perf_counter() - t0) / k
            if ref is None: ref = [x.copy() for x in out]
            same = all(np.array_equal(x, y) for x, y in zip(out, ref))
            ctx.host_unregister_all()
        print("%5d items per call, duplex %s%s: %.3f ms per call -> %.3e items/s (%.1f GB/s out, %.1f GB/s in)  identical: %s"
              % (nb, mode, " (sub-batches >= 128)" if dmin else "", dt * 1e3, nb / dt, nb * c["res"] * 4 / dt / 1e9, nb * c["nsamples"] * 8 / dt / 1e9, same), flush=True)
