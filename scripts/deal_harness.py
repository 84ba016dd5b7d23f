"""Config 4 inside ONE process, the way a GNU Radio flowgraph would run it: 64 music_doa block instances, each on its own
host thread (GNU Radio's thread-per-block scheduler), dealt over the visible gfx950 devices by the host block itself
(instance i -> device i mod G, baz_music_doa_deal_device) -- no torch.distributed, no launcher.  Every thread calls its
block's work() on host buffers (pybind releases the GIL), so G contexts per device run concurrently with their own
streams.  Prints one JSON line: items/s over all blocks, per-device instance counts, and whether every stream matched a
single-threaded run.  PCIe-inclusive (host buffers), not the headline metric.   argv: [blocks=64] [items per call=512] [calls=6]"""
import json
import os
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gr_baz_amd import synth
from gr_baz_amd.baz.music_doa_helper import calculate_antenna_array_response


def main():
    nblk = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    per_call = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    calls = int(sys.argv[3]) if len(sys.argv) > 3 else 6
    M, NE, N, RES = 4, 2, 1024, 3600
    os.environ.pop("BAZ_MUSIC_DEVICE", None)
    from gr_baz_amd import baz, capi
    G = capi.device_count()
    arr = synth.array_geometry(M)
    table = calculate_antenna_array_response([[0.5 * x, 0.5 * y] for x, y in arr], RES, 1.0)
    dev = torch.device("cuda:0")
    streams = []
    for s in range(nblk):        # stream s: seed 1000 + 2 + s
        x = synth.synth_stream(torch, dev, per_call, M, N, arr, synth.C_LIGHT, 0.5, seed=1002 + s)
        streams.append(np.ascontiguousarray(torch.view_as_complex(x.reshape(per_call, N, 2)).cpu().numpy()))
    blocks = [baz.music_doa(M, NE, N, table, RES) for _ in range(nblk)]
    devices = [b.device() for b in blocks]
    want = [b.work(streams[s], 3) for s, b in enumerate(blocks)]          # single-threaded reference (and warm-up)
    got = [None] * nblk
    start = threading.Barrier(nblk + 1)

    def worker(s):
        start.wait()
        for _ in range(calls):
            got[s] = blocks[s].work(streams[s], 3)

    th = [threading.Thread(target=worker, args=(s,)) for s in range(nblk)]
    for t in th:
        t.start()
    start.wait()
    t0 = time.perf_counter()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    same = all(g is not None and g[0] == w[0] and all(np.array_equal(a, b) for a, b in zip(g[1:], w[1:])) for g, w in zip(got, want))
    print(json.dumps({"blocks": nblk, "devices_visible": G, "instances_per_device": {str(d): devices.count(d) for d in sorted(set(devices))},
                      "dealing": "instance i -> device i mod G (round robin from wherever this process's counter stood)",
                      "items_per_call": per_call, "calls_per_block": calls, "seconds": dt,
                      "items_per_s_all_blocks": nblk * calls * per_call / dt, "streams_identical_to_single_threaded_run": bool(same)}), flush=True)


if __name__ == "__main__":
    main()
