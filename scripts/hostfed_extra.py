"""Host-fed secondary measurement for bench.py (config.extra.cfg2_host_fed_gr37_model; PCIe-inclusive, NOT the metric):
the MUSIC host block between a saturating source and draining sinks, driven as gnuradio-runtime 3.7 would drive it with the
block's DEFAULT scheduler hints (gr_baz_amd/host/gr_shim/gnuradio/flowgraph_model.h), config 2, stream buffers pageable and
page-locked, with and without the spectrum port.  Run in its own process by bench.py (a hang or crash here cannot take the
headline down); prints one JSON object.  argv: [items=8192]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from gr_baz_amd import synth
from gr_baz_amd.baz.music_doa_helper import calculate_antenna_array_response

M, NE, N, RES = 4, 2, 1024, 3600
ITEMS = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
for k in ("BAZ_MUSIC_OUTPUT_MULTIPLE", "BAZ_MUSIC_MIN_OUTPUT_BUFFER", "BAZ_MUSIC_MAX_NOUTPUT"):
    os.environ.pop(k, None)                      # the block's defaults are what is measured
dev = torch.device("cuda:0")
arr = synth.array_geometry(M)
table = calculate_antenna_array_response([[0.5 * x, 0.5 * y] for x, y in arr], RES, 1.0)
x = synth.synth_stream(torch, dev, ITEMS, M, N, arr, synth.C_LIGHT, 0.5, seed=1002)
items = np.ascontiguousarray(torch.view_as_complex(x.reshape(ITEMS, N, 2)).cpu().numpy())
del x
from gr_baz_amd import baz

blk = baz.music_doa(M, NE, N, table, RES)
out = {"workload": "cfg2 items through baz.music_doa's work(), host buffers, default scheduler hints (output multiple %d, "
                   "min output buffer %d); call sizes from the restated GNU Radio 3.7 buffer sizing / executor; 4 passes over %d "
                   "items, rates of passes 2-4; PCIe-inclusive, not the headline metric"
                   % (blk.output_multiple(), blk.min_output_buffer(), ITEMS),
       "runs": {}}
for n_outputs in (3, 2):
    for pin in (False, True):
        st, _, _, _ = blk.run_flowgraph(items, n_outputs, False, pin, 4)
        key = "%s_%s" % ("with_spectrum_port" if n_outputs == 3 else "ang_lvl_only", "page_locked_buffers" if pin else "pageable_buffers")
        if st["last_return"] < 0 or not st["steady_items"]:
            out["runs"][key] = {"error": "work() returned %d after %d items" % (st["last_return"], st["items"])}
            continue
        out["runs"][key] = {"items_per_s": st["steady_items"] / st["steady_work_seconds"],
                            "items_per_work_call": max(st["call_sizes"], key=lambda k: st["call_sizes"][k]),
                            "work_calls": st["calls"], "input_buffer_items": st["in_bufsize"],
                            "output_buffer_items": list(st["out_bufsize"]),
                            "page_locked_MiB": st["pinned_bytes_at_stop"] / 2.0**20}
print(json.dumps(out), flush=True)
