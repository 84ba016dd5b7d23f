"""Lab (GPU box): does the scan write EVERY spectrum value?  The output tensor is pre-filled with a sentinel; after process_device no element may still hold it.
usage: spectrum_fully_written.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from gr_baz_amd import capi
from oracle import music_oracle as mo

dev = torch.device("cuda:0")
SENT = 12345.678
bad = 0
for (m, n, K, res, batch, snr) in [(4, 3, 16, 64, 1000, 70.0), (4, 3, 16, 64, 1000, 20.0), (4, 2, 256, 3600, 300, 80.0), (4, 2, 16, 64, 1000, 70.0), (3, 2, 16, 64, 1000, 70.0),
                                   (4, 3, 16, 64, 64, 70.0), (4, 3, 16, 128, 1000, 70.0), (4, 3, 7, 1440, 257, 60.0), (8, 2, 64, 1000, 64, 70.0), (5, 4, 40, 361, 100, 90.0)]:
    arr = mo.array_geometry(m)
    table = mo.steering_table_c64(arr, res, mo.FREQUENCY, mo.SPACING)
    rng = np.random.default_rng(31)
    items = mo.synth_items(batch, m, m * K, arr, mo.FREQUENCY, mo.SPACING, angles_deg=tuple(rng.uniform(0, 360, n)), snr_db=snr, seed=77)
    with capi.Context(m, n, m * K, res, table) as ctx:
        x = torch.from_numpy(items.view(np.float32)).to(dev)
        ang = torch.zeros(batch, n, dtype=torch.float32, device=dev)
        lvl = torch.zeros_like(ang)
        spec = torch.full((batch, res), SENT, dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        ctx.process_device(x.data_ptr(), batch, ang.data_ptr(), lvl.data_ptr(), spec.data_ptr())
        ctx.sync()
        left = (spec == SENT)
        nleft = int(left.sum())
        rows = torch.nonzero(left.any(dim=1)).flatten().cpu().numpy()
        nan = int(torch.isnan(spec).sum())
        print("m=%d n=%d K=%d res=%d batch=%d snr=%g: %d values NOT written (rows %s...), %d NaN, refined %d" % (m, n, K, res, batch, snr, nleft, rows[:8].tolist(), nan, ctx.refined_values()), flush=True)
        if nleft:
            r0 = int(rows[0])
            print("   row %d unwritten bins: %s" % (r0, torch.nonzero(left[r0]).flatten().cpu().numpy()[:16].tolist()))
        bad += nleft
print("TOTAL unwritten:", bad)
