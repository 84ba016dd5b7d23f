"""Lab: where does the coarse form disagree with the exact one?  Dumps error / allowance for every (item, bin) of a small
batch and prints the pattern of the bad values."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from gr_baz_amd import capi
from oracle import music_oracle as mo
os.environ["BAZ_MUSIC_DEBUG_DUMP"] = "/tmp/coarse_dump.bin"
os.environ["BAZ_MUSIC_DEBUG_MARGIN"] = "1"
dev = torch.device("cuda:0")
for (m, n, N, res, B) in ((4, 2, 1024, 3600, 512), (3, 1, 96, 721, 512), (2, 1, 64, 90, 256)):
    arr = mo.array_geometry(m) if m >= 3 else [[0.0, 0.0], [1.0, 0.0]]
    table = mo.steering_table_c64(arr, res, mo.FREQUENCY, mo.SPACING)
    items = mo.synth_items(B, m, N, arr, mo.FREQUENCY, mo.SPACING, angles_deg=(40.3, 121.7)[:n], snr_db=20.0, seed=5)
    x = torch.from_numpy(items.view(np.float32)).to(dev)
    torch.cuda.synchronize()
    with capi.Context(m, n, N, res, table) as ctx:
        w = ctx.debug_coarse_margin(x.data_ptr(), B)
    r = np.fromfile("/tmp/coarse_dump.bin", dtype=np.float32).reshape(B, res)
    bad = np.argwhere(r > 0.5)
    print("m%d n%d res%d: worst %.4g, %d of %d values above 0.5; median %.3g" % (m, n, res, w, len(bad), r.size, np.median(r)))
    if len(bad):
        it, bn = bad[:, 0], bad[:, 1]
        print("  items: %d distinct, first %s ; item %% 16 histogram %s ; (item %% 64) // 16 histogram %s" %
              (len(set(it)), sorted(set(it))[:12], np.bincount(it % 16, minlength=16).tolist(), np.bincount((it % 64) // 16, minlength=4).tolist()))
        print("  bins: %d distinct, first %s ; bin %% 16 histogram %s ; tile %% 8 histogram %s" %
              (len(set(bn)), sorted(set(bn))[:12], np.bincount(bn % 16, minlength=16).tolist(), np.bincount((bn // 16) % 8, minlength=8).tolist()))
        print("  sample (item, bin, ratio):", [(int(a), int(b), float(r[a, b])) for a, b in bad[:8]])
