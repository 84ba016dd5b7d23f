"""Lab: which items does the gated scan get wrong?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from gr_baz_amd import capi
from test_coarse_scan import _scene, _run_nospec
dev = torch.device("cuda:0")
m, n, N, res, B, snr = 4, 2, 1024, 3600, 333, 0.0
table, items = _scene(m, n, N, res, B, snr, 9000 + int(snr) + 13 * m + n, True)
outs = {}
for label, env in (("full", {"BAZ_MUSIC_COARSE": "0"}), ("gated", {"BAZ_MUSIC_COARSE": "1"}), ("gated nsplit1", {"BAZ_MUSIC_COARSE": "1", "BAZ_MUSIC_NSPLIT": "1"}),
                   ("gated lazy0", {"BAZ_MUSIC_COARSE": "1", "BAZ_MUSIC_COARSE_LAZY": "0"}), ("gated rg2", {"BAZ_MUSIC_COARSE": "1", "BAZ_MUSIC_COARSE_RG": "2"})):
    for k in ("BAZ_MUSIC_NSPLIT", "BAZ_MUSIC_COARSE_LAZY", "BAZ_MUSIC_COARSE_RG"):
        os.environ.pop(k, None)
    os.environ.update(env)
    with capi.Context(m, n, N, res, table) as ctx:
        outs[label] = _run_nospec(ctx, items, dev)
a0, l0 = outs["full"]
for label in outs:
    if label == "full":
        continue
    a, l = outs[label]
    bad = np.nonzero((a != a0).any(axis=1) | (l.view(np.uint32) != l0.view(np.uint32)).any(axis=1))[0]
    print("%-14s %d items differ: %s" % (label, len(bad), bad[:40].tolist()))
    for i in bad[:6]:
        print("    item %d: full ang %s lvl %s | gated ang %s lvl %s" % (i, a0[i].tolist(), l0[i].tolist(), a[i].tolist(), l[i].tolist()))
