"""Lab (GPU box, quick / lab library): the scan of config 2 on an incoherent batch at 60 dB (every wave's 16 items have their
nulls in different steps: the literal-form refinement fires in a third of the steps) under lab switches given as K=V arguments.
usage: BAZ_MUSIC_LAB_LIB=quick python tests/lab/refine_cliff.py [K=V ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from gr_baz_amd import capi, synth
from gr_baz_amd.baz.music_doa_helper import calculate_antenna_array_response

for kv in sys.argv[1:]:
    k, v = kv.split("=")
    os.environ[k] = v
dev = torch.device("cuda:0")
M, NE, N, RES, B = 4, 2, 1024, 3600, 262144
arr = synth.array_geometry(M)
table = np.array(calculate_antenna_array_response([[0.5 * x, 0.5 * y] for x, y in arr], RES, 1.0)).astype(np.complex64)
for snr, spec_on in ((60.0, True), (20.0, True), (60.0, False)):
    x = synth.synth_scenes(torch, dev, B, M, N, arr, synth.C_LIGHT, 0.5, 2, snr_db=snr, seed=1007)
    ang = torch.zeros(B, NE, dtype=torch.float32, device=dev)
    lvl = torch.zeros_like(ang)
    spec = torch.zeros(B, RES, dtype=torch.float32, device=dev) if spec_on else None
    with capi.Context(M, NE, N, RES, table, lab=True) as ctx:
        ctx.reserve(B)
        sp = spec.data_ptr() if spec_on else None
        step = lambda: ctx.process_device(x.data_ptr(), B, ang.data_ptr(), lvl.data_ptr(), sp)
        for _ in range(3):
            step()
        ctx.sync()
        ctx.profile(1)
        for _ in range(5):
            step()
        ctx.sync()
        st = [ctx.stage_ms(s) for s in range(capi.NUM_STAGES)]
        ctx.profile(False)
        step()
        ref = ctx.refined_values()
    print("incoherent %2.0f dB %s %s: cov+evd %.3f scan %.3f merge %.3f ms | values recomputed per step %d" % (
        snr, "spectrum wired" if spec_on else "ang/lvl only  ", " ".join(sys.argv[1:]), st[0][0] / st[0][1], st[2][0] / st[2][1], st[3][0] / st[3][1], ref), flush=True)
    del x, spec
    torch.cuda.empty_cache()
