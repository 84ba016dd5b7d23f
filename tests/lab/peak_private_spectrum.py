"""Lab (GPU box; run it under BAZ_MUSIC_LAB_LIB=lab BAZ_MUSIC_GUARD=1 as well): peak mode WITHOUT the spectrum port -- the context keeps a private spectrum (dPeakSpec) that the
scan writes and the peak picker reads.  Compares ang / lvl with the same call given a caller-owned spectrum tensor, first call and second call, device-resident and host-fed."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from gr_baz_amd import capi
from oracle import music_oracle as mo

dev = torch.device("cuda:0")
for (m, n, K, res, batch, snr) in [(4, 3, 16, 64, 1000, 70.0), (4, 2, 16, 64, 1000, 20.0), (4, 3, 16, 360, 1000, 70.0), (8, 2, 64, 1000, 640, 20.0)]:
    arr = mo.array_geometry(m)
    table = mo.steering_table_c64(arr, res, mo.FREQUENCY, mo.SPACING)
    rng = np.random.default_rng(31)
    items = mo.synth_items(batch, m, m * K, arr, mo.FREQUENCY, mo.SPACING, angles_deg=tuple(rng.uniform(0, 360, n)), snr_db=snr, seed=77)
    with capi.Context(m, n, m * K, res, table) as ctx:
        ctx.set_peak_mode(1)
        x = torch.from_numpy(items.view(np.float32)).to(dev)
        mk = lambda: (torch.zeros(batch, n, dtype=torch.float32, device=dev), torch.zeros(batch, n, dtype=torch.float32, device=dev))
        a0, l0 = mk()
        spec = torch.zeros(batch, res, dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        ctx.process_device(x.data_ptr(), batch, a0.data_ptr(), l0.data_ptr(), spec.data_ptr()); ctx.sync()
        ref = a0.cpu().numpy()
        out = []
        for rep in range(3):
            a, l = mk()
            torch.cuda.synchronize()
            ctx.process_device(x.data_ptr(), batch, a.data_ptr(), l.data_ptr(), None); ctx.sync()
            d = a.cpu().numpy() != ref
            out.append("device private #%d: %d rows differ %s" % (rep + 1, int(d.any(axis=1).sum()), np.flatnonzero(d.any(axis=1))[:6].tolist()))
        pin = lambda z: torch.from_numpy(np.ascontiguousarray(z)).pin_memory().numpy()
        for rep in range(2):
            o = (pin(np.zeros((batch, n), np.float32)), pin(np.zeros((batch, n), np.float32)), None)
            ha, hl, _ = ctx.process(pin(items.view(np.float32)).view(np.complex64), out=o)
            d = ha != ref
            out.append("host pinned #%d: %d rows differ %s" % (rep + 1, int(d.any(axis=1).sum()), np.flatnonzero(d.any(axis=1))[:6].tolist()))
        ha, hl, _ = ctx.process(items, want_lvl=True, want_spectrum=False)
        d = ha != ref
        out.append("host pageable: %d rows differ" % int(d.any(axis=1).sum()))
    print("m=%d n=%d K=%d res=%d batch=%d snr=%g | " % (m, n, K, res, batch, snr) + " | ".join(out), flush=True)
