"""BASELINE config 5 on one GPU: 16-antenna wideband front-end (fractional resampler -> AGC, interleaving) fused
ahead of MUSIC-DoA (m16, n2, N4096 = 16 x 256, res3600), everything device resident on ONE stream.
argv: items per step [steps]."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gr_baz_amd import agc, capi, resamp, synth
from gr_baz_amd.baz.music_doa_helper import calculate_antenna_array_response

m, n, K, res = 16, 2, 256, 3600
N = m * K
nitems = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ratio = 1.25
dev = torch.device("cuda:0")
arr = synth.array_geometry(m)
lam = synth.C_LIGHT / 299792458.0
table = np.array(calculate_antenna_array_response([[0.5 * x, 0.5 * y] for x, y in arr], res, lam)).astype(np.complex64)
T_out = nitems * K
L = int(T_out * ratio) + 16
# planar capture: m streams of L samples (array snapshot generator, de-interleaved on the device)
it = synth.synth_stream(torch, dev, (L + K - 1) // K, m, N, arr, 299792458.0, 0.5, seed=1005)     # [items][N] complex64
raw = torch.view_as_real(it.view(torch.complex64).reshape(-1, m).t().contiguous()[:, :L].contiguous()).reshape(m, 2 * L)
d_rs = torch.zeros(m, 2 * T_out, dtype=torch.float32, device=dev)
d_items = torch.zeros(nitems, 2 * N, dtype=torch.float32, device=dev)
ang = torch.zeros(nitems, n, dtype=torch.float32, device=dev); lvl = torch.zeros_like(ang)
spec = torch.zeros(nitems, res, dtype=torch.float32, device=dev)
R = resamp.Resampler(0.0, ratio, nstreams=m); A = agc.Agc(1e-4, 1.0, nstreams=m); M = capi.Context(m, n, N, res, table)
M.reserve(nitems)
st = torch.cuda.Stream(device=dev)
for e in (R, A, M): e.set_stream(st.cuda_stream)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]

def step(timed=False):
    R.set_mu(0.0)                                    # every step resamples the same capture from its start
    if timed: ev[0].record(st)
    p, c = R.process_device(raw.data_ptr(), L, L, d_rs.data_ptr(), T_out, T_out)
    if timed: ev[1].record(st)
    A.process_device_interleaved(d_rs.data_ptr(), T_out, T_out, d_items.data_ptr())
    if timed: ev[2].record(st)
    M.process_device(d_items.data_ptr(), nitems, ang.data_ptr(), lvl.data_ptr(), spec.data_ptr())
    if timed: ev[3].record(st)
    return p

t_ramp = time.perf_counter()
while time.perf_counter() - t_ramp < 0.3:
    for _ in range(5): step()
    st.synchronize()
t0 = time.perf_counter()
for _ in range(steps): p = step()
st.synchronize()
dt = (time.perf_counter() - t0) / steps
step(True); st.synchronize()
ms = [ev[i].elapsed_time(ev[i + 1]) for i in range(3)]
M.profile(1); step(); st.synchronize()
stages = [M.stage_ms(s)[0] for s in range(4)]; M.profile(0)
in_bytes = m * L * 8
print("cfg5 fused pipeline: %d items/step (%d antennas x %d output samples, ratio %.2f): %.3f ms/step -> %.3e items/s, "
      "%.3e complex samples/s per antenna" % (nitems, m, T_out, ratio, dt * 1e3, nitems / dt, T_out / dt))
print("  stages (hipEvents, one step): resampler %.3f ms (%.0f GB/s), agc+interleave %.3f ms (%.0f GB/s), music %.3f ms "
      "[cov %.3f evd %.3f scan %.3f merge %.3f]" % (ms[0], (in_bytes + m * T_out * 8) / ms[0] / 1e6, ms[1],
                                                     (3 * m * T_out * 8) / ms[1] / 1e6, ms[2], *stages))
print("  produced %d of %d outputs per antenna; first DoA pairs %s" % (p, T_out, ang[:2].cpu().numpy().tolist()))
for e in (R, A, M): e.set_stream(None)
R.close(); A.close(); M.close()
