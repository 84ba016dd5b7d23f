"""Lab: does the config-5 chain (resampler -> AGC -> MUSIC, 16 antennas) gain from running a step as C sub-batches back
to back on one stream, so that the intermediate streams (resampled samples, items) are still in the 256-MiB Infinity
Cache when the next engine reads them?  Timing only: every sub-batch restarts the resampler phase at 0 (the production
chain is one call per engine and step).  argv: items per step [steps]."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from gr_baz_amd import agc, capi, resamp, synth
from gr_baz_amd.baz.music_doa_helper import calculate_antenna_array_response

m, n, K, res = 16, 2, 256, 3600
N = m * K
nitems = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
ratio = 1.25
dev = torch.device("cuda:0")
arr = synth.array_geometry(m)
table = np.array(calculate_antenna_array_response([[0.5 * x, 0.5 * y] for x, y in arr], res, 1.0)).astype(np.complex64)
T_out = nitems * K
L = int(T_out * ratio) + 64
it = synth.synth_stream(torch, dev, (L + K - 1) // K, m, N, arr, 299792458.0, 0.5, seed=1005)
raw = torch.view_as_real(it.view(torch.complex64).reshape(-1, m).t().contiguous()[:, :L].contiguous()).reshape(m, 2 * L)
d_rs = torch.zeros(m, 2 * T_out, dtype=torch.float32, device=dev)
d_items = torch.zeros(nitems, 2 * N, dtype=torch.float32, device=dev)
ang = torch.zeros(nitems, n, dtype=torch.float32, device=dev); lvl = torch.zeros_like(ang)
spec = torch.zeros(nitems, res, dtype=torch.float32, device=dev)
R = resamp.Resampler(0.0, ratio, nstreams=m); A = agc.Agc(1e-4, 1.0, nstreams=m); M = capi.Context(m, n, N, res, table)
M.reserve(nitems)
st = torch.cuda.Stream(device=dev)
for e in (R, A, M): e.set_stream(st.cuda_stream)
for C in (1, 2, 4, 8, 16, 32):
    ni = nitems // C
    To = ni * K
    Lc = int(To * ratio) + 16

    def step():
        for c in range(C):
            R.set_mu(0.0)
            # (stream stride stays the whole capture's / output's: sub-batch c is a window of every antenna's row)
            R.process_device(raw.data_ptr() + c * int(To * ratio) * 8, L, Lc, d_rs.data_ptr() + c * To * 8, T_out, To)
            A.process_device_interleaved(d_rs.data_ptr() + c * To * 8, To, T_out, d_items.data_ptr() + c * ni * N * 8)
            M.process_device(d_items.data_ptr() + c * ni * N * 8, ni, ang.data_ptr() + c * ni * n * 4, lvl.data_ptr() + c * ni * n * 4,
                             spec.data_ptr() + c * ni * res * 4)

    t_ramp = time.perf_counter()
    while time.perf_counter() - t_ramp < 0.3:
        for _ in range(3): step()
        st.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): step()
    t_cpu = time.perf_counter() - t0
    st.synchronize()
    dt = (time.perf_counter() - t0) / steps
    foot = (m * Lc * 8 + 2 * m * To * 8 + ni * res * 4) / 2**20
    print("%2d sub-batches of %5d items (%.0f MiB touched each): %.3f ms/step -> %.3e items/s   (launch loop alone %.3f ms/step)"
          % (C, ni, foot, dt * 1e3, nitems / dt, t_cpu / steps * 1e3), flush=True)
for e in (R, A, M): e.set_stream(None)
R.close(); A.close(); M.close()
