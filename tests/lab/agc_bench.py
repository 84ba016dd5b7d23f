"""AGC throughput on the GPU box: 16 streams (config 5's antenna count), device resident."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gr_baz_amd import agc
from oracle import agc_ref as ar
dev = torch.device("cuda:0")
S = 16
for n in (1 << 16, 1 << 20, 1 << 24):
    x = torch.randn(S, n, 2, device=dev, dtype=torch.float32) * 1.5
    out = torch.empty_like(x)
    env = torch.empty(S, n, device=dev, dtype=torch.float32)
    mul = torch.empty_like(env)
    side = torch.cuda.Stream()      # torch's default stream handle is 0 == "own stream" for set_stream
    with agc.Agc(1e-4, 1.0, nstreams=S) as blk, torch.cuda.stream(side):
        blk.set_stream(side.cuda_stream)
        for want in (False, True):
            e, m = (env.data_ptr(), mul.data_ptr()) if want else (None, None)
            for _ in range(2):
                blk.process_device(x.data_ptr(), n, n, out.data_ptr(), e, m)
            torch.cuda.synchronize()
            t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
            reps = 10
            t0.record()
            for _ in range(reps):
                blk.process_device(x.data_ptr(), n, n, out.data_ptr(), e, m)
            t1.record(); torch.cuda.synchronize()
            ms = t0.elapsed_time(t1) / reps
            bytes_alg = S * n * (16 + (8 if want else 0))
            print("agc 16 streams x %9d samples, env/mul ports %-5s: %.3f ms  %.3e samples/s  %.2f TB/s algorithmic (%.0f%% of 8 TB/s; the kernels read the input twice)"
                  % (n, want, ms, S * n / (ms * 1e-3), bytes_alg / (ms * 1e-3) / 1e12, bytes_alg / (ms * 1e-3) / 8e12 * 100), flush=True)
xh = (np.random.default_rng(0).standard_normal(1 << 20) + 1j * np.random.default_rng(1).standard_normal(1 << 20)).astype(np.complex64)
t = time.perf_counter(); ar.Agc().work(xh); dt = time.perf_counter() - t
print("cpu oracle (oracle/agc_ref.c, 1 thread): %.3e samples/s" % ((1 << 20) / dt))
