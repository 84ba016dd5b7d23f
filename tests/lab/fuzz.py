"""Randomised differential run: HIP path vs the C oracle over random (m, n, K, res, batch, snr, array) draws.
argv: number of cases [seed].  Prints every failure; exit code 1 if any."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gr_baz_amd import capi
from oracle import music_oracle as mo
from oracle import music_ref as mr
from helpers import assert_doa_match, assert_spectrum_close

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 12345)
dev = torch.device("cuda:0")
fails = 0
worst = 0.0
t0 = time.time()
for case in range(ncases):
    m = int(rng.integers(2, 17)) if rng.random() > 0.12 else int(rng.integers(17, 65))     # some wide arrays (run-time-m kernels)
    n = int(rng.integers(1, m))
    K = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 16, 31, 64, 100, 128, 256, 300]))
    res = int(rng.choice([1, 2, 3, 5, 63, 64, 65, 90, 127, 128, 129, 360, 361, 1000, 1440, 3600]))
    batch = int(rng.choice([1, 2, 15, 16, 17, 31, 33, 63, 64, 65, 100, 257]))
    snr = float(rng.choice([-5.0, 0.0, 10.0, 20.0, 40.0, 60.0]))
    N = m * K
    if rng.random() < 0.5:
        arr = mo.array_geometry(m)
    else:
        arr = (rng.random((m, 2)) * 3.0).tolist()
    nem = int(rng.integers(1, min(m, 5)))
    angles = tuple(float(a) for a in rng.uniform(0, 360, nem))
    try:
        table = mo.steering_table_c64(arr, res, mo.FREQUENCY, mo.SPACING)
        items = mo.synth_items(batch, m, N, arr, mo.FREQUENCY, mo.SPACING, angles_deg=angles, snr_db=snr, seed=int(rng.integers(1 << 30)))
        ao, lo, so = mr.work_batch(items, table, m, n)
        with capi.Context(m, n, N, res, table) as ctx:
            x = torch.from_numpy(items.view(np.float32)).to(dev)
            ang = torch.full((batch, n), -1.0, dtype=torch.float32, device=dev); lvl = torch.full_like(ang, -1.0)
            spec = torch.full((batch, res), -1.0, dtype=torch.float32, device=dev)
            ctx.process_device(x.data_ptr(), batch, ang.data_ptr(), lvl.data_ptr(), spec.data_ptr()); ctx.sync()
            a2 = torch.full_like(ang, -1.0)
            ctx.process_device(x.data_ptr(), batch, a2.data_ptr(), None, None); ctx.sync()
        sg = spec.cpu().numpy()
        fin = np.isfinite(so)
        # K < m: the covariance is rank deficient, the noise eigenvalues tie at ~0 and the reference's own answer depends
        # on LAPACK's choice inside the null space -> only well-posed cases are compared
        # n > emitters at high SNR: the n-th eigenvector is chosen among near-degenerate noise eigenvalues and the
        # reference's own spectrum differs by ~1e-5 between LAPACK and Jacobi (checked on the CPU) -> not compared
        if K >= m and not (n > nem and snr > 40.0):
            w = assert_spectrum_close(sg, so)
            worst = max(worst, w)
            assert_doa_match(ang.cpu().numpy(), lvl.cpu().numpy(), ao, lo, res, so.astype(np.float64))
            assert_doa_match(a2.cpu().numpy(), None, ao, lo, res, so.astype(np.float64))
        else:
            assert np.all(np.isfinite(sg) | ~fin)
    except AssertionError as e:
        fails += 1
        print("FAIL case %d: m=%d n=%d K=%d res=%d batch=%d snr=%g emitters=%d custom_array=%s: %s"
              % (case, m, n, K, res, batch, snr, nem, arr is not mo.array_geometry(m), str(e)[:300]), flush=True)
print("fuzz: %d cases, %d failures, worst spectrum rel err %.3g, %.1f s" % (ncases, fails, worst, time.time() - t0))
sys.exit(1 if fails else 0)
