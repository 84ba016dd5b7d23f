"""Randomised differential run of the matrix-core kernels for wide arrays (17 <= m <= 32: cov_wide_mfma_kernel, 33 <= m <= 64:
cov_wide_pairs_kernel; n <= 8: scan_wide_mfma_kernel) against the C oracle AND against the vector-unit kernels they replace (BAZ_MUSIC_WIDE_MFMA=0
BAZ_MUSIC_WIDE_COV_MFMA=0).  argv: number of cases [seed].  Prints every failure; exit code 1 if any."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from gr_baz_amd import capi
from oracle import music_oracle as mo
from oracle import music_ref as mr
from helpers import assert_doa_match, assert_spectrum_close

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 4242)
dev = torch.device("cuda:0")
fails = 0
worst = 0.0
worst_ab = 0.0
t0 = time.time()


def run(m, n, N, res, table, items, mfma):
    os.environ["BAZ_MUSIC_WIDE_MFMA"] = mfma
    os.environ["BAZ_MUSIC_WIDE_COV_MFMA"] = mfma
    batch = len(items)
    with capi.Context(m, n, N, res, table, lab=True) as ctx:          # (the lab form of the library: the release form does not read the switches)
        x = torch.from_numpy(items.view(np.float32)).to(dev)
        ang = torch.full((batch, n), -1.0, dtype=torch.float32, device=dev); lvl = torch.full_like(ang, -1.0)
        spec = torch.full((batch, res), -1.0, dtype=torch.float32, device=dev)
        ctx.process_device(x.data_ptr(), batch, ang.data_ptr(), lvl.data_ptr(), spec.data_ptr()); ctx.sync()
        a2 = torch.full_like(ang, -1.0)
        ctx.process_device(x.data_ptr(), batch, a2.data_ptr(), None, None); ctx.sync()
        names = (ctx.stage_name(0), ctx.stage_name(2))
    return ang.cpu().numpy(), lvl.cpu().numpy(), spec.cpu().numpy(), a2.cpu().numpy(), names


for case in range(ncases):
    m = int(rng.integers(17, 33)) if rng.random() < 0.4 else int(rng.integers(33, 65))
    n = int(rng.choice([1, 2, 2, 2, 3, 4, 5, 6, 7, 8, 11]))
    K = int(rng.choice([m, m + 1, 33, 40, 47, 64, 96, 100, 128, 131]))
    K = max(K, m)
    res = int(rng.choice([1, 3, 5, 63, 64, 65, 90, 127, 128, 129, 360, 361, 1000, 1440, 3600]))
    batch = int(rng.choice([1, 2, 3, 4, 5, 15, 16, 17, 31, 33, 63, 64, 65, 100] if m <= 32 else [1, 2, 3, 5, 15, 16, 17, 33]))
    snr = float(rng.choice([0.0, 10.0, 20.0, 40.0, 70.0]))
    N = m * K
    arr = mo.array_geometry(m) if rng.random() < 0.5 else (rng.random((m, 2)) * 4.0).tolist()
    nem = n if rng.random() < 0.7 else int(rng.integers(1, 4))
    angles = tuple(float(a) for a in rng.uniform(0, 360, nem))
    try:
        table = mo.steering_table_c64(arr, res, mo.FREQUENCY, mo.SPACING)
        items = mo.synth_items(batch, m, N, arr, mo.FREQUENCY, mo.SPACING, angles_deg=angles, snr_db=snr, seed=int(rng.integers(1 << 30)))
        ao, lo, so = mr.work_batch(items, table, m, n)
        a1, l1, s1, a1n, names1 = run(m, n, N, res, table, items, "1")
        a0, l0, s0, a0n, names0 = run(m, n, N, res, table, items, "0")
        assert names1[0].endswith("cov_wide_mfma_kernel" if m <= 32 else "cov_wide_pairs_kernel") and names0[0].endswith("cov_wide_kernel"), (names1, names0)
        assert names1[1].endswith("scan_wide_mfma_kernel") == (n <= 8), names1
        if not (n > nem and snr > 40.0):
            worst = max(worst, assert_spectrum_close(s1, so))
            assert_doa_match(a1, l1, ao, lo, res, so.astype(np.float64))
            assert_doa_match(a1n, None, ao, lo, res, so.astype(np.float64))
            ab = float(np.max(np.abs(s1.astype(np.float64) - s0) / s0))
            worst_ab = max(worst_ab, ab)
            assert ab <= 5e-6, "matrix-core vs vector-unit kernels: %.3g" % ab
        else:
            assert np.all(np.isfinite(s1) | ~np.isfinite(so))
        bins = np.rint(a1 * res / 360.0).astype(int) % res
        used = l1 > 0
        assert np.array_equal(l1[used], np.take_along_axis(s1, bins, axis=1)[used])
    except AssertionError as e:
        fails += 1
        print("FAIL case %d: m=%d n=%d K=%d res=%d batch=%d snr=%g emitters=%d: %s" % (case, m, n, K, res, batch, snr, nem, str(e)[:300]), flush=True)
print("fuzz_wide: %d cases, %d failures, worst spectrum rel err vs oracle %.3g, vs the vector-unit kernels %.3g, %.1f s"
      % (ncases, fails, worst, worst_ab, time.time() - t0))
sys.exit(1 if fails else 0)
