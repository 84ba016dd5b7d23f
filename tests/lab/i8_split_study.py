"""CPU study behind scan_i8_kernel (gr_baz_amd/csrc/scan_i8_kernels.hip.h): the scan's denominator d = sum_e q_e F_e with
both operands cut into NS balanced base-256 digits (int8), digit products accumulated EXACTLY per level s + t (what
v_mfma_i32_16x16x64_i8 does), levels >= NS dropped.  Checks the a-priori bound
    |d_int - d| <= E = MM * Fscale * NS * 1.01 * 2^(2 - 8 NS)
on real scenes and reports which share of values / (16-item x 64-bin) steps falls under the accuracy threshold
T = E (1 + 1/eps) and therefore stays on the fp64 matrix core.  numpy only; imports the oracle for the scenes (tests/lab may).

usage: python tests/lab/i8_split_study.py [m=8] [res=3600] [items=64] [NS=5] [snr_db=20]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import music_oracle as mo    # noqa: E402


def q_image(Q):
    """gr_baz_amd/csrc/music_kernels.hip.h evd_finish(): q[i*m+i] = Q_ii, q[i*m+j] = 2 Re Q_ij, q[j*m+i] = -2 Im Q_ij (i < j)."""
    m = Q.shape[-1]
    q = np.zeros(Q.shape[:-2] + (m * m,))
    for i in range(m):
        q[..., i * m + i] = Q[..., i, i].real
        for j in range(i + 1, m):
            q[..., i * m + j] = 2.0 * Q[..., i, j].real
            q[..., j * m + i] = -2.0 * Q[..., i, j].imag
    return q


def f_image(table):
    """baz_music_hip.hip build_F(): F[i*m+i] = |a_i|^2, F[i*m+j] = Re(conj(a_i) a_j), F[j*m+i] = Im(conj(a_i) a_j) (i < j)."""
    A = table.astype(np.complex128)
    res, m = A.shape
    F = np.zeros((res, m * m))
    for i in range(m):
        F[:, i * m + i] = np.abs(A[:, i]) ** 2
        for j in range(i + 1, m):
            c = np.conj(A[:, i]) * A[:, j]
            F[:, i * m + j] = c.real
            F[:, j * m + i] = c.imag
    return F


def digits(v, ns):
    """integer array v (|v| <= 2^(8 ns - 2) (1 + 2^-10)) -> ns balanced base-256 digits, most significant first;
    digits 1.. in [-128, 127], digit 0 what is left (|.| <= 65)."""
    v = v.astype(np.int64)
    out = []
    for _ in range(ns - 1):
        h = (v + 128) >> 8                 # floor((v + 128) / 256)
        out.append(v - (h << 8))
        v = h
    out.append(v)
    return out[::-1]


def scan_int(q, F, ns, fscale):
    sq = 2.0 ** (8 * ns - 2)
    sf = sq / fscale
    Qi = np.rint(q * sq)
    Fi = np.rint(F * sf)
    qd, fd = digits(Qi, ns), digits(Fi, ns)
    assert all(np.abs(d).max() <= 128 for d in qd + fd) and np.abs(qd[0]).max() <= 65 and np.abs(fd[0]).max() <= 65
    H = np.zeros((q.shape[0], F.shape[0]), dtype=np.int64)
    for l in range(ns):
        A = np.zeros_like(H)
        for s in range(l + 1):
            A += qd[s] @ fd[l - s].T       # exact (int64)
        assert np.abs(A).max() < 2 ** 31
        H += A << (8 * (ns - 1 - l))
    unit = 2.0 ** (8 * (ns - 1)) / (sq * sf)
    return H.astype(np.float64) * unit


def main():
    args = dict(a.split("=") for a in sys.argv[1:])
    m, res, items, ns, snr = int(args.get("m", 8)), int(args.get("res", 3600)), int(args.get("items", 64)), int(args.get("NS", 5)), float(args.get("snr_db", 20))
    n, K = 2, 512 if m == 8 else 256
    arr = mo.array_geometry(m)
    table = mo.steering_table_c64(arr, res, mo.FREQUENCY, mo.SPACING)
    F = f_image(table)
    mm = m * m
    fmax = np.abs(F).max()
    fscale = 2.0 ** np.ceil(np.log2(fmax))
    E = mm * fscale * ns * 1.01 * 2.0 ** (2 - 8 * ns)
    eps = 5e-7
    T = E * (1.0 + 1.0 / eps)
    print("m=%d res=%d NS=%d: fmax %.9g Fscale %g  E = %.3g  T(eps=%.0e) = %.3g  (||a||^2 = %d)" % (m, res, ns, fmax, fscale, E, eps, T, m))
    rng = np.random.default_rng(5)
    for scene in ("coherent", "incoherent"):
        if scene == "coherent":
            x = mo.synth_items(items, m, m * K, arr, mo.FREQUENCY, mo.SPACING, snr_db=snr, seed=1003)
        else:
            x = np.concatenate([mo.synth_items(1, m, m * K, arr, mo.FREQUENCY, mo.SPACING, snr_db=snr, seed=2000 + i,
                                               angles_deg=tuple(rng.uniform(0, 360, 2))) for i in range(items)])
        xs = x.astype(np.complex128).reshape(items, K, m).transpose(0, 2, 1)
        R = xs @ xs.conj().transpose(0, 2, 1) / K
        w, V = np.linalg.eigh(R)
        G = V[:, :, :m - n]
        Q = G @ G.conj().transpose(0, 2, 1)
        q = q_image(Q)
        assert np.abs(q).max() <= 1.0 + 1e-9, np.abs(q).max()
        d = (q.astype(np.longdouble) @ F.T.astype(np.longdouble)).astype(np.float64)   # ~exact
        di = scan_int(q, F, ns, fscale)
        err = np.abs(di - d)
        print(" %-10s worst |d_int - d| = %.3g = %.3f E;  worst relative where d_int > T: %.3g" %
              (scene, err.max(), err.max() / E, (err / np.abs(d))[di > T].max() if (di > T).any() else 0.0))
        low = di <= T
        steps = low[: items // 16 * 16].reshape(items // 16, 16, -1)
        ns64 = steps.shape[2] // 64
        st = steps[:, :, : ns64 * 64].reshape(items // 16, 16, ns64, 64).any(axis=(1, 3))
        print("            values under T: %.2f %%   (16 x 64) steps with one: %.2f %%   d/||a||^2 quantiles 1/10/50 %%: %s" %
              (100 * low.mean(), 100 * st.mean(), np.round(np.quantile(d / m, [0.01, 0.1, 0.5]), 4)))


if __name__ == "__main__":
    main()
