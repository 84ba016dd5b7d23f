"""CPU study behind scan_i8_kernel (gr_baz_amd/csrc/scan_i8_kernels.hip.h): the scan's denominator d = sum_e q_e F_e with
both operands cut into seven balanced base-256 digits (int8), digit products accumulated EXACTLY per level s + t (what
v_mfma_i32_16x16x64_i8 does).  Bulk form: the levels < 5 of the five leading digits; refined form: + levels 5, 6 of all seven.
Checks the a-priori bound  |d5 - d| <= E5 = MM * Fscale * 5 * 1.01 * 2^-38  on real scenes, reports which share of values /
(16-item x 16-bin) tiles falls under the accuracy threshold T = E5 (1 + 1/eps) and therefore adds the two digits, and how far
the refined form is from the true value.  numpy only; imports the oracle for the scenes (tests/lab may).

usage: python tests/lab/i8_split_study.py [m=8] [res=3600] [items=64] [snr_db=20]
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
        F[:, i * m + i] = A[:, i].real ** 2 + A[:, i].imag ** 2       # (not np.abs() ** 2: that takes a square root first)
        for j in range(i + 1, m):
            c = np.conj(A[:, i]) * A[:, j]
            F[:, i * m + j] = c.real
            F[:, j * m + i] = c.imag
    return F


NS, ND = 5, 7


def digits(v):
    """integer array v (|v| <= 2^54 (1 + 2^-10)) -> ND balanced base-256 digits, most significant first;
    digits 1.. in [-128, 127], digit 0 what is left (|.| <= 65)."""
    v = v.astype(np.int64)
    out = []
    for _ in range(ND - 1):
        h = (v + 128) >> 8                 # floor((v + 128) / 256)
        out.append(v - (h << 8))
        v = h
    out.append(v)
    return out[::-1]


def scan_int(q, F, fscale):
    """(d5, d7): bulk and refined form."""
    sq = 2.0 ** (8 * ND - 2)
    qd = digits(np.rint(q * sq).astype(np.int64))
    fd = digits(np.rint(F * (sq / fscale)).astype(np.int64))
    assert all(np.abs(d).max() <= 128 for d in qd + fd) and np.abs(qd[0]).max() <= 65 and np.abs(fd[0]).max() <= 65
    wt = [fscale * 2.0 ** (-12 - 8 * l) for l in range(ND)]
    d5 = np.zeros((q.shape[0], F.shape[0]))
    tail = np.zeros_like(d5)
    for l in range(ND):
        A = np.zeros(d5.shape, dtype=np.int64)
        for s in range(l + 1):
            A += qd[s] @ fd[l - s].T       # exact (int64)
        assert np.abs(A).max() < 2 ** 31
        if l < NS:
            d5 += A.astype(np.float64) * wt[l]
        else:
            tail += A.astype(np.float64) * wt[l]
    return d5, d5 + tail


def main():
    args = dict(a.split("=") for a in sys.argv[1:])
    m, res, items, snr = int(args.get("m", 8)), int(args.get("res", 3600)), int(args.get("items", 64)), float(args.get("snr_db", 20))
    ns = NS
    n, K = 2, 512 if m == 8 else 256
    arr = mo.array_geometry(m)
    table = mo.steering_table_c64(arr, res, mo.FREQUENCY, mo.SPACING)
    F = f_image(table)
    mm = m * m
    fmax = np.abs(F).max()
    fscale = 2.0 ** np.frexp(fmax / 1.0009765625)[1]
    E = mm * fscale * ns * 1.01 * 2.0 ** (2 - 8 * ns)
    eps = 7.5e-7
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
        dl = q.astype(np.longdouble) @ F.T.astype(np.longdouble)                       # ~exact
        d = dl.astype(np.float64)
        di, d7 = scan_int(q, F, fscale)
        err = np.abs(di - d)
        print(" %-10s worst |d5 - d| = %.3g = %.3f E5;  worst relative where d5 > T: %.3g" %
              (scene, err.max(), err.max() / E, (err / np.abs(d))[di > T].max() if (di > T).any() else 0.0))
        low = di <= T
        err7 = np.abs(d7 - dl).astype(np.float64)
        print("            refined form: worst |d7 - d| = %.3g = %.3f of (MM Fscale 7.07 2^-54 + 2^-53 d); worst relative among the values under T: %.3g" %
              (err7.max(), (err7 / (mm * fscale * 7.07 * 2.0 ** -54 + 2.0 ** -53 * d)).max(), (err7 / np.abs(d))[low].max() if low.any() else 0.0))
        rows = low[: items // 16 * 16].reshape(items // 16, 16, -1)
        ns64 = rows.shape[2] // 64
        # a tile = 16 items x the 16 bins 64 st + 4 c + t (c = 0 .. 15) of tile t
        tl = rows[:, :, : ns64 * 64].reshape(items // 16, 16, ns64, 16, 4).any(axis=(1, 3))
        print("            values under T: %.2f %%   (16 x 16) tiles with one: %.2f %%   d/||a||^2 quantiles 1/10/50 %%: %s" %
              (100 * low.mean(), 100 * tl.mean(), np.round(np.quantile(d / m, [0.01, 0.1, 0.5]), 4)))


if __name__ == "__main__":
    main()
