"""CPU study for DESIGN.md 10 item 5 (a lead for round 4, nothing here ships): how far is the scan's denominator
d = a^H Q a from its fp64 value when the projector GEMM runs in float32 (Q rounded to f32, f32 products, f32
accumulation in the matrix core's k order), as a function of d / ||a||^2 -- and what share of the (item, bin) values
of a cfg3-like scene lies above a given threshold?  numpy only (float32 arithmetic emulated operation by operation);
imports the oracle for the scenes, which tests/lab may do.

usage: python tests/lab/f32_bulk_study.py [items=24] [res=3600]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import music_oracle as mo    # noqa: E402


def real_image(Q):
    """q(item) in R^(m^2): d = sum_i Q_ii |a_i|^2 + 2 Re sum_{i<j} conj(a_i) Q_ij a_j as ONE real dot product with
    t(bin) = (|a_i|^2 ; 2 Re(conj(a_i) a_j) ; -2 Im(conj(a_i) a_j))  (the form the projector GEMM evaluates)."""
    m = Q.shape[-1]
    iu = np.triu_indices(m, 1)
    return np.concatenate([np.real(np.diagonal(Q, axis1=-2, axis2=-1)), np.real(Q[..., iu[0], iu[1]]),
                           np.imag(Q[..., iu[0], iu[1]])], axis=-1)


def table_image(A):
    m = A.shape[-1]
    iu = np.triu_indices(m, 1)
    p = np.conj(A[:, iu[0]]) * A[:, iu[1]]
    return np.concatenate([np.abs(A) ** 2, 2.0 * p.real, -2.0 * p.imag], axis=-1)


def dot_f32(q, t):
    """float32 products and a float32 running sum in index order (k = 0 .. m^2 - 1), one rounding per operation
    (the matrix core fuses multiply and add: this is the more pessimistic of the two)."""
    acc = np.zeros((q.shape[0], t.shape[0]), np.float32)
    q32, t32 = q.astype(np.float32), t.astype(np.float32)
    for k in range(q.shape[1]):
        acc = (acc + (q32[:, k, None] * t32[None, :, k]).astype(np.float32)).astype(np.float32)
    return acc


def bf16_round(x32):
    """float32 -> nearest bfloat16 (ties to even), returned as float32."""
    u = x32.astype(np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def split3(x):
    """x (float64) -> three bfloat16 parts h + m + l ~ x to 24 bits."""
    x32 = x.astype(np.float32)
    h = bf16_round(x32)
    r1 = (x32 - h).astype(np.float32)
    m = bf16_round(r1)
    l = bf16_round((r1 - m).astype(np.float32))
    return h, m, l


def dot_bf16x3(q, t):
    """six cross products hh, hm, mh, hl, lh, mm of the bf16 parts, exact products (bf16 x bf16 fits float32), float32
    accumulation in the order the concatenated-K GEMM would run them (part pair by part pair, k ascending)."""
    qh, qm, ql = split3(q)
    th, tm, tl = split3(t)
    acc = np.zeros((q.shape[0], t.shape[0]), np.float32)
    for a, b in ((ql, th), (qh, tl), (qm, tm), (qm, th), (qh, tm), (qh, th)):      # small terms first
        for k in range(q.shape[1]):
            acc = (acc + (a[:, k, None] * b[None, :, k]).astype(np.float32)).astype(np.float32)
    return acc


def split2_f16(x, scale):
    """x * scale -> two float16 parts (the product's coarse form uses the same kind of split, scan_coarse_kernels.hip.h)."""
    xs = (x * scale).astype(np.float32)
    h = xs.astype(np.float16).astype(np.float32)
    l = (xs - h).astype(np.float32).astype(np.float16).astype(np.float32)
    return h, l


def dot_f16x2(q, t):
    """two f16 parts per operand, all four cross products (exact in float32), float32 accumulation, small terms first."""
    sq, st = 2.0 ** 10, 2.0 ** 12
    qh, ql = split2_f16(q, sq)
    th, tl = split2_f16(t, st)
    acc = np.zeros((q.shape[0], t.shape[0]), np.float32)
    for a, b in ((ql, tl), (ql, th), (qh, tl), (qh, th)):
        for k in range(q.shape[1]):
            acc = (acc + (a[:, k, None] * b[None, :, k]).astype(np.float32)).astype(np.float32)
    return acc.astype(np.float64) / (sq * st)


def main():
    items = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    res = int(sys.argv[2]) if len(sys.argv) > 2 else 3600
    print("# f32 bulk study (CPU, numpy): m = 8, n = 2, K = 512, res = %d, %d items per SNR, emitters at 40.3 / 121.7 deg" % (res, items))
    print("# d = a^H Q a; rel = |d_f32 - d_f64| / d_f64; thresholds on d / ||a||^2 (||a||^2 = m = 8)")
    for snr in (10.0, 20.0, 40.0, 60.0):
        cfg = mo.make_config("cfg3", items, snr_db=snr)
        m, n = cfg["m"], cfg["n"]
        arr = cfg["array"]
        table = mo.steering_table_c64(arr, res, mo.FREQUENCY, mo.SPACING).astype(np.complex128)
        Qs = []
        for it in cfg["items"]:
            _, _, _, internals = mo.music_doa_work(it, table.astype(np.complex64), m, n, True, True)
            G = internals["G"]
            Qs.append(G @ G.conj().T)
        Q = np.stack(Qs)
        q, t = real_image(Q), table_image(table)
        d64 = q @ t.T
        d32 = dot_f32(q, t).astype(np.float64)
        rel = np.abs(d32 - d64) / d64
        relb = np.abs(dot_bf16x3(q, t).astype(np.float64) - d64) / d64
        relh = np.abs(dot_f16x2(q, t) - d64) / d64
        frac = d64 / float(m)
        line = "snr %4.0f dB: " % snr
        for thr in (0.5, 0.25, 0.125, 0.05, 0.02):
            sel = frac >= thr
            line += "| d/m >= %.3f: %5.1f %% of values, worst rel %.1e " % (thr, 100.0 * sel.mean(), rel[sel].max())
        print(line)
        # per 16-bin x 16-item tile (the matrix-core tile): a tile goes f32 only if ALL its values clear the threshold
        T = (items // 16) * 16
        if T:
            f = frac[:T].reshape(T // 16, 16, res // 16, 16).min(axis=(1, 3))
            print("           tiles (16 items x 16 bins, coherent items) entirely above 0.125: %.1f %%, above 0.05: %.1f %%"
                  % (100.0 * (f >= 0.125).mean(), 100.0 * (f >= 0.05).mean()))
        print("           three-part bf16 split (6 cross products, f32 accumulation): worst rel above 0.125: %.1e, above 0.05: %.1e, median %.1e"
              % (relb[frac >= 0.125].max(), relb[frac >= 0.05].max(), np.median(relb)))
        print("           two-part f16 split (4 cross products, f32 accumulation):   worst rel above 0.125: %.1e, above 0.05: %.1e, median %.1e"
              % (relh[frac >= 0.125].max(), relh[frac >= 0.05].max(), np.median(relh)))
        print("           overall worst rel %.1e at d/m = %.2e; median rel %.1e" % (rel.max(), frac.flat[rel.argmax()], np.median(rel)))


if __name__ == "__main__":
    main()
