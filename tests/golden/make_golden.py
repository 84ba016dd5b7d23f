#!/usr/bin/env python3
"""Generates tests/golden/*.npz -- run in the build container (needs /root/reference for oracle/_ref).

Each fixture holds: the complex64 input items, the complex64 steering table (helper formula,
python/music_doa_helper.py:32-46, rounded like SWIG does), and the float32 outputs of

    oracle/_ref/libbaz_music_ref.so  ==  the reference's OWN lib/baz_music_doa.cc::work(),
    compiled from /root/reference against oracle/ref_shim (Armadillo/GNU Radio API subset),
    eig_sym backed by LAPACK zheev (scipy's OpenBLAS) -- what Armadillo dispatches to.

plus the fp64 strengths of the numpy restatement (for tie analysis).  Inputs are stored, not
re-generated, so the fixtures do not depend on numpy's RNG stream.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import music_oracle as mo   # noqa: E402
from oracle import music_ref as mr      # noqa: E402


def fixture(name, m, n, nsamples, res, array, batch, snr_db, seed, angles=(40.3, 121.7),
            frequency=mo.FREQUENCY, spacing=mo.SPACING):
    table = mo.steering_table_c64(array, res, frequency, spacing)
    items = mo.synth_items(batch, m, nsamples, array, frequency, spacing, angles_deg=angles,
                           snr_db=snr_db, seed=seed)
    ang, lvl, spec = mr.ref_work_batch(items, table, m, n)
    _, _, _, strength = mo.music_doa_work_batch(items, table, m, n)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, items=items, table=table, ang=ang, lvl=lvl, spectrum=spec,
                        strength64=strength, m=m, n=n, nsamples=nsamples, res=res,
                        array=np.asarray(array, dtype=np.float64), snr_db=snr_db, seed=seed,
                        angles=np.asarray(angles), frequency=frequency, spacing=spacing,
                        generator="oracle/_ref (reference lib/baz_music_doa.cc::work on API shim, LAPACK zheev=%d)"
                                  % int(mr.ref().baz_ref_uses_lapack()))
    print("%-28s items %s  spectrum %s  %6.1f KiB" % (name, items.shape, spec.shape, os.path.getsize(path) / 1024))


def agc_signal(n, seed):
    rng = np.random.default_rng(seed)
    amp = 0.05 + 2.0 * np.abs(np.sin(np.linspace(0.0, 9.0, n))) + (np.arange(n) > n // 2) * 3.0
    return ((rng.standard_normal(n) + 1j * rng.standard_normal(n)) * amp).astype(np.complex64)


def agc_fixture(name, rate, reference, calls, seed):
    """calls: sample counts of consecutive work() calls on ONE block instance (state carries over).  Outputs
    come from oracle/_ref/libbaz_agc_ref.so == the reference's own lib/baz_agc_cc.cc::work()."""
    from oracle import agc_ref as ar
    x = agc_signal(int(sum(calls)), seed)
    blk = ar.Agc(rate, reference, use_reference_source=True)
    outs, envs, muls = [], [], []
    pos = 0
    for c in calls:
        o, e, m = blk.work(x[pos:pos + c])
        outs.append(o); envs.append(e); muls.append(m)
        pos += c
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, x=x, out=np.concatenate(outs), env=np.concatenate(envs), mul=np.concatenate(muls),
                        calls=np.asarray(calls), rate=np.float32(rate), reference=np.float32(reference),
                        generator="oracle/_ref (reference lib/baz_agc_cc.cc::work on API shim)")
    print("%-28s calls %s  %6.1f KiB" % (name, list(calls), os.path.getsize(path) / 1024))


def resamp_signal(n, seed):
    """complex baseband occupying about a quarter of the band (the MMSE table is designed for |f| <= 1/4)"""
    rng = np.random.default_rng(seed)
    w = rng.standard_normal(n + 64) + 1j * rng.standard_normal(n + 64)
    h = np.sinc(0.25 * (np.arange(65) - 32)) * np.hamming(65) * 0.25
    return np.convolve(w, h, mode="valid")[:n].astype(np.complex64)


RESAMP_EVENTS = {"set_mu": 1, "set_ratio": 2, "adjust": 3, "set_rational": 4}


def resamp_apply_event(blk, kind, a, b):
    if kind == 1: blk.set_mu(float(a))
    elif kind == 2: blk.set_resamp_ratio(float(a))
    elif kind == 3: blk.adjust(float(a))
    elif kind == 4: blk.set_resamp_ratio_rational(int(a), int(b))


def resamp_fixture(name, phase, ratio, num, denom, calls, events, seed, nin=18000):
    """calls: noutput of consecutive general_work() calls on ONE block instance; events: (before_call, kind, a, b).
    Outputs come from oracle/_ref/libbaz_resamp_ref.so == the reference's own
    lib/baz_fractional_resampler_cc.cc::general_work() (its MMSE interpolator is the shim, see oracle/resamp_ref.c:
    PARITY UNPINNED with respect to a real gnuradio-filter)."""
    from oracle import resamp_ref as rr
    x = resamp_signal(nin, seed)
    blk = rr.RefResampler(phase, ratio, num, denom)
    outs, cons, mus = [], [], []
    pos = 0
    for i, c in enumerate(calls):
        for (ci, kind, a, b) in events:
            if ci == i:
                resamp_apply_event(blk, kind, a, b)
        o, k = blk.work(x[pos:], c)
        outs.append(o); cons.append(k); mus.append(blk.mu())
        pos += k
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, x=x, out=np.concatenate(outs), calls=np.asarray(calls), consumed=np.asarray(cons),
                        mu_after=np.asarray(mus), events=np.asarray(events, dtype=np.float64).reshape(-1, 4),
                        phase=np.float64(phase), ratio=np.float64(ratio), num=np.uint64(num), denom=np.uint64(denom),
                        generator="oracle/_ref (reference lib/baz_fractional_resampler_cc.cc::general_work on API shim; "
                                  "MMSE interpolator = closed-form table, parity unpinned)")
    print("%-28s calls %s consumed %s  %6.1f KiB" % (name, list(calls), cons, os.path.getsize(path) / 1024))


def resamp_fixtures():
    resamp_fixture("resamp_ratio1.25", 0.0, 1.25, 0, 0, (4000, 1, 255, 256, 257, 9000), [], 4001)
    resamp_fixture("resamp_interp0.73_phase0.3", 0.3, 0.73, 0, 0, (5000, 7000), [], 4002)
    resamp_fixture("resamp_48k_to_44k1", 0.0, 0.0, 48000, 44100, (3000, 3000, 3000), [], 4003)
    resamp_fixture("resamp_setters", 0.5, 1.0000123, 0, 0, (2000, 2000, 2000, 2000, 2000),
                   [(1, 2, 1.5, 0), (2, 1, 0.125, 0), (3, 3, 0.75, 0), (4, 4, 3, 2), (4, 3, -0.25, 0)], 4004)


def wide_fixtures():
    """more than 16 antennas: the run-time-m kernels (gr_baz_amd/csrc/music_wide_kernels.hip.h)"""
    fixture("wide_m17_n2_N816_r360", 17, 2, 17 * 48, 360, mo.array_geometry(17), 3, 20.0, 2101)
    fixture("wide_m24_n5_N1536_r500", 24, 5, 24 * 64, 500, mo.array_geometry(24), 2, 25.0, 2102,
            angles=(20.0, 95.0, 170.0, 245.0, 320.0))
    fixture("wide_m32_n2_N2048_r720", 32, 2, 2048, 720, mo.array_geometry(32), 2, 20.0, 2103)
    fixture("wide_m33_n32_N2112_r90", 33, 32, 33 * 64, 90, mo.array_geometry(33), 2, 30.0, 2104,
            angles=tuple(np.linspace(3.0, 351.0, 32)))
    fixture("wide_m64_n3_N4096_r256", 64, 3, 4096, 256, mo.array_geometry(64), 1, 20.0, 2105, angles=(40.3, 121.7, 250.0))


def wide_fixtures_r3():
    """round 3: the shapes the matrix-core kernels for 33..64 antennas and 3..8 emitters take (cov_wide_pairs_kernel,
    scan_wide_mfma_kernel with four staged phases / two or one items per tile, sub_wide_kernel<5..8>)"""
    fixture("wide_m48_n8_N3072_r724", 48, 8, 48 * 64, 724, mo.array_geometry(48), 2, 20.0, 2106,
            angles=tuple(np.linspace(11.0, 331.0, 8)))
    fixture("wide_m40_n4_N2000_r1001", 40, 4, 40 * 50, 1001, mo.array_geometry(40), 2, 25.0, 2107, angles=(31.0, 122.5, 201.0, 299.0))
    fixture("wide_m64_n6_N4160_r360", 64, 6, 64 * 65, 360, mo.array_geometry(64), 1, 20.0, 2108,
            angles=(15.0, 70.0, 133.0, 190.0, 255.0, 320.0))


def main():
    if "--wide-r3-only" in sys.argv:
        if not mr.have_ref():
            mr.build()
        print("LAPACK zheev backend:", mr.ref_use_lapack(True))
        wide_fixtures_r3()
        return
    if "--resamp-only" in sys.argv:
        resamp_fixtures()
        return
    if "--wide-only" in sys.argv:
        if not mr.have_ref():
            mr.build()
        print("LAPACK zheev backend:", mr.ref_use_lapack(True))
        wide_fixtures()
        return
    if not mr.have_ref():
        mr.build()
    if not mr.have_ref():
        raise SystemExit("oracle/_ref could not be built (no /root/reference?)")
    print("LAPACK zheev backend:", mr.ref_use_lapack(True))
    sq = mo.array_geometry(4)
    # BASELINE.json configs
    fixture("cfg1_m4_n2_N256_r360", 4, 2, 256, 360, sq, 8, 20.0, 1001)
    fixture("cfg2_m4_n2_N1024_r3600", 4, 2, 1024, 3600, sq, 4, 20.0, 1002)
    fixture("cfg3_m8_n2_N4096_r36000", 8, 2, 4096, 36000, mo.array_geometry(8), 2, 20.0, 1003)
    # SNR sweep at cfg1 (Appendix C: 10 / 40 dB)
    fixture("cfg1_snr10", 4, 2, 256, 360, sq, 8, 10.0, 2001)
    fixture("cfg1_snr40", 4, 2, 256, 360, sq, 8, 40.0, 2002)
    # GRC defaults (grc/baz_music_doa.xml:13-55): m4 n1 N512 res360, ULA, frequency=1 spacing=1 is
    # degenerate (lambda = 3e8 m), so use lambda = 2 spacing like a real deployment
    fixture("grc_default_ula", 4, 1, 512, 360, mo.GRC_DEFAULT_ULA, 4, 20.0, 2003, angles=(63.0,),
            frequency=mo.C_LIGHT / 2.0, spacing=1.0)
    # shapes that exercise the generic paths: odd m, n = m-1, K % 32 != 0, res % 4 != 0, res % 16 != 0
    fixture("odd_m3_n1_N300_r357", 3, 1, 300, 357, mo.array_geometry(3), 5, 15.0, 2004, angles=(200.5,))
    fixture("m5_n3_N1000_r720", 5, 3, 1000, 720, mo.array_geometry(5), 4, 25.0, 2005, angles=(10.0, 95.5, 250.25))
    fixture("m8_n5_N1024_r1000", 8, 5, 1024, 1000, mo.array_geometry(8), 3, 25.0, 2006,
            angles=(15.0, 80.0, 140.0, 222.2, 300.0))
    fixture("m2_n1_N64_r90", 2, 1, 64, 90, [[0.0, 0.0], [1.0, 0.0]], 6, 20.0, 2007, angles=(60.0,))
    fixture("m6_n2_N1536_r1440", 6, 2, 1536, 1440, mo.array_geometry(6), 3, 20.0, 2008)
    fixture("m7_n4_N700_r500", 7, 4, 700, 500, mo.array_geometry(7), 3, 30.0, 2009,
            angles=(33.0, 111.0, 199.0, 287.0))
    # 9 <= m <= 16: two-tile covariance, LDS EVD, K = m*m up to 256 scan; BASELINE config 5 shape (m16, N4096, res3600)
    fixture("cfg5_m16_n2_N4096_r3600", 16, 2, 4096, 3600, mo.array_geometry(16), 2, 20.0, 1005)
    fixture("m12_n9_N1200_r720", 12, 9, 1200, 720, mo.array_geometry(12), 2, 30.0, 2010,
            angles=(10.0, 50.0, 90.0, 130.0, 170.0, 210.0, 250.0, 290.0, 330.0))
    fixture("m10_n3_N1000_r1001", 10, 3, 1000, 1001, mo.array_geometry(10), 3, 20.0, 2011, angles=(70.0, 190.0, 300.5))
    fixture("m9_n1_N630_r250", 9, 1, 630, 250, mo.array_geometry(9), 3, 15.0, 2012, angles=(222.0,))
    wide_fixtures()
    wide_fixtures_r3()
    # baz_agc_cc: defaults of lib/baz_agc_cc.h:41 and a fast loop; stateful call sequences
    # that straddle the 4096-sample chunk of the HIP scan
    agc_fixture("agc_default_rate1e-4", 1e-4, 1.0, (12000,), 3001)
    agc_fixture("agc_stateful_rate1e-2", 1e-2, 0.5, (1, 4095, 4096, 4097, 3, 5000), 3002)
    agc_fixture("agc_fast_rate0.5", 0.5, 2.0, (100, 9000), 3003)
    # fractional_resampler_cc
    resamp_fixtures()


if __name__ == "__main__":
    main()
