"""Shared assertions for the parity tests (tolerances are the ones BASELINE.json / SURVEY.md 8d state)."""
import numpy as np

SPECTRUM_RTOL = 1e-5   # north_star: "within 1e-5 relative float tolerance"
TIE_RTOL = 2e-5        # ang may differ only where the reference's own competing
                       # strengths differ by less than this


def assert_spectrum_close(spec, spec_ref, rtol=SPECTRUM_RTOL, what="spectrum"):
    spec = np.asarray(spec, dtype=np.float64)
    ref = np.asarray(spec_ref, dtype=np.float64)
    assert spec.shape == ref.shape, (spec.shape, ref.shape)
    fin = np.isfinite(ref)
    assert np.array_equal(np.isfinite(spec), fin), "%s: finite/non-finite pattern differs" % what
    err = np.abs(spec[fin] - ref[fin])
    bound = rtol * np.abs(ref[fin])
    worst = float(np.max(err / np.maximum(np.abs(ref[fin]), 1e-300))) if err.size else 0.0
    assert np.all(err <= bound), "%s: max relative error %.3g > %.1g" % (what, worst, rtol)
    return worst


def assert_doa_match(ang, lvl, ang_ref, lvl_ref, res, strength64=None):
    """ang/lvl: (B,n).  Bins must be identical, except where the reference's own strengths at the two
    competing bins are within TIE_RTOL of each other (then either order/bin is acceptable)."""
    ang = np.asarray(ang); ang_ref = np.asarray(ang_ref)
    assert ang.shape == ang_ref.shape
    if lvl is not None:
        assert_spectrum_close(lvl, lvl_ref, what="lvl")
    if np.array_equal(ang, ang_ref):
        return
    assert strength64 is not None, "ang differs and no fp64 strengths were supplied for tie analysis"
    B, n = ang.shape
    for b in range(B):
        for i in range(n):
            if ang[b, i] == ang_ref[b, i]:
                continue
            bin_a = int(round(float(ang[b, i]) * res / 360.0)) % res
            bin_r = int(round(float(ang_ref[b, i]) * res / 360.0)) % res
            sa, sr = strength64[b, bin_a], strength64[b, bin_r]
            assert abs(sa - sr) <= TIE_RTOL * max(abs(sa), abs(sr)), \
                "item %d slot %d: bin %d (%.6g) vs reference bin %d (%.6g) is not a tie" % (b, i, bin_a, sa, bin_r, sr)
