"""CPU tests: the oracle restatements against the golden vectors (made by the reference's own
work() source, oracle/_ref), against each other, and against analytic known answers
(SURVEY.md 8c "pins the new repo must author")."""
import numpy as np
import pytest

from conftest import golden_names, load_golden
from helpers import assert_doa_match, assert_spectrum_close
from oracle import music_oracle as mo
from oracle import music_ref as mr


@pytest.mark.parametrize("name", golden_names())
def test_numpy_oracle_matches_golden(name):
    g = load_golden(name)
    ang, lvl, spec, strength = mo.music_doa_work_batch(g["items"], g["table"], g["m"], g["n"])
    assert_spectrum_close(spec, g["spectrum"], rtol=1e-6)
    assert_doa_match(ang, lvl, g["ang"], g["lvl"], g["res"], strength)


@pytest.mark.parametrize("name", golden_names())
def test_c_oracle_matches_golden(name):
    g = load_golden(name)
    ang, lvl, spec = mr.work_batch(g["items"], g["table"], g["m"], g["n"])
    assert_spectrum_close(spec, g["spectrum"], rtol=1e-6)
    assert_doa_match(ang, lvl, g["ang"], g["lvl"], g["res"], g["strength64"])


@pytest.mark.parametrize("name", ["cfg1_m4_n2_N256_r360", "m5_n3_N1000_r720", "odd_m3_n1_N300_r357"])
def test_per_item_loop_equals_batched_oracle(name):
    g = load_golden(name)
    for b in range(min(3, g["items"].shape[0])):
        a, l, s = mo.music_doa_work(g["items"][b], g["table"], g["m"], g["n"])
        assert_spectrum_close(s, g["spectrum"][b], rtol=1e-6)
        assert np.array_equal(a, g["ang"][b])


@pytest.mark.skipif(not mr.have_ref(), reason="oracle/_ref not built (no /root/reference on this box)")
@pytest.mark.parametrize("lapack", [False, True])
@pytest.mark.parametrize("name", ["cfg1_m4_n2_N256_r360", "cfg2_m4_n2_N1024_r3600", "m8_n5_N1024_r1000"])
def test_reference_source_build_matches_golden(name, lapack, capfd):
    """oracle/_ref = the reference's own lib/baz_music_doa.cc; both eig_sym backends (LAPACK zheev
    and the shim's Jacobi) must reproduce the stored vectors: the result is eigensolver-independent."""
    g = load_golden(name)
    mr.ref_use_lapack(lapack)
    ang, lvl, spec = mr.ref_work_batch(g["items"], g["table"], g["m"], g["n"])
    capfd.readouterr()
    assert_spectrum_close(spec, g["spectrum"], rtol=1e-6)
    assert_doa_match(ang, lvl, g["ang"], g["lvl"], g["res"], g["strength64"])


def test_jacobi_eig_matches_lapack():
    rng = np.random.default_rng(7)
    for m in (2, 3, 4, 5, 8, 16):
        A = rng.standard_normal((m, m)) + 1j * rng.standard_normal((m, m))
        A = A @ A.conj().T / m
        w, V = mr.eig(A)
        w2, _ = np.linalg.eigh(A)
        assert np.allclose(w, w2, rtol=1e-12, atol=1e-13)
        assert np.allclose(V.conj().T @ V, np.eye(m), atol=1e-13)
        assert np.allclose(A @ V, V * w, atol=1e-12)


@pytest.mark.parametrize("theta", [33.0, 77.0, 123.5, 200.0, 290.5, 340.0])
def test_single_emitter_known_answer(theta):
    """Analytic pin: one strong emitter on a 2-D array -> argmax bin = round(theta*res/360).
    (The half-wavelength square is ambiguous exactly on its axes -- a(0)=a(180), a(90)=a(270) -- so the
    pins stay off 0/90/180/270 degrees.)"""
    m, n, N, res = 4, 1, 512, 720
    arr = mo.array_geometry(m)
    table = mo.steering_table_c64(arr, res, mo.FREQUENCY, mo.SPACING)
    items = mo.synth_items(3, m, N, arr, mo.FREQUENCY, mo.SPACING, angles_deg=(theta,), snr_db=30.0, seed=11)
    ang, lvl, spec = mr.work_batch(items, table, m, n)
    expect = (round(theta * res / 360.0) % res) * 360.0 / res
    assert np.all(ang[:, 0] == np.float32(expect))
    assert np.all(spec.argmax(axis=1) == round(theta * res / 360.0) % res)
    assert np.all(lvl[:, 0] == spec.max(axis=1))


def test_input_layout_is_antenna_interleaved():
    """x(r,c) = in[c*m + r] (lib/baz_music_doa.cc:82-84): permuting antennas in the data AND the table
    leaves the spectrum unchanged; permuting only the data moves it."""
    m, n, N, res = 4, 1, 256, 360
    arr = mo.array_geometry(m)
    table = mo.steering_table_c64(arr, res, mo.FREQUENCY, mo.SPACING)
    items = mo.synth_items(2, m, N, arr, mo.FREQUENCY, mo.SPACING, angles_deg=(70.0,), snr_db=25.0, seed=5)
    perm = [2, 0, 3, 1]
    items_p = items.reshape(2, N // m, m)[:, :, perm].reshape(2, N)
    table_p = table[:, perm]
    _, _, s0 = mr.work_batch(items, table, m, n)
    _, _, s1 = mr.work_batch(items_p, table_p, m, n)
    _, _, s2 = mr.work_batch(items_p, table, m, n)
    assert_spectrum_close(s1, s0, rtol=1e-6)
    assert s2.argmax(axis=1).tolist() != s0.argmax(axis=1).tolist()
    # a de-interleaved (antenna-major) buffer is NOT what the block expects
    items_wrong = items.reshape(2, N // m, m).transpose(0, 2, 1).reshape(2, N)
    _, _, s3 = mr.work_batch(np.ascontiguousarray(items_wrong), table, m, n)
    assert not np.allclose(s3, s0, rtol=1e-3)


def test_top_n_insertion_semantics():
    """lib/baz_music_doa.cc:95,129-141: n largest BINS (not peaks), descending, earliest bin wins
    ties, strict '>' against the initial (0,0) entries, NaN never inserted, +inf is."""
    res = 10
    s = np.array([1.0, 5.0, 5.0, 3.0, 7.0, 7.0, 2.0, 0.0, 6.9, 1.0])
    ang, lvl = mo.top_n_insertion(s, 3, res)
    assert lvl.tolist() == [7.0, 7.0, 6.9] and ang.tolist() == [4 * 36.0, 5 * 36.0, 8 * 36.0]
    ang4, lvl4 = mo.top_n_insertion(s, 5, res)
    assert lvl4.tolist() == [7.0, 7.0, 6.9, 5.0, 5.0] and ang4.tolist()[3:] == [1 * 36.0, 2 * 36.0]
    ang2, lvl2 = mo.top_n_fast(s, 3, res)
    assert np.array_equal(ang, ang2) and np.array_equal(lvl, lvl2)
    # fewer than n positive entries: the rest stay (0, 0)
    ang, lvl = mo.top_n_insertion(np.array([0.0, 0.0, 4.0, 0.0]), 2, 4)
    assert lvl.tolist() == [4.0, 0.0] and ang.tolist() == [180.0, 0.0]
    # NaN never inserts, +inf does
    ang, lvl = mo.top_n_insertion(np.array([np.nan, 2.0, np.inf, np.nan]), 2, 4)
    assert lvl.tolist() == [np.inf, 2.0] and ang.tolist() == [180.0, 90.0]
    # adjacent bins of one lobe are both reported (quirk a10)
    ang, lvl = mo.top_n_insertion(np.array([1.0, 9.0, 10.0, 9.5, 1.0, 3.0]), 2, 6)
    assert ang.tolist() == [120.0, 180.0]


def test_helper_table_formula_and_retune():
    """python/music_doa_helper.py:32-46,55-56,100-103."""
    arr = [[0, 0], [1, 0], [0, 1]]
    res, f, sp = 8, 150e6, 0.75
    l = mo.C_LIGHT / f
    tab = np.array(mo.calculate_antenna_array_response(mo.scaled_array(arr, sp), res, l))
    assert tab.shape == (res, 3)
    for step in range(res):
        th = step * 360.0 / res * np.pi / 180.0
        for t, (x, y) in enumerate(arr):
            expect = np.exp(-2j * np.pi * ((sp * x) * np.cos(th) + (sp * y) * np.sin(th)) / l)
            assert abs(tab[step, t] - expect) < 1e-14
    assert np.allclose(np.abs(tab), 1.0)
    t1 = mo.steering_table_c64(arr, res, f, sp)
    t2 = mo.steering_table_c64(arr, res, 2 * f, sp)
    assert t1.dtype == np.complex64 and not np.allclose(t1, t2)
    assert np.allclose(t2, mo.steering_table_c64(arr, res, f, 2 * sp))   # only spacing/lambda matters


def test_c_oracle_rejects_bad_arguments():
    lib = mr.lib()
    z = np.zeros(16, np.float32)
    import ctypes
    p = z.ctypes.data_as(ctypes.POINTER(ctypes.c_float))
    assert lib.music_ref_work(p, p, 4, 4, 8, 1, p, p, None, None) == -1   # n == m (.cc:93 underflow)
    assert lib.music_ref_work(p, p, 4, 0, 8, 1, p, p, None, None) == -1
    assert lib.music_ref_work(p, p, 4, 2, 6, 1, p, p, None, None) == -1   # nsamples % m != 0
    assert lib.music_ref_work(p, p, 0, 0, 8, 1, p, p, None, None) == -1


@pytest.mark.parametrize("seed", range(6))
def test_three_restatements_agree_on_random_shapes(seed, capfd):
    """Fresh random cases (not the stored vectors): the numpy restatement, the plain-C restatement and -- where it is built
    -- the reference's own baz_music_doa.cc (oracle/_ref, both eig_sym backends) give the same spectra to 1e-6 and the same
    DoA bins up to the tie rule, over antennas 2..12, emitters 1..m-1, odd resolutions, -5..40 dB and 0..n actual emitters."""
    rng = np.random.default_rng(4242 + seed)
    m = int(rng.integers(2, 13))
    n = int(rng.integers(1, m))
    K = int(rng.integers(max(2 * m, 8), 200))
    res = int(rng.integers(16, 900))
    snr = float(rng.uniform(-5.0, 40.0))
    emitters = int(rng.integers(0, n + 1))
    arr = mo.array_geometry(m)
    table = mo.steering_table_c64(arr, res, mo.FREQUENCY, mo.SPACING)
    items = mo.synth_items(5, m, K * m, arr, mo.FREQUENCY, mo.SPACING, angles_deg=tuple(rng.uniform(0, 360, emitters)),
                           snr_db=snr, seed=int(rng.integers(1 << 30)))
    a0, l0, s0, strength = mo.music_doa_work_batch(items, table, m, n)
    a1, l1, s1 = mr.work_batch(items, table, m, n)
    assert_spectrum_close(s1, s0, rtol=1e-6)
    assert_doa_match(a1, l1, a0, l0, res, strength)
    if mr.have_ref():
        for lapack in (False, True):
            mr.ref_use_lapack(lapack)
            a2, l2, s2 = mr.ref_work_batch(items, table, m, n)
            capfd.readouterr()
            assert_spectrum_close(s2, s0, rtol=1e-6)
            assert_doa_match(a2, l2, a0, l0, res, strength)
