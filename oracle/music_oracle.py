"""CPU ORACLE (test infrastructure, NOT the product) for gr-baz's MUSIC-DoA block.

numpy restatement of
  * /root/reference/lib/baz_music_doa.cc:72-161   (baz_music_doa::work)
  * /root/reference/python/music_doa_helper.py:29-46 (unit_vect,
    calculate_antenna_array_response)
plus the seeded synthetic-input generators of SURVEY.md section 8(d).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product path (gr_baz_amd/) never does.

PARITY PINNING: the reference ships no tests/golden vectors for this path
(lib/qa_baz.cc:32-41 and python/qa_baz.py:34-56 are empty) and its arithmetic
lives in Armadillo -> LAPACK zheev(d), an un-vendored, un-versioned system
dependency (cmake/Modules/FindArmadillo.cmake:35-73).  This restatement is pinned
by (1) oracle/_ref: the reference's OWN lib/baz_music_doa.cc compiled in place
against a minimal Armadillo/GNU-Radio API shim (see oracle/Makefile), (2) the
plain-C restatement oracle/music_ref.c, (3) analytic known-answer tests
(tests/test_oracle.py).  Where oracle/_ref cannot be built the status is
"parity unpinned" (see DESIGN.md section 3).

Eigen step: numpy.linalg.eigh == LAPACK zheevd (the family Armadillo's eig_sym
dispatches to).  The spectrum depends only on the noise-subspace projector, so the
eigensolver's phase/basis choice is immaterial.
"""
from __future__ import annotations

import math

import numpy as np

C_LIGHT = 299792458.0  # python/music_doa_helper.py:55


# --------------------------------------------------------------------------
# python/music_doa_helper.py:29-46
# --------------------------------------------------------------------------
def unit_vect(theta):
    """python/music_doa_helper.py:29-30."""
    return np.array([np.cos(theta), np.sin(theta)])


def calculate_antenna_array_response(antenna_array, angular_resolution, l):
    """python/music_doa_helper.py:32-46, python-3 restatement, same arithmetic order.

    Returns a list (len angular_resolution) of lists (len m) of python complex
    (fp64), exactly what the helper hands to SWIG.
    """
    response = []
    for step in range(0, angular_resolution):
        angle = (step * 360.0 / angular_resolution) * (np.pi / 180.0)
        response_step = []
        for antenna in antenna_array:
            phase_offset = np.inner(antenna, unit_vect(angle)) / l
            antenna_response = np.exp(-1j * 2.0 * np.pi * phase_offset)
            response_step += [antenna_response]
        response += [response_step]
    return response


def scaled_array(antenna_array, array_spacing):
    """python/music_doa_helper.py:56."""
    return [[array_spacing * x, array_spacing * y] for [x, y] in antenna_array]


def steering_table_c64(antenna_array, angular_resolution, frequency, array_spacing):
    """The table the native block actually sees: helper formula in fp64, then the
    SWIG std::vector<std::vector<gr_complex>> typemap rounds to complex64
    (swig/baz_swig.i:564).  Shape (res, m), dtype complex64."""
    l = C_LIGHT / frequency
    arr = scaled_array(antenna_array, array_spacing)
    tab = np.array(calculate_antenna_array_response(arr, angular_resolution, l),
                   dtype=np.complex128)
    return tab.astype(np.complex64)


# --------------------------------------------------------------------------
# lib/baz_music_doa.cc:72-161
# --------------------------------------------------------------------------
def music_doa_work(in_c64, table_c64, m, n, want_spectrum=True, return_internals=False):
    """One item of baz_music_doa::work.

    in_c64    : (N,) complex64, antenna-interleaved: x(r,c) = in[c*m + r]   (.cc:82-84)
    table_c64 : (res, m) complex64                                          (.h:32-33)
    returns (ang[n] float32, lvl[n] float32, spectrum[res] float32 | None)
    """
    in_c64 = np.asarray(in_c64, dtype=np.complex64)
    table_c64 = np.asarray(table_c64, dtype=np.complex64)
    N = in_c64.shape[0]
    res = table_c64.shape[0]
    assert table_c64.shape[1] == m and 0 < n < m and N > 0 and N % m == 0

    data = in_c64.astype(np.complex128)                      # .cc:74-77 (exact widening)
    K = N // m                                               # .cc:83
    x = data.reshape(K, m).T                                 # .cc:82-84 column-major reshape
    R = (x @ x.conj().T) / float(K)                          # .cc:85
    w, V = np.linalg.eigh(R)                                 # .cc:88-90 ascending
    G = V[:, 0:m - n]                                        # .cc:93

    A = table_c64.astype(np.complex128)                      # .cc:110-112
    c = A @ G.conj()                                         # (res, m-n): c[s,k] = sum_t conj(G[t,k]) a[t]
    nrm = np.sqrt(np.sum(c.real * c.real + c.imag * c.imag, axis=1))   # arma::norm(.,2)
    with np.errstate(divide="ignore"):
        strength = 1.0 / (nrm * nrm)                         # .cc:114-119 (pow(.,2))

    ang, lvl = top_n_insertion(strength, n, res)             # .cc:95,129-141
    ang32 = ang.astype(np.float32)                           # .cc:146-155
    lvl32 = lvl.astype(np.float32)
    spec32 = strength.astype(np.float32) if want_spectrum else None    # .cc:120-121
    if return_internals:
        return ang32, lvl32, spec32, dict(R=R, w=w, V=V, G=G, strength=strength)
    return ang32, lvl32, spec32


def top_n_insertion(strength, n, res):
    """lib/baz_music_doa.cc:95,129-141 verbatim semantics: list of n (angle,strength)
    initialised to (0,0); for each bin in ascending order, insert before the first
    entry with strictly smaller strength, drop the last.  NaN never inserts."""
    ang = [0.0] * n
    lvl = [0.0] * n
    for step in range(res):
        s = float(strength[step])
        for i in range(n):
            if s > lvl[i]:
                angle = float(step) * 360.0 / float(res)
                ang.insert(i, angle)
                lvl.insert(i, s)
                ang.pop()
                lvl.pop()
                break
    return np.array(ang, dtype=np.float64), np.array(lvl, dtype=np.float64)


def top_n_fast(strength, n, res):
    """Vectorised equivalent of top_n_insertion for NaN-free input (used for the big
    batches): n largest bins, descending strength, earliest bin wins ties, entries
    must beat the initial 0.0 strictly."""
    strength = np.asarray(strength, dtype=np.float64)
    order = np.lexsort((np.arange(res), -strength))          # by -strength, then bin
    ang = np.zeros(n)
    lvl = np.zeros(n)
    k = 0
    for b in order[:n]:
        if strength[b] > 0.0:
            ang[k] = float(b) * 360.0 / float(res)
            lvl[k] = strength[b]
            k += 1
    return ang, lvl


def music_doa_work_batch(in_c64, table_c64, m, n, want_spectrum=True):
    """Vectorised oracle over a batch (B, N) of items; identical arithmetic per item
    (batched eigh == zheevd per matrix).  Used where the per-item loop is too slow."""
    in_c64 = np.asarray(in_c64, dtype=np.complex64)
    table_c64 = np.asarray(table_c64, dtype=np.complex64)
    B, N = in_c64.shape
    res = table_c64.shape[0]
    K = N // m
    x = in_c64.astype(np.complex128).reshape(B, K, m).transpose(0, 2, 1)   # (B,m,K)
    R = (x @ x.conj().transpose(0, 2, 1)) / float(K)
    w, V = np.linalg.eigh(R)
    G = V[:, :, 0:m - n]                                                   # (B,m,m-n)
    A = table_c64.astype(np.complex128)
    c = np.einsum("st,btk->bsk", A, G.conj())
    nrm = np.sqrt(np.sum(c.real ** 2 + c.imag ** 2, axis=2))
    with np.errstate(divide="ignore"):
        strength = 1.0 / (nrm * nrm)
    ang = np.zeros((B, n), dtype=np.float32)
    lvl = np.zeros((B, n), dtype=np.float32)
    for b in range(B):
        a, l = top_n_fast(strength[b], n, res)
        ang[b] = a
        lvl[b] = l
    spec = strength.astype(np.float32) if want_spectrum else None
    return ang, lvl, spec, strength


# --------------------------------------------------------------------------
# seeded synthetic inputs
# --------------------------------------------------------------------------
def array_geometry(m):
    """Unambiguous 2-D arrays in units of array_spacing: m=4 unit
    square; otherwise a uniform circle whose adjacent-element chord is 1 unit."""
    if m == 4:
        return [[0.0, 0.0], [1.0, 0.0], [1.0, 1.0], [0.0, 1.0]]
    r = 0.5 / math.sin(math.pi / m)
    return [[r * math.cos(2 * math.pi * k / m), r * math.sin(2 * math.pi * k / m)]
            for k in range(m)]


GRC_DEFAULT_ULA = [[0, 0], [1, 0], [2, 0], [3, 0]]          # grc/baz_music_doa.xml:55


def steer(theta_deg, antenna_array_scaled, l):
    """Array response of an emitter at theta (same sign convention as the table,
    python/music_doa_helper.py:40-41)."""
    th = theta_deg * np.pi / 180.0
    u = np.array([np.cos(th), np.sin(th)])
    p = np.asarray(antenna_array_scaled, dtype=np.float64)
    return np.exp(-1j * 2.0 * np.pi * (p @ u) / l)


def synth_items(batch, m, nsamples, antenna_array, frequency, array_spacing,
                angles_deg=(40.3, 121.7), snr_db=20.0, seed=1000):
    """(batch, nsamples) complex64, antenna-interleaved in[c*m+r]; uncorrelated
    unit-power complex-Gaussian emitters + AWGN."""
    rng = np.random.default_rng(seed)
    K = nsamples // m
    l = C_LIGHT / frequency
    arr = scaled_array(antenna_array, array_spacing)
    x = np.zeros((batch, K, m), dtype=np.complex128)
    for th in angles_deg:
        s = (rng.standard_normal((batch, K)) + 1j * rng.standard_normal((batch, K))) / np.sqrt(2.0)
        x += s[:, :, None] * steer(th, arr, l)[None, None, :]
    sigma = 10.0 ** (-snr_db / 20.0)
    noise = (rng.standard_normal((batch, K, m)) + 1j * rng.standard_normal((batch, K, m))) / np.sqrt(2.0)
    x += sigma * noise
    return x.reshape(batch, K * m).astype(np.complex64)


# BASELINE.json configs -> concrete parameters
CONFIGS = {
    "cfg1": dict(m=4, n=2, nsamples=256, res=360, seed=1001),
    "cfg2": dict(m=4, n=2, nsamples=1024, res=3600, seed=1002),
    "cfg3": dict(m=8, n=2, nsamples=4096, res=36000, seed=1003),
}
FREQUENCY = 299792458.0   # lambda = 1 m
SPACING = 0.5


def make_config(name, batch, snr_db=20.0, seed=None, angles_deg=(40.3, 121.7)):
    cfg = dict(CONFIGS[name])
    m = cfg["m"]
    arr = array_geometry(m)
    table = steering_table_c64(arr, cfg["res"], FREQUENCY, SPACING)
    items = synth_items(batch, m, cfg["nsamples"], arr, FREQUENCY, SPACING,
                        angles_deg=angles_deg, snr_db=snr_db,
                        seed=cfg["seed"] if seed is None else seed)
    cfg.update(array=arr, table=table, items=items)
    return cfg


def peak_pick(spectrum, n, res=None):
    """EXTENSION, no reference counterpart: the n strongest circular local maxima of one float32
    spectrum row -- b is a peak when s[b] > s[b-1] and s[b] >= s[b+1] -- as (angle_deg, strength) float32 arrays in
    descending strength (earlier bin first on ties), (0, 0) where fewer than n peaks exist.  Definition used by
    tests/test_gpu_parity.py for baz_music_set_peak_mode(ctx, 1)."""
    s = np.asarray(spectrum, dtype=np.float32)
    res = s.shape[0] if res is None else res
    prev, nxt = np.roll(s, 1), np.roll(s, -1)
    with np.errstate(invalid="ignore"):
        is_peak = (s > prev) & (s >= nxt) & (s > 0)
    bins = np.nonzero(is_peak)[0]
    order = sorted(bins.tolist(), key=lambda b: (-float(s[b]), b))[:n]
    ang = np.zeros(n, np.float32)
    lvl = np.zeros(n, np.float32)
    for i, b in enumerate(order):
        ang[i] = np.float32(b * 360.0 / res)
        lvl[i] = s[b]
    return ang, lvl
