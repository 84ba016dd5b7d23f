"""MUSIC DOA helper -- python-3 counterpart of /root/reference/python/music_doa_helper.py.

Same public surface: unit_vect(theta), calculate_antenna_array_response(antenna_array,
angular_resolution, l) (:29-46) and class music_doa_helper(m, n, nsamples, angular_resolution,
frequency, array_spacing, antenna_array, output_spectrum=False) with set_frequency() (:48-103), used
by grc/baz_music_doa.xml:6-8.  The steering-table formula, wavelength, array scaling, the
nsamples % m check and the 2-or-3 output signature are the reference's (python-3 syntax).

With GNU Radio installed the class is a gr.hier_block2 wired exactly like the reference (:61-96).
Without it (this container) it is a plain object with the same attributes plus work(items), so the
helper -> baz.music_doa -> host block -> C-ABI -> HIP chain can be exercised end to end.
"""
import os

import numpy

try:                                    # real GNU Radio host
    from gnuradio import gr             # noqa: F401
    _HAVE_GR = True
except ImportError:                     # this container / the GPU box
    gr = None
    _HAVE_GR = False

C_LIGHT = 299792458.0                   # python/music_doa_helper.py:55


def unit_vect(theta):
    return numpy.array([numpy.cos(theta), numpy.sin(theta)])


def _fma(a, b, c):
    """round(a * b + c) in float64 without an fma instruction: the product exactly as p + e (Veltkamp / Dekker), the sum of the
    three terms by a compensated addition.  (Not correctly rounded in every halfway case; it is only ever used to CONFIRM a table
    bit for bit, so an error here can only send the helper to its per-element loop.)"""
    p = a * b
    ta, tb = 134217729.0 * a, 134217729.0 * b
    ah = ta - (ta - a)
    bh = tb - (tb - b)
    al, bl = a - ah, b - bh
    e = ((ah * bh - p) + ah * bl + al * bh) + al * bl
    s = p + c
    bb = s - p
    t = (p - (s - bb)) + (c - bb)
    return s + (t + e)


# u0 a0 + u1 a1 as a BLAS dot of length two may round it: the second product fused onto the first, the first onto the second, or unfused
_TWO_TERM_FORMS = (lambda u0, a0, u1, a1: _fma(u1, a1, u0 * a0),
                   lambda u0, a0, u1, a1: _fma(u0, a0, u1 * a1),
                   lambda u0, a0, u1, a1: u0 * a0 + u1 * a1)


def _phase_by_element(antennas, angular_resolution, l, steps=None):
    """The reference's loop (python/music_doa_helper.py:35-41): one numpy.inner per (step, antenna)."""
    steps = range(0, angular_resolution) if steps is None else steps
    phase = numpy.empty((len(steps), len(antennas)), dtype=numpy.float64)
    for k, step in enumerate(steps):
        angle = (step * 360.0 / angular_resolution) * (numpy.pi / 180.0)
        u = unit_vect(angle)
        for t, antenna in enumerate(antennas):
            phase[k, t] = numpy.inner(antenna, u) / l
    return phase


def calculate_antenna_array_response(antenna_array, angular_resolution, l):
    """response[step][antenna] = exp(-j 2 pi (p_antenna . u(theta_step)) / l), theta_step =
    step*360/angular_resolution degrees (python/music_doa_helper.py:32-46).  Returns a list of lists of
    python complex, which is what baz.music_doa's vector<vector<gr_complex>> typemap takes.

    The reference forms every phase with its own numpy.inner call: 288,000 of them at 8 antennas x 36,000 bins, 0.4 - 1.1 s
    per retune -- next to a set_array_response that takes 0.24 ms.  Here ONE numpy.inner over all steps and
    antennas forms them (the same two-term dot product per element, by the same BLAS: bit-identical wherever it was tried,
    20 x faster).  It is accepted only when (i) a sample of 64 rows recomputed the reference's way agrees bit for bit and (ii)
    EVERY element agrees bit for bit with a deterministic restatement of the rounding the sample shows (fused either way, or
    unfused: _TWO_TERM_FORMS) -- a BLAS whose blocked product, thread partition or tail rounds some rows differently fails
    (ii); in every other case the whole table is formed per element.  BAZ_MUSIC_HELPER_PER_ELEMENT=1 forces that loop."""
    antennas = [numpy.asarray(a, dtype=numpy.float64) for a in antenna_array]
    res = int(angular_resolution)
    phase = None
    if (res > 0 and len(antennas) > 0 and all(a.shape == (2,) for a in antennas)
            and os.environ.get("BAZ_MUSIC_HELPER_PER_ELEMENT", "0") in ("", "0")):     # (=1: the reference's loop, whatever it costs)
        angle = (numpy.arange(res) * 360.0 / res) * (numpy.pi / 180.0)
        u = numpy.stack([numpy.cos(angle), numpy.sin(angle)], axis=1)
        A = numpy.stack(antennas)
        fast = numpy.inner(u, A) / l
        sample = sorted(set([0, res - 1] + [(k * 2654435761) % res for k in range(1, 63)]))
        by_element = _phase_by_element(antennas, res, l, sample)
        if numpy.array_equal(fast[sample].view(numpy.int64), by_element.view(numpy.int64)):
            # The sample says what ONE dot product of this BLAS rounds like.  The matrix product may still round some block,
            # thread partition or tail differently: the WHOLE table is therefore compared with a deterministic
            # restatement of that rounding -- whichever of the three two-term forms reproduces the sample -- and any differing
            # element sends the table through the per-element loop.
            for form in _TWO_TERM_FORMS:
                cand = form(u[:, 0:1], A[None, :, 0], u[:, 1:2], A[None, :, 1]) / l
                if numpy.array_equal(cand[sample].view(numpy.int64), by_element.view(numpy.int64)):
                    if numpy.array_equal(cand.view(numpy.int64), fast.view(numpy.int64)):
                        phase = fast
                    break
    if phase is None:
        phase = _phase_by_element(antennas, res, l)
    response = numpy.exp(-1j * 2.0 * numpy.pi * phase)
    return response.tolist()


def _banner(h):
    print("MUSIC DOA Helper: M: %d, N: %d, # samples: %d, steps of %f degress, lambda: %f, array: %s" % (
        h.m, h.n, h.nsamples, (360.0 / h.angular_resolution), h.l, str(h.antenna_array)))


def _configure(h, m, n, nsamples, angular_resolution, frequency, array_spacing, antenna_array):
    h.m = m
    h.n = n
    h.nsamples = nsamples
    h.angular_resolution = angular_resolution
    h.l = C_LIGHT / frequency
    h.antenna_array = [[array_spacing * x, array_spacing * y] for [x, y] in antenna_array]
    if (nsamples % m) != 0:
        raise Exception("nsamples must be multiple of m")


def _make_impl(h):
    from . import music_doa as _music_doa
    h.array_response = calculate_antenna_array_response(h.antenna_array, h.angular_resolution, h.l)
    h.impl = _music_doa(h.m, h.n, h.nsamples, h.array_response, h.angular_resolution)


def _retune(h, frequency):
    h.l = C_LIGHT / frequency
    h.array_response = calculate_antenna_array_response(h.antenna_array, h.angular_resolution, h.l)
    h.impl.set_array_response(h.array_response)


if _HAVE_GR:

    class music_doa_helper(gr.hier_block2):
        def __init__(self, m, n, nsamples, angular_resolution, frequency, array_spacing, antenna_array,
                     output_spectrum=False):
            _configure(self, m, n, nsamples, angular_resolution, frequency, array_spacing, antenna_array)
            if output_spectrum:
                output_sig = gr.io_signature3(3, 3, (gr.sizeof_float * n), (gr.sizeof_float * n),
                                              (gr.sizeof_float * angular_resolution))
            else:
                output_sig = gr.io_signature2(2, 2, (gr.sizeof_float * n), (gr.sizeof_float * n))
            gr.hier_block2.__init__(self, "music_doa_helper",
                                    gr.io_signature(1, 1, (gr.sizeof_gr_complex * nsamples)), output_sig)
            _banner(self)
            _make_impl(self)
            self.connect(self, self.impl)
            self.connect((self.impl, 0), (self, 0))
            self.connect((self.impl, 1), (self, 1))
            if output_spectrum:
                self.connect((self.impl, 2), (self, 2))

        def set_frequency(self, frequency):
            _retune(self, frequency)

else:

    class music_doa_helper(object):
        """GNU-Radio-less stand-in with the reference's constructor, attributes and set_frequency()."""

        def __init__(self, m, n, nsamples, angular_resolution, frequency, array_spacing, antenna_array,
                     output_spectrum=False):
            _configure(self, m, n, nsamples, angular_resolution, frequency, array_spacing, antenna_array)
            self.output_spectrum = bool(output_spectrum)
            # (itemsize, ...) of the ports the hier block would declare (:61-71)
            self.input_item_sizes = [8 * nsamples]
            self.output_item_sizes = [4 * n, 4 * n] + ([4 * angular_resolution] if output_spectrum else [])
            _banner(self)
            _make_impl(self)

        def set_frequency(self, frequency):
            _retune(self, frequency)

        def work(self, items):
            """Runs the wrapped block on (k, nsamples) complex64 items: returns (ang, lvl[, spectrum])."""
            produced, ang, lvl, spec = self.impl.work(numpy.ascontiguousarray(items, dtype=numpy.complex64),
                                                      3 if self.output_spectrum else 2)
            if produced != ang.shape[0]:
                raise RuntimeError("music_doa work() returned %d" % produced)
            return (ang, lvl, spec) if self.output_spectrum else (ang, lvl)
