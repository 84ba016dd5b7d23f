"""How gnuradio-runtime 3.7 would drive the host block: buffer sizing and call planning as restated in
gr_baz_amd/host/gr_shim/gnuradio/flowgraph_model.h, pinned on hand-computed cases (CPU), and the MUSIC block run through
it -- persistent doubly mapped stream buffers, call sizes decided by the block's hints, optional page-locking of those
buffers -- against the golden vectors (GPU)."""
import numpy as np
import pytest

from conftest import load_golden
from helpers import assert_doa_match, assert_spectrum_close


def _baz():
    from gr_baz_amd import baz
    return baz._native


# --------------------------------------------------------------------------- CPU: the arithmetic of the model
def test_buffer_sizing_follows_allocate_buffer():
    b = _baz()
    # no hints: 2 x 32 KiB per port, rounded UP to page / gcd(item, page) items
    assert b.gr37_buffer_items(8) == 8192                      # gr_complex stream: the familiar 8,192 items
    assert b.gr37_buffer_items(4) == 16384 and b.gr37_buffer_items(2) == 32768 and b.gr37_buffer_items(1) == 65536
    assert b.gr37_buffer_items(24) == 3072                     # 2730 -> granularity 512
    assert b.gr37_buffer_items(14400) == 64                    # 4 items -> granularity 4096 / 64
    # at least two output multiples
    assert b.gr37_buffer_items(8192, output_multiple=64) == 128
    # min output buffer raises, is cut to a multiple; max output buffer clamps and wins over min
    assert b.gr37_buffer_items(14400, 64, 512) == 512
    assert b.gr37_buffer_items(14400, 64, 600) == 576
    assert b.gr37_buffer_items(8, 64, 512) == 8192
    assert b.gr37_buffer_items(8, 64, 512, 1000) == 1024       # min(8192, 1000) -> 960 -> granularity 512
    # a downstream block with a large multiple / decimation / history enlarges the upstream buffer
    assert b.gr37_buffer_items(8192, 1, -1, -1, [(1.0, 64, 1)]) == 130          # cfg2 input: 2 x (64 + 1)
    assert b.gr37_buffer_items(8192, 1, -1, -1, [(1.0, 1024, 1)]) == 2050
    assert b.gr37_buffer_items(8, 1, -1, -1, [(10.0, 1000, 5)]) == 20480        # 2 x (10 x 1000 + 5) = 20010 -> 512s
    assert b.gr37_buffer_items(12, 1, -1, -1, [], 4096) == 6144                 # 5461 -> granularity 1024


def test_call_planning_follows_run_one_iteration():
    b = _baz()
    # output side: at most half a buffer, in multiples; the fuller port decides
    assert b.gr37_plan_noutput(10**6, [8191, 511], [8192, 512], 64) == 256
    assert b.gr37_plan_noutput(10**6, [8191, 100], [8192, 512], 64) == 64
    assert b.gr37_plan_noutput(10**6, [8191, 63], [8192, 512], 64) == 0         # blocked on output
    # input side: what is there, rounded down to the multiple, when that is less
    assert b.gr37_plan_noutput(129, [8191, 511], [8192, 512], 64) == 128
    assert b.gr37_plan_noutput(70, [8191, 511], [8192, 512], 64) == 64
    assert b.gr37_plan_noutput(63, [8191, 511], [8192, 512], 64) == 0           # blocked on input
    # the cap, never below one multiple
    assert b.gr37_plan_noutput(10**6, [8191], [8192], 64, 1, 128) == 128
    assert b.gr37_plan_noutput(10**6, [8191], [8192], 64, 1, 10) == 64
    # ... and NOT rounded to a multiple: the runtime takes min(noutput, max_noutput_items) as it is, so a cap that is no
    # multiple of the output multiple reaches work() -- the host block therefore rounds its cap down to a multiple
    assert b.gr37_plan_noutput(10**6, [8191], [8192], 2, 1, 5) == 5
    # history: noutput + history - 1 inputs needed
    assert b.gr37_plan_noutput(100, [8191], [8192], 1, 11) == 90
    # the reference's block (no hints at all): one page of 8-byte items at a time is NOT the limit, the input is
    assert b.gr37_plan_noutput(7, [8191, 8191, 63], [8192, 8192, 64], 1) == 7


def test_look_back_items_enlarge_the_input_buffer_and_lose_nothing():
    """The MUSIC / AGC blocks' buffer request: history h with output multiple 1.  The upstream buffer is sized for
    2 (1 + h) items, calls grow to what half an output buffer and the input's h - 1 look-back items leave, a stream of ANY
    length arrives complete (the reader starts h - 1 zero items behind the write pointer and treats every real item as
    the newest of its window), and a single item is a valid call."""
    b = _baz()
    item = 24
    for n, h, min_buffer in ((1, 100, 256), (5, 100, 256), (5000, 100, 256), (5000, 1000, 2048), (3000, 64, -1)):
        data = np.random.default_rng(n + h).integers(0, 256, size=item * n, dtype=np.uint8)
        st, outs = b.gr37_model_selftest(data, item, 2, 1, min_buffer, 0, h)
        assert st["items"] == n and st["dropped_at_end"] == 0
        for o in outs:
            assert np.array_equal(o[:n * item], data)
        assert st["in_bufsize"] >= 2 * (1 + h)
        if n == 5000:
            half_out = min(x // 2 for x in st["out_bufsize"])
            assert max(st["call_sizes"]) == min(half_out, st["in_bufsize"] - 1 - (h - 1))
    # the arithmetic behind it
    assert b.gr37_buffer_items(8192, 1, -1, -1, [(1.0, 1, 1025)]) == 2052                    # cfg2 input behind H = 1024
    assert b.gr37_plan_noutput(2051, [2047, 2047, 2047], [2048, 2048, 2048], 1, 1025) == 1024
    assert b.gr37_plan_noutput(1025, [2047], [2048], 1, 1025) == 1                          # one real item: a call
    assert b.gr37_plan_noutput(1024, [2047], [2048], 1, 1025) == 0                          # only look-back: blocked on input


@pytest.mark.parametrize("item, nout, multiple, min_buffer, cap, want_sizes", [
    (24, 2, 64, 512, 0, {1536: 3, 384: 1}),          # half of the 3,072-item buffers, then what is left in multiples
    (24, 1, 1, -1, 0, {1536: 3, 392: 1}),
    (24, 3, 64, 512, 256, {256: 19, 128: 1}),        # capped
    (8192, 1, 64, -1, 0, None),
])
def test_model_moves_every_item_through_wrapping_buffers(item, nout, multiple, min_buffer, cap, want_sizes):
    b = _baz()
    n = 5000 if item < 100 else 700
    data = np.random.default_rng(item + nout).integers(0, 256, size=item * n, dtype=np.uint8)
    st, outs = b.gr37_model_selftest(data, item, nout, multiple, min_buffer, cap)
    done = st["items"]
    assert done == n - n % multiple and st["dropped_at_end"] == n - done        # the tail below one multiple is dropped
    assert sum(k * v for k, v in st["call_sizes"].items()) == done and st["calls"] == sum(st["call_sizes"].values())
    assert all(k % multiple == 0 for k in st["call_sizes"])
    for o in outs:
        assert np.array_equal(o[:done * item], data[:done * item])             # across many wraps of the double mapping
    if want_sizes is not None:
        assert st["call_sizes"] == want_sizes
    else:      # cfg2's input item on both sides: 130-item input buffer, 128-item output buffer -> half of it per call
        assert st["in_bufsize"] == 130 and st["out_bufsize"] == [128] and set(st["call_sizes"]) == {64}


# --------------------------------------------------------------------------- GPU: the MUSIC block under the model
@pytest.mark.gpu
@pytest.mark.parametrize("name, n_outputs", [("cfg1_m4_n2_N256_r360", 3), ("cfg2_m4_n2_N1024_r3600", 3), ("cfg2_m4_n2_N1024_r3600", 2),
                                             ("cfg5_m16_n2_N4096_r3600", 3)])
@pytest.mark.parametrize("pin", [False, True])
@pytest.mark.parametrize("lookback", ["0", "5"])
def test_music_block_under_the_scheduler_model(name, n_outputs, pin, lookback, gpu_device, monkeypatch):
    g = load_golden(name)
    monkeypatch.setenv("BAZ_MUSIC_OUTPUT_MULTIPLE", "4")       # the goldens hold a few dozen items: many calls, many wraps
    monkeypatch.setenv("BAZ_MUSIC_MIN_OUTPUT_BUFFER", "8")
    monkeypatch.setenv("BAZ_MUSIC_INPUT_LOOKBACK", lookback)   # 5: the look-back window travels through the wraps too
    from gr_baz_amd import baz
    blk = baz.music_doa(g["m"], g["n"], g["nsamples"], [list(map(complex, r)) for r in g["table"]], g["res"])
    assert blk.pin_buffers() is False and blk.pinned_bytes() == 0
    reps = 6
    items = np.concatenate([g["items"]] * reps)
    k = items.shape[0] - items.shape[0] % 4
    st, ang, lvl, spec = blk.run_flowgraph(items, n_outputs, True, pin)
    assert st["items"] == k and st["last_return"] > 0 and st["calls"] >= (2 if lookback == "0" else 1)
    assert all(c % 4 == 0 for c in st["call_sizes"])
    want = lambda key: np.concatenate([g[key]] * reps)[:k]
    if n_outputs > 2:
        assert_spectrum_close(spec[:k], want("spectrum"))
    assert_doa_match(ang[:k], lvl[:k], want("ang"), want("lvl"), g["res"], want("strength64"))
    # the same items in one plain work() call: identical bits, whatever the call sizes and buffer kind were
    p, a1, l1, s1 = blk.work(items[:k], n_outputs)
    assert p == k and np.array_equal(a1, ang[:k]) and np.array_equal(l1, lvl[:k])
    if n_outputs > 2:
        assert np.array_equal(s1, spec[:k])
    if pin:
        assert st["pinned_bytes_at_stop"] > 0                  # the stream buffers were page-locked during the run ...
    else:
        assert st["pinned_bytes_at_stop"] == 0
    assert blk.pinned_bytes() == 0 and blk.pin_buffers() is False             # ... and released by stop()


def test_model_invariants_on_random_hints():
    """Random item sizes, hints and lengths: every item arrives once and in order through the wrapping buffers, every call
    is a multiple, no call exceeds half an output buffer, the cap (never below one multiple) or what the input buffer
    can hold, and only a tail below one multiple is dropped."""
    from hypothesis import given, settings, strategies as st

    b = _baz()

    @settings(max_examples=60, deadline=None)
    @given(item=st.sampled_from([4, 8, 12, 24, 100, 1000, 8192, 14400]), nout=st.integers(1, 3),
           multiple=st.sampled_from([1, 2, 3, 7, 16, 64, 100]), min_buffer=st.sampled_from([-1, 10, 512, 3000]),
           capm=st.sampled_from([0, 1, 2, 9]), n=st.integers(1, 3000), seed=st.integers(0, 2**31 - 1))
    def check(item, nout, multiple, min_buffer, capm, n, seed):
        cap = capm * multiple            # (a cap that is no multiple of the output multiple is passed through: see above)
        data = np.random.default_rng(seed).integers(0, 256, size=item * n, dtype=np.uint8)
        stt, outs = b.gr37_model_selftest(data, item, nout, multiple, min_buffer, cap)
        done = stt["items"]
        assert done == n - n % multiple and stt["dropped_at_end"] == n - done
        for o in outs:
            assert np.array_equal(o[:done * item], data[:done * item])
        limit = min(x // 2 for x in stt["out_bufsize"])
        if cap > 0:
            limit = min(limit, max(cap, multiple))
        limit = min(limit, stt["in_bufsize"] - 1)
        for size, count in stt["call_sizes"].items():
            assert size % multiple == 0 and 0 < size <= limit and count > 0
        assert sum(k * v for k, v in stt["call_sizes"].items()) == done
        assert all(x >= 2 * multiple for x in stt["out_bufsize"]) and stt["in_bufsize"] >= 2 * (multiple + 1)

    check()


def test_stand_in_module_exposes_the_scheduling_surface():
    b = _baz()
    for name in ("gr37_buffer_items", "gr37_plan_noutput", "gr37_model_selftest"):
        assert callable(getattr(b, name))
    for name in ("run_flowgraph", "set_pin_buffers", "pin_buffers", "pinned_bytes", "output_multiple", "min_output_buffer",
                 "max_noutput_items", "work"):
        assert hasattr(b.baz_music_doa_sptr, name), name
    for name in ("output_multiple", "min_output_buffer", "work"):
        assert hasattr(b.baz_agc_cc_sptr, name), name
    # what the AGC block's hints buy under the runtime's rules: 4,096-sample calls without them, 16,384-sample calls with its
    # 16,384 look-back samples and 32,768-sample output buffers -- and a 3-sample capture is still a call
    assert b.gr37_plan_noutput(8191, [8191, 16383], [8192, 16384], 1) == 4096
    out = [b.gr37_buffer_items(8, 1, 2 * 16384), b.gr37_buffer_items(4, 1, 2 * 16384)]
    assert out == [32768, 32768]
    upstream = b.gr37_buffer_items(8, 1, -1, -1, [(1.0, 1, 16385)])
    assert upstream == 32772 + (-32772) % 512 and b.gr37_plan_noutput(upstream - 1, [x - 1 for x in out], out, 1, 16385) == 16384
    assert b.gr37_plan_noutput(16384 + 3, [x - 1 for x in out], out, 1, 16385) == 3
