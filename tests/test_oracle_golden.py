"""Pins the oracle (oracle/mls_oracle.py + oracle/mls_oracle.c) to the real
reference through the fixtures tests/golden/make_golden.py captured from it, and
to the one known-answer example the reference carries itself (mls.py:179-205).
CPU only."""
import json
import os

import numpy as np
import pytest

import inputs
from oracle import c_oracle, mls_oracle

G = os.path.join(os.path.dirname(__file__), "golden")


def load_json(name):
    with open(os.path.join(G, name)) as f:
        return json.load(f)


KATS = load_json("seq_kat.json")
IR_SMALL = load_json("ir_small.json")["cases"]
IR_LARGE = load_json("ir_large.json")["cases"]


def test_reference_comment_example():
    # mls.py:179-205: N=3, taps (3,2)
    s = mls_oracle.full_sequence(3, (3, 2))
    assert s.shape == (7, 1) and s.dtype == np.float64
    assert s[:, 0].tolist() == [1, -1, -1, -1, 1, 1, -1]
    lin = np.convolve(s[::-1, 0], s[:, 0])
    assert np.allclose(lin, [-1, 2, 1, -2, -3, 0, 7, 0, -3, -2, 1, 2, -1])
    h = mls_oracle.get_impulse(3, (3, 2), np.tile(s, (2, 1)))
    assert h.shape == (7, 1)
    assert np.allclose(h[:, 0], [1] + [-1 / 7] * 6, atol=1e-15)


@pytest.mark.parametrize("kat", [k for k in KATS["kats"] if k["N"] <= 14],
                         ids=lambda k: "N%d-%s" % (k["N"], "_".join(map(str, k["taps"]))))
def test_python_oracle_sequence_kat(kat):
    packed = mls_oracle.packed_bits(kat["N"], tuple(kat["taps"]))
    assert packed.size == kat["nbytes"]
    assert inputs.digest(packed) == kat["sha256"]


@pytest.mark.parametrize("kat", KATS["kats"],
                         ids=lambda k: "N%d-%s" % (k["N"], "_".join(map(str, k["taps"]))))
def test_c_oracle_sequence_kat(kat):
    N, taps = kat["N"], kat["taps"]
    packed = c_oracle.bits_packed(N, taps)
    assert inputs.digest(packed) == kat["sha256"]
    bits = np.unpackbits(packed, bitorder="little")[:(1 << N) - 1]
    assert int(bits.sum()) == kat["ones"] == 1 << (N - 1)
    assert "".join(map(str, bits[:64])) == kat["first64"]


PREFIXES = load_json("seq_prefix.json")["prefixes"]


@pytest.mark.parametrize("kat", PREFIXES, ids=lambda k: "N%d-%s" % (k["N"], "_".join(map(str, k["taps"]))))
def test_c_oracle_sequence_prefix_of_the_largest_orders(kat):
    """N = 25..32 (the end of the reference's tap table): the first 2^20 bits as the reference's generator_chunk emits
    them, from both C restatements (byte cells, and the register in one word that the N = 31, 32 GPU tests lean on)."""
    N, taps, nbits = kat["N"], kat["taps"], kat["nbits"]
    for fn in (c_oracle.bits_packed, c_oracle.bits_packed_word):
        packed = fn(N, taps, nbits)
        assert inputs.digest(packed) == kat["sha256"], fn.__name__
    bits = np.unpackbits(packed, bitorder="little")[:nbits]
    assert int(bits.sum()) == kat["ones"] and "".join(map(str, bits[:64])) == kat["first64"]


@pytest.mark.parametrize("kat", [k for k in KATS["kats"] if k["N"] <= 20],
                         ids=lambda k: "N%d-%s" % (k["N"], "_".join(map(str, k["taps"]))))
def test_word_register_oracle_sequence_kat(kat):
    packed = c_oracle.bits_packed_word(kat["N"], kat["taps"])
    assert inputs.digest(packed) == kat["sha256"]
    # ... and on across the period boundary, like the cell version
    n = (1 << kat["N"]) + 77
    assert np.array_equal(c_oracle.bits_packed_word(kat["N"], kat["taps"], n), c_oracle.bits_packed(kat["N"], kat["taps"], n))


SEQ_ANY = load_json("seq_any.json")


@pytest.mark.parametrize("kat", SEQ_ANY["sets"], ids=lambda k: "N%d-%s" % (k["N"], "_".join(map(str, k["taps"]))))
def test_oracles_on_tap_sets_outside_the_table(kat):
    """Taps the reference accepts without being in its table (not primitive, without N, N+1..2N through the negative
    deque index, a tap twice): the first L bits and 3L+17 bits of the running register, from all three restatements."""
    N, taps, L = kat["N"], tuple(kat["taps"]), (1 << kat["N"]) - 1
    first = np.packbits(((1.0 - mls_oracle.full_sequence(N, taps)[:, 0]) / 2).astype(np.uint8), bitorder="little")
    assert inputs.digest(first) == kat["sha256"]
    run = next(mls_oracle.chunk_stream(N, taps, kat["run_bits"]))[:, 0].astype(np.uint8) if N <= 12 else None
    for fn in (c_oracle.bits_packed, c_oracle.bits_packed_word):
        assert inputs.digest(fn(N, list(taps), L)) == kat["sha256"], fn.__name__
        packed = fn(N, list(taps), kat["run_bits"])
        assert inputs.digest(packed) == kat["run_sha256"], fn.__name__
        bits = np.unpackbits(packed, bitorder="little")[:kat["run_bits"]]
        assert int(bits.sum()) == kat["run_ones"] and int(bits[:L].sum()) == kat["ones"]
        assert bool(np.array_equal(bits[L:2 * L], bits[:L])) == kat["wraps_at_L"]
        if run is not None:
            assert np.array_equal(bits, run)


def test_oracles_refuse_taps_the_register_cannot_index():
    for e in SEQ_ANY["errors"]:
        assert e["generator_bit"] == e["get_full_sequence"] == "IndexError"
        with pytest.raises(IndexError):
            next(mls_oracle.bit_stream(e["N"], tuple(e["taps"])))
        with pytest.raises(Exception):
            c_oracle.bits_packed(e["N"], e["taps"], 8)


def test_c_oracle_matches_python_oracle_samples():
    for N, taps in ((5, (5, 4, 3, 2)), (9, (9, 5)), (12, (12, 11, 10, 4))):
        a = mls_oracle.full_sequence(N, taps, 3)
        b = c_oracle.sequence(N, taps, 3)
        assert a.shape == b.shape and np.array_equal(a, b)


def test_generator_chunk_continuity():
    g = KATS["generator_chunk"]
    gen = mls_oracle.chunk_stream(g["N"], tuple(g["taps"]), g["chunk"])
    got = [int(v) for _ in range(4) for v in next(gen)[:, 0]]
    assert got == g["bits"]
    assert got == c_oracle.bits(g["N"], g["taps"], len(got)).tolist()
    gs = mls_oracle.sample_stream(g["N"], tuple(g["taps"]), 7)
    sam = [float(v) for _ in range(3) for v in next(gs)[:, 0]]
    assert sam == g["samples_chunk7"]


def _case_input(case, arrays=None):
    seq = c_oracle.sequence(case["N"], case["taps"])[:, 0]
    x = inputs.make_input(case["kind"], seq, case["R"], case["seed"], case["shape"], case["dtype"])
    assert inputs.digest(x) == case["x_sha256"], "input generator drifted from the one the golden was made with"
    if arrays is not None and case.get("x_stored"):
        assert np.array_equal(arrays[case["id"] + "_x"], x)
    return x


@pytest.mark.parametrize("case", IR_SMALL, ids=lambda c: "%s-N%d-%s" % (c["id"], c["N"], c["kind"]))
def test_oracle_ir_small(case):
    arrays = np.load(os.path.join(G, "ir_small.npz"))
    x = _case_input(case, arrays)
    L = (1 << case["N"]) - 1
    assert str(mls_oracle.trim_and_average(x, L).dtype) == case["avg_dtype"]
    h = mls_oracle.get_impulse(case["N"], tuple(case["taps"]), x)
    assert h.shape == (L, 1) and h.dtype == np.float64
    want = arrays[case["id"] + "_h"]
    # same arithmetic as the reference -> bit-identical on the same numpy/scipy
    assert np.max(np.abs(h[:, 0] - want)) <= 1e-15 * case["peak"]
    if case["N"] <= 11:
        avg = mls_oracle.trim_and_average(x, L)
        ref = c_oracle.sequence(case["N"], case["taps"])
        direct = c_oracle.xcorr_direct(avg.astype(np.float64), ref)
        # float32 captures: the reference's fftconvolve transforms the float32 average with a
        # single-precision FFT (complex64), so the REFERENCE itself is only ~1e-7*peak from the
        # exact correlation there; float64/int captures agree to round-off.
        tol = 1e-6 if case["dtype"] == "float32" else 1e-12
        assert np.max(np.abs(direct - want)) <= tol * case["peak"]


@pytest.mark.parametrize("case", [c for c in IR_LARGE if c["N"] <= 18],
                         ids=lambda c: "%s-N%d-%s" % (c["id"], c["N"], c["kind"]))
def test_oracle_ir_large_sparse(case):
    arrays = np.load(os.path.join(G, "ir_large.npz"))
    x = _case_input(case)
    ref = c_oracle.sequence(case["N"], case["taps"])
    h = mls_oracle.get_impulse(case["N"], tuple(case["taps"]), x, ref=ref)[:, 0]
    tol = 1e-13 * case["peak"]
    assert np.max(np.abs(h[arrays[case["id"] + "_idx"]] - arrays[case["id"] + "_h"])) <= tol
    lo = case["peakwin_lo"]
    assert np.max(np.abs(h[lo:lo + 256] - arrays[case["id"] + "_peakwin"])) <= tol
    assert int(np.argmax(np.abs(h))) == case["argmax"]
    assert abs(float(np.sum(h)) - case["sum"]) <= 1e-9 * max(1.0, abs(case["sum"]))


def test_oracle_error_table():
    table = load_json("errors.json")
    N, taps = 5, (5, 3)
    L = 31
    good = mls_oracle.full_sequence(N, taps, 3)

    def name(fn):
        try:
            fn()
        except BaseException as e:  # noqa: BLE001
            return type(e).__name__
        return None

    assert name(lambda: mls_oracle.check_ctor_args(None, taps)) == table["ctor_N_none"]["type"]
    assert name(lambda: mls_oracle.check_ctor_args(N, None)) == table["ctor_taps_none"]["type"]
    assert name(lambda: mls_oracle.check_ctor_args(N, ())) == table["ctor_taps_empty"]["type"]
    assert name(lambda: mls_oracle.check_ctor_args(N, np.array(taps))) == table["ctor_taps_ndarray"]["type"]
    assert name(lambda: mls_oracle.check_ctor_args(N, list(taps))) == table["ctor_taps_list_ok"]["type"]
    assert name(lambda: mls_oracle.get_impulse(N, taps, good[:L])) == table["impulse_len_eq_L"]["type"]
    assert name(lambda: mls_oracle.get_impulse(N, taps, np.zeros(2 * L + 3))) == \
        table["impulse_len_not_multiple"]["type"]
    assert name(lambda: mls_oracle.get_impulse(N, taps, np.zeros((2 * L, 2)))) == \
        table["impulse_two_channels"]["type"]
    assert name(lambda: mls_oracle.get_impulse(N, taps, list(good[:, 0]))) == table["impulse_list"]["type"]
    assert name(lambda: mls_oracle.get_impulse(N, taps, good)) is None


def test_audio_fft_restatement_against_reference_spectra():
    """oracle.audio_fft vs spectra captured from the reference's Audio.fft (tests/golden/make_golden.py: dump_fft),
    including the Mersenne-prime size bump. Same numpy/scipy routines, so bit-for-bit."""
    meta = json.load(open(os.path.join(G, "fft.json")))
    arrays = np.load(os.path.join(G, "fft.npz"))
    for c in meta["cases"]:
        rng = np.random.default_rng(c["seed"])
        x = rng.standard_normal((c["n"], c["ch"])) * np.exp(-np.arange(c["n"]) / (c["n"] / 8.0))[:, None]
        assert inputs.digest(x) == c["x_sha256"]
        frq, mag = mls_oracle.audio_fft(x, c["fs"], c["window"], c["normalise"])
        assert mag.shape == (c["bins"], c["ch"])
        assert np.array_equal(frq, arrays[c["id"] + "_frq"])
        assert np.allclose(mag, arrays[c["id"] + "_mag"], rtol=0, atol=1e-9)


def test_oracle_chain_against_reference_mls_simple_short_orders():
    """lfilter + the oracle's get_impulse against the real reference's MLS_simple.get_impulse at N = 5..10
    (tests/golden/simple_short.*): pins the de-emphasis + impulse chain the GPU tests compare the fused path with."""
    import scipy.signal
    from zignal_amd.filters.biquads import RBJ
    cases = load_json("simple_short.json")["cases"]
    arrays = np.load(os.path.join(G, "simple_short.npz"))
    from zignal_amd.measure.mlstaps import TAPS
    for c in cases:
        B, A = RBJ(filtertype="highshelf", gaindb=-10, f0=100, Q=0.707, fs=c["fs"]).get_coefficients()
        x = arrays[c["id"] + "_x"]
        assert inputs.digest(x) == c["x_sha256"]
        y = scipy.signal.lfilter(A, B, x, axis=0)                  # de-emphasis = the inverse filter (mls.py:366)
        h = mls_oracle.get_impulse(c["N"], TAPS[c["N"]][0], y)
        want = arrays[c["id"] + "_ir"]
        assert np.max(np.abs(h - want)) <= 1e-12 * np.max(np.abs(want))
