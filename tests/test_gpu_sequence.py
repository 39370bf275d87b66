"""GPU parity, integer part: the LFSR kernel's bit sequence must equal the reference's bit for
bit (golden SHA-256 digests captured from the reference + the C oracle), and the permutation
tables it builds must be the bijections SURVEY.md §A.2 defines."""
import numpy as np
import pytest

import inputs
from gpu_common import load_json, plan
from oracle import c_oracle
from zignal_amd.measure.mls import TAPS, _MLS_base

pytestmark = pytest.mark.gpu

KATS = load_json("seq_kat.json")


def _kat_id(k):
    return "N%d-%s" % (k["N"], "_".join(map(str, k["taps"])))


# every tap set for N <= 8, then first/last(/middle) per order up to 24
SEL = [k for k in KATS["kats"] if k["N"] <= 8 or k["N"] >= 13] + \
      [k for i, k in enumerate(KATS["kats"]) if 9 <= k["N"] <= 12 and i % 7 == 0]


@pytest.mark.parametrize("kat", SEL, ids=_kat_id)
def test_sequence_bits_match_reference_digest(kat):
    p = plan(kat["N"], kat["taps"])
    packed = p.bits_packed()
    assert packed.size == kat["nbytes"]
    assert inputs.digest(packed) == kat["sha256"]
    bits = np.unpackbits(packed, bitorder="little")[:p.L]
    assert int(bits.sum()) == kat["ones"]
    assert "".join(map(str, bits[:64])) == kat["first64"]


@pytest.mark.parametrize("N", [25, 26])
def test_sequence_beyond_reference_reach_matches_c_oracle(N):
    # the reference's Python loop would need minutes here; the C oracle runs the same recurrence
    taps = TAPS[N][0]
    assert np.array_equal(plan(N, taps).bits_packed(), c_oracle.bits_packed(N, taps))


PREFIXES = load_json("seq_prefix.json")["prefixes"]


def _ones(packed, nbits):
    full, total = nbits // 8, 0
    for i in range(0, full, 1 << 26):
        total += int(np.bitwise_count(packed[i:min(full, i + (1 << 26))]).sum(dtype=np.int64))
    if nbits % 8:
        total += bin(int(packed[full]) & ((1 << (nbits % 8)) - 1)).count("1")
    return total


def _bit(packed, i):
    return (int(packed[i >> 3]) >> (i & 7)) & 1


@pytest.mark.parametrize("kat", [k for k in PREFIXES if k["N"] <= 30], ids=_kat_id)
def test_sequence_prefix_of_large_orders_matches_reference_digest(kat):
    """N = 25..30: the first 2^20 bits against the digest of what the reference's own generator_chunk emits"""
    from zignal_amd.device import Plan
    p = Plan(kat["N"], kat["taps"], device=0)
    packed = p.bits_packed(kat["nbits"])
    p.close()
    assert inputs.digest(packed) == kat["sha256"]


@pytest.mark.parametrize("N", [31, 32])
def test_orders_31_and_32_generate_the_sequence_and_decline_the_transform(N):
    """The end of zignal's tap table (mlstaps.py): the generator covers it - the whole 2^N - 1 bit period equals the C
    oracle's (word register; its first 2^20 bits are pinned to the reference's in tests/test_oracle_golden.py) and the
    reference's own first 2^20 bits - while impulse responses end at N = 30 (the workspace of one channel slab alone
    would be 137 / 275 GB) and say so."""
    import ctypes
    from zignal_amd import _lib
    from zignal_amd.device import DeviceBuffer, Plan
    for taps in (TAPS[N][0], TAPS[N][-1]):
        p = Plan(N, taps, device=0)
        L = p.L
        kat = [k for k in PREFIXES if k["N"] == N and tuple(k["taps"]) == tuple(taps)][0]
        assert inputs.digest(p.bits_packed(kat["nbits"])) == kat["sha256"]
        if taps == TAPS[N][0]:
            got = p.bits_packed(L + 32)               # one period and on across its end (mls.py:117-128)
            want = c_oracle.bits_packed_word(N, taps, L + 32)
            assert got.size == want.size and np.array_equal(got, want)
            # a maximum-length sequence has 2^(N-1) ones per period and closes on itself
            assert _ones(got, L) == 1 << (N - 1)
            assert [_bit(got, L + i) for i in range(32)] == [_bit(got, i) for i in range(32)]
            del want
        assert p.info()["passes"] == 0
        d = DeviceBuffer(1 << 20)
        rc = _lib.lib().zmls_impulse_device(p._h, d.ptr, 0, 2 * L, 2, 2, 1, d.ptr, 2, 1, None)
        assert rc == _lib.E_UNSUPPORTED and "up to order 30" in _lib.last_error()
        assert _lib.lib().zmls_plan_reserve(p._h, 2) == _lib.E_UNSUPPORTED
        # the +1/-1 stimulus on the device follows the bits (first 2^16 samples of a float32 copy)
        if taps == TAPS[N][0]:
            d_s = DeviceBuffer(L * 4)
            _lib.check(_lib.lib().zmls_sequence_device(p._h, d_s.ptr, _lib.ZMLS_F32, 1, 1, None))
            s = d_s.download((1 << 16,), np.float32)
            bits = np.unpackbits(got[:1 << 13], bitorder="little")
            assert np.array_equal(s, 1.0 - 2.0 * bits.astype(np.float32))
            d_s.free()
        d.free()
        p.close()
    with pytest.raises(_lib.ZmlsError):
        Plan(33, (33, 20), device=0)


def test_sequence_samples_and_generators():
    g = KATS["generator_chunk"]
    m = _MLS_base(g["N"], tuple(g["taps"]), device=0)
    gen = m.generator_chunk(g["chunk"])
    blocks = [next(gen) for _ in range(4)]
    assert blocks[0].shape == (g["chunk"], 1) and blocks[0].dtype == np.int64
    assert [int(v) for b in blocks for v in b[:, 0]] == g["bits"]
    gs = m.generator_samples(7)
    sam = [next(gs) for _ in range(3)]
    assert sam[0].dtype == np.float64 and sam[0].shape == (7, 1)
    assert [float(v) for b in sam for v in b[:, 0]] == g["samples_chunk7"]
    gb = m.generator_bit()
    assert [next(gb) for _ in range(80)] == g["bits"]
    s = m.get_full_sequence(3)
    assert s.shape == (3 * m.L, 1) and s.dtype == np.float64 and s.flags.c_contiguous
    assert np.array_equal(s, c_oracle.sequence(g["N"], g["taps"], 3))
    # running past the period boundary through the C ABI (mls.py:117-128)
    p = plan(g["N"], g["taps"])
    nb = p.L + 32
    assert np.array_equal(p.bits_packed(nb), c_oracle.bits_packed(g["N"], g["taps"], nb))


@pytest.mark.parametrize("N,taps", [(3, (3, 2)), (8, (8, 7, 6, 1)), (11, TAPS[11][5]), (16, TAPS[16][0]),
                                     (20, TAPS[20][0]), (22, TAPS[22][-1])])
def test_permutation_tables(N, taps):
    p = plan(N, taps)
    L = p.L
    q, m = p.tables()
    assert q[0] == 0xFFFFFFFF and m[0] == 0xFFFFFFFF
    # bijections onto 0..L-1
    assert np.array_equal(np.sort(q[1:]), np.arange(L, dtype=np.uint32))
    assert np.array_equal(np.sort(m[1:]), np.arange(L, dtype=np.uint32))
    bits = np.unpackbits(c_oracle.bits_packed(N, taps, L + N), bitorder="little").astype(np.uint32)
    # Ps[j] = N-bit window of the sequence starting at j  ->  Q[Ps[j]] = j
    rng = np.random.default_rng(N)
    js = np.concatenate(([0, 1, L - 1], rng.integers(0, L, 2000)))
    for j in js:
        w = int(sum(int(bits[j + i]) << i for i in range(N)))
        assert q[w] == j
    # Hadamard identity: a[(j - mm) mod L] = parity(Ps[j] & Pl[mm]), checked on random pairs
    ps = np.empty(L, dtype=np.uint32)
    ps[q[1:]] = np.arange(1, L + 1, dtype=np.uint32)
    pl = np.empty(L, dtype=np.uint32)
    pl[m[1:]] = np.arange(1, L + 1, dtype=np.uint32)
    jj = rng.integers(0, L, 20000)
    mm = rng.integers(0, L, 20000)
    par = np.bitwise_count(ps[jj] & pl[mm]) & 1
    assert np.array_equal(par.astype(np.uint32), bits[(jj - mm) % L])


def test_non_maximal_taps_generate_the_sequence_and_refuse_the_transform():
    """x^8 + x^4 + 1 is not primitive: the reference generates its (short-period) sequence all the same, and so does the
    plan; only the TRANSFORM - whose Hadamard identity needs a maximum-length sequence - is refused, loudly."""
    from zignal_amd import _lib
    from zignal_amd.device import Plan
    p = Plan(8, (8, 4), device=0)
    assert p.kind() == (False, False)
    assert p.sequence(2).shape == (2 * 255, 1)
    x = np.zeros((2 * 255, 1))
    with pytest.raises(_lib.ZmlsError) as ei:
        p.impulse_host(x)
    assert ei.value.code == _lib.E_UNSUPPORTED and "maximum-length" in str(ei.value)
    with pytest.raises(_lib.ZmlsError) as ei:
        p.reserve(4)
    assert ei.value.code == _lib.E_UNSUPPORTED
    p.close()
    q = Plan(8, TAPS[8][0], device=0)
    assert q.kind() == (True, True)
    q.close()


SEQ_ANY = load_json("seq_any.json")


@pytest.mark.parametrize("kat", SEQ_ANY["sets"], ids=lambda k: "N%d-%s" % (k["N"], "_".join(map(str, k["taps"]))))
def test_tap_sets_outside_the_table_bit_exact_against_the_reference(kat):
    """rows a3-a6 for every tap set the reference accepts (it validates nothing, mls.py:64-68, 101-108): fixtures captured
    from the real reference (tests/golden/make_golden.py --only seq_any) - the first L bits get_full_sequence tiles, and
    3L+17 bits of generator_chunk's register, which runs on without wrapping at L when it is not maximum-length."""
    from zignal_amd.measure.mls import _MLS_base
    N, taps, L = kat["N"], tuple(kat["taps"]), (1 << kat["N"]) - 1
    m = _MLS_base(N, taps)
    s = m.get_full_sequence(2)
    assert s.shape == (2 * L, 1) and s.dtype == np.float64 and np.array_equal(s[:L], s[L:])
    first = ((1.0 - s[:L, 0]) / 2).astype(np.uint8)
    assert inputs.digest(np.packbits(first, bitorder="little")) == kat["sha256"] and int(first.sum()) == kat["ones"]
    assert "".join(map(str, first[:64])) == kat["first64"]
    run = next(m.generator_chunk(kat["run_bits"]))
    assert run.shape == (kat["run_bits"], 1) and run.dtype == np.int64
    assert inputs.digest(np.packbits(run[:, 0].astype(np.uint8), bitorder="little")) == kat["run_sha256"]
    # chunks continue one another, whatever their size, and generator_bit / generator_samples hand out the same stream
    g = m.generator_chunk(1000)
    pieces = np.concatenate([next(g)[:, 0] for _ in range(-(-kat["run_bits"] // 1000))])[:kat["run_bits"]]
    assert np.array_equal(pieces, run[:, 0])
    gb = m.generator_bit()
    assert [next(gb) for _ in range(min(3000, kat["run_bits"]))] == run[:3000, 0].tolist()
    gs = m.generator_samples(777)
    assert np.array_equal(next(gs)[:, 0], 1.0 - 2.0 * run[:777, 0]) or kat["run_bits"] < 777
    # far along the stream: the jump-ahead start state against the C oracle's bit-serial register
    p = m._plan()
    if not kat["wraps_at_L"] or kat["ones"] != 1 << (N - 1):
        assert p.kind() == (False, False)
    start = 5 * L + 12345
    want = c_oracle.bits(N, list(taps), start + 4096)[start:]
    assert np.array_equal(p.bits_from(start, 4096), want)


def test_taps_the_register_cannot_index_raise_index_error_like_the_reference():
    from zignal_amd.measure.mls import _MLS_base
    for e in SEQ_ANY["errors"]:
        m = _MLS_base(e["N"], tuple(e["taps"]))                  # the constructor accepts them (mls.py:64-68)
        with pytest.raises(IndexError):
            next(m.generator_bit())
        with pytest.raises(IndexError):
            m.get_full_sequence(1)


def test_bits_from_far_positions_on_maximal_and_other_registers():
    from zignal_amd.device import Plan
    for N, taps in ((16, TAPS[16][0]), (16, (16, 8)), (9, (3,)), (20, TAPS[20][0])):
        p = Plan(N, taps, device=0)
        for start in (0, 1, (1 << N) - 2, (1 << N) - 1, 3 * (1 << N) + 5, (1 << 33) + 7, (1 << 45) + 123):
            if start < (1 << 24):
                want = c_oracle.bits(N, list(taps), start + 2000)[start:]
            elif p.kind()[0]:
                pos = start % ((1 << N) - 1)
                want = c_oracle.bits(N, list(taps), pos + 2000)[pos:]
            else:
                continue
            assert np.array_equal(p.bits_from(start, 2000), want), (N, taps, start)
        p.close()


def test_every_tap_set_of_the_reference_table():
    """The bit-exactness contract is 'against the reference taps': all tap sets of zignal's table up to
    N = 22 give, bit for bit, the C oracle's period (which the goldens pin to the reference); for N = 23..26 every
    set (27..30: the first and last) must build a plan (= the kernel's walk visits every state: maximum length) and agree on the first 64 Ki bits."""
    from zignal_amd.device import Plan
    n_sets = 0
    for N in sorted(TAPS):
        if N > 30:
            continue                                  # orders 31, 32: test_orders_31_and_32_generate_the_sequence_...
        for taps in (TAPS[N] if N <= 26 else (TAPS[N][0], TAPS[N][-1])):
            p = Plan(N, taps, device=0)
            if N <= 22:
                assert np.array_equal(p.bits_packed(), c_oracle.bits_packed(N, taps)), "N=%d taps=%s" % (N, taps)
            else:
                nb = 1 << 16
                assert np.array_equal(p.bits_packed(nb), c_oracle.bits_packed(N, taps, nb)), "N=%d taps=%s" % (N, taps)
            p.close()
            n_sets += 1
    assert n_sets == sum(len(TAPS[N]) for N in TAPS if N <= 26) + 8
