"""CPU tier: the C-ABI library loads and exports everything include/zmls.h declares, the
host-side logic (pass planner, GF(2) jump-ahead, tile geometry, argument checks, error
types) behaves, and the product package never touches the oracle. No GPU compute here."""
import ctypes
import hashlib
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
G = os.path.join(ROOT, "tests", "golden")

from oracle import c_oracle  # noqa: E402
from zignal_amd import _lib  # noqa: E402


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "zignal_amd", "csrc"), "-j8"])
    return _lib.lib()


HEADERS = ("zmls.h", "zmls_testing.h")      # the product ABI, and the hooks of the tests / the benchmark


def header_functions(name=None):
    names = set()
    for h in ([name] if name else HEADERS):
        txt = open(os.path.join(ROOT, "include", h)).read()
        txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
        names |= set(re.findall(r"\b(zmls_[a-z0-9_]+)\s*\(", txt))
    return sorted(names)


def test_library_exports_every_declared_symbol(lib):
    names = header_functions()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), "libzmls.so does not export %s" % n
    # and the ctypes table covers them all (plus zmls_last_error, bound separately)
    assert set(names) - {"zmls_last_error"} == set(_lib.SIGNATURES)
    assert lib.zmls_version() >= 100


def test_product_header_holds_no_test_hooks():
    """A maintainer binding the library reads zmls.h: every entry point there takes over a piece of the reference (or is
    needed to run those pieces on a GPU); memory / stream / event wrappers, synthetic input, timing and plan internals
    live in zmls_testing.h."""
    product, hooks = set(header_functions("zmls.h")), set(header_functions("zmls_testing.h"))
    assert not product & hooks
    assert not [n for n in product if n.startswith(("zmls_dev_", "zmls_stream_", "zmls_event_", "zmls_host_", "zmls_synth_"))]
    assert {"zmls_plan_export_table", "zmls_plan_force_tile", "zmls_impulse_device_timed"} <= hooks
    assert {"zmls_plan_create", "zmls_sequence_bits", "zmls_impulse_device", "zmls_impulse_host", "zmls_impulse_device_static"} <= product


def _placement(ms, max_live, tries=4, model=12.5):
    arr = (ctypes.c_float * len(ms))(*ms)
    out = (ctypes.c_int64 * 8)()
    rc = _lib.lib().zmls_host_placement_sim(arr, len(ms), max_live, tries, model, out)
    return rc, dict(zip(("kept", "rated", "first_us", "final_us", "rounds", "alloc_failed", "live_at_end", "peak_live"), list(out)))


def test_capture_placement_search_host_logic(lib):
    """zmls_capture_alloc's search (zmls_host.cpp::placement_search) on a simulated machine."""
    # the first candidate already sits on the floor the model predicts: taken at once, rated again, nothing else allocated
    rc, r = _placement([12.6, 13.5, 14.5], max_live=4)
    assert rc == 0 and r["kept"] == 0 and r["rated"] == 1 and r["live_at_end"] == 1 and r["peak_live"] == 1 and not r["alloc_failed"]
    # a slow first candidate: the search goes on until one reaches the floor, and gives the rest back
    rc, r = _placement([14.5, 13.5, 12.6, 12.6], max_live=4)
    assert rc == 0 and r["kept"] == 2 and r["first_us"] == 14500 and r["final_us"] == 12600 and r["live_at_end"] == 1
    assert r["peak_live"] <= 3                                     # the best and at most two rejected candidates
    # a plateau above a floor that is in reach needs a third vote, then another round is tried with the winner kept
    rc, r = _placement([13.5], max_live=4)
    assert rc == 0 and r["final_us"] == 13500 and r["rounds"] == 3 and r["live_at_end"] == 1 and r["peak_live"] <= 4
    # shapes the model does not fit (everything far above it): two candidates that agree are enough, one round
    rc, r = _placement([20.0, 20.1], max_live=4)
    assert rc == 0 and r["rated"] == 2 and r["rounds"] == 1 and r["final_us"] == 20000
    # FALLBACK, memory holds one capture only (a capture above half of the free memory): the first allocation is the
    # answer, rated once, and the search says that it had no choice
    rc, r = _placement([14.5, 12.6], max_live=1)
    assert rc == 0 and r["kept"] == 0 and r["rated"] == 1 and r["alloc_failed"] == 1 and r["final_us"] == 14500 and r["live_at_end"] == 1
    # memory holds two: the better of two, never more than two alive
    rc, r = _placement([14.5, 13.5, 12.6], max_live=2)
    assert rc == 0 and r["kept"] in (1, 2) and r["peak_live"] == 2 and r["alloc_failed"] == 1 and r["live_at_end"] == 1
    # memory holds none: an error, nothing leaked
    rc, r = _placement([12.6], max_live=0)
    assert rc == _lib.E_HIP and r["kept"] == -1 and r["live_at_end"] == 0


def test_pass_split(lib):
    for N in range(2, 31):
        bits = (ctypes.c_int * 4)()
        n = lib.zmls_host_pass_split(N, bits)
        b = list(bits)[:n]
        assert sum(b) == N and max(b) <= 10 and n == -(-N // 10)
        assert max(b) - min(b) <= 1 and b == sorted(b, reverse=True)
    assert lib.zmls_host_pass_split(31, (ctypes.c_int * 4)()) == _lib.E_ARG


@pytest.mark.parametrize("first", [0, 1])
@pytest.mark.parametrize("K", range(1, 11))
@pytest.mark.parametrize("cfg", [0, 1, 2])
def test_tile_geometry_is_a_tiling(lib, cfg, K, first):
    assert lib.zmls_host_geometry_selfcheck(K, cfg, first) == 0, _lib.last_error()


@pytest.mark.parametrize("first", [0, 1])
@pytest.mark.parametrize("K", [10, 11])
def test_tall_tile_geometry_is_a_tiling(lib, K, first):
    assert lib.zmls_host_geometry_selfcheck(K, 3, first) == 0, _lib.last_error()
    assert lib.zmls_host_geometry_selfcheck(12, 3, first) != 0
    assert lib.zmls_host_geometry_selfcheck(11, 2, first) != 0      # 11-bit passes are the tall tile's alone


def _window(N, taps, steps):
    bits = c_oracle.bits(N, taps, steps + N)
    return int(sum(int(b) << i for i, b in enumerate(bits[steps:steps + N])))


@pytest.mark.parametrize("N,taps", [(3, (3, 2)), (5, (5, 4, 3, 2)), (10, (10, 7)), (16, (16, 15, 13, 4)),
                                     (20, (20, 17)), (24, (24, 23, 22, 17)), (30, (30, 29, 28, 7))])
def test_jump_ahead_matches_bit_serial_oracle(lib, N, taps):
    arr, n = _lib.taps_array(taps)
    rng = np.random.default_rng(N)
    steps = [0, 1, 2, N - 1, N, 1023, 1024, 4097] + [int(v) for v in rng.integers(0, 200000, 6)]
    for s in steps:
        out = ctypes.c_uint32()
        assert lib.zmls_host_lfsr_state(N, arr, n, s, ctypes.byref(out)) == 0
        assert out.value == _window(N, taps, s), (N, s)


def test_bad_plan_arguments_without_gpu(lib):
    h = ctypes.c_void_p()
    arr, n = _lib.taps_array((5, 3))
    # order out of range / bad dtype
    assert lib.zmls_plan_create(ctypes.byref(h), 1, arr, n, 0, 0) == _lib.E_ARG
    assert lib.zmls_plan_create(ctypes.byref(h), 33, arr, n, 0, 0) == _lib.E_ARG
    # taps the reference's register cannot index (its IndexError, mls.py:107): 0, negative, beyond 2N
    for N, bad in ((5, (0,)), (5, (5, 11)), (6, (6, -1)), (4, (4, 9))):
        arr2, n2 = _lib.taps_array(bad)
        assert lib.zmls_plan_create(ctypes.byref(h), N, arr2, n2, 0, 0) == _lib.E_ARG
        assert "deque index out of range" in _lib.last_error()
    # everything else is a tap set the reference generates a sequence for - without N, not primitive, N+1..2N - and gets
    # as far as the device (none in this tier)
    if _lib.device_count() == 0:
        for N, ok in ((4, (5, 3)), (5, (4, 3)), (8, (8, 4)), (6, (12, 7))):
            arr2, n2 = _lib.taps_array(ok)
            assert lib.zmls_plan_create(ctypes.byref(h), N, arr2, n2, 0, 0) == _lib.E_NODEVICE
    assert lib.zmls_plan_create(ctypes.byref(h), 5, arr, n, 7, 0) == _lib.E_ARG
    assert "dtype" in _lib.last_error()


def test_taps_table_digest():
    from zignal_amd.measure.mlstaps import TAPS
    want = open(os.path.join(G, "taps.json")).read()
    got = json.dumps({str(N): [list(t) for t in TAPS[N]] for N in sorted(TAPS)}, separators=(",", ":"), sort_keys=True)
    assert hashlib.sha256(got.encode()).hexdigest() == hashlib.sha256(want.strip().encode()).hexdigest()
    assert sum(len(v) for v in TAPS.values()) == 1014 and sorted(TAPS) == list(range(3, 33))
    assert all(t[0] == N for N, sets in TAPS.items() for t in sets)


def test_api_argument_errors_match_reference_table():
    """Exception types of the reference for malformed arguments (tests/golden/errors.json) —
    all raised before any device work, so they can be checked without a GPU."""
    from zignal_amd.measure.mls import MLS, TAPS, _MLS_base, get_random_taps
    table = json.load(open(os.path.join(G, "errors.json")))

    def name(fn):
        try:
            fn()
        except BaseException as e:  # noqa: BLE001
            return type(e).__name__, str(e)
        return None, None

    N, taps, L = 5, (5, 3), 31
    m = _MLS_base(N, taps)
    assert (m.N, m.L, m.taps) == (5, 31, (5, 3))
    checks = {
        "ctor_N_none": lambda: _MLS_base(None, taps),
        "ctor_taps_none": lambda: _MLS_base(N, None),
        "ctor_taps_empty": lambda: _MLS_base(N, ()),
        "ctor_taps_ndarray": lambda: _MLS_base(N, np.array(taps)),
        "ctor_taps_list_ok": lambda: _MLS_base(N, list(taps)),
        "impulse_len_eq_L": lambda: m.get_impulse(np.zeros((L, 1))),
        "impulse_len_not_multiple": lambda: m.get_impulse(np.zeros(2 * L + 3)),
        "impulse_two_channels": lambda: m.get_impulse(np.zeros((2 * L, 2))),
        "impulse_list": lambda: m.get_impulse([0.0] * (3 * L)),
        "mls_repeats_1": lambda: MLS(N, taps, repeats=1),
        "random_taps_unknown_N": lambda: get_random_taps(99),
        "taps_keyerror": lambda: TAPS[99],
    }
    for key, fn in checks.items():
        t, msg = name(fn)
        assert t == table[key]["type"], key
        if table[key]["msg"]:
            assert msg.strip("'") == table[key]["msg"].strip("'"), key


def test_no_gpu_means_loud_failure_not_fallback(lib):
    n = ctypes.c_int()
    lib.zmls_device_count(ctypes.byref(n))
    if n.value > 0:
        pytest.skip("a GPU is present")
    from zignal_amd.measure.mls import _MLS_base
    m = _MLS_base(5, (5, 3))
    with pytest.raises(_lib.ZmlsError) as ei:
        m.get_full_sequence(1)
    assert ei.value.code == _lib.E_NODEVICE and "no CPU fallback" in str(ei.value)
    with pytest.raises(_lib.ZmlsError):
        m.get_impulse(np.zeros((62, 1)))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "zignal_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".h", ".inc")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
                assert "mls_oracle" not in txt and "/root/reference" not in txt, f
    # importing the package must not drag in torch or the oracle
    code = "import sys; import zignal_amd.measure.mls; print('torch' in sys.modules, any(m.startswith('oracle') for m in sys.modules))"
    out = subprocess.check_output([sys.executable, "-c", code], cwd=ROOT).decode().split()
    assert out == ["False", "False"]


def test_argument_errors_need_no_device(lib):
    """Null handles and out-of-range values are refused before anything touches a GPU."""
    assert lib.zmls_plan_set_input_scale_log2(None, 15) == _lib.E_ARG
    assert lib.zmls_dev_copy(0, None, None, 16, None) == _lib.E_ARG
    assert lib.zmls_device_count(None) == _lib.E_ARG
    assert lib.zmls_plan_info(None, None) == _lib.E_ARG
    assert lib.zmls_plan_force_tile(None, 0) == _lib.E_ARG
    assert b"" != _lib.last_error().encode()
    assert lib.zmls_impulse_sources_device(None, 2, None, None, 0, 8, 1, 1, 1, None, 1, 1, 1, None) == _lib.E_ARG
    assert lib.zmls_sources_reserve((ctypes.c_void_p * 2)(None, None), 2, None, 8, 1) == _lib.E_ARG and "null" in _lib.last_error()
    assert lib.zmls_sources_reserve((ctypes.c_void_p * 1)(None), 65, None, 8, 1) == _lib.E_ARG


def test_audio_append_builds_the_multichannel_stimulus():
    """Audio.append as the reference's multichannel recipe uses it (audio.py:376-421, mls.py:338-343): channels to the
    right, the shorter side padded with zeros, mismatching rates and types refused."""
    from zignal_amd.audio import Audio
    a = Audio(fs=48000, initialdata=np.arange(6.0).reshape(6, 1))
    b = Audio(fs=48000, initialdata=np.ones((4, 2)))
    a.append(b, Audio(fs=48000, initialdata=np.full((8, 1), 7.0)))
    assert a.samples.shape == (8, 4) and a.ch == 4 and a.nofsamples == 8 and a.duration == 8 / 48000
    assert a.samples[:, 0].tolist() == [0, 1, 2, 3, 4, 5, 0, 0]
    assert a.samples[:, 1].tolist() == [1, 1, 1, 1, 0, 0, 0, 0] and a.samples[:, 3].tolist() == [7] * 8
    with pytest.raises(AssertionError):
        a.append(Audio(fs=44100, initialdata=np.zeros((8, 1))))
    with pytest.raises(AssertionError):
        a.append(Audio(fs=48000, initialdata=np.zeros((8, 1), dtype=np.float32)))
    with pytest.raises(AssertionError):
        a.append(np.zeros((8, 1)))


def test_plain_c_program_compiles_against_the_product_header_and_fails_loudly_without_a_gpu(lib):
    """include/zmls.h is C (C99, -pedantic -Werror), not only C++: examples/c_api.c uses nothing else. Without a GPU it
    reports that and leaves with status 3 - no CPU path behind the ABI either."""
    import __graft_entry__ as g
    exe = g.build_c_example()
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    n = ctypes.c_int()
    lib.zmls_device_count(ctypes.byref(n))
    if n.value > 0:
        assert r.returncode == 0 and "c_api ok" in r.stdout, r.stdout + r.stderr
    else:
        assert r.returncode == 3 and "no CPU path" in r.stdout, r.stdout + r.stderr


@pytest.mark.parametrize("m", list(range(1, 23)))
def test_transform_passes_on_the_host_against_a_reference_fft(m):
    """the tile code of csrc/fft_tile.h - the very functions the kernels call - run thread by thread on the CPU: forward passes
    read back through the passes' own output order, every bin against a textbook radix-2 transform, then the inverse passes
    (2^13 ... 2^21: one strided level of 4 ... 9 bits + the contiguous one; 2^22: two strided levels)"""
    lib = _lib.lib()
    assert lib.zmls_host_fft_selfcheck(0, 1 << m) == 0, _lib.last_error()


@pytest.mark.parametrize("F", [2, 3, 5, 31, 100, 127, 1000, 1023, 2049, 4095, 8191, 16383, 48000])
def test_bluestein_chain_on_the_host_against_a_direct_dft(F):
    """chirp -> forward passes -> chirp filter's spectrum -> inverse passes -> chirp, for lengths that are not powers of two"""
    lib = _lib.lib()
    assert lib.zmls_host_fft_selfcheck(1, F) == 0, _lib.last_error()
    assert lib.zmls_host_fft_selfcheck(0, 1000) != 0 and lib.zmls_host_fft_selfcheck(2, 8) != 0
