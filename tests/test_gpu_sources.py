"""GPU parity: several MLS sources measured at once (the multichannel recipe of
zignal/measure/mls.py:338-343). zmls_impulse_sources_device through the C ABI against

  * the reference itself: tests/golden/multi_source.* holds captures of 2-4 sources playing together into small mixing
    systems and the responses the reference's own get_impulse returns pair by pair (with and without emphasis);
  * the oracle, pair by pair;
  * zmls_impulse_device, one call per source (the path every other test pins), in every layout the entry point takes.

Tolerance: 1e-12 x peak (float64 plans), 1e-5 x peak (float32 plans)."""
import ctypes
import os

import numpy as np
import pytest
from scipy.signal import lfilter

from gpu_common import G, load_json, oracle_ir, rel_err
from oracle import c_oracle
from zignal_amd import _lib
from zignal_amd.device import DeviceBuffer, IirFilter, Plan, Sources, sync
from zignal_amd.filters.biquads import RBJ
from zignal_amd.measure.mls import MLS, TAPS
from zignal_amd.measure.multimls import MultiMLS

pytestmark = pytest.mark.gpu

CASES = load_json("multi_source.json")["cases"]


@pytest.mark.parametrize("case", CASES, ids=lambda c: "%s-N%d-K%d-C%d-R%d%s" % (c["id"], c["N"], len(c["taps"]), c["channels"], c["repeats"],
                                                                                "-emph" if c["emphasis"] else ""))
def test_golden_multi_source(case):
    arrays = np.load(os.path.join(G, "multi_source.npz"))
    x, want = arrays[case["id"] + "_x"], arrays[case["id"] + "_h"]
    taps = [tuple(t) for t in case["taps"]]
    m = MultiMLS(N=case["N"], taps=taps, fs=case["fs"], repeats=case["repeats"], B=case["B"], A=case["A"], device=0)
    stim = m.get_stimulus(emphasis=case["emphasis"])
    assert stim.samples.shape == (case["repeats"] * m.L, len(taps)) and stim.ch == len(taps)
    if case["id"] + "_stim" in arrays:
        assert rel_err(stim.samples, arrays[case["id"] + "_stim"]) <= (1e-12 if case["emphasis"] else 0.0)
    h = m.get_impulses(x, deemphasis=case["emphasis"])
    assert h.shape == want.shape and h.dtype == np.float64
    for k in range(len(taps)):
        assert rel_err(h[k], want[k]) <= 1e-12, k
    # the pairing front end: an Audio with one channel per (recorded channel, source) pair, in the order asked for
    pairs = [(case["channels"] - 1, 0), (0, len(taps) - 1), (0, 0)]
    a = m.get_impulse(x, pairs=pairs, deemphasis=case["emphasis"])
    assert a.samples.shape == (m.L, 3) and a.fs == case["fs"]
    for i, (c, k) in enumerate(pairs):
        assert rel_err(a.samples[:, i], want[k, :, c]) <= 1e-12
    everything = m.get_impulse(x, deemphasis=case["emphasis"])
    assert everything.ch == case["channels"] * len(taps)
    assert rel_err(everything.samples[:, len(taps) + 1], want[1, :, 1]) <= 1e-12        # pair (c=1, k=1)
    # the reference's way: one single-source MLS per pair
    single = MLS(N=case["N"], taps=taps[1], fs=case["fs"], repeats=case["repeats"], B=case["B"], A=case["A"], device=0)
    col = x[:, :1]
    if case["emphasis"]:
        col = single.apply_deemphasis(col)
    assert rel_err(single.get_impulse(col).samples[:, 0], want[1, :, 0]) <= 1e-12
    m.close()


def _mixture(N, taps, R, C, seed, dtype=np.float64):
    """all sources playing together into C recorded channels: a delay and a gain per pair, plus noise"""
    L = (1 << N) - 1
    rng = np.random.default_rng(seed)
    x = 0.02 * rng.standard_normal((R * L, C))
    for k, t in enumerate(taps):
        s = np.tile(c_oracle.sequence(N, t)[:, 0], R)
        for c in range(C):
            x[:, c] += (0.3 + 0.1 * ((c + k) % 5)) * np.roll(s, int(rng.integers(0, L)))
    if np.issubdtype(dtype, np.integer):
        return np.round(x * (np.iinfo(dtype).max / 8.0)).astype(dtype)
    return x.astype(dtype)


_group = {}


def _plans(N, taps, dtype="f64"):
    """the plans of one test, kept across tests; evicted only between groups (gpu_common.plan may close plans handed out
    earlier in the same list)"""
    key = (N, tuple(taps), dtype)
    if key not in _group:
        if len(_group) > 6:
            for ps in _group.values():
                for p in ps:
                    p.close()
            _group.clear()
        _group[key] = [Plan(N, t, dtype=dtype, device=0) for t in taps]
    return _group[key]


def _per_source(plans, x):
    return np.stack([p.impulse_host(x) for p in plans])


LAYOUTS = ["frames", "stacked", "strided"]


@pytest.mark.parametrize("N,K,C,R", [(9, 2, 4, 5), (10, 3, 3, 2), (12, 4, 8, 6), (12, 2, 5, 9), (14, 3, 16, 4), (16, 4, 64, 5), (16, 2, 7, 3),
                                     (13, 1, 4, 6)])
@pytest.mark.parametrize("layout", LAYOUTS)
def test_sources_call_equals_one_call_per_source(N, K, C, R, layout):
    """Same responses as K zmls_impulse_device calls, whichever way the entry point gets there: capture read once per
    source (R = 2, single-pass orders) or periods added up once (the rest); aligned frames, a stack of single-channel
    captures with responses to match, or padded rows with an odd channel count."""
    taps = [tuple(t) for t in TAPS[N][:K]]
    plans = _plans(N, taps)
    L = plans[0].L
    x = _mixture(N, taps, R, C, seed=N * 100 + K)
    want = _per_source(plans, x)
    group = Sources(plans)
    group.reserve(R * L, C)
    if layout == "frames":
        got = group.impulse_host(x)
    else:
        if layout == "stacked":                         # (C, R*L) in, (K, C, L) out
            xd, rs, cs = np.ascontiguousarray(x.T), 1, R * L
            out_shape, src_stride, ors, ocs = (K, C, L), C * L, 1, L
        else:                                           # rows padded to C+3 columns, responses to C+1
            xd = np.zeros((R * L, C + 3))
            xd[:, :C] = x
            rs, cs = C + 3, 1
            out_shape, src_stride, ors, ocs = (K, L, C + 1), L * (C + 1), C + 1, 1
        d_x = DeviceBuffer(xd.nbytes, 0).upload(xd)
        d_h = DeviceBuffer(int(np.prod(out_shape)) * 8, 0)
        d_h.fill(0)
        group.impulse_device(d_x.ptr, _lib.ZMLS_F64, R * L, C, rs, cs, d_h.ptr, src_stride, ors, ocs)
        sync(0)
        raw = d_h.download(out_shape, np.float64)
        got = raw.transpose(0, 2, 1) if layout == "stacked" else raw[:, :, :C]
        if layout == "strided":
            assert not raw[:, :, C].any()                # the pad column of the responses is not ours to write
        d_x.free()
        d_h.free()
    assert got.shape == want.shape
    assert rel_err(got, want) <= 1e-14
    for k in (0, K - 1):
        assert rel_err(got[k, :, C - 1], oracle_ir(N, taps[k], x[:, C - 1:C])) <= 1e-12


@pytest.mark.parametrize("in_dtype,plan_dtype", [(np.int16, "f64"), (np.int32, "f64"), (np.float32, "f64"), (np.float64, "f32"), (np.float32, "f32")])
def test_sources_other_sample_types(in_dtype, plan_dtype):
    N, K, C, R = 13, 3, 6, 5
    taps = [tuple(t) for t in TAPS[N][:K]]
    plans = _plans(N, taps, plan_dtype)
    x = _mixture(N, taps, R, C, seed=77, dtype=in_dtype)
    got = Sources(plans).impulse_host(x)
    assert got.dtype == (np.float32 if plan_dtype == "f32" else np.float64)
    tol = 1e-5 if (plan_dtype == "f32" or in_dtype == np.float32) else 1e-12
    assert rel_err(got, _per_source(plans, x)) <= (1e-6 if plan_dtype == "f32" else 1e-14)
    assert rel_err(got[1, :, 2], oracle_ir(N, taps[1], x[:, 2:3].astype(np.float64))) <= tol


@pytest.mark.parametrize("N,K,C,R,order", [(8, 2, 3, 6, 2), (12, 3, 4, 4, 2), (14, 2, 16, 3, 4), (12, 2, 2, 5, 1)])
def test_sources_with_deemphasis(N, K, C, R, order):
    """`f` given: every source's response of the de-emphasised capture, against scipy's lfilter + one call per source"""
    taps = [tuple(t) for t in TAPS[N][:K]]
    plans = _plans(N, taps)
    if order == 2:
        A, B = RBJ(filtertype="highshelf", gaindb=-10, f0=100, Q=0.707, fs=48000).get_coefficients()     # the inverse shelf
    else:
        rng = np.random.default_rng(order)
        B = rng.standard_normal(order + 1)
        A = np.poly(0.6 * rng.uniform(-1, 1, order))
    x = _mixture(N, taps, R, C, seed=5 + N)
    f = IirFilter(B, A, device=0)
    got = Sources(plans).impulse_host(x, iir=f)
    y = lfilter(B, A, x, axis=0)
    assert rel_err(got, _per_source(plans, y)) <= 1e-12
    assert rel_err(got[K - 1, :, 0], oracle_ir(N, taps[K - 1], y[:, :1])) <= 1e-12
    # time-contiguous capture: the stacked layout of the fused sweep
    L = plans[0].L
    group = Sources(plans)
    group.reserve(R * L, C, iir=f)
    xd = np.ascontiguousarray(x.T)
    d_x, d_h = DeviceBuffer(xd.nbytes, 0).upload(xd), DeviceBuffer(K * L * C * 8, 0)
    group.impulse_device(d_x.ptr, _lib.ZMLS_F64, R * L, C, 1, R * L, d_h.ptr, C * L, 1, L, iir=f)
    sync(0)
    assert rel_err(d_h.download((K, C, L), np.float64).transpose(0, 2, 1), got) <= 1e-12
    d_x.free()
    d_h.free()
    f.close()


def test_sources_share_one_workspace():
    """the plans behind a Sources call need no scratch of their own: only plan 0 is ever reserved"""
    N, K, C, R = 15, 3, 32, 4
    taps = [tuple(t) for t in TAPS[N][3:3 + K]]
    fresh = [Plan(N, t, device=0) for t in taps]
    try:
        x = _mixture(N, taps, R, C, seed=3)
        got = Sources(fresh).impulse_host(x)
        assert fresh[0].info()["workspace_bytes"] > 0 and fresh[0].info()["staging_bytes"] > 0
        assert all(p.info()["workspace_bytes"] == 0 and p.info()["staging_bytes"] == 0 for p in fresh[1:])
        assert rel_err(got, _per_source(_plans(N, taps), x)) <= 1e-14
        # R = 2: nothing to add up, every source reads the capture itself - still on plan 0's workspace
        got2 = Sources(fresh).impulse_host(x[:2 * fresh[0].L])
        assert all(p.info()["workspace_bytes"] == 0 for p in fresh[1:])
        assert rel_err(got2, _per_source(_plans(N, taps), x[:2 * fresh[0].L])) <= 1e-14
    finally:
        for p in fresh:
            p.close()


def test_sources_argument_errors():
    lib = _lib.lib()
    a, b = _plans(10, TAPS[10][:2])
    c32, other = _plans(10, TAPS[10][1:2], "f32")[0], _plans(11, TAPS[11][:1])[0]
    L = a.L
    d_x, d_h = DeviceBuffer(3 * L * 4 * 8, 0), DeviceBuffer(2 * L * 4 * 8, 0)

    def call(plans, n_rows=3 * L, n_ch=4, rs=4, cs=1, src_stride=L * 4, x=None, h=None, n=None):
        arr = (ctypes.c_void_p * max(1, len(plans)))(*[p._h if p is not None else None for p in plans])
        return lib.zmls_impulse_sources_device(arr, len(plans) if n is None else n, None, d_x.ptr if x is None else x, _lib.ZMLS_F64, n_rows, n_ch,
                                               rs, cs, d_h.ptr if h is None else h, src_stride, 4, 1, None)

    assert call([a, b]) == 0
    sync(0)
    assert call([a, other]) == _lib.E_ARG and "differs from plan 0" in _lib.last_error()
    assert call([a, c32]) == _lib.E_ARG
    assert call([a, None]) == _lib.E_ARG and "null" in _lib.last_error()
    assert call([a], n=0) == _lib.E_ARG
    assert call([a], n=65) == _lib.E_ARG
    assert call([a, b], n_rows=L) == _lib.E_ARG and "thrown away" in _lib.last_error()
    assert call([a, b], n_rows=3 * L + 1) == _lib.E_ARG
    assert call([a, b], src_stride=0) == _lib.E_ARG
    assert call([a, b], n_ch=0) == _lib.E_ARG
    assert call([a, b], x=0) == _lib.E_ARG and call([a, b], h=0) == _lib.E_ARG
    assert lib.zmls_sources_reserve((ctypes.c_void_p * 2)(a._h, other._h), 2, None, 3 * L, 4) == _lib.E_ARG
    with pytest.raises(AssertionError):
        MultiMLS(N=10, taps=2, B=(1, -2), A=(1, 0), device=0)                       # a zero outside the unit circle: no inverse
    m = MultiMLS(N=10, taps=2, device=0)
    assert m.taps == [tuple(t) for t in TAPS[10][:2]] and "MultiMLS(N=10" in repr(m)
    with pytest.raises(AssertionError):
        m.get_impulse(np.zeros((2 * m.L, 1)), pairs=[(1, 0)])
    m.close()
    d_x.free()
    d_h.free()


def test_cross_talk_between_sources_is_small():
    """why the reference says "different taps": two sources at once barely leak into each other's response"""
    N = 14
    m = MultiMLS(N=N, taps=2, fs=48000, repeats=3, device=0)
    stim = m.get_stimulus().samples
    x = stim[:, :1] + np.roll(stim[:, 1:2], 100, axis=0)          # one microphone hears both, the second 100 late
    h = m.get_impulses(x)
    assert abs(h[0, 0, 0] - 1.0) < 0.02 and abs(h[1, 100, 0] - 1.0) < 0.02
    leak = np.abs(h[0, :, 0]).copy()
    leak[0] = 0
    assert leak.max() < 0.1                                                 # the two sequences' cross-correlation: small, not zero
    assert rel_err(h[0, :, 0], oracle_ir(N, m.taps[0], x)) <= 1e-12         # and exactly what the reference algorithm sees
    m.close()


def test_full_size_sources():
    """N=20, 4 sources x 32 channels x 5 periods (1.3 GB capture): the one-sweep path against one call per source on whole
    slabs, and the loop-back structure of every response."""
    N, K, C, R = 20, 4, 32, 5
    taps = [tuple(t) for t in TAPS[N][:K]]
    plans = _plans(N, taps)
    L = plans[0].L
    group = Sources(plans)
    group.reserve(R * L, C)
    d_x = DeviceBuffer(R * L * C * 8, 0)
    # channel c carries source (c mod K) alone, delayed by 1000 c
    cap = np.zeros((R * L, C))
    for k in range(K):
        seq = plans[k].sequence(1)[:, 0]
        for c in range(k, C, K):
            cap[:, c] = np.tile(np.roll(seq, 1000 * c), R)
    d_x.upload(cap)
    d_h, d_ref = DeviceBuffer(K * L * C * 8, 0), DeviceBuffer(L * C * 8, 0)
    group.impulse_device(d_x.ptr, _lib.ZMLS_F64, R * L, C, C, 1, d_h.ptr, L * C, C, 1)
    sync(0)
    h = d_h.download((K, L, C), np.float64)
    for k in range(K):
        plans[k].reserve(C)
        plans[k].impulse_device(d_x.ptr, _lib.ZMLS_F64, R * L, C, C, 1, d_ref.ptr, C, 1)
        sync(0)
        assert np.array_equal(h[k], d_ref.download((L, C), np.float64)), k       # exact loop-back data: identical sums
    for c in range(0, C, 7):
        k = c % K
        assert abs(h[k, 1000 * c, c] - 1.0) <= 1e-12
        rest = np.delete(h[k, :, c], 1000 * c)
        assert np.max(np.abs(rest + 1.0 / L)) <= 1e-12
    for q in (d_x, d_h, d_ref):
        q.free()


def _random_source_cases():
    rng = np.random.default_rng(20261111)
    out = []
    for i in range(36):
        N = int(rng.choice([7, 9, 10, 11, 12, 13, 14, 16]))
        out.append(dict(N=N, K=int(rng.integers(1, 5)), C=int(rng.choice([1, 2, 3, 5, 8, 16, 17, 24, 32])), R=int(rng.choice([2, 3, 4, 5, 7, 9])),
                        plan=str(rng.choice(["f64", "f64", "f32"])), x=str(rng.choice(["f64", "f64", "f32", "i16", "i32"])),
                        lin=str(rng.choice(["frame", "stack", "padded"])), lout=str(rng.choice(["frame", "stack"])),
                        filt=int(rng.choice([0, 0, 1, 2, 3])), seed=i))
    return out


@pytest.mark.parametrize("c", _random_source_cases(), ids=lambda c: "N%d-K%d-C%d-R%d-%s-%s-%s-%s-f%d" % (
    c["N"], c["K"], c["C"], c["R"], c["plan"], c["x"], c["lin"], c["lout"], c["filt"]))
def test_random_source_cases_against_the_oracle(c):
    """Randomised sweep over what selects a path inside zmls_impulse_sources_device: single / multi-pass orders, one to four
    stimuli, period counts either side of the shared-sum rule, sample types, layouts in and out, filter orders 0-3.
    Every (stimulus, channel) response is held to the oracle (after scipy's lfilter when a filter is given)."""
    NPT = {"f64": np.float64, "f32": np.float32, "i16": np.int16, "i32": np.int32}
    N, K, C, R = c["N"], c["K"], c["C"], c["R"]
    taps = [tuple(t) for t in TAPS[N][:K]]
    plans = _plans(N, taps, c["plan"])
    L = plans[0].L
    x = _mixture(N, taps, R, C, seed=500 + c["seed"], dtype=NPT[c["x"]])
    f, y = None, x.astype(np.float64)
    if c["filt"]:
        rng = np.random.default_rng(c["seed"])
        B, A = rng.standard_normal(c["filt"] + 1), np.poly(0.7 * rng.uniform(-1, 1, c["filt"]))
        f = IirFilter(B, A, device=0)
        y = lfilter(B, A, y, axis=0)
    if c["lin"] == "frame":
        xm, rs, cs = np.ascontiguousarray(x), C, 1
    elif c["lin"] == "stack":
        xm, rs, cs = np.ascontiguousarray(x.T), 1, R * L
    else:
        xm = np.zeros((R * L, C + 3), dtype=x.dtype)
        xm[:, :C] = x
        rs, cs = C + 3, 1
    group = Sources(plans)
    group.reserve(R * L, C, iir=f)
    npdt = plans[0].np_dtype
    d_x = DeviceBuffer(xm.nbytes, 0).upload(xm)
    d_h = DeviceBuffer(K * L * C * np.dtype(npdt).itemsize, 0)
    ors, ocs = (C, 1) if c["lout"] == "frame" else (1, L)
    group.impulse_device(d_x.ptr, _lib.dtype_code(x.dtype), R * L, C, rs, cs, d_h.ptr, L * C, ors, ocs, iir=f)
    sync(0)
    h = d_h.download((K, L, C) if c["lout"] == "frame" else (K, C, L), npdt)
    h = h if c["lout"] == "frame" else h.transpose(0, 2, 1)
    tol = 1e-12 if c["plan"] == "f64" else 2e-5
    for k in range(K):
        ref = c_oracle.sequence(N, taps[k])
        for ch in sorted({0, C // 2, C - 1}):
            assert rel_err(h[k, :, ch], oracle_ir(N, taps[k], np.ascontiguousarray(y[:, ch:ch + 1]), ref)) <= tol, (k, ch)
    d_x.free()
    d_h.free()
    if f is not None:
        f.close()
