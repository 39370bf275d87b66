"""GPU parity, floating-point part: impulse responses from the HIP path (through the C ABI)
against the golden fixtures captured from the reference and against the oracle.

Tolerances (BASELINE.json north_star): fp64 max-abs error <= 1e-12 x peak; fp32 mode <= 1e-5 x
peak. float32 CAPTURES are special: the reference transforms a float32 average with a
single-precision FFT, so its own output is ~1e-7 x peak off the exact correlation; those cases
are held to 1e-6 x peak."""
import os

import numpy as np
import pytest

import inputs
from gpu_common import G, case_input, load_json, oracle_ir, plan, rel_err
from oracle import c_oracle, mls_oracle
from zignal_amd import _lib
from zignal_amd import _lib as L_
from zignal_amd.device import DeviceBuffer, sync
from zignal_amd.measure.mls import MLS, TAPS, MLS_simple, _MLS_base

pytestmark = pytest.mark.gpu

TOL64 = 1e-12
TOL32 = 1e-5
IR_SMALL = load_json("ir_small.json")["cases"]
IR_LARGE = load_json("ir_large.json")["cases"]


def _tol(case):
    return 1e-6 if case["dtype"] == "float32" else TOL64


@pytest.mark.parametrize("case", IR_SMALL, ids=lambda c: "%s-N%d-%s-%s" % (c["id"], c["N"], c["kind"], c["dtype"]))
def test_golden_ir_small(case):
    arrays = np.load(os.path.join(G, "ir_small.npz"))
    x = case_input(case)
    m = _MLS_base(case["N"], tuple(case["taps"]), device=0)
    h = m.get_impulse(x)
    assert h.shape == (m.L, 1) and h.dtype == np.float64 and h.flags.c_contiguous
    assert rel_err(h[:, 0], arrays[case["id"] + "_h"]) <= _tol(case)
    avg = m.trim_and_avgerage(x)
    assert avg.shape == (m.L, 1) and str(avg.dtype) == case["avg_dtype"]
    assert rel_err(avg, mls_oracle.trim_and_average(x, m.L)) <= (1e-6 if case["dtype"] == "float32" else 1e-15)
    if case["dtype"] == "float64":
        assert rel_err(m.xcorr_circular(avg)[:, 0], arrays[case["id"] + "_h"]) <= TOL64
    # optional fp32 arithmetic
    h32 = _MLS_base(case["N"], tuple(case["taps"]), dtype="f32", device=0).get_impulse(x)
    assert h32.dtype == np.float64 and rel_err(h32[:, 0], arrays[case["id"] + "_h"]) <= TOL32


@pytest.mark.parametrize("case", IR_LARGE, ids=lambda c: "%s-N%d-%s-R%d" % (c["id"], c["N"], c["kind"], c["R"]))
def test_golden_ir_large(case):
    arrays = np.load(os.path.join(G, "ir_large.npz"))
    x = case_input(case)
    h = plan(case["N"], case["taps"]).impulse_host(x)[:, 0]
    tol = _tol(case) * case["peak"]
    k = case["id"]
    assert np.max(np.abs(h[arrays[k + "_idx"]] - arrays[k + "_h"])) <= tol
    lo = case["peakwin_lo"]
    assert np.max(np.abs(h[lo:lo + 256] - arrays[k + "_peakwin"])) <= tol
    assert int(np.argmax(np.abs(h))) == case["argmax"]
    assert abs(float(h[case["argmax"]]) - case["max"]) <= tol
    agg = 1e-5 if case["dtype"] == "float32" else 1e-9      # float32 captures: see module docstring
    assert abs(float(np.sum(h)) - case["sum"]) <= agg * max(1.0, abs(case["sum"]))
    assert abs(float(np.sum(h * h)) - case["sumsq"]) <= agg * case["sumsq"]
    if case["N"] <= 20 and case["dtype"] == "float64":
        h32 = plan(case["N"], case["taps"], "f32").impulse_host(x)[:, 0]
        assert np.max(np.abs(h32[arrays[k + "_idx"]] - arrays[k + "_h"])) <= TOL32 * case["peak"]


def test_reference_comment_example_on_device():
    m = _MLS_base(3, (3, 2), device=0)
    s = m.get_full_sequence(2)
    assert s[:7, 0].tolist() == [1, -1, -1, -1, 1, 1, -1]          # mls.py:183
    h = m.get_impulse(s)
    assert np.allclose(h[:, 0], [1] + [-1 / 7] * 6, atol=1e-15)     # mls.py:205


@pytest.mark.parametrize("N", list(range(3, 21)))
def test_every_order_against_oracle(N):
    """All pass splits (1 pass up to N=10, 2 passes above), first and last tap set, noise input."""
    rng = np.random.default_rng(1000 + N)
    for taps in (TAPS[N][0], TAPS[N][-1]):
        L = (1 << N) - 1
        R = 3 if N < 18 else 2
        x = rng.standard_normal((R * L, 1))
        want = oracle_ir(N, taps, x)
        assert rel_err(plan(N, taps).impulse_host(x)[:, 0], want) <= TOL64
        assert rel_err(plan(N, taps, "f32").impulse_host(x)[:, 0], want) <= TOL32


@pytest.mark.parametrize("N,C", [(5, 1), (7, 3), (9, 8), (10, 37), (12, 1), (12, 5), (13, 16), (14, 24), (16, 10), (17, 33)])
def test_multichannel_layouts(N, C):
    """(R*L, C) frame-major (vector path when C is a multiple of the vector width, element path
    otherwise), a transposed view (capture-major strides) and float32 captures: each column must
    equal the single-channel oracle result."""
    taps = TAPS[N][1 % len(TAPS[N])]
    L = (1 << N) - 1
    R = 4
    rng = np.random.default_rng(N * 100 + C)
    x = rng.standard_normal((R * L, C))
    ref = c_oracle.sequence(N, taps)
    want = np.stack([oracle_ir(N, taps, np.ascontiguousarray(x[:, c:c + 1]), ref) for c in range(C)], axis=1)
    p = plan(N, taps)
    h = p.impulse_host(x)
    assert h.shape == (L, C) and h.dtype == np.float64
    assert rel_err(h, want) <= TOL64
    xt = np.ascontiguousarray(x.T).T            # same values, capture-major memory
    assert xt.strides[0] == 8 and rel_err(p.impulse_host(xt), want) <= TOL64
    xs = np.zeros((R * L, 2 * C + 3))
    xs[:, 1:2 * C + 1:2] = x                    # every other column of a wider buffer
    assert rel_err(p.impulse_host(xs[:, 1:2 * C + 1:2]), want) <= TOL64
    x32 = x.astype(np.float32)
    want32 = np.stack([oracle_ir(N, taps, np.ascontiguousarray(x32[:, c:c + 1]).astype(np.float64), ref)
                       for c in range(C)], axis=1)
    assert rel_err(p.impulse_host(x32), want32) <= 1e-6
    p32 = plan(N, taps, "f32")
    assert rel_err(p32.impulse_host(x32), want32) <= TOL32
    m = _MLS_base(N, taps, device=0)
    assert rel_err(m.get_impulses(x), want) <= TOL64


def test_device_entry_point_chunks_channels_with_small_workspace():
    """zmls_impulse_device never allocates: with a workspace of one slab it must loop over the
    channel slabs itself and still give the same answer; with none it must refuse."""
    N, C, R = 13, 72, 3
    taps = TAPS[N][0]
    L = (1 << N) - 1
    rng = np.random.default_rng(7)
    x = rng.standard_normal((R * L, C))
    want = plan(N, taps).impulse_host(x)
    from zignal_amd.device import Plan
    p = Plan(N, taps, device=0)
    p.force_tile(0)                               # small tile: 8 float64 channels per slab
    dx = DeviceBuffer(x.nbytes).upload(x)
    dh = DeviceBuffer(L * C * 8)
    with pytest.raises(_lib.ZmlsError) as ei:
        p.impulse_device(dx.ptr, _lib.ZMLS_F64, R * L, C, C, 1, dh.ptr, C, 1)
    assert ei.value.code == _lib.E_WORKSPACE
    p.reserve(8)                                  # room for one slab of the small tile
    last = (1 << N) - 1                           # rows are padded: one extra row per 2^10 (and per 2^20) rows
    assert p.info()["workspace_bytes"] == (last + (last >> 10) + (last >> 20) + 1) * 64
    p.impulse_device(dx.ptr, _lib.ZMLS_F64, R * L, C, C, 1, dh.ptr, C, 1)     # 9 slabs -> 9 chunks
    got = dh.download((L, C), np.float64)
    assert rel_err(got, want) <= 1e-14
    p.force_tile(1)                               # big tile: 32 channels per slab
    with pytest.raises(_lib.ZmlsError) as ei:     # a small slab of workspace holds no big slab
        p.impulse_device(dx.ptr, _lib.ZMLS_F64, R * L, C, C, 1, dh.ptr, C, 1)
    assert ei.value.code == _lib.E_WORKSPACE
    p.reserve(8)                                  # forced tile: room for one of its slabs -> 3 chunks of one slab
    assert p.info()["workspace_bytes"] == (last + (last >> 10) + (last >> 20) + 1) * 256
    dh.fill(0)
    p.impulse_device(dx.ptr, _lib.ZMLS_F64, R * L, C, C, 1, dh.ptr, C, 1)
    assert rel_err(dh.download((L, C), np.float64), want) <= 1e-14
    p.close()


def test_linearity_shift_and_sum_properties():
    N, taps = 15, TAPS[15][0]
    L = (1 << N) - 1
    rng = np.random.default_rng(3)
    p = plan(N, taps)
    a = rng.standard_normal((2 * L, 1))
    b = rng.standard_normal((2 * L, 1))
    ha, hb = p.impulse_host(a), p.impulse_host(b)
    hab = p.impulse_host(2.5 * a - 0.75 * b)
    assert rel_err(hab, 2.5 * ha - 0.75 * hb) <= 1e-13
    d = 12345
    a_shift = np.concatenate((np.roll(a[:L], d, axis=0), np.roll(a[L:], d, axis=0)))
    assert rel_err(p.impulse_host(a_shift), np.roll(ha, d, axis=0)) <= 1e-13
    # sum_m h[m] = -(1/L) sum_j avg[j]
    assert abs(float(ha.sum()) + float(a[L:].sum()) / L) <= 1e-12 * max(1.0, abs(float(ha.sum())))
    # loop-back: two-valued autocorrelation
    s = p.sequence(3)
    h = p.impulse_host(s)[:, 0]
    assert h[0] == 1.0 and np.max(np.abs(h[1:] + 1.0 / L)) <= 1e-18 + 1e-16 / L
    # a pure delay moves the peak
    hd = p.impulse_host(np.roll(s, 777, axis=0))[:, 0]
    assert int(np.argmax(hd)) == 777 and hd[777] == 1.0


def test_integer_captures():
    N, taps = 12, TAPS[12][3]
    L = (1 << N) - 1
    rng = np.random.default_rng(5)
    p = plan(N, taps)
    for dt in (np.int16, np.int32):
        x = rng.integers(-20000, 20000, size=(3 * L, 2)).astype(dt)
        want = np.stack([oracle_ir(N, taps, np.ascontiguousarray(x[:, c:c + 1])) for c in range(2)], axis=1)
        assert rel_err(p.impulse_host(x), want) <= TOL64


def test_error_behaviour_through_the_c_abi():
    p = plan(5, (5, 3))
    L = 31
    lib = _lib.lib()
    x = np.zeros((3 * L, 1))
    h = np.zeros((L, 1))
    a = lambda n_rows: lib.zmls_impulse_host(p._h, x.ctypes.data, 0, n_rows, 1, 1, 1, h.ctypes.data, 1, 1)  # noqa: E731
    assert a(L) == _lib.E_ARG and "thrown away" in _lib.last_error()
    assert a(2 * L + 3) == _lib.E_ARG and "multiple" in _lib.last_error()
    assert a(3 * L) == 0
    assert lib.zmls_impulse_host(p._h, x.ctypes.data, 9, 3 * L, 1, 1, 1, h.ctypes.data, 1, 1) == _lib.E_ARG
    assert lib.zmls_impulse_host(p._h, None, 0, 3 * L, 1, 1, 1, h.ctypes.data, 1, 1) == _lib.E_ARG


def test_mls_and_mls_simple_classes_against_reference_goldens():
    meta = load_json("api.json")
    arrays = np.load(os.path.join(G, "api.npz"))
    g = meta["mls"]
    mls = MLS(N=g["N"], taps=tuple(g["taps"]), fs=g["fs"], repeats=g["repeats"], B=g["B"], A=g["A"], device=0)
    assert list(mls.samples.shape) == g["samples_shape"] and str(mls.samples.dtype) == g["samples_dtype"]
    assert (mls.nofsamples, mls.ch, mls.duration) == (g["nofsamples"], g["ch"], g["duration"])
    assert np.array_equal(mls.samples, arrays["mls_raw"])
    assert str(mls) == g["str"]
    mls.apply_emphasis()
    # emphasis runs on the device: lfilter's recurrence cut into parallel blocks, equal to rounding (floating point: tolerance)
    assert rel_err(mls.samples, arrays["mls_emph"]) <= TOL64
    y = np.roll(mls.samples, 100, axis=0) * 0.5
    z = mls.apply_deemphasis(y)
    assert rel_err(z, arrays["mls_deemph"]) <= TOL64
    k = mls.get_impulse(z)
    assert list(k.samples.shape) == g["ir_shape"] and (k.fs, k.ch, k.nofsamples) == (g["ir_fs"], g["ir_ch"], g["ir_nofsamples"])
    assert rel_err(k.samples, arrays["mls_ir"]) <= TOL64
    s = meta["simple"]
    ms = MLS_simple(N=s["N"], fs=s["fs"], repeats=s["repeats"], device=0)
    assert repr(ms) == s["repr"] and str(ms) == s["str"]
    assert rel_err(ms.samples, arrays["simple_samples"]) <= TOL64
    ir = ms.get_impulse(ms.samples)
    assert rel_err(ir.samples, arrays["simple_ir"]) <= TOL64
    with pytest.raises(AssertionError):
        MLS(5, (5, 3), repeats=2, B=(1, 2, 0), A=(1, 0, 0), device=0)      # not minimum phase


@pytest.mark.parametrize("N,C,dtype", [(11, 32, "f64"), (12, 64, "f64"), (13, 96, "f64"), (15, 40, "f64"), (16, 64, "f64"),
                                        (17, 32, "f64"), (18, 64, "f64"), (19, 32, "f64"), (20, 32, "f64"), (21, 32, "f64"),
                                        (22, 16, "f64"), (14, 64, "f32"), (18, 128, "f32"), (20, 64, "f32"), (21, 64, "f32"),
                                        (22, 32, "f32")])
def test_big_tile_against_small_tile_and_oracle(N, C, dtype):
    """Wide, contiguous channel blocks go through the register-resident tiles (128- and 256-byte row
    segments); the results must equal the small tile's to round-off and the oracle's within tolerance.
    Covers 2-pass (N<=20) and 3-pass (N=21, 22) plans, a ragged last slab (C=40, C=96), and the tall tile's
    two-pass schedule (11+10, 11+11 bits) that orders 21 and 22 take by default."""
    taps = TAPS[N][0]
    L = (1 << N) - 1
    R = 3 if N <= 18 else 2
    np_dt = np.float64 if dtype == "f64" else np.float32
    rng = np.random.default_rng(N * 1000 + C)
    x = rng.standard_normal((R * L, C)).astype(np_dt)
    p = plan(N, taps, dtype)
    p.force_tile(1)
    h_big = p.impulse_host(x)
    p.force_tile(2)
    h_mid = p.impulse_host(x)
    assert rel_err(h_mid, h_big.astype(np.float64)) <= (1e-14 if dtype == "f64" else 1e-5)
    p.force_tile(0)
    h_small = p.impulse_host(x)
    p.force_tile(-1)
    h_auto = p.impulse_host(x)
    if N in (21, 22):
        assert p.info()["tall_passes"] == 2
        p.force_tile(3)
        h_tall = p.impulse_host(x)
        p.force_tile(-1)
        assert np.array_equal(h_auto, h_tall)                 # auto picks the tall tile where it saves a pass
        assert rel_err(h_tall, h_mid.astype(np.float64)) <= (1e-14 if dtype == "f64" else 1e-5)
    else:
        assert p.info()["tall_passes"] == 0
        with pytest.raises(Exception):
            p.force_tile(3)
        # auto picks the mid tile for such input, unless its grid (2^(N-10) macro tiles per 16 f64 / 32 f32 channels)
        # has fewer than 512 workgroups: then the small tile's finer grid
        mid_grid = -(-C // p.info()["channels_per_slab_mid"]) << (N - 10)
        assert np.array_equal(h_auto, h_mid if mid_grid >= 512 else h_small)
    tol = TOL64 if dtype == "f64" else TOL32
    assert rel_err(h_big, h_small.astype(np.float64)) <= (1e-14 if dtype == "f64" else 1e-5)
    ref = c_oracle.sequence(N, taps)
    for c in (0, C // 2 + 1, C - 1):
        want = oracle_ir(N, taps, np.ascontiguousarray(x[:, c:c + 1]).astype(np.float64), ref)
        assert rel_err(h_big[:, c], want) <= tol, (N, c)


def test_capture_alloc_chooses_a_placement_and_the_buffer_works():
    """zmls_capture_alloc: under 1 GiB a plain allocation (probe time 0); above, candidates are rated with the
    plan's gather pass. Either way the buffer is an ordinary capture buffer."""
    N = 18
    taps = TAPS[N][0]
    p = plan(N, taps)
    L = p.L
    for C, R, probed in ((16, 2, False), (256, 3, True)):
        d_x, ms = p.alloc_capture(R * L, C, tries=2)
        assert (ms > 0.0) == probed
        assert d_x.nbytes == R * L * C * 8
        p.reserve(C)
        p.synth_device(d_x.ptr, _lib.ZMLS_F64, R, C, C, 1, seed=11, sigma=0.0)
        d_h = DeviceBuffer(L * C * 8)
        p.impulse_device(d_x.ptr, _lib.ZMLS_F64, R * L, C, C, 1, d_h.ptr, C, 1)
        h = d_h.download((L, C), np.float64)
        for c in (0, C - 1):
            g, d = p.synth_system(c, seed=11)
            want = np.zeros(L)
            want[(d + np.arange(16)) % L] = g
            # two-valued autocorrelation (L at lag 0, -1 elsewhere): h = g (1 + 1/L) - sum(g)/L
            want = want * (1.0 + 1.0 / L) - g.sum() / L
            assert np.max(np.abs(h[:, c] - want)) <= 1e-12
        d_x.free()
    with pytest.raises(_lib.ZmlsError):
        p.alloc_capture(L, 4)            # one period only: "The first sequence will be thrown away"


@pytest.mark.parametrize("N,C,R,dtype,stage", [(12, 40, 4, "f64", 40), (13, 33, 3, "f64", 33), (14, 70, 5, "f64", 16), (13, 48, 3, "f32", 48),
                                               (12, 1, 3, "f64", 1)])
def test_layout_adapters_on_the_device_entry_point(N, C, R, dtype, stage):
    """Captures stacked as (C, R*L) single-channel arrays in, responses (C, L) out, through the staged adapters
    (periods summed + transposed in, transposed out), also in channel chunks when the staging arrays are smaller
    than the batch, mixed with frame-major on one side, and int16 (frames, channels) input through the staged sum."""
    from zignal_amd.device import Plan
    taps = TAPS[N][0]
    L = (1 << N) - 1
    npdt = np.float64 if dtype == "f64" else np.float32
    tol = TOL64 if dtype == "f64" else TOL32
    rng = np.random.default_rng(N * 7 + C)
    x = rng.standard_normal((R * L, C)).astype(npdt)
    ref = c_oracle.sequence(N, taps)
    want = np.stack([oracle_ir(N, taps, np.ascontiguousarray(x[:, c:c + 1]).astype(np.float64), ref) for c in range(C)], axis=1)
    p = Plan(N, taps, dtype=dtype, device=0)
    p.reserve(C)
    L_.check(L_.lib().zmls_plan_reserve_staging(p._h, stage))
    xt = np.ascontiguousarray(x.T)                         # (C, R*L)
    d_x = DeviceBuffer(xt.nbytes).upload(xt)
    d_h = DeviceBuffer(L * C * x.itemsize)
    # capture-major in, capture-major out
    d_h.fill(0xFF)
    p.impulse_device(d_x.ptr, p.dtype_code, R * L, C, 1, R * L, d_h.ptr, 1, L)
    assert rel_err(d_h.download((C, L), npdt).T, want) <= tol
    # capture-major in, frame-major out
    d_h.fill(0xFF)
    p.impulse_device(d_x.ptr, p.dtype_code, R * L, C, 1, R * L, d_h.ptr, C, 1)
    assert rel_err(d_h.download((L, C), npdt), want) <= tol
    # frame-major in, capture-major out
    d_f = DeviceBuffer(x.nbytes).upload(x)
    d_h.fill(0xFF)
    p.impulse_device(d_f.ptr, p.dtype_code, R * L, C, C, 1, d_h.ptr, 1, L)
    assert rel_err(d_h.download((C, L), npdt).T, want) <= tol
    # int16 frames (what a sound card hands over), frame-major both sides
    xi = rng.integers(-30000, 30000, size=(R * L, C)).astype(np.int16)
    want_i = np.stack([oracle_ir(N, taps, xi[:, c:c + 1].astype(np.float64), ref) for c in range(C)], axis=1)
    d_i = DeviceBuffer(xi.nbytes).upload(xi)
    d_h.fill(0xFF)
    p.impulse_device(d_i.ptr, _lib.ZMLS_I16, R * L, C, C, 1, d_h.ptr, C, 1)
    assert rel_err(d_h.download((L, C), npdt), want_i) <= tol
    p.close()


@pytest.mark.parametrize("N,C,R,pad", [(11, 33, 3, 0), (12, 40, 4, 1), (12, 16, 2, 3), (13, 64, 5, 2), (14, 48, 3, 1)])
def test_stacked_float64_captures_at_every_alignment(N, C, R, pad):
    """round 3 (layout.hip, sum_transpose_pairs_kernel / transpose_out_pairs_kernel): stacked float64 captures are summed in 16-byte
    loads and stacked responses written in 16-byte stores, with the odd-started runs (L is odd) shifted by one frame. Buffers that
    start 8 bytes off a 16-byte boundary and captures with `pad` unused samples between them must give the very same responses."""
    from zignal_amd.device import Plan
    taps = TAPS[N][0]
    L = (1 << N) - 1
    rng = np.random.default_rng(N * 11 + C)
    x = rng.standard_normal((R * L, C))
    ref = c_oracle.sequence(N, taps)
    want = np.stack([oracle_ir(N, taps, np.ascontiguousarray(x[:, c:c + 1]), ref) for c in range(C)], axis=1)
    p = Plan(N, taps, device=0)
    p.reserve(C)
    L_.check(L_.lib().zmls_plan_reserve_staging(p._h, C))
    stride = R * L + pad
    results = []
    for off_in in (0, 1):
        for off_out in (0, 1):
            host = np.full(off_in + C * stride, np.nan)
            for c in range(C):
                host[off_in + c * stride: off_in + c * stride + R * L] = x[:, c]
            d_x = DeviceBuffer(host.nbytes).upload(host)
            d_h = DeviceBuffer((off_out + C * L) * 8)
            d_h.fill(0xFF)
            p.impulse_device(d_x.ptr + 8 * off_in, p.dtype_code, R * L, C, 1, stride, d_h.ptr + 8 * off_out, 1, L)
            sync(0)
            out = d_h.download((off_out + C * L,), np.float64)
            if off_out:
                assert np.isnan(out[0])                      # 0xFF fill: the element before the first response is untouched
            h = out[off_out:].reshape(C, L).T
            assert rel_err(h, want) <= TOL64
            results.append(h)
            d_x.free(); d_h.free()
    for h in results[1:]:
        assert np.array_equal(h, results[0])
    p.close()


def test_order_26_single_capture_needs_only_a_small_slab_of_workspace():
    """Orders beyond the BASELINE range (the reference's tap table goes to 32; this library to 30): a three-pass plan
    of 2^26 points, two channels, checked against the known synthetic system; the workspace is one small-tile slab."""
    from zignal_amd.device import Plan
    N, C, R = 26, 2, 2
    p = Plan(N, TAPS[N][0], device=0)
    L = p.L
    assert p.info()["passes"] == 3
    d_x = DeviceBuffer(R * L * C * 8)
    d_h = DeviceBuffer(L * C * 8)
    p.reserve(C)
    assert p.info()["workspace_bytes"] <= (1 << N) * 65
    p.synth_device(d_x.ptr, _lib.ZMLS_F64, R, C, C, 1, seed=3, sigma=0.0)
    p.impulse_device(d_x.ptr, _lib.ZMLS_F64, R * L, C, C, 1, d_h.ptr, C, 1)
    h = d_h.download((L, C), np.float64)
    for c in range(C):
        g, d = p.synth_system(c, seed=3)
        want = np.zeros(L)
        want[(d + np.arange(16)) % L] = g
        want = want * (1.0 + 1.0 / L) - g.sum() / L
        assert np.max(np.abs(h[:, c] - want)) <= 1e-12
    assert np.array_equal(p.bits_packed(), c_oracle.bits_packed(N, TAPS[N][0]))
    p.close()


@pytest.mark.gpu
@pytest.mark.parametrize("N,C,np_dt", [(10, 1, np.int16), (14, 1, np.int16), (16, 1, np.int32), (17, 6, np.int16),
                                        (18, 16, np.int32), (12, 3, np.int32)])
def test_raw_pcm_captures_equal_convert_to_float_then_get_impulse(N, C, np_dt):
    """int16/int32 frames straight from the sound card / a wav file. The fused path must give
    what Audio.convert_to_float (audio.py:582, samples / 2^(bits-1)) followed by get_impulse gives: against the oracle on
    the converted capture within 1e-12 of the peak, and exactly 2^-(bits-1) times the response of the raw integers."""
    from zignal_amd.audio import Audio
    taps = TAPS[N][0]
    L = (1 << N) - 1
    R = 3
    bits = 8 * np.dtype(np_dt).itemsize
    rng = np.random.default_rng(N * 31 + C)
    seq = c_oracle.sequence(N, taps)
    one = np.asarray(seq, dtype=np.float64).reshape(-1)
    g = rng.standard_normal(8) * 0.2
    cols = []
    for c in range(C):
        y = np.zeros(L)
        for k, gk in enumerate(g):
            y += gk * np.roll(one, 100 * c + k)
        y = np.tile(y, R) + 1e-3 * rng.standard_normal(R * L)
        cols.append(y / np.max(np.abs(y)) * 0.9)
    a = Audio(fs=48000, initialdata=np.stack(cols, axis=1))
    a.convert_to_integer(targetbits=bits)
    x_int = a.samples
    assert x_int.dtype == np_dt
    m = _MLS_base(N, taps, device=0)
    h = m.get_impulse_pcm(x_int if C > 1 else x_int[:, 0])
    assert h.shape == (L, C) and h.dtype == np.float64
    raw = m.get_impulses(x_int)
    assert np.array_equal(h, raw * 2.0 ** -(bits - 1))
    assert np.array_equal(m.get_impulses(x_int), raw)              # the scale does not stick to the plan
    b = Audio(fs=48000, initialdata=x_int)
    b.convert_to_float()
    assert b.samples.dtype == np.float64 and np.array_equal(b.samples, x_int / 2.0 ** (bits - 1))
    for c in sorted({0, C - 1}):
        want = oracle_ir(N, taps, np.ascontiguousarray(b.samples[:, c:c + 1]), seq)
        assert rel_err(h[:, c], want) <= TOL64
    with pytest.raises(TypeError):
        m.get_impulse_pcm(b.samples)


@pytest.mark.parametrize("N,dtype", [(2, "f64"), (3, "f64"), (4, "f64"), (5, "f64"), (6, "f64"), (7, "f32"), (8, "f64"), (9, "f64"),
                                      (10, "f64"), (10, "f32"), (6, "f32")])
def test_single_pass_orders_in_the_register_tile(N, dtype):
    """Orders up to 10 are one pass. Wide aligned batches take it in the mid tile (whole transforms of 2^(10-N) slabs of
    16 f64 / 32 f32 channels per workgroup, 16-byte vectors); channels that do not fill a workgroup's set of slabs, and
    everything else, go through the small tile's element kernel. Both must agree to round-off and with the oracle."""
    taps = TAPS[N][0] if N in TAPS else (2, 1)          # zignal's table starts at N = 3
    L = (1 << N) - 1
    R = 4
    np_dt = np.float64 if dtype == "f64" else np.float32
    p = plan(N, taps, dtype)
    per_wg = p.info()["channels_per_slab_mid"] << (10 - N)
    tol_pair = 1e-14 if dtype == "f64" else 1e-5
    for C in (per_wg, 3 * per_wg + 6 * (2 if dtype == "f64" else 4), per_wg // 2 if per_wg >= 64 else 2 * per_wg):
        if C * R * L * 8 > (1 << 28):
            continue
        rng = np.random.default_rng(N * 7 + C)
        x = rng.standard_normal((R * L, C)).astype(np_dt)
        p.force_tile(2)
        h_mid = p.impulse_host(x)
        p.force_tile(0)
        h_small = p.impulse_host(x)
        p.force_tile(-1)
        assert rel_err(h_mid, h_small.astype(np.float64)) <= tol_pair, (N, C)
        ref = c_oracle.sequence(N, taps)
        for c in sorted({0, C // 2, C - 1}):
            want = oracle_ir(N, taps, np.ascontiguousarray(x[:, c:c + 1]).astype(np.float64), ref)
            assert rel_err(h_mid[:, c], want) <= (TOL64 if dtype == "f64" else TOL32), (N, C, c)
    # enough channels for 512 workgroups: the mid tile without being forced; an odd period count exercises the exact mean
    if N >= 7:
        C = 512 * per_wg
        x = np.random.default_rng(N).standard_normal((3 * L, C)).astype(np_dt)
        h_auto = p.impulse_host(x)
        p.force_tile(0)
        h_small = p.impulse_host(x[:, :64])
        p.force_tile(2)
        assert np.array_equal(p.impulse_host(x), h_auto)
        p.force_tile(-1)
        assert rel_err(h_auto[:, :64], h_small.astype(np.float64)) <= tol_pair


def test_capture_alloc_stats_and_free_through_the_c_abi():
    """small captures are simply allocated (nothing to choose: zero candidates rated); zmls_capture_free gives the buffer back"""
    import ctypes
    p = plan(12, TAPS[12][0])
    lib = _lib.lib()
    ptr, stats = ctypes.c_void_p(), (ctypes.c_float * 8)()
    assert lib.zmls_capture_alloc_stats(p._h, _lib.ZMLS_F64, 3 * p.L, 8, 0, ctypes.byref(ptr), stats) == 0 and ptr.value
    assert list(stats)[:3] == [0.0, 0.0, 0.0]
    assert lib.zmls_dev_memset(0, ptr, 0, 3 * p.L * 8 * 8) == 0
    assert lib.zmls_capture_free(p._h, ptr) == 0
    assert lib.zmls_capture_free(p._h, None) == 0 and lib.zmls_capture_free(None, None) == _lib.E_ARG
    assert lib.zmls_capture_alloc_stats(p._h, _lib.ZMLS_F64, p.L, 8, 0, ctypes.byref(ptr), stats) == _lib.E_ARG      # one period: nothing to average


@pytest.mark.parametrize("np_in,plan_dtype", [(np.int16, "f64"), (np.int32, "f64"), (np.float32, "f64"), (np.int16, "f32"), (np.int32, "f32"),
                                              (np.float64, "f32")])
@pytest.mark.parametrize("N,C,R", [(12, 24, 6), (13, 64, 3), (12, 8, 11)])
def test_frame_major_captures_of_other_sample_types_take_the_vector_adapter(np_in, plan_dtype, N, C, R):
    """(frames, channels) captures whose sample type is not the plan's - raw PCM, float32 into a float64 plan, float64 into a
    float32 plan - are added up by period_sum_cvt_kernel (16-byte loads, converted in registers). Same additions in the same order
    as the element adapter, which the same capture takes as soon as its rows are padded by one sample: bit-identical; and the
    oracle on the converted capture."""
    taps = TAPS[N][0]
    p = plan(N, taps, plan_dtype)
    L = p.L
    rng = np.random.default_rng(N * 1000 + C + R)
    seq = c_oracle.sequence(N, taps)[:, 0]
    x = np.stack([np.tile(np.roll(seq, 37 * c + 1), R) * (0.2 + 0.01 * c) for c in range(C)], axis=1) + 0.05 * rng.standard_normal((R * L, C))
    if np.issubdtype(np_in, np.integer):
        x = np.round(x * (np.iinfo(np_in).max / 4.0)).astype(np_in)
    else:
        x = x.astype(np_in)
    code = _lib.dtype_code(np_in)
    p.reserve(C, staging=True)
    es_out = np.dtype(p.np_dtype).itemsize
    d_h = DeviceBuffer(L * C * es_out, 0)
    got = []
    for pad in (0, 1):                                   # pad 1: rows of C+1 samples, not a whole number of vectors -> element adapter
        xd = np.zeros((R * L, C + pad), dtype=np_in)
        xd[:, :C] = x
        d_x = DeviceBuffer(xd.nbytes, 0).upload(xd)
        p.impulse_device(d_x.ptr, code, R * L, C, C + pad, 1, d_h.ptr, C, 1)
        _lib.check(_lib.lib().zmls_dev_sync(0))
        got.append(d_h.download((L, C), p.np_dtype))
        d_x.free()
    d_h.free()
    assert np.array_equal(got[0], got[1])
    tol = TOL32 if (plan_dtype == "f32" or np_in == np.float32) else TOL64
    for c in (0, C // 2, C - 1):
        assert rel_err(got[0][:, c], oracle_ir(N, taps, x[:, c:c + 1].astype(np.float64))) <= tol
