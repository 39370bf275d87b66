"""GPU parity for the emphasis / de-emphasis filter: zmls_iir_* against
scipy.signal.lfilter, the very routine the reference calls (zignal/filters/linearfilter.py:62-63), on the same
inputs. Tolerance 1e-12 x peak (float64 arithmetic on both sides; the block-parallel start states differ from the
sequential ones by rounding only)."""
import numpy as np
import pytest
import scipy.signal

from gpu_common import rel_err
from zignal_amd import _lib
from zignal_amd.device import DeviceBuffer, IirFilter, Plan, sync
from zignal_amd.filters.biquads import RBJ
from zignal_amd.measure.mls import MLS, TAPS, MLS_simple

pytestmark = pytest.mark.gpu
TOL = 1e-12


def stable_filter(order, rng):
    """random minimum-phase B and A of the given order (roots inside the unit circle)"""
    def poly():
        r = rng.uniform(0.2, 0.97, size=order) * np.exp(1j * rng.uniform(0, np.pi, size=order))
        roots = []
        for k in range(0, order - 1, 2):
            roots += [r[k], np.conj(r[k])]
        if order % 2:
            roots.append(rng.uniform(-0.95, 0.95))
        return np.real(np.poly(roots)) if roots else np.array([1.0])
    return rng.uniform(0.5, 2.0) * poly(), rng.uniform(0.5, 2.0) * poly()


@pytest.mark.parametrize("order", range(0, 9))
def test_every_order_against_lfilter_single_channel(order):
    rng = np.random.default_rng(100 + order)
    B, A = stable_filter(order, rng)
    f = IirFilter(B, A, device=0)
    for n in (1, 31, 32, 33, 8191, 8192, 8193, 100003):          # around the sub-block and chunk sizes
        x = rng.standard_normal(n)
        want = scipy.signal.lfilter(B, A, x, axis=0)
        assert rel_err(f.filter_host(x), want) <= TOL
        assert f.filter_host(x.reshape(-1, 1)).shape == (n, 1)
    f.close()


@pytest.mark.parametrize("shape,dtype", [((70000, 3), np.float64), ((20011, 16), np.float64), ((9000, 37), np.float32),
                                         ((50000, 5), np.int16), ((4097, 1), np.int32),
                                         # the chunks' start states come from the second-level scan: 586 chunks = 3 groups of
                                         # 256 frame-major, 37 chunks = 1 group time-contiguous; then 18 and 2 groups
                                         ((300007, 6), np.float64), ((2300000, 2), np.float32)])
def test_frames_by_channels_and_dtypes(shape, dtype):
    rng = np.random.default_rng(shape[0] + shape[1])
    B, A = RBJ(filtertype="highshelf", gaindb=-10, f0=100, Q=0.707, fs=48000).get_coefficients()
    if np.issubdtype(dtype, np.integer):
        x = rng.integers(-30000, 30000, size=shape).astype(dtype)
    else:
        x = rng.standard_normal(shape).astype(dtype)
    for b, a in ((B, A), (A, B)):                                # emphasis and its inverse (mls.py:365-366)
        f = IirFilter(b, a, device=0)
        want = scipy.signal.lfilter(b, a, x.astype(np.float64), axis=0)
        got = f.filter_host(x)
        assert got.dtype == np.float64 and got.shape == x.shape
        assert rel_err(got, want) <= TOL
        xt = np.ascontiguousarray(x.T).T                          # same values, time-contiguous memory
        assert rel_err(f.filter_host(xt), want) <= TOL
        f.close()


def test_device_entry_point_in_place_and_workspace_error():
    rng = np.random.default_rng(3)
    n, C = 40000, 8
    x = rng.standard_normal((n, C))
    B, A = stable_filter(2, rng)
    f = IirFilter(B, A, device=0)
    d = DeviceBuffer(x.nbytes).upload(x)
    with pytest.raises(_lib.ZmlsError) as ei:
        f.filter_device(d.ptr, _lib.ZMLS_F64, n, C, C, 1, d.ptr, C, 1)
    assert ei.value.code == _lib.E_WORKSPACE
    f.reserve(n, C)
    f.filter_device(d.ptr, _lib.ZMLS_F64, n, C, C, 1, d.ptr, C, 1)          # in place
    sync(0)
    assert rel_err(d.download((n, C), np.float64), scipy.signal.lfilter(B, A, x, axis=0)) <= TOL
    f.close()
    with pytest.raises(_lib.ZmlsError) as ei:
        IirFilter(np.ones(10), np.ones(3), device=0)
    assert ei.value.code == _lib.E_UNSUPPORTED


def test_one_filter_object_on_a_long_narrow_then_a_shorter_wide_signal():
    """The group arrays of the second-level scan are sized apart from the chunk arrays: a long one-channel signal needs
    MORE chunk states and FEWER group states than a shorter signal of many channels (round-2 advisor finding: the second
    call's group scan wrote past arrays sized by the first)."""
    rng = np.random.default_rng(77)
    B, A = stable_filter(2, rng)
    f = IirFilter(B, A, device=0)
    for shape in ((17_800_000, 1), (8704, 1000), (1_200_000, 2), (9000, 700)):     # 34766 chunk states, then 17000 with 1000 group states
        x = rng.standard_normal(shape)
        want = scipy.signal.lfilter(B, A, x, axis=0)
        assert rel_err(f.filter_host(x), want) <= TOL                          # frame-major: 512-frame chunks, 17+ of them
    d = DeviceBuffer(8 * 8704 * 2400)
    with pytest.raises(_lib.ZmlsError) as ei:                                  # wider than anything reserved: refused, not overrun
        f.filter_device(d.ptr, _lib.ZMLS_F64, 8704, 2400, 2400, 1, d.ptr, 2400, 1)
    assert ei.value.code == _lib.E_WORKSPACE
    d.free()
    f.close()


def test_long_capture_like_the_headline_stimulus():
    """(R*L, 1) at N=20, R=3: 3.1 M samples, 384 chunks; de-emphasis of an emphasised MLS gives the MLS back."""
    N, R = 20, 3
    B, A = RBJ(filtertype="highshelf", gaindb=-10, f0=100, Q=0.707, fs=96000).get_coefficients()
    m = MLS(N=N, taps=TAPS[N][0], fs=96000, repeats=R, B=B, A=A, device=0)
    s0 = m.samples.copy()
    m.apply_emphasis()
    assert rel_err(m.samples, scipy.signal.lfilter(B, A, s0, axis=0)) <= TOL
    back = m.apply_deemphasis(m.samples)
    assert np.max(np.abs(back - s0)) <= 1e-9
    k = m.get_impulse(back)
    assert abs(k.samples[0, 0] - 1.0) <= 1e-9


def test_mls_simple_fused_deemphasis_and_impulse_matches_the_two_steps():
    ms = MLS_simple(N=14, fs=48000, repeats=4, device=0)
    rng = np.random.default_rng(9)
    y = ms.samples + 1e-3 * rng.standard_normal(ms.samples.shape)
    k = ms.get_impulse(y)
    mls = ms._mls
    two = mls.get_impulse(mls.apply_deemphasis(y))
    assert k.samples.shape == two.samples.shape == (mls.L, 1)
    assert rel_err(k.samples, two.samples) <= 1e-14
    want = scipy.signal.lfilter(*mls._filter_deemphasis.get_coefficients(), y, axis=0)
    assert rel_err(mls.apply_deemphasis(y), want) <= TOL
    assert abs(k.samples[0, 0] - 1.0) < 1e-3


# ---------------------------------------------------------------- de-emphasis fused with the impulse response
def _fused(plan, iir, x, rs=None, cs=None, host=None):
    """zmls_impulse_filtered_device on capture x (frames, channels); `host` is the array actually uploaded when the
    memory layout differs from x's"""
    n_rows, C = x.shape
    src = x if host is None else host
    d_x = DeviceBuffer(src.nbytes).upload(src)
    d_h = DeviceBuffer(plan.L * C * plan.np_dtype().itemsize)
    d_h.fill(0xFF)
    iir.reserve_fused(plan, n_rows, C)
    plan.impulse_filtered_device(iir, d_x.ptr, _lib.dtype_code(src.dtype), n_rows, C, C if rs is None else rs, 1 if cs is None else cs,
                                 d_h.ptr, C, 1)
    sync(0)
    h = d_h.download((plan.L, C), plan.np_dtype).astype(np.float64)
    d_x.free()
    d_h.free()
    return h


def _reference_chain(N, taps, b, a, x):
    """what the reference does (mls.py:401-414): lfilter over the whole capture, then get_impulse, channel by channel"""
    from gpu_common import oracle_ir
    from oracle import c_oracle
    ref = c_oracle.sequence(N, taps)
    y = scipy.signal.lfilter(b, a, x.astype(np.float64), axis=0)
    return np.stack([oracle_ir(N, taps, np.ascontiguousarray(y[:, c:c + 1]), ref) for c in range(x.shape[1])], axis=1)


@pytest.mark.parametrize("N,C,R,order", [(5, 3, 4, 2), (6, 16, 2, 1), (8, 5, 6, 3), (10, 32, 3, 2), (12, 7, 5, 4), (13, 48, 4, 8),
                                         (14, 16, 17, 2), (16, 32, 3, 2), (17, 4, 2, 5)])
def test_fused_deemphasis_and_impulse_against_lfilter_then_oracle(N, C, R, order):
    """Short orders on purpose: at N = 5..8 the filter's memory spans several periods (poles up to 0.97: thousands of
    samples against L = 31..255), so the summed period-boundary states carry most of the answer."""
    from gpu_common import plan
    taps = TAPS[N][0]
    p = plan(N, taps)
    rng = np.random.default_rng(1000 * N + C)
    b, a = stable_filter(order, rng)
    f = IirFilter(b, a, device=0)
    x = rng.standard_normal((R * p.L, C))
    want = _reference_chain(N, taps, b, a, x)
    assert rel_err(_fused(p, f, x), want) <= TOL
    # the same capture as a stack of single-channel arrays (channels, frames): time-contiguous memory
    xt = np.ascontiguousarray(x.T)
    assert rel_err(_fused(p, f, x, rs=1, cs=R * p.L, host=xt), want) <= TOL
    f.close()


@pytest.mark.parametrize("dtype", [np.float32, np.int16, np.int32])
def test_fused_path_input_types_and_float32_plan(dtype):
    from gpu_common import plan
    N, C, R = 11, 12, 3
    taps = TAPS[N][1]
    rng = np.random.default_rng(77)
    b, a = RBJ(filtertype="highshelf", gaindb=-10, f0=100, Q=0.707, fs=48000).get_coefficients()
    x = (rng.standard_normal((R * ((1 << N) - 1), C)) * 3000).astype(dtype)
    want = _reference_chain(N, taps, a, b, x)                       # de-emphasis: the inverse filter
    f = IirFilter(a, b, device=0)
    assert rel_err(_fused(plan(N, taps), f, x), want) <= TOL
    assert rel_err(_fused(plan(N, taps, "f32"), f, x), want) <= 1e-5
    f.close()


def test_fused_path_equals_the_two_device_steps_at_n20():
    """N=20, 8 channels x 3 periods: against zmls_iir_device followed by zmls_impulse_device on the same buffers"""
    from gpu_common import plan
    N, C, R = 20, 8, 3
    p = plan(N, TAPS[N][0])
    L = p.L
    b, a = RBJ(filtertype="highshelf", gaindb=-10, f0=100, Q=0.707, fs=96000).get_coefficients()
    f = IirFilter(a, b, device=0)
    d_x, d_y, d_h = DeviceBuffer(R * L * C * 8), DeviceBuffer(R * L * C * 8), DeviceBuffer(L * C * 8)
    p.synth_device(d_x.ptr, _lib.ZMLS_F64, R, C, C, 1, sigma=1e-2)
    f.reserve(R * L, C)
    p.reserve(C)
    f.filter_device(d_x.ptr, _lib.ZMLS_F64, R * L, C, C, 1, d_y.ptr, C, 1)
    p.impulse_device(d_y.ptr, _lib.ZMLS_F64, R * L, C, C, 1, d_h.ptr, C, 1)
    sync(0)
    two = d_h.download((L, C), np.float64)
    f.reserve_fused(p, R * L, C)
    d_h.fill(0xFF)
    p.impulse_filtered_device(f, d_x.ptr, _lib.ZMLS_F64, R * L, C, C, 1, d_h.ptr, C, 1)
    sync(0)
    assert rel_err(d_h.download((L, C), np.float64), two) <= TOL
    # without the reservation: a clear error, nothing launched
    f2 = IirFilter(a, b, device=0)
    with pytest.raises(_lib.ZmlsError) as ei:
        p.impulse_filtered_device(f2, d_x.ptr, _lib.ZMLS_F64, R * L, C, C, 1, d_h.ptr, C, 1)
    assert ei.value.code == _lib.E_WORKSPACE
    for d in (d_x, d_y, d_h):
        d.free()
    f.close()
    f2.close()


def _fused_form(plan, iir, x, form, stacked_in=False, stacked_out=False):
    """zmls_impulse_filtered_device in one of its two forms on capture x (frames, channels); either side frames x channels
    or a stack of single-channel arrays. Returns the responses (L, channels) and the form the call took."""
    n_rows, C = x.shape
    src = np.ascontiguousarray(x.T) if stacked_in else x
    d_x = DeviceBuffer(src.nbytes).upload(src)
    d_h = DeviceBuffer(plan.L * C * 8)
    d_h.fill(0xFF)
    iir.reserve_fused(plan, n_rows, C)
    iir.set_fused_form(form)
    rs, cs = (1, n_rows) if stacked_in else (C, 1)
    ors, ocs = (1, plan.L) if stacked_out else (C, 1)
    plan.impulse_filtered_device(iir, d_x.ptr, _lib.dtype_code(src.dtype), n_rows, C, rs, cs, d_h.ptr, ors, ocs)
    sync(0)
    took = iir.last_fused_form()
    iir.set_fused_form(-1)
    h = d_h.download((C, plan.L) if stacked_out else (plan.L, C), np.float64)
    d_x.free()
    d_h.free()
    return (h.T if stacked_out else h), took


@pytest.mark.parametrize("N,C,R,order,stacked_in,stacked_out", [(12, 7, 5, 4, False, False), (13, 16, 4, 2, False, False), (14, 5, 3, 3, False, False),
                                                               (14, 1, 3, 2, True, True), (15, 3, 2, 1, True, False), (16, 32, 17, 2, False, False),
                                                               (16, 4, 3, 8, False, True), (17, 4, 2, 5, True, True), (18, 2, 3, 2, False, False)])
def test_post_filter_form_of_the_fused_call(N, C, R, order, stacked_in, stacked_out):
    """round 3: filter and transform commute up to a rank-S term (csrc/iir.hip, impulse_filtered_post), so the capture takes the
    ordinary gather pass and the filter runs over the responses. Forced (form 1) it runs wherever the filter's memory fits a
    period; against the sweep form (form 0) on the same buffers and against lfilter-then-oracle (mls.py:401-414)."""
    from gpu_common import plan
    taps = TAPS[N][0]
    p = plan(N, taps)
    rng = np.random.default_rng(2000 * N + C)
    b, a = stable_filter(order, rng)
    f = IirFilter(b, a, device=0)
    x = rng.standard_normal((R * p.L, C))
    sweep, took0 = _fused_form(p, f, x, 0, stacked_in, stacked_out)
    post, took1 = _fused_form(p, f, x, 1, stacked_in, stacked_out)
    assert (took0, took1) == (0, 1)
    assert rel_err(post, sweep) <= 1e-13
    if N <= 16:
        assert rel_err(post, _reference_chain(N, taps, b, a, x)) <= TOL
    for mode, m in ((1, 3), (0, 0)):                        # the filter over the responses in its one-sweep form (forced), and never
        f.set_one_sweep(mode, m)
        got, took = _fused_form(p, f, x, 1, stacked_in, stacked_out)
        assert took == 1 and rel_err(got, sweep) <= 1e-13, (mode, m)
    f.set_one_sweep(-1)
    # again on the same filter and plan (the tail and state arrays are reused), with a shorter capture
    if R > 2:
        x2 = x[:2 * p.L]
        again, took = _fused_form(p, f, x2, 1, stacked_in, stacked_out)
        ref2, _ = _fused_form(p, f, x2, 0, stacked_in, stacked_out)
        assert took == 1 and rel_err(again, ref2) <= 1e-13
    f.close()


def test_post_filter_form_choice_and_raw_pcm_scale():
    """auto: the post-filter form when the filter's memory is a small part of a period (N=16: 65535 frames against ~8000 for the
    reference's shelving filter), the sweep form at short orders and for float32 plans; int16 captures with the 2^-15 scale
    folded into the plan (zmls_plan_set_input_scale_log2) come out scaled in both forms."""
    from gpu_common import plan
    b, a = RBJ(filtertype="highshelf", gaindb=-10, f0=100, Q=0.707, fs=48000).get_coefficients()
    f = IirFilter(a, b, device=0)
    rng = np.random.default_rng(31)
    for N, want_form in ((16, 1), (11, 0)):
        p = plan(N, TAPS[N][0])
        x = rng.standard_normal((3 * p.L, 4))
        got, took = _fused_form(p, f, x, -1)
        assert took == want_form
        assert rel_err(got, _reference_chain(N, TAPS[N][0], a, b, x)) <= TOL
    p32 = plan(16, TAPS[16][0], "f32")
    f.reserve_fused(p32, 3 * p32.L, 4)
    d_x, d_h = DeviceBuffer(3 * p32.L * 4 * 8), DeviceBuffer(p32.L * 4 * 4)
    d_x.fill(0)
    p32.impulse_filtered_device(f, d_x.ptr, _lib.ZMLS_F64, 3 * p32.L, 4, 4, 1, d_h.ptr, 4, 1)
    sync(0)
    assert f.last_fused_form() == 0
    d_x.free(); d_h.free()
    # raw PCM
    N = 16
    p = Plan(N, TAPS[N][0], device=0)
    xi = (rng.standard_normal((3 * p.L, 6)) * 3000).astype(np.int16)
    p.set_input_scale_log2(15)
    want = _reference_chain(N, TAPS[N][0], a, b, xi.astype(np.float64) / 32768.0)
    for form in (0, 1):
        got, took = _fused_form(p, f, xi, form)
        assert took == form and rel_err(got, want) <= TOL, form
    p.close()
    f.close()


def test_post_filter_form_with_segments_of_the_time_axis_and_float32_captures():
    """the rank-S term rides in the filter's second sweep: with the time axis cut into segments (zmls_iir_set_segment) every
    segment must pick its own rows of psi; float32 captures go through the layout adapter first"""
    from gpu_common import plan
    N, C, R = 17, 256, 2                  # 16 channel groups: a segment is at least 128 chunks = 65536 of the 131071 frames (launch_iir_s)
    taps = TAPS[N][0]
    p = plan(N, taps)
    rng = np.random.default_rng(5)
    b, a = stable_filter(3, rng)
    f = IirFilter(b, a, device=0)
    x = rng.standard_normal((R * p.L, C))
    want, _ = _fused_form(p, f, x, 0)
    whole, took = _fused_form(p, f, x, 1)
    assert took == 1 and rel_err(whole, want) <= 1e-13
    for seg in (1 << 20, 160 << 20):      # two segments (the smallest there are); 80000 frames and the rest
        f.set_segment(seg)
        got, took = _fused_form(p, f, x, 1)
        assert took == 1 and rel_err(got, want) <= 1e-13, seg
        assert not np.array_equal(got, whole), "the segments were not cut: the start states of the second one come by another route"
    x = x[:, :8].copy()
    want, _ = _fused_form(p, f, x, 0)
    f.set_segment(None)
    x32 = x.astype(np.float32)
    assert rel_err(_fused_form(p, f, x, 1)[0], want) <= 1e-13
    want32, _ = _fused_form(p, f, x32, 0)
    got32, took = _fused_form(p, f, x32, 1)
    assert took == 1 and rel_err(got32, want32) <= 1e-13
    f.close()


@pytest.mark.parametrize("shape,layout,dtype,order,M", [((70000, 3), "frames", np.float64, 2, 1), ((20011, 16), "frames", np.float64, 5, 3),
                                                        ((300007, 6), "frames", np.float64, 8, 8), ((1 << 20, 1), "time", np.float64, 2, 2),
                                                        ((1 << 17, 3), "stacked", np.float64, 3, 5), ((90000, 37), "frames", np.float32, 4, 2),
                                                        ((50000, 5), "frames", np.int16, 1, 7), ((700, 5), "frames", np.float64, 2, 4)])
def test_one_sweep_form_for_short_memory_filters(shape, layout, dtype, order, M):
    """round 3 (csrc/iir.hip, iir_once_kernel): a workgroup walks M chunks carrying the state after a warm-up over the chunks that
    still ring at its first one; forced here with small M (many workgroups, many warm-ups) against the two sweeps and lfilter.
    In place the library must fall back to the two sweeps by itself."""
    rng = np.random.default_rng(shape[0] % 9973 + M)
    n, C = shape
    x = (rng.standard_normal(shape) * 1000).astype(dtype)
    B, A = stable_filter(order, rng)
    f = IirFilter(B, A, device=0)
    f.reserve(n, C)
    host = np.ascontiguousarray(x.T) if layout == "stacked" else x
    rs, cs = (1, n) if layout == "stacked" else (C, 1)
    dx = DeviceBuffer(host.nbytes).upload(host)
    dy = DeviceBuffer(n * C * 8)
    want = scipy.signal.lfilter(B, A, x.astype(np.float64), axis=0)

    def run(mode, m=0, src=None):
        f.set_one_sweep(mode, m)
        if src is None:
            dy.fill(0xFF)
        f.filter_device((src or dx).ptr, _lib.dtype_code(host.dtype) if src is None else _lib.ZMLS_F64, n, C, rs, cs, dy.ptr, rs, cs)
        sync(0)
        out = dy.download(host.shape, np.float64)
        return out.T if layout == "stacked" else out

    two = run(0)
    assert rel_err(two, want) <= TOL
    one = run(1, M)
    assert rel_err(one, want) <= TOL and rel_err(one, two) <= 1e-13
    assert np.array_equal(run(1, M), one), "same bits every time"
    assert rel_err(run(1, 4096), want) <= TOL               # one workgroup per channel group walks the whole signal: no warm-up at all
    assert rel_err(run(-1), want) <= TOL
    if dtype == np.float64:
        dy.upload(host)                                      # in place: the warm-up would read outputs - the two sweeps run instead
        assert np.array_equal(run(1, M, src=dy), two)
    f.set_one_sweep(-1)
    dx.free(); dy.free(); f.close()


def test_mls_get_impulses_deemphasised_many_channels():
    N, R, C = 12, 4, 6
    b, a = RBJ(filtertype="highshelf", gaindb=-10, f0=100, Q=0.707, fs=48000).get_coefficients()
    m = MLS(N=N, taps=TAPS[N][0], fs=48000, repeats=R, B=b, A=a, device=0)
    rng = np.random.default_rng(4)
    x = rng.standard_normal((R * m.L, C))
    got = m.get_impulses_deemphasised(x)
    assert got.shape == (m.L, C) and got.dtype == np.float64
    assert rel_err(got, _reference_chain(N, TAPS[N][0], a, b, x)) <= TOL


def test_mls_simple_short_orders_against_reference_goldens():
    """MLS_simple.get_impulse as the REAL reference computes it (tests/golden/simple_short.*, made by make_golden.py) at
    N = 5..10, where the emphasis filter's memory spans many periods: the fused path's summed period-boundary states."""
    import json
    import os
    G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    cases = json.load(open(os.path.join(G, "simple_short.json")))["cases"]
    arrays = np.load(os.path.join(G, "simple_short.npz"))
    for c in cases:
        ms = MLS_simple(N=c["N"], fs=c["fs"], repeats=c["repeats"], device=0)
        x = arrays[c["id"] + "_x"]
        got = ms.get_impulse(x)
        assert got.samples.shape == arrays[c["id"] + "_ir"].shape
        assert rel_err(got.samples, arrays[c["id"] + "_ir"]) <= TOL, c


@pytest.mark.parametrize("shape,layout", [((1 << 20, 1), "time"), ((100003, 16), "frames"), ((40000, 37), "frames"), ((1 << 17, 3), "stacked"), ((700, 5), "frames")])
def test_segments_of_the_time_axis_carry_the_filter_state(shape, layout):
    """round 3: the two sweeps run segment by segment so that the second one reads from the Infinity Cache; the state after a
    segment's last chunk is the next one's initial state (iir_carry_kernel). Every segment size must give lfilter's output
    (to 1e-12 x peak), the same bits every time, in place as well."""
    rng = np.random.default_rng(shape[0] % 9973)
    n, C = shape
    x = rng.standard_normal(shape)
    for order in (2, 5, 8):
        B, A = stable_filter(order, rng) if order != 2 else RBJ(filtertype="lowshelf", gaindb=12, f0=60, Q=0.707, fs=96000).get_coefficients()
        f = IirFilter(B, A, device=0)
        f.reserve(n, C)
        host = np.ascontiguousarray(x.T) if layout == "stacked" else x
        rs, cs = (1, n) if layout == "stacked" else (C, 1)
        dx = DeviceBuffer(x.nbytes).upload(host)
        dy = DeviceBuffer(x.nbytes)
        want = scipy.signal.lfilter(B, A, x, axis=0)

        def run(seg, src=None):
            f.set_segment(seg)
            dy.fill(0xFF)
            f.filter_device(dx.ptr, _lib.ZMLS_F64, n, C, rs, cs, dy.ptr, rs, cs)
            sync(0)
            out = dy.download(host.shape, np.float64)
            return out.T if layout == "stacked" else out

        whole = run(0)
        assert rel_err(whole, want) <= TOL
        for seg in (None, 1, 8192 * 8 * 3, n * C * 8 // 3, n * C * 8 - 8):      # built-in; smallest there is; three chunks of one channel; thirds; almost all
            got = run(seg)
            assert np.isfinite(got).all() and rel_err(got, want) <= TOL, (order, seg, rel_err(got, want))
            assert np.array_equal(run(seg), got), "same segment size, different bits"
        f.set_segment(n * C * 8 // 3)
        one = run(n * C * 8 // 3)
        f.filter_device(dx.ptr, _lib.ZMLS_F64, n, C, rs, cs, dx.ptr, rs, cs)          # in place
        sync(0)
        got = dx.download(host.shape, np.float64)
        assert np.array_equal(got.T if layout == "stacked" else got, one)
        dx.free(); dy.free(); f.close()
