"""zmls_impulse_device_static (include/zmls.h): the impulse-response launch sequence whose intermediate lives in the
library's static device array instead of an allocated workspace, for callers that cannot allocate. Checked through
the C ABI against zmls_impulse_device on the same inputs: bit-identical (the same kernels run on the same tiles; only
the place of the intermediate and the chunking over channel slabs differ), and against the oracle to the fp64 bar.
Two calls in a row on one plan, the second with a smaller input, so that what the first call left in the array is
there when the second runs. Collected last (file name): a failure here must not hide the tests before it."""
import ctypes

import numpy as np
import pytest

from gpu_common import oracle_ir, rel_err
from zignal_amd import _lib
from zignal_amd.device import DeviceBuffer, Plan, sync
from zignal_amd.measure.mls import TAPS

pytestmark = pytest.mark.gpu
TOL64 = 1e-12          # BASELINE.json north_star: max-abs error relative to peak, fp64


def _static(p, d_x, code, n_rows, C, rs, cs, d_h, ors, ocs, stream=None):
    return _lib.lib().zmls_impulse_device_static(p._h, d_x, code, n_rows, C, rs, cs, d_h, ors, ocs, stream)


def _both(p, ref_plan, x, C):
    """responses of capture x (frames, C) from the static entry point on plan p and from zmls_impulse_device on ref_plan"""
    L = p.L
    d_x = DeviceBuffer(x.nbytes).upload(x)
    d_a, d_b = DeviceBuffer(L * C * 8), DeviceBuffer(L * C * 8)
    d_a.fill(0xFF)
    d_b.fill(0xFF)
    _lib.check(_static(p, d_x.ptr, _lib.ZMLS_F64, x.shape[0], C, C, 1, d_a.ptr, C, 1))
    ref_plan.impulse_device(d_x.ptr, _lib.ZMLS_F64, x.shape[0], C, C, 1, d_b.ptr, C, 1)
    sync(0)
    got, want = d_a.download((L, C), np.float64), d_b.download((L, C), np.float64)
    for b in (d_x, d_a, d_b):
        b.free()
    return got, want


@pytest.mark.parametrize("N,C1,R1,C2,R2", [(18, 96, 3, 32, 2),      # mid tile; 6 slabs of 33.6 MB: two chunks, then one
                                            (14, 24, 4, 5, 2),       # small tile, then an odd channel count (element path)
                                            (10, 256, 3, 40, 2)])    # single pass: the array is not touched at all
def test_static_entry_point_matches_impulse_device_twice_in_a_row(N, C1, R1, C2, R2):
    taps = TAPS[N][0]
    L = (1 << N) - 1
    p = Plan(N, taps, device=0)                 # never reserves anything: the static entry point must not need it
    ref_plan = Plan(N, taps, device=0)
    ref_plan.reserve(max(C1, C2))
    assert p.info()["workspace_bytes"] == 0
    rng = np.random.default_rng(100 + N)
    for C, R in ((C1, R1), (C2, R2)):
        x = rng.standard_normal((R * L, C))
        got, want = _both(p, ref_plan, x, C)
        assert np.isfinite(got).all()
        assert np.array_equal(got, want), "static workspace result differs from zmls_impulse_device (rel %.3e)" % rel_err(got, want)
        for c in (0, C - 1):
            assert rel_err(got[:, c], oracle_ir(N, taps, np.ascontiguousarray(x[:, c:c + 1]))) <= TOL64
    assert p.info()["workspace_bytes"] == 0     # and it did not quietly reserve one
    p.close()
    ref_plan.close()


@pytest.mark.parametrize("N,C,tile", [(21, 16, 0), (20, 32, 0), (21, 8, -1)])
def test_wide_tiles_give_way_to_the_narrow_one_when_their_slab_does_not_fit(N, C, tile):
    """N = 21: a 128-byte-row slab (mid / tall tile) is 268 MB against the 144 MiB array, so the batch runs on the small
    tile, one 64-byte-row slab per chunk, with vector access (api.hip::run_passes, the fallback for caller workspaces).
    Held bit for bit to zmls_impulse_device on the small tile, and to the oracle."""
    taps = TAPS[N][0]
    L = (1 << N) - 1
    p = Plan(N, taps, device=0)
    ref_plan = Plan(N, taps, device=0)
    ref_plan.force_tile(0)
    ref_plan.reserve(C)
    if tile >= 0:
        p.force_tile(tile)
    x = np.random.default_rng(N + C).standard_normal((2 * L, C))
    got, want = _both(p, ref_plan, x, C)
    assert np.array_equal(got, want), "rel %.3e" % rel_err(got, want)
    for c in (0, C - 1):
        assert rel_err(got[:, c], oracle_ir(N, taps, np.ascontiguousarray(x[:, c:c + 1]))) <= TOL64
    assert p.info()["workspace_bytes"] == 0
    p.close()
    ref_plan.close()


def test_one_channel_is_served_beyond_the_order_limit_through_the_pseudo_channel_layout():
    """the reference's own call shape: one channel of N = 22 or 24 takes slabs a sixteenth of the ordinary ones"""
    for N in (22, 24):
        taps = TAPS[N][0]
        L = (1 << N) - 1
        p = Plan(N, taps, device=0)
        ref_plan = Plan(N, taps, device=0)
        ref_plan.reserve(1)
        d_x, d_a, d_b = DeviceBuffer(2 * L * 8), DeviceBuffer(L * 8), DeviceBuffer(L * 8)
        p.synth_device(d_x.ptr, _lib.ZMLS_F64, 2, 1, 1, 1, seed=5, sigma=1e-3)
        _lib.check(_static(p, d_x.ptr, _lib.ZMLS_F64, 2 * L, 1, 1, 1, d_a.ptr, 1, 1))
        ref_plan.impulse_device(d_x.ptr, _lib.ZMLS_F64, 2 * L, 1, 1, 1, d_b.ptr, 1, 1)
        sync(0)
        got = d_a.download((L,), np.float64)
        assert np.array_equal(got, d_b.download((L,), np.float64))
        g, d = p.synth_system(0, 5)
        exp = np.full(L, -g.sum() / L)
        exp[(d + np.arange(16)) % L] += g * (1.0 + 1.0 / L)
        assert np.max(np.abs(got - exp)) <= 1e-4
        assert p.info()["workspace_bytes"] == 0
        for b in (d_x, d_a, d_b):
            b.free()
        p.close()
        ref_plan.close()


def test_static_entry_point_under_global_stream_capture():
    """Capture in global mode is invalidated by any allocation, free or synchronisation on any thread."""
    h = ctypes.CDLL("libamdhip64.so")
    vp = ctypes.c_void_p
    h.hipStreamBeginCapture.argtypes = [vp, ctypes.c_int]
    h.hipStreamEndCapture.argtypes = [vp, ctypes.POINTER(vp)]
    h.hipGraphInstantiate.argtypes = [ctypes.POINTER(vp), vp, vp, vp, ctypes.c_size_t]
    h.hipGraphLaunch.argtypes = [vp, vp]
    h.hipGraphExecDestroy.argtypes = [vp]
    h.hipGraphDestroy.argtypes = [vp]
    N, C, R = 16, 32, 3
    taps = TAPS[N][0]
    L = (1 << N) - 1
    p = Plan(N, taps, device=0)
    x = np.random.default_rng(7).standard_normal((R * L, C))
    d_x = DeviceBuffer(x.nbytes).upload(x)
    d_direct, d_graph = DeviceBuffer(L * C * 8), DeviceBuffer(L * C * 8)
    d_graph.fill(0)
    _lib.check(_static(p, d_x.ptr, _lib.ZMLS_F64, R * L, C, C, 1, d_direct.ptr, C, 1))
    sync(0)
    stream, graph, exe = vp(), vp(), vp()
    _lib.check(_lib.lib().zmls_stream_create(0, ctypes.byref(stream)))
    assert h.hipStreamBeginCapture(stream, 0) == 0
    try:
        rc_call = _static(p, d_x.ptr, _lib.ZMLS_F64, R * L, C, C, 1, d_graph.ptr, C, 1, stream)
    finally:
        rc = h.hipStreamEndCapture(stream, ctypes.byref(graph))
    assert rc_call == 0, _lib.last_error()
    assert rc == 0 and graph.value, "the capture was invalidated: something allocated or synchronised"
    assert h.hipGraphInstantiate(ctypes.byref(exe), graph, None, None, 0) == 0
    assert h.hipGraphLaunch(exe, stream) == 0
    _lib.check(_lib.lib().zmls_stream_sync(0, stream))
    assert np.array_equal(d_graph.download((L, C), np.float64), d_direct.download((L, C), np.float64))
    h.hipGraphExecDestroy(exe)
    h.hipGraphDestroy(graph)
    _lib.check(_lib.lib().zmls_stream_destroy(0, stream))
    for b in (d_x, d_direct, d_graph):
        b.free()
    p.close()


def test_static_entry_point_refuses_orders_beyond_its_array():
    N = 22                                       # one 64-byte-row slab of the intermediate is 268 MB > 144 MiB
    p = Plan(N, TAPS[N][0], device=0)
    p.force_pseudo_channels(0)                   # (two channels of N = 22 would otherwise take the pseudo-channel layout, which fits)
    L = p.L
    d_x, d_h = DeviceBuffer(2 * L * 2 * 8), DeviceBuffer(L * 2 * 8)
    d_x.fill(0)
    assert _static(p, d_x.ptr, _lib.ZMLS_F64, 2 * L, 2, 2, 1, d_h.ptr, 2, 1) == _lib.E_WORKSPACE
    msg = _lib.last_error()
    assert "order 22" in msg and "static" in msg and "21" in msg
    # argument errors are the same as zmls_impulse_device's
    assert _static(p, d_x.ptr, _lib.ZMLS_F64, L, 2, 2, 1, d_h.ptr, 2, 1) == _lib.E_ARG and "thrown away" in _lib.last_error()
    d_x.free()
    d_h.free()
    p.close()
