"""The *_device entry points promise not to allocate, free or synchronise: a caller can put them
under HIP stream capture and replay the graph. Capture is started in global mode, where any such call on any thread
invalidates it, so a successful capture + replay with the right answer is the proof."""
import ctypes

import numpy as np
import pytest

from gpu_common import plan, rel_err
from zignal_amd import _lib
from zignal_amd.device import DeviceBuffer, sync
from zignal_amd.measure.mls import TAPS

pytestmark = pytest.mark.gpu


def hip():
    lib = ctypes.CDLL("libamdhip64.so")
    vp = ctypes.c_void_p
    lib.hipStreamBeginCapture.argtypes = [vp, ctypes.c_int]
    lib.hipStreamEndCapture.argtypes = [vp, ctypes.POINTER(vp)]
    lib.hipGraphInstantiate.argtypes = [ctypes.POINTER(vp), vp, vp, vp, ctypes.c_size_t]
    lib.hipGraphLaunch.argtypes = [vp, vp]
    lib.hipGraphExecDestroy.argtypes = [vp]
    lib.hipGraphDestroy.argtypes = [vp]
    lib.hipGraphGetNodes.argtypes = [vp, vp, ctypes.POINTER(ctypes.c_size_t)]
    return lib


@pytest.mark.parametrize("N,C,layout", [(16, 32, "frame"), (12, 5, "frame"), (17, 16, "stacked"), (9, 8, "frame")])
def test_impulse_device_can_be_captured_and_replayed(N, C, layout):
    h = hip()
    L = (1 << N) - 1
    R = 3
    rng = np.random.default_rng(N)
    x = rng.standard_normal((R * L, C))
    p = plan(N, TAPS[N][0])
    p.reserve(C, staging=True)
    if layout == "frame":
        host_in, rs, cs, ors, ocs = x, C, 1, C, 1
    else:
        host_in, rs, cs, ors, ocs = np.ascontiguousarray(x.T), 1, R * L, 1, L
    d_x = DeviceBuffer(x.nbytes).upload(host_in)
    d_direct, d_graph = DeviceBuffer(L * C * 8), DeviceBuffer(L * C * 8)
    d_direct.fill(0); d_graph.fill(0)
    p.impulse_device(d_x.ptr, _lib.ZMLS_F64, R * L, C, rs, cs, d_direct.ptr, ors, ocs)
    sync(0)

    stream = ctypes.c_void_p()
    _lib.check(_lib.lib().zmls_stream_create(0, ctypes.byref(stream)))
    graph, exe = ctypes.c_void_p(), ctypes.c_void_p()
    assert h.hipStreamBeginCapture(stream, 0) == 0                      # hipStreamCaptureModeGlobal
    try:
        p.impulse_device(d_x.ptr, _lib.ZMLS_F64, R * L, C, rs, cs, d_graph.ptr, ors, ocs, stream=stream)
    finally:
        rc = h.hipStreamEndCapture(stream, ctypes.byref(graph))
    assert rc == 0 and graph.value, "the capture was invalidated: something allocated or synchronised"
    n_nodes = ctypes.c_size_t()
    assert h.hipGraphGetNodes(graph, None, ctypes.byref(n_nodes)) == 0 and n_nodes.value >= 1
    out0 = d_graph.download((L, C) if layout == "frame" else (C, L), np.float64)
    assert not out0.any()                                               # capturing ran nothing
    assert h.hipGraphInstantiate(ctypes.byref(exe), graph, None, None, 0) == 0
    for _ in range(2):
        assert h.hipGraphLaunch(exe, stream) == 0
    _lib.check(_lib.lib().zmls_stream_sync(0, stream))
    shape = (L, C) if layout == "frame" else (C, L)
    assert np.array_equal(d_graph.download(shape, np.float64), d_direct.download(shape, np.float64))
    h.hipGraphExecDestroy(exe)
    h.hipGraphDestroy(graph)
    _lib.check(_lib.lib().zmls_stream_destroy(0, stream))
    for b in (d_x, d_direct, d_graph):
        b.free()


def _capture_and_replay(h, launch, stream):
    """launch(stream) under global-mode capture; the instantiated graph is replayed twice. Returns the node count."""
    graph, exe = ctypes.c_void_p(), ctypes.c_void_p()
    assert h.hipStreamBeginCapture(stream, 0) == 0
    try:
        launch(stream)
    finally:
        rc = h.hipStreamEndCapture(stream, ctypes.byref(graph))
    assert rc == 0 and graph.value, "the capture was invalidated: something allocated or synchronised"
    n_nodes = ctypes.c_size_t()
    assert h.hipGraphGetNodes(graph, None, ctypes.byref(n_nodes)) == 0 and n_nodes.value >= 1
    assert h.hipGraphInstantiate(ctypes.byref(exe), graph, None, None, 0) == 0
    for _ in range(2):
        assert h.hipGraphLaunch(exe, stream) == 0
    _lib.check(_lib.lib().zmls_stream_sync(0, stream))
    h.hipGraphExecDestroy(exe)
    h.hipGraphDestroy(graph)
    return n_nodes.value


@pytest.mark.parametrize("entry", ["static", "filtered", "sources", "sources_filtered"])
def test_the_other_launch_entry_points_can_be_captured_too(entry):
    """zmls_impulse_device_static (made for exactly this caller), zmls_impulse_filtered_device and
    zmls_impulse_sources_device: captured in global mode, replayed, same bytes as the direct call."""
    from zignal_amd.device import IirFilter, Plan, Sources
    h = hip()
    N, C, R, K = 14, 16, 5, 3
    L = (1 << N) - 1
    rng = np.random.default_rng(7)
    x = rng.standard_normal((R * L, C))
    d_x = DeviceBuffer(x.nbytes).upload(x)
    plans = [Plan(N, t, device=0) for t in TAPS[N][:K]]
    f = IirFilter([1.0, -0.4, 0.1], [1.0, -0.7, 0.2], device=0)
    n_out = K if entry.startswith("sources") else 1
    d_direct, d_graph = DeviceBuffer(n_out * L * C * 8), DeviceBuffer(n_out * L * C * 8)
    d_direct.fill(0); d_graph.fill(0)
    lib = _lib.lib()
    group = Sources(plans)
    if entry == "static":
        def launch(dst, stream):
            _lib.check(lib.zmls_impulse_device_static(plans[0]._h, d_x.ptr, _lib.ZMLS_F64, R * L, C, C, 1, dst, C, 1, stream))
    elif entry == "filtered":
        f.reserve_fused(plans[0], R * L, C)
        def launch(dst, stream):
            plans[0].impulse_filtered_device(f, d_x.ptr, _lib.ZMLS_F64, R * L, C, C, 1, dst, C, 1, stream=stream)
    else:
        iir = f if entry == "sources_filtered" else None
        group.reserve(R * L, C, iir=iir)
        def launch(dst, stream):
            group.impulse_device(d_x.ptr, _lib.ZMLS_F64, R * L, C, C, 1, dst, L * C, C, 1, iir=iir, stream=stream)
    launch(d_direct.ptr, None)
    sync(0)
    stream = ctypes.c_void_p()
    _lib.check(lib.zmls_stream_create(0, ctypes.byref(stream)))
    nodes = _capture_and_replay(h, lambda s: launch(d_graph.ptr, s), stream)
    assert nodes >= 2
    got, want = d_graph.download((n_out, L, C), np.float64), d_direct.download((n_out, L, C), np.float64)
    assert want.any() and np.array_equal(got, want)
    _lib.check(lib.zmls_stream_destroy(0, stream))
    for b in (d_x, d_direct, d_graph):
        b.free()
    f.close()
    for p in plans:
        p.close()
