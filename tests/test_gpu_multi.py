"""GPU parity for the period-sharded multi-GPU mode, on the one GPU a test box has:
the three device steps (period sums in the exchange layout, RCCL reduce-scatter, transform of summed
periods) one by one with two emulated ranks, and the fused, chunked entry point on a one-rank RCCL
communicator against zmls_impulse_device and the oracle. Tolerance: 1e-12 x peak (f64), 1e-5 (f32)."""
import numpy as np
import pytest

from gpu_common import oracle_ir, plan, rel_err
from oracle import c_oracle
from zignal_amd import _lib, multi
from zignal_amd.device import DeviceBuffer, sync
from zignal_amd.measure.mls import TAPS

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def comm1():
    c = multi.Comm(multi.make_unique_id(), 0, 1, device=0)
    yield c
    c.close()


def _oracle_all(N, taps, x):
    ref = c_oracle.sequence(N, taps)
    return np.stack([oracle_ir(N, taps, np.ascontiguousarray(x[:, c:c + 1], dtype=np.float64), ref) for c in range(x.shape[1])], axis=1)


@pytest.mark.parametrize("N,C,R,dtype", [(9, 8, 5, "f64"), (12, 32, 6, "f64"), (13, 64, 4, "f32"), (12, 6, 3, "f64")])
def test_two_emulated_ranks_step_by_step(N, C, R, dtype):
    """Rank r holds periods r, r+2, ...; each adds its periods up on the GPU into the send layout; the host
    adds the two send buffers (what ncclReduceScatter(sum) does) and gives block q back to 'rank' q, which
    transforms it on the GPU."""
    taps = TAPS[N][0]
    p = plan(N, taps, dtype)
    L, P = p.L, 2
    npdt = p.np_dtype
    rng = np.random.default_rng(N * 100 + C)
    x = rng.standard_normal((R * L, C)).astype(npdt)
    share = C // P
    sends = []
    for r in range(P):
        periods, skip, total = multi.period_shard(R, P, r)
        local = np.ascontiguousarray(np.concatenate([x[q * L:(q + 1) * L] for q in periods], axis=0))
        d_x = DeviceBuffer(local.nbytes).upload(local)
        d_s = DeviceBuffer(P * L * share * local.itemsize)
        d_s.fill(0xFF)
        multi.period_sum_device(p, d_x.ptr, p.dtype_code, local.shape[0], skip, C, 1, P, share, 0, share, d_s.ptr)
        got = d_s.download((P, L, share), npdt)
        want = np.zeros((L, C), dtype=npdt)
        for i in range(skip, len(periods)):
            want += local[i * L:(i + 1) * L]
        assert np.array_equal(got, multi.send_layout(want, P)), "partial period sums of rank %d differ" % r
        sends.append(got)
    summed = sends[0] + sends[1]
    want_h = _oracle_all(N, taps, x)
    tol = 1e-12 if dtype == "f64" else 1e-5
    for q in range(P):
        first, n = multi.exchange_block(C, P, q)
        d_blk = DeviceBuffer(summed[q].nbytes).upload(summed[q])
        d_h = DeviceBuffer(L * n * summed.itemsize)
        p.reserve(n)
        multi.impulse_sums_device(p, d_blk.ptr, R - 1, n, n, 1, d_h.ptr, n, 1)
        h = d_h.download((L, n), npdt)
        assert rel_err(h, want_h[:, first:first + n]) <= tol


@pytest.mark.parametrize("P,chunk", [(2, 3), (4, 2), (8, 1)])
def test_period_sums_in_the_exchange_layout_chunk_by_chunk(P, chunk):
    """What the fused entry point asks of zmls_period_sum_device on P > 1 ranks: P channel blocks per call, a chunk of
    each block's channels at a time (n_blocks = P, block stride = share, ch_first = k * chunk): block q of the send
    buffer must hold the sums of channels q*share + k*chunk ... of this rank's periods."""
    N, R = 10, 5
    taps = TAPS[N][0]
    p = plan(N, taps)
    L = p.L
    share = 8
    C = P * share
    rng = np.random.default_rng(P)
    x = rng.standard_normal((R * L, C))
    d_x = DeviceBuffer(x.nbytes).upload(x)
    want = x[L:].reshape(R - 1, L, C).sum(axis=0)
    for k in range((share + chunk - 1) // chunk):
        cc = min(chunk, share - k * chunk)
        d_s = DeviceBuffer(P * L * cc * 8)
        d_s.fill(0xFF)
        multi.period_sum_device(p, d_x.ptr, _lib.ZMLS_F64, R * L, 1, C, 1, P, share, k * chunk, cc, d_s.ptr)
        got = d_s.download((P, L, cc), np.float64)
        for q in range(P):
            assert np.allclose(got[q], want[:, q * share + k * chunk:q * share + k * chunk + cc], rtol=0, atol=1e-12), (k, q)
        d_s.free()
    d_x.free()


@pytest.mark.parametrize("np_in,plan_dtype", [(np.int16, "f64"), (np.int32, "f64"), (np.float32, "f64"), (np.int16, "f32"), (np.float64, "f32")])
@pytest.mark.parametrize("P,share,chunk", [(1, 16, 16), (2, 64, 32), (4, 1024, 512), (4, 24, 8), (2, 512, 512)])
def test_period_sums_of_other_sample_types_in_the_exchange_layout(np_in, plan_dtype, P, share, chunk):
    """the converting vector kernel (period_sum_cvt_kernel) in the layouts the exchange asks for: one block, blocks whose rows are
    whole waves of lanes (stores transposed through LDS) and blocks whose rows are not (strided stores); exact integer sums for
    PCM, sums in the plan's precision otherwise"""
    N, R = 7, 4
    p = plan(N, TAPS[N][0], plan_dtype)
    L, C = p.L, P * share
    rng = np.random.default_rng(P * share + chunk)
    if np.issubdtype(np_in, np.integer):
        x = rng.integers(-3000, 3000, size=(R * L, C)).astype(np_in)
    else:
        x = rng.standard_normal((R * L, C)).astype(np_in)
    d_x = DeviceBuffer(x.nbytes).upload(x)
    acc = np.zeros((L, C), dtype=p.np_dtype)
    for r in range(1, R):
        acc += x[r * L:(r + 1) * L].astype(p.np_dtype)              # period order, in the plan's precision
    for k in range(share // chunk):
        d_s = DeviceBuffer(P * L * chunk * acc.itemsize)
        d_s.fill(0xFF)
        multi.period_sum_device(p, d_x.ptr, _lib.dtype_code(np_in), R * L, 1, C, 1, P, share, k * chunk, chunk, d_s.ptr)
        got = d_s.download((P, L, chunk), p.np_dtype)
        for q in range(P):
            assert np.array_equal(got[q], acc[:, q * share + k * chunk:q * share + (k + 1) * chunk]), (k, q)
        d_s.free()
    d_x.free()


@pytest.mark.parametrize("P,N,C,R,chunk,dtype", [(2, 12, 48, 5, 8, "f64"), (4, 14, 64, 9, 16, "f64"), (8, 10, 64, 3, 3, "f64"), (3, 13, 48, 7, 16, "f32"),
                                                  (4, 16, 128, 4, 8, "f64")])
def test_fused_entry_point_on_several_ranks_through_the_loopback_transport(P, N, C, R, chunk, dtype):
    """zmls_impulse_period_sharded_device with P > 1: every rank is a thread with its own plan, communicator and stream on
    the one GPU, the exchange is the library's loopback transport (same stream semantics as the RCCL call it stands in
    for). Rank r holds periods r, r+P, ... (rank 0 the discarded one; with R < P some ranks hold none) and must end up
    with channel block r of what zmls_impulse_device makes of the whole capture; also held to the oracle."""
    import ctypes
    import threading
    from zignal_amd.device import Plan
    taps = TAPS[N][0]
    ref_plan = plan(N, taps, dtype)
    L, npdt = ref_plan.L, ref_plan.np_dtype
    rng = np.random.default_rng(P * 100 + N)
    x = rng.standard_normal((R * L, C)).astype(npdt)
    d_all, d_ref = DeviceBuffer(x.nbytes).upload(x), DeviceBuffer(L * C * x.itemsize)
    ref_plan.reserve(C)
    ref_plan.impulse_device(d_all.ptr, ref_plan.dtype_code, R * L, C, C, 1, d_ref.ptr, C, 1)
    sync(0)
    want = d_ref.download((L, C), npdt).astype(np.float64)
    hub = multi.LoopbackHub(P)
    share = C // P
    results, errors = [None] * P, []

    def rank_main(r):
        try:
            p = Plan(N, taps, dtype=dtype, device=0)
            comm = multi.Comm.loopback(hub, r, 0)
            stream = ctypes.c_void_p()
            _lib.check(_lib.lib().zmls_stream_create(0, ctypes.byref(stream)))
            mine, skip, total = multi.period_shard(R, P, r)
            local = np.ascontiguousarray(np.concatenate([x[q * L:(q + 1) * L] for q in mine], axis=0)) if mine else np.zeros((1, C), npdt)
            d_x = DeviceBuffer(local.nbytes).upload(local)
            d_h = DeviceBuffer(L * share * x.itemsize)
            d_h.fill(0xFF)
            ps = multi.PeriodSharded(p, comm, chunk)
            for _ in range(2):                        # twice in a row on the same staging buffers
                ps.impulse_device(d_x.ptr, p.dtype_code, len(mine) * L, skip, total, C, C, 1, d_h.ptr, share, 1, stream=stream)
            _lib.check(_lib.lib().zmls_stream_sync(0, stream))
            results[r] = d_h.download((L, share), npdt).astype(np.float64)
            _lib.check(_lib.lib().zmls_stream_destroy(0, stream))
            comm.close()
            p.close()
        except Exception as e:  # noqa: BLE001
            errors.append((r, repr(e)))

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(P)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(120)
    assert not any(t.is_alive() for t in threads), "a rank is stuck"
    assert not errors, errors
    hub.close()
    got = np.concatenate(results, axis=1)
    peak = np.max(np.abs(want))
    # same samples added up in another order (partial sums per rank, then across ranks): equal to rounding
    assert np.max(np.abs(got - want)) <= (1e-14 if dtype == "f64" else 2e-6) * peak
    ora = _oracle_all(N, taps, x[:, ::max(1, C // 4)])
    assert rel_err(got[:, ::max(1, C // 4)], ora) <= (1e-12 if dtype == "f64" else 1e-5)
    d_all.free()
    d_ref.free()


def test_two_ranks_on_two_gpus_over_rccl():
    """The fused period-sharded entry point with a real two-rank communicator: needs two GPUs (the driver's multi-GPU
    node; a one-GPU test box skips it - RCCL refuses two ranks on one device). Each rank is a process of its own
    (tests/rccl_rank_worker.py), the id travels through a file, the result is held to zmls_impulse_device."""
    import os
    import subprocess
    import sys
    import tempfile
    if _lib.device_count() < 2:
        pytest.skip("one GPU: RCCL needs a device per rank (parity of the P > 1 exchange is unpinned on this box)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as tmp:
        procs = [subprocess.Popen([sys.executable, os.path.join(root, "tests", "rccl_rank_worker.py"), str(r), "2", os.path.join(tmp, "id.bin")],
                                  stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
        outs = []
        for pr in procs:
            try:
                outs.append(pr.communicate(timeout=240)[0])
            except subprocess.TimeoutExpired:
                for q in procs:
                    q.kill()
                raise
    for r, (pr, out) in enumerate(zip(procs, outs)):
        assert pr.returncode == 0 and "rank %d ok" % r in out, out[-3000:]


def test_rccl_collectives_on_a_one_rank_communicator(comm1):
    a = np.arange(1000, dtype=np.float64)
    d_a, d_b = DeviceBuffer(a.nbytes).upload(a), DeviceBuffer(a.nbytes)
    d_b.fill(0)
    comm1.reduce_scatter_device(d_a.ptr, d_b.ptr, a.size, _lib.ZMLS_F64)
    sync(0)
    assert np.array_equal(d_b.download(a.shape, np.float64), a)
    d_b.fill(0)
    comm1.all_gather_device(d_a.ptr, d_b.ptr, a.size, _lib.ZMLS_F64)
    sync(0)
    assert np.array_equal(d_b.download(a.shape, np.float64), a)
    assert comm1.info()["world"] == 1


@pytest.mark.parametrize("N,C,R,chunk,dtype", [(12, 48, 5, 16, "f64"), (14, 32, 3, 32, "f64"), (9, 12, 4, 5, "f64"),
                                              (13, 64, 9, 16, "f32")])
def test_fused_entry_point_one_rank_against_impulse_device_and_oracle(comm1, N, C, R, chunk, dtype):
    taps = TAPS[N][0]
    p = plan(N, taps, dtype)
    L = p.L
    npdt = p.np_dtype
    rng = np.random.default_rng(N + C + R)
    x = rng.standard_normal((R * L, C)).astype(npdt)
    d_x = DeviceBuffer(x.nbytes).upload(x)
    d_h1, d_h2 = DeviceBuffer(L * C * x.itemsize), DeviceBuffer(L * C * x.itemsize)
    d_h1.fill(0xFF)
    d_h2.fill(0xFF)
    p.reserve(C)
    p.impulse_device(d_x.ptr, p.dtype_code, R * L, C, C, 1, d_h1.ptr, C, 1)
    ps = multi.PeriodSharded(p, comm1, chunk)
    ps.impulse_device(d_x.ptr, p.dtype_code, R * L, 1, R - 1, C, C, 1, d_h2.ptr, C, 1)
    sync(0)
    h1, h2 = d_h1.download((L, C), npdt), d_h2.download((L, C), npdt)
    tol = 1e-12 if dtype == "f64" else 1e-5
    assert rel_err(h2, h1.astype(np.float64)) <= (4e-16 if dtype == "f64" else 4e-7)     # same sums, same transform
    assert rel_err(h2, _oracle_all(N, taps, x)) <= tol
    # twice in a row on the same staging buffers, second time with fewer channels
    C2 = chunk
    d_h2.fill(0xFF)
    ps.impulse_device(d_x.ptr, p.dtype_code, R * L, 1, R - 1, C2, C, 1, d_h2.ptr, C2, 1)
    sync(0)
    assert rel_err(d_h2.download((L, C2), npdt), h1[:, :C2].astype(np.float64)) <= (4e-16 if dtype == "f64" else 4e-7)


def test_fused_entry_point_integer_capture_major_input(comm1):
    N, C, R = 12, 8, 4
    taps = TAPS[N][0]
    p = plan(N, taps)
    L = p.L
    rng = np.random.default_rng(5)
    x = rng.integers(-2000, 2000, size=(C, R * L)).astype(np.int16)            # (C, R*L): a stack of single-channel captures
    d_x = DeviceBuffer(x.nbytes).upload(x)
    d_h = DeviceBuffer(L * C * 8)
    ps = multi.PeriodSharded(p, comm1, 8)
    ps.impulse_device(d_x.ptr, _lib.ZMLS_I16, R * L, 1, R - 1, C, 1, R * L, d_h.ptr, C, 1)
    sync(0)
    h = d_h.download((L, C), np.float64)
    assert rel_err(h, _oracle_all(N, taps, x.T.astype(np.float64))) <= 1e-12


def test_fused_entry_point_argument_errors(comm1):
    N = 10
    p = plan(N, TAPS[N][0])
    L = p.L
    d = DeviceBuffer(2 * L * 8 * 8)
    ps = multi.PeriodSharded(p, comm1, 8)
    with pytest.raises(_lib.ZmlsError):     # frames not a multiple of L
        ps.impulse_device(d.ptr, _lib.ZMLS_F64, 2 * L - 1, 1, 1, 8, 8, 1, d.ptr, 8, 1)
    with pytest.raises(_lib.ZmlsError):     # more local periods than the stated total
        ps.impulse_device(d.ptr, _lib.ZMLS_F64, 2 * L, 0, 1, 8, 8, 1, d.ptr, 8, 1)
    p2 = plan(11, TAPS[11][0])
    with pytest.raises(_lib.ZmlsError):     # staging was reserved for another plan
        multi.PeriodSharded.impulse_device(type("X", (), {"plan": p2, "comm": comm1})(), d.ptr, _lib.ZMLS_F64, 2 * p2.L, 1, 1, 8, 8, 1, d.ptr, 8, 1)
