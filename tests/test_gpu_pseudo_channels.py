"""The pseudo-channel layout of few-channel batches (csrc/fwht_kernels.inc: IN_GATHER_PC / OUT_SCATTER_PC, round 3).

The reference's own call is ONE channel (zignal/measure/mls.py:338-343, "limited to one channel of audio";
get_impulse, mls.py:315-328). A single channel fills a sixteenth of a 128-byte row of adjacent channels, so one channel from N = 18 on,
and up to four at N = 20..22 (where it pays, zmls_internal.h::pc_pays), lay the 2^N points of each channel out as 2^(N-4) rows of 16 (f32: 2^(N-5) rows of 32): the low bits of
the transform index ride where adjacent channels ride otherwise. It is another butterfly order of the same sums, so the
responses agree with the ordinary layout's to rounding (not bit for bit) and with the oracle within the north_star
tolerance. Every layout, sample type and period count the entry points take goes through it."""
import numpy as np
import pytest

from gpu_common import oracle_ir, rel_err
from oracle import c_oracle
from zignal_amd import _lib
from zignal_amd.device import DeviceBuffer, Plan, sync
from zignal_amd.measure.mls import TAPS

pytestmark = pytest.mark.gpu

NP = {_lib.ZMLS_F64: np.float64, _lib.ZMLS_F32: np.float32, _lib.ZMLS_I16: np.int16, _lib.ZMLS_I32: np.int32}


def capture(rng, n_rows, C, code):
    if code in (_lib.ZMLS_I16, _lib.ZMLS_I32):
        return rng.integers(-20000, 20000, size=(n_rows, C)).astype(NP[code])
    return rng.standard_normal((n_rows, C)).astype(NP[code])


def run(p, x, layout, staging):
    """x: (rows, C) host array; layout 'frames' (C-order), 'stacked' (one capture after the other, responses too),
    'padded' (rows padded to C + 3 elements)"""
    n_rows, C = x.shape
    L, es = p.L, np.dtype(p.np_dtype).itemsize
    p.reserve(C, staging=staging)
    code = _lib.dtype_code(x.dtype)
    if layout == "frames":
        host, rs, cs, ors, ocs = x, C, 1, C, 1
    elif layout == "stacked":
        host, rs, cs, ors, ocs = np.ascontiguousarray(x.T), 1, n_rows, 1, L
    else:
        host = np.zeros((n_rows, C + 3), dtype=x.dtype)
        host[:, :C] = x
        rs, cs, ors, ocs = C + 3, 1, C + 3, 1
    dx = DeviceBuffer(host.nbytes).upload(host)
    out_elems = L * (C + 3) if layout == "padded" else L * C
    dh = DeviceBuffer(out_elems * es)
    dh.fill(0xFF)
    p.impulse_device(dx.ptr, code, n_rows, C, rs, cs, dh.ptr, ors, ocs)
    sync(0)
    raw = dh.download((out_elems,), p.np_dtype)
    dx.free(); dh.free()
    if layout == "frames":
        return raw.reshape(L, C)
    if layout == "stacked":
        return raw.reshape(C, L).T
    return raw.reshape(L, C + 3)[:, :C]


@pytest.mark.parametrize("N", [6, 7, 8, 10, 11, 13, 14, 15, 16, 18, 20])
@pytest.mark.parametrize("dtype", ["f64", "f32"])
def test_few_channels_take_the_layout_and_agree_with_the_ordinary_one_and_the_oracle(N, dtype):
    if dtype == "f32" and N < 7:
        pytest.skip("f32 rows hold 32 points: orders from 7")
    taps = TAPS[N][0]
    p = Plan(N, taps, dtype=dtype, device=0)
    L = p.L
    tol, tol_pair = (1e-12, 1e-13) if dtype == "f64" else (1e-5, 2e-6)
    ref = c_oracle.sequence(N, taps)
    rng = np.random.default_rng(1000 + N)
    cases = [(1, 2, _lib.ZMLS_F64, "frames", True), (1, 5, _lib.ZMLS_F64, "frames", False), (2, 3, _lib.ZMLS_F32, "stacked", True),
             (3, 2, _lib.ZMLS_I16, "padded", True), (4, 4, _lib.ZMLS_I32, "frames", False), (4, 3, _lib.ZMLS_F64, "stacked", False)]
    if N >= 18:
        cases = cases[:3] + cases[4:5]
    for C, R, code, layout, staging in cases:
        if dtype == "f32" and code == _lib.ZMLS_F64:
            code = _lib.ZMLS_F32
        x = capture(rng, R * L, C, code)
        p.force_pseudo_channels(0)
        ordinary = run(p, x, layout, staging)
        p.force_pseudo_channels(1)
        got = run(p, x, layout, staging)
        assert np.all(np.isfinite(got)), (C, R, code, layout)
        assert rel_err(got.astype(np.float64), ordinary.astype(np.float64)) <= tol_pair, (C, R, code, layout, staging)
        p.force_pseudo_channels(-1)                  # the library's own choice (zmls_internal.h::pc_pays) is one of the two
        auto = run(p, x, layout, staging)
        pays = (C == 1 and N >= 18) or (N in (20, 21, 22))
        assert np.array_equal(auto, got if pays else ordinary), "the measured rule: one channel from N = 18, up to four at N = 20..22"
        if code == _lib.ZMLS_F32 and dtype == "f64":
            continue                                 # the reference transforms float32 captures in single precision
        for c in {0, C - 1}:
            want = oracle_ir(N, taps, x[:, c:c + 1].astype(np.float64), ref)
            assert rel_err(got[:, c], want) <= tol, (C, R, code, layout, c)
    p.close()


@pytest.mark.parametrize("N,C,dtype", [(9, 7, "f64"), (12, 20, "f64"), (14, 33, "f32"), (16, 9, "f64"), (17, 6, "f32")])
def test_forced_on_wider_batches(N, C, dtype):
    """whole transforms of several channels per workgroup (single pass) and more slabs than four"""
    taps = TAPS[N][0]
    p = Plan(N, taps, dtype=dtype, device=0)
    R = 3
    x = capture(np.random.default_rng(N * C), R * p.L, C, p.dtype_code)
    p.force_pseudo_channels(0)
    ordinary = run(p, x, "frames", False)
    p.force_pseudo_channels(1)
    got = run(p, x, "frames", False)
    assert rel_err(got.astype(np.float64), ordinary.astype(np.float64)) <= (1e-13 if dtype == "f64" else 2e-6)
    p.close()


def test_orders_too_small_for_the_layout_say_so():
    p = Plan(4, TAPS[4][0], device=0)
    with pytest.raises(_lib.ZmlsError):
        p.force_pseudo_channels(1)
    with pytest.raises(_lib.ZmlsError):
        p.force_pseudo_channels(2)
    x = capture(np.random.default_rng(4), 3 * p.L, 1, _lib.ZMLS_F64)
    got = run(p, x, "frames", False)                 # one channel at N = 4 still works, through the ordinary tiles
    assert rel_err(got[:, 0], oracle_ir(4, TAPS[4][0], x)) <= 1e-12
    p.close()


@pytest.mark.parametrize("N,R", [(22, 2), (24, 2), (25, 2)])
def test_one_channel_of_a_large_order_recovers_the_known_system(N, R):
    """N = 24 is the N = 20 two-pass schedule on one slab; N = 25 takes three passes (resident schedule included)"""
    taps = TAPS[N][0]
    p = Plan(N, taps, device=0)
    L = p.L
    p.reserve(1, staging=True)
    assert p.info()["workspace_bytes"] <= 1.01 * (1 << N) * 8           # one pseudo-channel slab: a sixteenth of a 128-byte-row one
    dx = DeviceBuffer(R * L * 8)
    dh = DeviceBuffer(L * 8)
    p.synth_device(dx.ptr, _lib.ZMLS_F64, R, 1, 1, 1, seed=99, sigma=1e-3)
    p.impulse_device(dx.ptr, _lib.ZMLS_F64, R * L, 1, 1, 1, dh.ptr, 1, 1)
    sync(0)
    got = dh.download((L,), np.float64)
    g, d = p.synth_system(0, 99)
    exp = np.full(L, -g.sum() / L)
    exp[(d + np.arange(16)) % L] += g * (1.0 + 1.0 / L)
    assert np.max(np.abs(got - exp)) <= 1e-4         # the capture carries noise of sigma 1e-3, averaged by the correlation
    p.force_pseudo_channels(0)
    p.reserve(1, staging=True)                       # the ordinary slabs are larger
    p.impulse_device(dx.ptr, _lib.ZMLS_F64, R * L, 1, 1, 1, dh.ptr, 1, 1)
    sync(0)
    assert rel_err(got, dh.download((L,), np.float64)) <= 1e-13
    dx.free(); dh.free(); p.close()


def test_static_workspace_entry_point_with_one_channel():
    N, R = 18, 3
    p = Plan(N, TAPS[N][0], device=0)
    L = p.L
    x = capture(np.random.default_rng(18), R * L, 1, _lib.ZMLS_F64)
    dx = DeviceBuffer(x.nbytes).upload(x)
    dh = DeviceBuffer(L * 8)
    p.reserve(1)
    p.impulse_device(dx.ptr, _lib.ZMLS_F64, R * L, 1, 1, 1, dh.ptr, 1, 1)      # one channel at N = 18: the layout is the library's choice
    sync(0)
    want = dh.download((L, 1), np.float64)
    dh.fill(0)
    _lib.check(_lib.lib().zmls_impulse_device_static(p._h, dx.ptr, _lib.ZMLS_F64, R * L, 1, 1, 1, dh.ptr, 1, 1, None))
    sync(0)
    assert np.array_equal(dh.download((L, 1), np.float64), want)
    assert rel_err(want[:, 0], oracle_ir(N, TAPS[N][0], x)) <= 1e-12
    dx.free(); dh.free(); p.close()
