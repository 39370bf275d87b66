"""The cache-resident schedule (csrc/api.hip::run_passes, round 3): batches of one or two averaged periods run in chunks
whose intermediate stays in the 256 MiB Infinity Cache - channel slabs for two-pass orders, ranges of the top bits for
passes 1 and 2 of three-pass orders. It is another launch order of the same kernels on the same data, so the responses
must be BIT-IDENTICAL to the streaming schedule's for every chunk size, and equal to the oracle's within the
north_star tolerance. Reference: the arithmetic is trim_and_avgerage + xcorr_circular (zignal/measure/mls.py:169-313)."""
import numpy as np
import pytest

from gpu_common import oracle_ir, rel_err
from oracle import c_oracle
from zignal_amd import _lib
from zignal_amd.device import DeviceBuffer, Plan, sync
from zignal_amd.measure.mls import TAPS

pytestmark = pytest.mark.gpu


def responses(p, dx, dh, n_rows, C, dt=np.float64):
    dh.fill(0xFF)
    p.impulse_device(dx.ptr, p.dtype_code, n_rows, C, C, 1, dh.ptr, C, 1)
    sync(0)
    return dh.download((p.L, C), dt)


@pytest.mark.parametrize("N,C,R,dtype,tile", [(14, 64, 2, "f64", -1), (16, 48, 2, "f64", -1), (18, 64, 2, "f64", -1), (18, 40, 3, "f64", 0),
                                             (20, 32, 2, "f64", -1), (17, 128, 2, "f32", -1), (21, 32, 2, "f64", -1), (15, 24, 5, "f64", 0)])
def test_two_pass_orders_chunked_by_channel_slabs(N, C, R, dtype, tile):
    taps = TAPS[N][0]
    np_dt = np.float64 if dtype == "f64" else np.float32
    p = Plan(N, taps, dtype=dtype, device=0)
    if tile >= 0:
        p.force_tile(tile)
    L = p.L
    rng = np.random.default_rng(N * 100 + C)
    x = rng.standard_normal((R * L, C)).astype(np_dt)
    dx = DeviceBuffer(x.nbytes).upload(x)
    dh = DeviceBuffer(L * C * x.itemsize)
    p.reserve(C)
    p.set_resident(0)
    want = responses(p, dx, dh, R * L, C, np_dt)
    base = (1 << N) * 128                                             # a 128-byte-row slab without its padding rows
    for nbytes in (0, int(1.6 * base), int(3.3 * base), 1 << 12):     # built-in size; one wide slab (three narrow) per chunk; three; none fits: not resident
        p.set_resident(1, nbytes)
        assert np.array_equal(responses(p, dx, dh, R * L, C, np_dt), want), nbytes
    p.set_resident(-1)                                                # the library's own choice: resident for one or two periods
    assert np.array_equal(responses(p, dx, dh, R * L, C, np_dt), want)
    p.set_resident(1, 0, 2)                                           # tuning flag: resident stores, streaming loads
    assert np.array_equal(responses(p, dx, dh, R * L, C, np_dt), want)
    for nbytes in (0, int(3.3 * base), int(6.6 * base)):              # two lanes: chunks alternate between two streams
        p.set_resident(1, nbytes, 4)
        assert np.array_equal(responses(p, dx, dh, R * L, C, np_dt), want), ("two lanes", nbytes)
        assert np.array_equal(responses(p, dx, dh, R * L, C, np_dt), want), ("two lanes, again", nbytes)
    ref = c_oracle.sequence(N, taps)
    for c in (0, C - 1):
        assert rel_err(want[:, c], oracle_ir(N, taps, x[:, c:c + 1].astype(np.float64), ref)) <= (1e-12 if dtype == "f64" else 1e-5)
    with pytest.raises(_lib.ZmlsError):
        p.set_resident(2)
    dx.free(); dh.free(); p.close()


@pytest.mark.parametrize("N,C,dtype,tile", [(23, 32, "f64", -1), (23, 8, "f64", 0), (24, 16, "f32", -1), (25, 16, "f64", -1)])
def test_three_pass_orders_chunked_by_ranges_of_the_top_bits(N, C, dtype, tile):
    taps = TAPS[N][0]
    np_dt = np.float64 if dtype == "f64" else np.float32
    p = Plan(N, taps, dtype=dtype, device=0)
    assert p.info()["passes"] == 3
    if tile >= 0:
        p.force_tile(tile)
    L, R = p.L, 2
    dx = DeviceBuffer(R * L * C * np.dtype(np_dt).itemsize)
    dh = DeviceBuffer(L * C * np.dtype(np_dt).itemsize)
    p.reserve(C)
    p.synth_device(dx.ptr, p.dtype_code, R, C, C, 1, seed=77, sigma=1e-3)
    p.set_resident(0)
    want = responses(p, dx, dh, R * L, C, np_dt)
    # the built-in chunk size; a chunk of one top-bit value of one slab (the smallest there is); a few of them; all slabs at once
    for nbytes, flags in ((0, 0), (1 << 20, 0), (40 << 20, 0), (0, 1), (700 << 20, 0), (0, 4), (40 << 20, 4), (1 << 20, 5)):
        p.set_resident(1, nbytes, flags)
        assert np.array_equal(responses(p, dx, dh, R * L, C, np_dt), want), (nbytes, flags)
    p.set_resident(-1)
    assert np.array_equal(responses(p, dx, dh, R * L, C, np_dt), want)
    # the known synthetic system of two channels (size-independent check, tests/test_gpu_full_size.py)
    for c in (0, C - 1):
        g, d = p.synth_system(c, 77)
        exp = np.full(L, -g.sum() / L)
        exp[(d + np.arange(16)) % L] += g * (1.0 + 1.0 / L)
        assert np.max(np.abs(want[:, c] - exp)) <= 1e-4
    # timed variant: the chunked passes 1 and 2 are reported together
    p.set_resident(1)
    ms = p.impulse_device_timed(dx.ptr, p.dtype_code, R * L, C, C, 1, dh.ptr, C, 1)
    assert ms[0] > 0 and ms[1] == 0 and ms[2] > 0
    dx.free(); dh.free(); p.close()


def test_static_workspace_entry_point_is_resident_by_construction():
    """zmls_impulse_device_static runs in chunks of its 144 MiB array: the same kernels with the resident policy"""
    N, C, R = 18, 64, 2
    p = Plan(N, TAPS[N][0], device=0)
    L = p.L
    x = np.random.default_rng(5).standard_normal((R * L, C))
    dx = DeviceBuffer(x.nbytes).upload(x)
    dh = DeviceBuffer(L * C * 8)
    p.reserve(C)
    p.set_resident(0)
    want = responses(p, dx, dh, R * L, C)
    p.set_resident(-1)
    dh.fill(0)
    _lib.check(_lib.lib().zmls_impulse_device_static(p._h, dx.ptr, _lib.ZMLS_F64, R * L, C, C, 1, dh.ptr, C, 1, None))
    sync(0)
    assert np.array_equal(dh.download((L, C), np.float64), want)
    dx.free(); dh.free(); p.close()
