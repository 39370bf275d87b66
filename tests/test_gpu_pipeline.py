"""One launch for both passes of a two-pass order (tuning flag 8 of zmls_plan_set_resident; csrc/fwht_kernels.inc::
fwht_pipeline_kernel): the same tiles in another order, so the responses must be BIT-IDENTICAL to the streaming schedule's.
Reference for the arithmetic: trim_and_avgerage + xcorr_circular (zignal/measure/mls.py:169-313)."""
import ctypes

import numpy as np
import pytest

from zignal_amd import _lib
from zignal_amd.device import DeviceBuffer, Plan, sync
from zignal_amd.measure.mls import TAPS

pytestmark = pytest.mark.gpu


def responses(p, dx, dh, n_rows, C, dt):
    dh.fill(0xFF)
    p.impulse_device(dx.ptr, p.dtype_code, n_rows, C, C, 1, dh.ptr, C, 1)
    sync(0)
    return dh.download((p.L, C), dt)


def status(p):
    out = (ctypes.c_int32 * 2)()
    _lib.check(_lib.lib().zmls_plan_pipeline_status(p._h, out))
    return out[0], out[1]


@pytest.mark.parametrize("N,C,R,dtype", [(14, 64, 2, "f64"), (16, 48, 3, "f64"), (17, 32, 2, "f64"), (18, 64, 2, "f64"), (19, 32, 2, "f64"),
                                         (20, 48, 2, "f64"), (15, 64, 2, "f64"), (18, 64, 2, "f32"), (16, 1024, 2, "f64")])
def test_one_launch_schedule_is_bit_identical(N, C, R, dtype):
    taps = TAPS[N][0]
    np_dt = np.float64 if dtype == "f64" else np.float32
    p = Plan(N, taps, dtype=dtype, device=0)
    p.force_tile(2)                                             # the mid tile, whatever the batch size
    L = p.L
    rng = np.random.default_rng(N * 100 + C)
    x = rng.standard_normal((R * L, C)).astype(np_dt)
    dx = DeviceBuffer(x.nbytes).upload(x)
    dh = DeviceBuffer(L * C * x.itemsize)
    p.reserve(C)
    p.set_resident(0)
    want = responses(p, dx, dh, R * L, C, np_dt)
    launches_before = p.last_launches()
    slab_ch = 16 if dtype == "f64" else 32
    tiles = (1 << N) >> 10 if N >= 10 else 1
    for delay in (0, 1, 100000):                                # default; scatter tiles right behind their slab's gather tiles (they do wait); far behind
        _lib.check(_lib.lib().zmls_plan_set_pipeline_delay(p._h, delay))
        p.set_resident(1, 0, 8)
        for again in range(2):
            got = responses(p, dx, dh, R * L, C, np_dt)
            tickets, timed_out = status(p)
            assert timed_out == 0
            assert p.last_launches() == [1, 0, 0], (p.last_launches(), launches_before)
            assert np.array_equal(got, want), (delay, again)
    dx.free(); dh.free(); p.close()
