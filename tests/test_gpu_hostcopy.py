"""Chunked host <-> device copies (csrc/hostcopy.hip): captures handed over as numpy arrays - the reference's own call,
get_impulse(x), zignal/measure/mls.py:315-328 - cross PCIe in chunks through pinned bounce buffers on persistent threads.
It is plumbing: every byte must arrive, at every size around the chunk and engine thresholds, in both directions, and
the entry points built on it must return what they returned through plain hipMemcpy."""
import threading

import numpy as np
import pytest

from gpu_common import oracle_ir, rel_err
from zignal_amd.device import DeviceBuffer, Plan, sync
from zignal_amd.measure.mls import TAPS, _MLS_base

pytestmark = pytest.mark.gpu

MiB = 1 << 20


@pytest.mark.parametrize("nbytes", [1, 4095, 8 * MiB - 1, 8 * MiB, 8 * MiB + 1, 12 * MiB + 7, 4 * 4 * MiB, 37 * MiB + 3, 200 * MiB + 12345])
def test_round_trip_of_every_byte(nbytes):
    rng = np.random.default_rng(nbytes % 1000)
    a = rng.integers(0, 256, size=nbytes, dtype=np.uint8)
    d = DeviceBuffer(nbytes + 64)
    d.fill(0xEE)
    d.upload(a)
    back = d.download((nbytes + 64,), np.uint8)
    assert np.array_equal(back[:nbytes], a)
    assert np.all(back[nbytes:] == 0xEE), "the copy wrote past its end"
    d.free()


def test_copies_from_several_threads_do_not_mix():
    """the engine serves one copy at a time; callers on several host threads queue up"""
    out = {}

    def one(k):
        a = np.full(24 * MiB + k, k, dtype=np.uint8)
        d = DeviceBuffer(a.nbytes)
        d.upload(a)
        out[k] = bool(np.all(d.download((a.nbytes,), np.uint8) == k))
        d.free()

    ts = [threading.Thread(target=one, args=(k,)) for k in range(1, 5)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert out == {1: True, 2: True, 3: True, 4: True}


def test_host_entry_point_on_a_capture_large_enough_for_the_engine():
    N, R = 20, 5                                    # 4 periods x 8 MiB go up, 8 MiB comes back
    taps = TAPS[N][0]
    m = _MLS_base(N, taps, device=0)
    L = m.L
    rng = np.random.default_rng(20)
    x = np.roll(m.get_full_sequence(R), 12345, axis=0) * 0.25 + 1e-3 * rng.standard_normal((R * L, 1))
    h = m.get_impulse(x)
    p = Plan(N, taps, device=0)                     # the same through device buffers filled by small plain copies
    p.reserve(1, staging=True)
    dx = DeviceBuffer(x.nbytes)
    step = 4 * MiB // 8
    from zignal_amd import _lib
    for i in range(0, x.shape[0], step):
        part = np.ascontiguousarray(x[i:i + step])
        _lib.check(_lib.lib().zmls_dev_upload(0, dx.ptr + i * 8, part.ctypes.data, part.nbytes))
    dh = DeviceBuffer(L * 8)
    p.impulse_device(dx.ptr, _lib.ZMLS_F64, R * L, 1, 1, 1, dh.ptr, 1, 1)
    sync(0)
    assert np.array_equal(dh.download((L, 1), np.float64), h)
    assert abs(h[12345, 0] - 0.25) < 1e-3
    dx.free(); dh.free(); p.close()


def test_stacked_captures_large_enough_for_the_engine():
    N, R, C = 20, 3, 2                              # (channels, frames) view: one run of 16 MiB per channel
    taps = TAPS[N][0]
    p = Plan(N, taps, device=0)
    L = p.L
    x = np.random.default_rng(7).standard_normal((C, R * L))
    h = p.impulse_host(x.T)
    for c in range(C):
        assert rel_err(h[:, c], oracle_ir(N, taps, np.ascontiguousarray(x[c]).reshape(-1, 1))) <= 1e-12
    p.close()
