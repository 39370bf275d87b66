#!/usr/bin/env python3
"""Which buffer's placement moves the passes of a single-period batch (N=20, 1024 channels)? The capture is placed
once; then the plan (tables + workspace) is rebuilt several times with everything else fixed, then the response
buffer is reallocated several times with everything else fixed. Prints [pass 1, pass 2] milliseconds per variant."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zignal_amd import _lib
from zignal_amd.device import DeviceBuffer, Plan, sync
from zignal_amd.measure.mlstaps import TAPS

N, C, R = 20, 1024, 2
L = (1 << N) - 1


def timed(p, d_x, d_h, reps=5):
    p.impulse_device(d_x.ptr, _lib.ZMLS_F64, R * L, C, C, 1, d_h.ptr, C, 1)
    sync(0)
    ms = np.zeros(2)
    for _ in range(reps):
        ms += np.array(p.impulse_device_timed(d_x.ptr, _lib.ZMLS_F64, R * L, C, C, 1, d_h.ptr, C, 1))[:2]
    return np.round(ms / reps, 3).tolist()


p = Plan(N, TAPS[N][0], device=0)
p.reserve(C)
d_x, probe = p.alloc_capture(R * L, C, _lib.ZMLS_F64)
p.synth_device(d_x.ptr, _lib.ZMLS_F64, R, C, C, 1, seed=1, sigma=1e-3)
d_h = DeviceBuffer(L * C * 8)
print("capture probe %.3f ms" % probe)
print("same buffers, measured three times:", [timed(p, d_x, d_h) for _ in range(3)])
for i in range(6):
    p.close()
    p = Plan(N, TAPS[N][0], device=0)
    p.reserve(C)
    print("new plan (tables + workspace) %d:" % i, timed(p, d_x, d_h), flush=True)
for i in range(6):
    d_h.free()
    d_h = DeviceBuffer(L * C * 8)
    print("new response buffer %d:" % i, timed(p, d_x, d_h), flush=True)
held = []
for i in range(4):
    held.append(d_h)                       # keep the old one, so the next comes from other memory
    d_h = DeviceBuffer(L * C * 8)
    print("new response buffer, old ones kept %d:" % i, timed(p, d_x, d_h), flush=True)
