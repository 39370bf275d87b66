#!/usr/bin/env python3
"""Per-step device time of the headline batch over many consecutive steps (HIP events around every step): shows
warm-up drift and jitter that a K-step average hides."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zignal_amd import _lib
from zignal_amd.device import DeviceBuffer, Plan, Timer, sync
from zignal_amd.measure.mlstaps import TAPS

N, C, R = 20, 512, 17
p = Plan(N, TAPS[N][0], device=0)
L = p.L
p.reserve(C)
d_x, probe = p.alloc_capture(R * L, C, _lib.ZMLS_F64)
d_h = DeviceBuffer(L * C * 8)
p.synth_device(d_x.ptr, _lib.ZMLS_F64, R, C, C, 1, seed=1, sigma=1e-3)
sync(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
timers = [Timer(0) for _ in range(n)]
for t in timers:
    t.start()
    p.impulse_device(d_x.ptr, _lib.ZMLS_F64, R * L, C, C, 1, d_h.ptr, C, 1)
    t.stop()
sync(0)
ms = np.array([t.elapsed_ms() for t in timers])
print("probe %.2f ms; steps: first 5 %s ... min %.3f median %.3f max %.3f; > median+2%%: %s" %
      (probe, np.round(ms[:5], 3), ms.min(), np.median(ms), ms.max(), np.nonzero(ms > np.median(ms) * 1.02)[0].tolist()))
print(np.round(ms, 2).tolist())
