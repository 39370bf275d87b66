import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from zignal_amd import _lib
from zignal_amd.device import DeviceBuffer, Plan, download_channel, sync
from zignal_amd.measure.mls import TAPS
for N in (28, 30):
    t0 = time.perf_counter()
    p = Plan(N, TAPS[N][0], device=0)
    L, C, R = p.L, 2, 2
    d_x = DeviceBuffer(R * L * C * 8); d_h = DeviceBuffer(L * C * 8)
    p.reserve(C)
    p.synth_device(d_x.ptr, _lib.ZMLS_F64, R, C, C, 1, seed=5, sigma=0.0)
    p.impulse_device(d_x.ptr, _lib.ZMLS_F64, R * L, C, C, 1, d_h.ptr, C, 1)
    sync(0)
    worst = 0.0
    for c in range(C):
        h = download_channel(0, d_h.ptr, _lib.ZMLS_F64, L, C, 1, c)
        g, d = p.synth_system(c, seed=5)
        idx = (d + np.arange(16)) % L
        want = g * (1.0 + 1.0 / L) - g.sum() / L
        worst = max(worst, float(np.max(np.abs(h[idx] - want))))
        h[idx] = -g.sum() / L
        worst = max(worst, float(np.max(np.abs(h + g.sum() / L))))
    print("N=%d passes %s workspace %.1f GB: max abs err vs known system %.2e, %.1f s" % (N, p.info()["pass_bits"], p.info()["workspace_bytes"] / 1e9, worst, time.perf_counter() - t0), flush=True)
    d_x.free(); d_h.free(); p.close()
