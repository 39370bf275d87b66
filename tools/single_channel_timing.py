#!/usr/bin/env python3
"""Device time of the reference's own shape - ONE channel, (R*L, 1) - at several orders and period counts (capture resident)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zignal_amd import _lib
from zignal_amd.device import DeviceBuffer, Plan, Timer, sync
from zignal_amd.measure.mlstaps import TAPS

for N, R in ((16, 2), (16, 5), (18, 5), (20, 2), (20, 17), (22, 5), (24, 2)):
    p = Plan(N, TAPS[N][0], device=0)
    L = p.L
    p.reserve(1, staging=True)
    x = DeviceBuffer(R * L * 8); x.fill(0)
    h = DeviceBuffer(L * 8)
    def run():
        p.impulse_device(x.ptr, _lib.ZMLS_F64, R * L, 1, 1, 1, h.ptr, 1, 1)
    run(); sync(0)
    t = Timer(0); t.start()
    for _ in range(20):
        run()
    t.stop(); sync(0)
    ms = t.elapsed_ms() / 20
    print("N=%d, 1 channel x %d periods: %8.4f ms  (%.0f GB/s of capture read)" % (N, R, ms, (R - 1) * L * 8 / ms / 1e6), flush=True)
    x.free(); h.free(); p.close()
