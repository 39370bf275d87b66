#!/usr/bin/env python3
"""Device time of the reference's own shape - ONE channel, (R*L, 1) - at several orders and period counts (capture resident),
through the ordinary channel slabs (a sixteenth of every row in use) and through the pseudo-channel layout.
usage: single_channel_timing.py [channels]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zignal_amd import _lib
from zignal_amd.device import DeviceBuffer, Plan, Timer, sync
from zignal_amd.measure.mlstaps import TAPS

C = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for N, R in ((8, 2), (10, 2), (11, 2), (12, 2), (13, 2), (14, 2), (15, 2), (16, 2), (16, 5), (17, 2), (18, 2), (18, 5), (19, 2), (20, 2), (20, 17),
             (21, 2), (22, 5), (23, 2), (24, 2), (26, 2)):
    p = Plan(N, TAPS[N][0], device=0)
    L = p.L
    p.reserve(C, staging=True)
    x = DeviceBuffer(R * L * 8 * C); x.fill(0)
    h = DeviceBuffer(L * 8 * C)
    def run():
        p.impulse_device(x.ptr, _lib.ZMLS_F64, R * L, C, C, 1, h.ptr, C, 1)
    out = []
    for mode in (0, 1):
        p.force_pseudo_channels(mode)
        run(); sync(0)
        t = Timer(0)
        best = 1e9
        for _ in range(3):
            t.start()
            for _ in range(20):
                run()
            t.stop(); sync(0)
            best = min(best, t.elapsed_ms() / 20)
        out.append(best)
    print("N=%d, %d channel(s) x %d periods: ordinary %8.4f ms, pseudo-channel %8.4f ms  (%.0f GB/s of capture read)"
          % (N, C, R, out[0], out[1], C * (R - 1) * L * 8 / out[1] / 1e6), flush=True)
    x.free(); h.free(); p.close()
