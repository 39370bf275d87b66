#!/usr/bin/env python3
"""Device time of the impulse entry point on captures of other sample types (raw PCM, float32 into a float64 plan), frame-major and
stacked: N=20 x 512 channels x 17 periods. The periods are added up by the layout adapters, then the passes run on one period of sums."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zignal_amd import _lib
from zignal_amd.device import DeviceBuffer, Plan, Timer, sync
from zignal_amd.measure.mlstaps import TAPS

N, C, R = 20, 512, 17
p = Plan(N, TAPS[N][0], device=0)
L = p.L
p.reserve(C, staging=True)
h = DeviceBuffer(L * C * 8)
for code, es, name in ((_lib.ZMLS_F64, 8, "float64"), (_lib.ZMLS_F32, 4, "float32"), (_lib.ZMLS_I32, 4, "int32"), (_lib.ZMLS_I16, 2, "int16")):
    x = DeviceBuffer(R * L * C * es)
    x.fill(0)
    for layout, (rs, cs) in (("frames x channels", (C, 1)), ("stacked", (1, R * L))):
        def run():
            p.impulse_device(x.ptr, code, R * L, C, rs, cs, h.ptr, C, 1)
        run(); sync(0)
        t = Timer(0); t.start()
        for _ in range(5):
            run()
        t.stop(); sync(0)
        ms = t.elapsed_ms() / 5
        print("%-8s %-18s %8.3f ms  %6.0f GB/s of capture read (%.1f GB)" % (name, layout, ms, (R - 1) * L * C * es / ms / 1e6, (R - 1) * L * C * es / 1e9), flush=True)
    x.free()
