#!/usr/bin/env python3
"""BASELINE config 5: float32 plan against float64 plan at N=20, 1024 channels, one averaged period - device time of
one step and the error of the float32 responses against the float64 ones (relative to the peak), same synthetic
capture rounded to float32."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zignal_amd import _lib
from zignal_amd.device import DeviceBuffer, Plan, Timer, download_channel, sync
from zignal_amd.measure.mlstaps import TAPS

N, C, R = 20, 1024, 2
res = {}
for name, code, es in (("f64", _lib.ZMLS_F64, 8), ("f32", _lib.ZMLS_F32, 4)):
    p = Plan(N, TAPS[N][0], dtype=name, device=0)
    L = p.L
    p.reserve(C)
    d_x, _ = p.alloc_capture(R * L, C, code)
    d_h = DeviceBuffer(L * C * es)
    p.synth_device(d_x.ptr, code, R, C, C, 1, seed=0x5A17, sigma=1e-3)
    p.impulse_device(d_x.ptr, code, R * L, C, C, 1, d_h.ptr, C, 1)
    sync(0)
    t = Timer(0); t.start()
    for _ in range(20):
        p.impulse_device(d_x.ptr, code, R * L, C, C, 1, d_h.ptr, C, 1)
    t.stop()
    ms = t.elapsed_ms() / 20
    res[name] = (ms, {c: download_channel(0, d_h.ptr, code, L, C, 1, c).astype(np.float64) for c in (0, 333, 777, 1023)})
    print("%s: %.3f ms per step, %.1f Gsamples/s, %.0f GB/s algorithmic" % (name, ms, C * L / ms / 1e6, 2 * C * L * es / ms / 1e6), flush=True)
    d_x.free(); d_h.free(); p.close()
errs = [float(np.max(np.abs(res["f32"][1][c] - res["f64"][1][c])) / np.max(np.abs(res["f64"][1][c]))) for c in res["f64"][1]]
print("float32 against float64 responses: max |error| / peak = %s; speed-up %.2fx" % (["%.2e" % e for e in errs], res["f64"][0] / res["f32"][0]))
