#!/usr/bin/env python3
"""Does the speed of the gather pass depend on which physical memory hipMalloc hands out? Allocates the headline capture
several times in one process and times the passes on each.
    alloc_probe.py N plain    plain allocations, freed in between
    alloc_probe.py N chosen   Plan.alloc_capture (placement chosen by probing, zmls_capture_alloc)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zignal_amd import _lib
from zignal_amd.device import DeviceBuffer, Plan, sync
from zignal_amd.measure.mlstaps import TAPS

N, C, R = 20, 512, 17
plan = Plan(N, TAPS[N][0])
L = plan.L
d_h = DeviceBuffer(L * C * 8)
plan.reserve(C)
mode = sys.argv[2] if len(sys.argv) > 2 else "plain"
for trial in range(int(sys.argv[1]) if len(sys.argv) > 1 else 5):
    t0 = time.perf_counter()
    if mode == "chosen":
        d_x, probe_ms = plan.alloc_capture(R * L, C)
    else:
        d_x, probe_ms = DeviceBuffer(R * L * C * 8), 0.0
    t_alloc = time.perf_counter() - t0
    pre = np.zeros(2)
    if mode == "chosen":          # same buffer, still holding the probe's zeros, after the rejected candidates were freed
        for rep in range(3):
            t = np.array(plan.impulse_device_timed(d_x.ptr, _lib.ZMLS_F64, R * L, C, C, 1, d_h.ptr, C, 1))
            if rep:
                pre += t / 2
    plan.synth_device(d_x.ptr, _lib.ZMLS_F64, R, C, C, 1)
    sync(0)
    ms = np.zeros(2)
    for rep in range(4):
        t = np.array(plan.impulse_device_timed(d_x.ptr, _lib.ZMLS_F64, R * L, C, C, 1, d_h.ptr, C, 1))
        if rep:
            ms += t
    ms /= 3
    print("trial %d %s ptr %#x alloc %.3f s probe %.3f ms zeros-after-cleanup %.3f pass_ms %s" % (trial, mode, d_x.ptr, t_alloc, probe_ms, pre[0], np.round(ms, 3)), flush=True)
    d_x.free()
