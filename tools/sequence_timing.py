#!/usr/bin/env python3
"""Device time of the stimulus expansion (zmls_sequence_device: packed bits -> +1/-1 samples, `repeats` periods)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zignal_amd import _lib
from zignal_amd.device import DeviceBuffer, Plan, Timer, sync
from zignal_amd.measure.mlstaps import TAPS
for N, rep in ((16, 5), (20, 17), (24, 17), (24, 2)):
    p = Plan(N, TAPS[N][0], device=0); L = p.L
    d = DeviceBuffer(rep * L * 8)
    lib = _lib.lib()
    def run(): _lib.check(lib.zmls_sequence_device(p._h, d.ptr, _lib.ZMLS_F64, rep, 1, None))
    run(); sync(0)
    t = Timer(0); t.start()
    for _ in range(10): run()
    t.stop(); sync(0)
    ms = t.elapsed_ms() / 10
    print("sequence N=%d x %d repeats: %.4f ms, %.0f GB/s written" % (N, rep, ms, rep * L * 8 / ms / 1e6))
    d.free(); p.close()
