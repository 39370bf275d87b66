#!/usr/bin/env python3
"""Device time of the magnitude response alone (zmls_spectrum_device) for MLS-sized responses: ms per call, and the bytes the
transform passes move per call. tools/secondary_bench.py prints the same three shapes among everything else."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zignal_amd.device import DeviceBuffer, Spectrum, Timer, sync


def timed(fn, reps=10):
    fn(); sync(0)
    t = Timer(0)
    t.start()
    for _ in range(reps):
        fn()
    t.stop()
    sync(0)
    return t.elapsed_ms() / reps


for N, C in ((20, 1), (20, 2), (20, 64), (16, 512), (18, 64), (19, 64), (22, 8), (14, 1024)):
    L = 2**N - 1
    sp = Spectrum(L, C, device=0)
    h = DeviceBuffer(L * C * 8); h.fill(0)
    m = DeviceBuffer(sp.n_bins * C * 8)
    ms = timed(lambda: sp.magnitude_db_device(h.ptr, C, 1, None, False, m.ptr))
    print("spectrum N=%d x %d channels (fftsize %d)  %8.3f ms  %7.0f GB/s of response in + spectrum out" % (N, C, sp.fftsize, ms, 1.5 * L * C * 8 / ms / 1e6), flush=True)
    sp.close(); h.free(); m.free()
