#!/usr/bin/env python3
"""Only the magnitude response (zmls_spectrum_device), N=20 x 64 channels and N=16 x 512, a few times each: run under
rocprofv3 --kernel-trace --stats to see how the time splits between the transform passes (fft.hip; hipFFT's Bluestein until
round 3) and the kernels around them (peak, window + pack, dB + unpack); tools/kernel_times.py lists the result.
usage: spectrum_profile.py [N:C ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zignal_amd.device import DeviceBuffer, Spectrum, sync

shapes = [tuple(int(v) for v in a.split(':')) for a in sys.argv[1:]] or [(20, 64), (16, 512)]
for N, C in shapes:
    L = 2**N - 1
    sp = Spectrum(L, C, device=0)
    h = DeviceBuffer(L * C * 8); h.fill(0)
    m = DeviceBuffer(sp.n_bins * C * 8)
    for _ in range(5):
        sp.magnitude_db_device(h.ptr, C, 1, None, False, m.ptr)
    sync(0)
    sp.close(); h.free(); m.free()
print("done")
