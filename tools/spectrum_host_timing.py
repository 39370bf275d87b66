#!/usr/bin/env python3
"""The drop-in call Audio.fft (zignal/audio.py:678-704) numpy to numpy: median of 20 calls on one-channel responses of
L = 2^N - 1 points (what MLS_simple.plot_fft asks for) and on a 64-channel one. ZMLS_LIBRARY=<other build> for an A/B."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zignal_amd.audio import Audio

for N, C in ((14, 1), (16, 1), (18, 1), (20, 1), (16, 64), (20, 8)):
    L = 2**N - 1
    a = Audio(fs=96000, initialdata=np.random.default_rng(N).standard_normal((L, C)))
    a.fft(window='rectangular', normalise=False, device=0)
    ts = []
    for _ in range(20):
        t0 = time.perf_counter(); a.fft(window='rectangular', normalise=False, device=0); ts.append(time.perf_counter() - t0)
    print("Audio.fft N=%d x %d channels  median %.3f ms  min %.3f ms" % (N, C, 1e3 * np.median(ts), 1e3 * min(ts)), flush=True)
