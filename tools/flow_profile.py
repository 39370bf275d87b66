#!/usr/bin/env python3
"""The reference's own flow (MLS_simple docstring, zignal/measure/mls.py:423-428) step by step, numpy to numpy, with where the
host time goes (cProfile of the second round): construct, get_impulse, fft."""
import cProfile, os, pstats, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zignal_amd.measure.mls import MLS_simple

N, R = int(sys.argv[1]) if len(sys.argv) > 1 else 18, 4


def flow():
    t = [time.perf_counter()]
    ms = MLS_simple(N=N, fs=96000, repeats=R, device=0); t.append(time.perf_counter())
    y = ms.samples; imp = ms.get_impulse(y); t.append(time.perf_counter())
    frq, mag = imp.fft(window='rectangular', normalise=False, device=0); t.append(time.perf_counter())
    return [1e3 * (b - a) for a, b in zip(t, t[1:])]


flow()
for _ in range(3):
    print("N=%d repeats=%d: construct %.3f ms, get_impulse %.3f ms, fft %.3f ms" % ((N, R) + tuple(flow())), flush=True)
pr = cProfile.Profile(); pr.enable(); flow(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
