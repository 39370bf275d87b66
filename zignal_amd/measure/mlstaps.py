"""Feedback-tap table `TAPS[N] -> tuple of tap tuples`, N = 3..32.

Same content and order as zignal/measure/mlstaps.py:18-1093 (1014 maximum-length tap sets,
1-indexed, each starting with N), shipped as data in taps.json (written by
tests/golden/make_golden.py from the reference's table; tests/test_host_cpu.py checks its
digest and tests/test_gpu_sequence.py proves on the device that every set used is
maximum-length).
"""
import json
import os

__all__ = ['TAPS']

with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "taps.json")) as _f:
    TAPS = {int(N): tuple(tuple(t) for t in sets) for N, sets in json.load(_f).items()}
del _f
