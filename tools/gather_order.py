#!/usr/bin/env python3
"""How much of the gather pass's time is the scatter of its rows? The headline batch with the plan's own gather table,
then with tables that keep the same kernel but change where the rows lie: sorted (row d-1 for index d: every tile reads
1024 consecutive rows), sorted in runs of 2 / 8 / 64 rows at random places, and a random permutation."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zignal_amd import _lib
from zignal_amd.device import DeviceBuffer, Plan, sync
from zignal_amd.measure.mlstaps import TAPS

N, C, R = 20, 512, 17
p = Plan(N, TAPS[N][0], device=0)
L = p.L
p.reserve(C)
d_x, ms0 = p.alloc_capture(R * L, C)
d_h = DeviceBuffer(L * C * 8)
d_x.fill(0)


def timed(label):
    t = np.zeros(2)
    for _ in range(5):
        t += np.array(p.impulse_device_timed(d_x.ptr, _lib.ZMLS_F64, R * L, C, C, 1, d_h.ptr, C, 1))
    print("%-46s gather pass %.3f ms (%.0f GB/s)  scatter pass %.3f ms" % (label, t[0] / 5, (R - 1) * L * C * 8 / (t[0] / 5) / 1e6, t[1] / 5), flush=True)


timed("the plan's table (MLS permutation)")
rng = np.random.default_rng(1)
n = 1 << N
for run in (1, 2, 8, 64, 1024, n):
    q = np.empty(n, dtype=np.uint32)
    if run == n:
        q[1:] = np.arange(L, dtype=np.uint32)
        label = "sorted: index d reads row d-1"
    else:
        blocks = rng.permutation(n // run).astype(np.uint32)
        q = (np.repeat(blocks, run) * run + np.tile(np.arange(run, dtype=np.uint32), n // run)) % np.uint32(L)
        label = "random runs of %d consecutive rows" % run
    q[0] = 0xFFFFFFFF
    _lib.check(_lib.lib().zmls_plan_import_table(p._h, 0, q.ctypes.data))
    timed(label)
