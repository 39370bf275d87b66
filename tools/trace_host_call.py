import os, sys, numpy as np
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo/zignal_amd") else os.getcwd())
from zignal_amd.measure.mls import TAPS, _MLS_base
for N, R in ((16, 2), (16, 5), (18, 2)):
    m = _MLS_base(N, TAPS[N][0])
    x = m.get_full_sequence(R) * 0.5
    for _ in range(3):
        m.get_impulse(x)
    import time
    for _ in range(3):
        t0 = time.perf_counter(); m.get_full_sequence(R); print("get_full_sequence %.0f us" % ((time.perf_counter() - t0) * 1e6), file=sys.stderr)
    print("--", N, R, file=sys.stderr, flush=True)
