#!/usr/bin/env python3
"""Host <-> device copy rates of the library's buffer helpers (csrc/hostcopy.hip) on pageable numpy memory."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zignal_amd.device import DeviceBuffer, sync

for kib in (4, 16, 64, 128, 256, 512, 1024, 2048, 3072, 4096, 8192, 16384, 32768, 65536, 131072, 262144, 524288, 1048576):
    n = kib << 10
    a = np.random.default_rng(0).integers(0, 256, size=n, dtype=np.uint8)
    d = DeviceBuffer(n)
    d.upload(a); d.download((n,), np.uint8)
    up, down = [], []
    for _ in range(5):
        t0 = time.perf_counter(); d.upload(a); up.append(time.perf_counter() - t0)
        out = np.empty(n, dtype=np.uint8)
        out[::4096] = 0                               # touch the pages: a fresh numpy array is not resident yet
        import ctypes
        from zignal_amd import _lib
        t0 = time.perf_counter(); _lib.check(_lib.lib().zmls_dev_download(0, out.ctypes.data, d.ptr, n)); down.append(time.perf_counter() - t0)
    print("%8d KiB: up %.3f ms = %.1f GB/s | down %.3f ms = %.1f GB/s" % (kib, min(up) * 1e3, n / min(up) / 1e9, min(down) * 1e3, n / min(down) / 1e9), flush=True)
    d.free()
