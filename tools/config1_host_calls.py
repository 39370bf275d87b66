#!/usr/bin/env python3
"""BASELINE config 1 through the host entry point, call by call (what bench.py reports as host_api_ms_per_call), after one
of the things a bench.py process has done before it gets there.
usage: config1_host_calls.py [none|bigalloc|oracle20|dlchan|subproc|sleep][@late]   (@late: after the library's copy engine exists)"""
import os, subprocess, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zignal_amd import _lib
from zignal_amd.device import DeviceBuffer, Plan, download_channel, sync
from zignal_amd.measure.mls import TAPS


def cpu_now():
    return int(open("/proc/self/stat").read().rsplit(")", 1)[1].split()[36])


def do(what):
    if what == "bigalloc":
        b = DeviceBuffer(60 << 30); b.fill(0); sync(0); b.free()
    if what == "oracle20":
        from oracle import c_oracle, mls_oracle
        ref = c_oracle.sequence(20, TAPS[20][0])
        xx = np.random.default_rng(1).standard_normal((3 * ((1 << 20) - 1), 1))
        mls_oracle.get_impulse(20, TAPS[20][0], xx, ref=ref)
    if what == "dlchan":
        b = DeviceBuffer(8 * 64 * ((1 << 18) - 1)); b.fill(0)
        for c in (0, 5, 63):
            download_channel(0, b.ptr, _lib.ZMLS_F64, (1 << 18) - 1, 64, 1, c)
        b.free()
    if what in ("subproc", "subproc8"):
        code = "import time\nt=time.time()\nwhile time.time()-t<%d: sum(range(10000))" % (8 if what == "subproc8" else 2)
        procs = [subprocess.Popen([sys.executable, "-c", code]) for _ in range(16)]
        for pr in procs:
            pr.wait()
    if what == "sleep":
        time.sleep(3.0)


if __name__ == "__main__":
    arg = sys.argv[1] if len(sys.argv) > 1 else "none"
    what, late = (arg.split("@") + [""])[:2]
    N = 16
    if not late:
        do(what)
    p = Plan(N, TAPS[N][0], device=0)
    x = np.random.default_rng(0).standard_normal((2 * p.L, 1))
    p.impulse_host(x)
    print("   started on cpu %d" % cpu_now(), flush=True)
    big = np.zeros(4 << 20); d = DeviceBuffer(big.nbytes); d.upload(big); d.free()      # the copy engine's threads exist from here
    if late:
        do(what)
    print("   running on cpu %d (affinity: %d cpus)" % (cpu_now(), len(os.sched_getaffinity(0))), flush=True)
    ts = []
    for _ in range(40):
        t0 = time.perf_counter(); p.impulse_host(x); ts.append((time.perf_counter() - t0) * 1e6)
    print("%-14s median %.0f us, min %.0f, max %.0f" % (arg, np.median(ts), np.min(ts), np.max(ts)), flush=True)
