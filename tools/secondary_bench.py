#!/usr/bin/env python3
"""Device-time of the kernels either side of the headline path (HIP events, inputs resident): emphasis filter,
magnitude response, layout adapters. Prints algorithmic GB/s (bytes each step must read + write once)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zignal_amd import _lib
from zignal_amd.device import DeviceBuffer, IirFilter, Plan, Spectrum, Timer, sync
from zignal_amd.filters.biquads import RBJ
from zignal_amd.measure.mlstaps import TAPS


def timed(fn, reps=5):
    fn(); sync(0)
    t = Timer(0); t.start()
    for _ in range(reps):
        fn()
    t.stop()
    return t.elapsed_ms() / reps


B, A = RBJ(filtertype="highshelf", gaindb=-10, f0=100, Q=0.707, fs=96000).get_coefficients()
f = IirFilter(A, B, device=0)
if os.environ.get("SEGMENT_MIB"):
    f.set_segment(int(os.environ["SEGMENT_MIB"]) << 20)       # A/B runs: 0 = one segment, else MiB of signal per segment of the time axis
for n, C, rs, cs, name in ((17 * (2**20 - 1), 1, 1, 1, "one channel, 17 periods of N=20"),
                           (2 * (2**20 - 1), 512, 512, 1, "(2L, 512) frames x channels"),
                           (17 * (2**20 - 1), 64, 64, 1, "(17L, 64) frames x channels"),
                           (2 * (2**20 - 1), 512, 1, 2 * (2**20 - 1), "(512, 2L) stacked captures")):
    d = DeviceBuffer(n * C * 8); d.fill(0)
    y = DeviceBuffer(n * C * 8)
    f.reserve(n, C)
    ors, ocs = (C, 1) if cs == 1 else (1, n)
    ms = timed(lambda: f.filter_device(d.ptr, _lib.ZMLS_F64, n, C, rs, cs, y.ptr, ors, ocs))
    print("iir      %-34s %8.3f ms  %7.0f GB/s (read + write once = %.2f GB)" % (name, ms, 2 * n * C * 8 / ms / 1e6, 2 * n * C * 8 / 1e9), flush=True)
    d.free(); y.free()

for N, C in ((20, 1), (20, 64), (16, 512)):
    L = 2**N - 1
    sp = Spectrum(L, C, device=0)
    h = DeviceBuffer(L * C * 8); h.fill(0)
    m = DeviceBuffer(sp.n_bins * C * 8)
    ms = timed(lambda: sp.magnitude_db_device(h.ptr, C, 1, None, False, m.ptr))
    print("spectrum N=%d x %d channels             %8.3f ms  %7.0f GB/s (response in + spectrum out = %.3f GB)" % (N, C, ms, 1.5 * L * C * 8 / ms / 1e6, 1.5 * L * C * 8 / 1e9), flush=True)
    sp.close(); h.free(); m.free()

N, C, R = 20, 512, 17
p = Plan(N, TAPS[N][0], device=0)
L = p.L
p.reserve(C, staging=True)
x = DeviceBuffer(R * L * C * 8); x.fill(0)
h = DeviceBuffer(L * C * 8)
for name, args in (("frame-major in/out (fused gather)", (C, 1, C, 1)), ("stacked captures in, frame-major out", (1, R * L, C, 1)),
                   ("stacked captures in and out", (1, R * L, 1, L)), ("frame-major in, stacked out", (C, 1, 1, L))):
    rs, cs, ors, ocs = args
    ms = timed(lambda: p.impulse_device(x.ptr, _lib.ZMLS_F64, R * L, C, rs, cs, h.ptr, ors, ocs), 3)
    print("impulse  %-38s %8.3f ms  %7.0f GB/s algorithmic" % (name, ms, R * L * C * 8 / ms / 1e6), flush=True)

# de-emphasis + impulse response (MLS.apply_deemphasis + get_impulse, mls.py:401-414): the two device steps against the fused call
y = DeviceBuffer(R * L * C * 8)
f.reserve(R * L, C)
two = timed(lambda: (f.filter_device(x.ptr, _lib.ZMLS_F64, R * L, C, C, 1, y.ptr, C, 1),
                     p.impulse_device(y.ptr, _lib.ZMLS_F64, R * L, C, C, 1, h.ptr, C, 1)), 3)
y.free()
f.reserve_fused(p, R * L, C)
f.set_fused_form(0)
sweep = timed(lambda: p.impulse_filtered_device(f, x.ptr, _lib.ZMLS_F64, R * L, C, C, 1, h.ptr, C, 1), 3)
f.set_fused_form(-1)
one = timed(lambda: p.impulse_filtered_device(f, x.ptr, _lib.ZMLS_F64, R * L, C, C, 1, h.ptr, C, 1), 3)
print("de-emphasis + impulse, N=%d x %d channels x %d periods: filter then transform %8.3f ms, fused: sweep form %8.3f ms, library's choice (form %d) %8.3f ms (%.0f GB/s of capture read)"
      % (N, C, R, two, sweep, f.last_fused_form(), one, R * L * C * 8 / one / 1e6), flush=True)
x1 = DeviceBuffer(R * L * 8); x1.fill(0)
h1 = DeviceBuffer(L * 8)
f.reserve(R * L, 1); f.reserve_fused(p, R * L, 1)
y1 = DeviceBuffer(R * L * 8)
two = timed(lambda: (f.filter_device(x1.ptr, _lib.ZMLS_F64, R * L, 1, 1, 1, y1.ptr, 1, 1),
                     p.impulse_device(y1.ptr, _lib.ZMLS_F64, R * L, 1, 1, 1, h1.ptr, 1, 1)), 5)
f.set_fused_form(0)
sweep = timed(lambda: p.impulse_filtered_device(f, x1.ptr, _lib.ZMLS_F64, R * L, 1, 1, 1, h1.ptr, 1, 1), 5)
f.set_fused_form(-1)
one = timed(lambda: p.impulse_filtered_device(f, x1.ptr, _lib.ZMLS_F64, R * L, 1, 1, 1, h1.ptr, 1, 1), 5)
print("de-emphasis + impulse, N=%d, one channel x %d periods (the reference's shape): filter then transform %8.3f ms, fused: sweep form %8.3f ms, library's choice (form %d) %8.3f ms"
      % (N, R, two, sweep, f.last_fused_form(), one), flush=True)

# several sources at once (mls.py:338-343): K plans over one capture, one call per source against zmls_impulse_sources_device
from zignal_amd.device import Sources
for K, Rk in ((4, 17), (4, 5), (2, 17), (4, 2)):
    plans = [p] + [Plan(N, TAPS[N][k], device=0) for k in range(1, K)]
    for q in plans[1:]:
        q.reserve(C)
    hk = DeviceBuffer(K * L * C * 8)
    each = timed(lambda: [q.impulse_device(x.ptr, _lib.ZMLS_F64, Rk * L, C, C, 1, hk.ptr + k * L * C * 8, C, 1) for k, q in enumerate(plans)], 3)
    grp = Sources(plans)
    grp.reserve(Rk * L, C)
    once = timed(lambda: grp.impulse_device(x.ptr, _lib.ZMLS_F64, Rk * L, C, C, 1, hk.ptr, L * C, C, 1), 3)
    grp.reserve(Rk * L, C, iir=f)
    once_f = timed(lambda: grp.impulse_device(x.ptr, _lib.ZMLS_F64, Rk * L, C, C, 1, hk.ptr, L * C, C, 1, iir=f), 3)
    print("sources  N=%d x %d channels x %d periods, %d sources: one call per source %8.3f ms, one call %8.3f ms (%.0f GB/s of capture read + responses written); "
          "with de-emphasis %8.3f ms" % (N, C, Rk, K, each, once, ((Rk - 1) + K) * L * C * 8 / once / 1e6, once_f), flush=True)
    hk.free()
    for q in plans[1:]:
        q.close()
