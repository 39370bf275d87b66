#!/usr/bin/env python3
"""Only the fused de-emphasis + impulse-response call (zmls_impulse_filtered_device), N=20 x 512 channels x 17 periods, a few
times: for rocprofv3 --kernel-trace (tools/kernel_times.py lists the sweep, the filter kernels and the passes) and for the
ZMLS_IIR_DOT=0/1 A/B of the sweep's two forms; FORMS=0,1 times the sweep form and the post-filter form one after the other."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zignal_amd import _lib
from zignal_amd.device import DeviceBuffer, IirFilter, Plan, Timer, sync
from zignal_amd.measure.mls import TAPS

N, C, R = 20, 512, 17
p = Plan(N, TAPS[N][0], device=0)
L = p.L
p.reserve(C, staging=True)
x = DeviceBuffer(R * L * C * 8)
if os.environ.get("SYNTH"):                                               # SYNTH=1: the synthetic capture of bench.py instead of zeros
    p.synth_device(x.ptr, _lib.ZMLS_F64, R, C, C, 1, seed=7, sigma=1e-3)
else:
    x.fill(0)
h = DeviceBuffer(L * C * 8)
f = IirFilter([1.0, -1.9, 0.9], [1.2, -1.8, 0.7], device=0)
f.reserve_fused(p, R * L, C)
call = lambda: p.impulse_filtered_device(f, x.ptr, _lib.ZMLS_F64, R * L, C, C, 1, h.ptr, C, 1)
t = Timer(0)
forms = [int(v) for v in os.environ.get("FORMS", "-1").split(",")]       # FORMS=0,1: the sweep form and the post-filter form (zmls_iir_set_fused_form)
if os.environ.get("ONCE_M"):                                              # ONCE_M=16: the one-sweep filter forced with that many chunks per workgroup (0: never)
    m = int(os.environ["ONCE_M"])
    f.set_one_sweep(1 if m else 0, m)
for form in forms:
    f.set_fused_form(form)
    call(); sync(0)
    for _ in range(3):
        t.start(); call(); t.stop(); sync(0)
        print("fused call, form %d (took %d): %.3f ms" % (form, f.last_fused_form(), t.elapsed_ms()), flush=True)
