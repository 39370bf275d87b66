#!/usr/bin/env python3
"""Does device memory come back? Every kind of object the library hands out is created, used and destroyed over and over,
stage by stage, and the free device memory is compared before and after each stage. Measured (round 2): plans, captures,
filters (incl. the fused scratch), loopback communicators, MLS / MLS_simple objects: +0 bytes once the first round of a
stage has loaded what the runtime keeps (10 MiB of runtime staging once per process, 2 MiB once for the filter kernels);
spectra (hipFFT plans until round 3: 25 MB once per set of sizes inside that library; the library's own passes since: +0); RCCL communicators: about 19 MB stay
with librccl after every ncclCommDestroy (inside the library, not in libzmls's buffers)."""
import ctypes, os, sys, gc
import numpy as np
sys.path.insert(0, os.getcwd())
from zignal_amd import _lib, multi
from zignal_amd.device import DeviceBuffer, IirFilter, Plan, Spectrum, sync
from zignal_amd.measure.mls import MLS_simple, TAPS, _MLS_base
def free_bytes():
    f, t = ctypes.c_size_t(), ctypes.c_size_t()
    _lib.check(_lib.lib().zmls_dev_mem_info(0, ctypes.byref(f), ctypes.byref(t)))
    return f.value
Plan(10, TAPS[10][0], device=0).close(); sync(0)
rng = np.random.default_rng(0)
def stage(name, fn, n=30):
    a = free_bytes()
    for i in range(n): fn(i)
    sync(0); gc.collect()
    print("%-28s %+d bytes over %d rounds" % (name, a - free_bytes(), n), flush=True)
def plans(i):
    N = 3 + i % 16
    p = Plan(N, TAPS[N][0], dtype=("f64","f32")[i%2], device=0)
    x = rng.standard_normal((3 * p.L, 1 + i % 7))
    p.impulse_host(x); p.impulse_host(np.ascontiguousarray(x.T).T); p.close()
stage("plan + impulse_host", plans, 200)
def caps(i):
    p = Plan(14, TAPS[14][0], device=0); d, ms = p.alloc_capture(3 * p.L, 8); d.free(); p.close()
stage("plan + impulse_host again", plans, 200)
stage("alloc_capture", caps)
def iir(i):
    p = Plan(12, TAPS[12][0], device=0)
    f = IirFilter([1.0, -0.5], [1.0, -0.9], device=0); f.filter_host(rng.standard_normal((5000, 3))); f.reserve_fused(p, 3 * p.L, 4); f.close(); p.close()
stage("iir", iir, 300)
def spec(i):
    s = Spectrum(4095 + 2 * (i % 3), 1 + i % 3, device=0); s.magnitude_db_host(np.ones((s.n_samples, s.n_ch))); s.close()
stage("spectrum", spec, 120)
def comm(i):
    p = Plan(10, TAPS[10][0], device=0)
    c = multi.Comm(multi.make_unique_id(), 0, 1, device=0); multi.PeriodSharded(p, c, 8); c.close(); p.close()
stage("rccl comm", comm, 5)
def loop(i):
    hub = multi.LoopbackHub(1); cl = multi.Comm.loopback(hub, 0, 0); cl.close(); hub.close()
stage("loopback comm", loop)
def simple(i):
    ms_ = MLS_simple(N=10, fs=48000, repeats=3, device=0); ms_.get_impulse(ms_.samples); del ms_
stage("MLS_simple", simple)
def base(i):
    m = _MLS_base(9, TAPS[9][0], device=0); m.get_impulse(m.get_full_sequence(2)); del m
stage("_MLS_base", base)
