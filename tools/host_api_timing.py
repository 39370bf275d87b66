#!/usr/bin/env python3
"""End-to-end timings of the drop-in Python API (numpy in, numpy out, PCIe included) next to the oracle
(the reference's algorithm restated, as the reference ships it: sequence regenerated on every call) on this box's
host: BASELINE.json config 1 (N=16, one channel, one averaged period) and its larger siblings."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import mls_oracle
from zignal_amd.measure.mls import MLS, TAPS, _MLS_base


def best(f, n=5):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); f(); ts.append(time.perf_counter() - t0)
    return min(ts)


for N, R in ((16, 2), (16, 5), (18, 2), (20, 2), (20, 17)):
    taps = TAPS[N][0]
    L = (1 << N) - 1
    t0 = time.perf_counter()
    m = _MLS_base(N, taps)
    seq = m.get_full_sequence(R)
    t_first = time.perf_counter() - t0
    rng = np.random.default_rng(N)
    x = np.roll(seq, 1000, axis=0) * 0.5 + 1e-3 * rng.standard_normal(seq.shape)
    h = m.get_impulse(x)
    t_gpu = best(lambda: m.get_impulse(x))
    t_seq = best(lambda: m.get_full_sequence(R))
    n_cpu = 1 if N >= 20 else 2
    t_cpu = best(lambda: mls_oracle.get_impulse(N, taps, x), n_cpu)
    h_cpu = mls_oracle.get_impulse(N, taps, x)
    err = np.max(np.abs(h - h_cpu)) / np.max(np.abs(h_cpu))
    print("N=%d R=%d: zignal_amd get_impulse %.2f ms (%.1f M IR-samples/s), get_full_sequence %.2f ms, first plan+sequence %.0f ms | "
          "oracle get_impulse %.0f ms (%.2f M IR-samples/s) | x%.0f | max rel err %.1e"
          % (N, R, t_gpu * 1e3, L / t_gpu / 1e6, t_seq * 1e3, t_first * 1e3, t_cpu * 1e3, L / t_cpu / 1e6, t_cpu / t_gpu, err), flush=True)

# raw int16 frames (sound card / wav) against the same capture converted to float64 on the host first
for N, R, C in ((20, 17, 1), (18, 3, 64)):
    taps = TAPS[N][0]
    L = (1 << N) - 1
    m = _MLS_base(N, taps)
    rng = np.random.default_rng(N)
    x16 = rng.integers(-20000, 20000, size=(R * L, C), dtype=np.int16)
    m.get_impulse_pcm(x16)
    t_pcm = best(lambda: m.get_impulse_pcm(x16))
    t_conv = best(lambda: m.get_impulses(x16 / 32768.0))
    xf = x16 / 32768.0
    t_f64 = best(lambda: m.get_impulses(xf))
    print("N=%d R=%d C=%d int16: get_impulse_pcm %.2f ms | convert on host + get_impulses %.2f ms | float64 already there %.2f ms"
          % (N, R, C, t_pcm * 1e3, t_conv * 1e3, t_f64 * 1e3), flush=True)
