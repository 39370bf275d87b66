#!/usr/bin/env python3
"""Measure a known system with an emphasised MLS, end to end on the GPU.

The scenario of zignal's examples/ex_mls.py (N=16, fs=48 kHz, 5 repeats; BASELINE.json config 1 is its template):
a stimulus with a -10 dB high-shelf emphasis goes through a "system" (three peaking filters and a delay, simulated
on the host with scipy, as a sound card and a room would be outside this library), the capture is de-emphasised, the
first period is dropped, the others are averaged, and the impulse response and its magnitude response are
recovered. Stimulus, emphasis, de-emphasis, averaging, correlation and the spectrum run on the MI355X through
libzmls; nothing here falls back to the CPU.

    python examples/ex_mls.py [--order 16] [--fs 48000] [--repeats 5] [--plot response.png]
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zignal_amd.audio import Audio  # noqa: E402
from zignal_amd.filters.biquads import RBJ  # noqa: E402
from zignal_amd.measure.mls import MLS, TAPS  # noqa: E402

SYSTEM = ((-30, 10), (50, 100), (-60, 1000))      # (gain dB, centre Hz) of the peaking filters, Q = 7.07
DELAY = 24000                                      # samples


def system_under_test(x, fs):
    """The device being measured: host-side simulation, not part of the library's path."""
    y = Audio(fs=fs, initialdata=x)
    for gaindb, f0 in SYSTEM:
        y.samples = RBJ(filtertype="peak", gaindb=gaindb, f0=f0, Q=0.707 * 10, fs=fs).filter_samples(y.samples)
    y.delay(DELAY)
    return y


def measure(N=16, fs=48000, repeats=5, device=None):
    """Returns (mls, capture after de-emphasis, impulse response Audio, (frequencies, magnitude dB))."""
    B, A = RBJ(filtertype="highshelf", gaindb=-10, f0=100, Q=0.707, fs=fs).get_coefficients()
    mls = MLS(N=N, taps=TAPS[N][0], fs=fs, repeats=repeats, B=B, A=A, device=device)
    mls.apply_emphasis()
    y = system_under_test(mls.samples, fs)
    z = mls.apply_deemphasis(y.samples)
    ir = mls.get_impulse(z)
    # no window: it would scale the impulse (zignal/measure/mls.py:464-466)
    freqs, mag = ir.fft(window='rectangular', normalise=False)
    return mls, z, ir, (freqs, mag)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--order", type=int, default=16)
    ap.add_argument("--fs", type=int, default=48000)
    ap.add_argument("--repeats", type=int, default=5)
    ap.add_argument("--plot", default=None, help="write the magnitude response to this image file (needs matplotlib)")
    args = ap.parse_args()
    mls, z, ir, (freqs, mag) = measure(args.order, args.fs, args.repeats)
    print(repr(mls))
    print(mls)
    peak, where = ir.peak()
    print("impulse response: %d samples, peak %.4f at sample %d (system delay %d)" % (len(ir), peak[0], where[0], DELAY))
    for f in (100, 1000, 5000):
        k = int(round(f * 2 * len(mag) / args.fs))
        print("magnitude at %5d Hz: %+7.2f dB" % (f, mag[k, 0]))
    if args.plot:
        ir.plot_fft(plotname=args.plot, window='rectangular', normalise=False)
        print("wrote", args.plot)


if __name__ == '__main__':
    main()
