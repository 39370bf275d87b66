"""ORACLE — test infrastructure, not product code.

One worker of bench.py's all-cores CPU baseline: runs the oracle's `get_impulse` (the reference's algorithm as the
reference ships it: per-bit LFSR regenerated on every call, then fftconvolve; zignal/measure/mls.py:315-328) over and
over on one capture-shaped array for a given number of seconds and prints `<calls> <seconds>`. The reference is
single-threaded and captures are independent, so P such processes are what a user with P cores can do.
Timing does not depend on the sample values, so the capture is seeded noise of the right shape."""
import sys
import time

import numpy as np

import mls_oracle


def main():
    N, R, seconds, seed = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4])
    taps = tuple(int(t) for t in sys.argv[5].split(","))
    L = (1 << N) - 1
    x = np.random.default_rng(seed).standard_normal((R * L, 1))
    calls, t0 = 0, time.perf_counter()
    while True:
        mls_oracle.get_impulse(N, taps, x)
        calls += 1
        if time.perf_counter() - t0 >= seconds:
            break
    print(calls, time.perf_counter() - t0)


if __name__ == "__main__":
    main()
