"""Helpers shared by the GPU parity tests (they call the HIP library through the C ABI and
check it against the oracle / the golden fixtures)."""
import json
import os

import numpy as np

import inputs
from oracle import c_oracle, mls_oracle
from zignal_amd import _lib
from zignal_amd.device import Plan

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_plans = {}


def load_json(name):
    with open(os.path.join(G, name)) as f:
        return json.load(f)


def plan(N, taps, dtype="f64"):
    key = (N, tuple(taps), dtype)
    if key not in _plans:
        if len(_plans) > 12:          # keep device memory bounded across a long test session
            for p in _plans.values():
                p.close()
            _plans.clear()
        _plans[key] = Plan(N, taps, dtype=dtype, device=0)
    return _plans[key]


def oracle_ir(N, taps, x, ref=None):
    """Reference-algorithm impulse response of a single-channel capture (float64 column)."""
    if ref is None:
        ref = c_oracle.sequence(N, taps)
    return mls_oracle.get_impulse(N, tuple(taps), x, ref=ref)[:, 0]


def rel_err(got, want):
    peak = float(np.max(np.abs(want)))
    return float(np.max(np.abs(np.asarray(got, dtype=np.float64) - want))) / (peak if peak else 1.0)


def case_input(case):
    seq = c_oracle.sequence(case["N"], case["taps"])[:, 0]
    x = inputs.make_input(case["kind"], seq, case["R"], case["seed"], case["shape"], case["dtype"])
    assert inputs.digest(x) == case["x_sha256"], "input generator drifted from the one the golden was made with"
    return x
