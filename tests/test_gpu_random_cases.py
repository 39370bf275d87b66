"""Randomised sweep of the device entry point over everything that selects a code path: order (single / two / three
passes), channel count (ragged slabs, odd counts), period count (the exact-mean path for non powers of two), input
type, input and output layout, plan arithmetic, staging reserved or not (adapters vs element path), forced tiles.
Every case is held to the oracle on every channel: 1e-12 x peak (f64 plans), 1e-5 (f32 plans)."""
import numpy as np
import pytest

from gpu_common import oracle_ir, rel_err
from oracle import c_oracle
from zignal_amd import _lib
from zignal_amd.device import DeviceBuffer, Plan
from zignal_amd.measure.mls import TAPS

pytestmark = pytest.mark.gpu

NP = {"f64": np.float64, "f32": np.float32, "i16": np.int16, "i32": np.int32}
CODE = {"f64": _lib.ZMLS_F64, "f32": _lib.ZMLS_F32, "i16": _lib.ZMLS_I16, "i32": _lib.ZMLS_I32}


def cases():
    rng = np.random.default_rng(20261110)
    out = []
    for i in range(60):
        N = int(rng.choice([6, 9, 10, 11, 12, 13, 14, 15, 18, 19, 21]))
        big = N >= 15
        C = int(rng.choice([1, 2, 3, 16, 32]) if big else rng.choice([1, 2, 3, 5, 8, 15, 16, 17, 24, 31, 32, 33, 48, 70]))
        R = int(rng.choice([2, 3]) if big else rng.choice([2, 3, 4, 6, 8]))
        out.append(dict(N=N, C=C, R=R, plan=str(rng.choice(["f64", "f64", "f32"])), x=str(rng.choice(["f64", "f64", "f32", "i16", "i32"])),
                        lin=str(rng.choice(["frame", "stack", "padded"])), lout=str(rng.choice(["frame", "stack"])),
                        staging=bool(rng.integers(0, 2)), tile=int(rng.choice([-1, -1, 0, 2])), tapset=int(rng.integers(0, 2)), seed=i))
    return out


@pytest.mark.parametrize("c", cases(), ids=lambda c: "N%d-C%d-R%d-%s-%s-%s-%s-%s-t%d" % (c["N"], c["C"], c["R"], c["plan"], c["x"], c["lin"], c["lout"],
                                                                                 "st" if c["staging"] else "el", c["tile"]))
def test_random_case(c):
    N, C, R = c["N"], c["C"], c["R"]
    taps = TAPS[N][c["tapset"] % len(TAPS[N])]
    L = (1 << N) - 1
    rng = np.random.default_rng(1000 + c["seed"])
    if c["x"] in ("i16", "i32"):
        x = rng.integers(-20000, 20000, size=(R * L, C)).astype(NP[c["x"]])
    else:
        x = rng.standard_normal((R * L, C)).astype(NP[c["x"]])
    ref = c_oracle.sequence(N, taps)
    want = np.stack([oracle_ir(N, taps, x[:, k:k + 1].astype(np.float64), ref) for k in range(C)], axis=1)
    p = Plan(N, taps, dtype=c["plan"], device=0)
    p.reserve(C, staging=c["staging"])
    if c["tile"] >= 0 and p.info()["passes"] > 1:
        p.force_tile(c["tile"])
        p.reserve(C)
    # input layout
    if c["lin"] == "frame":
        xm, rs, cs = np.ascontiguousarray(x), C, 1
    elif c["lin"] == "stack":
        xm, rs, cs = np.ascontiguousarray(x.T), 1, R * L
    else:                                            # frames of a wider buffer: row stride C + 3
        xm = np.zeros((R * L, C + 3), dtype=x.dtype)
        xm[:, :C] = x
        rs, cs = C + 3, 1
    d_x = DeviceBuffer(xm.nbytes).upload(xm)
    npdt = p.np_dtype
    d_h = DeviceBuffer(L * C * npdt().itemsize)
    d_h.fill(0xFF)
    ors, ocs = (C, 1) if c["lout"] == "frame" else (1, L)
    wide = c["tile"] == 2 and p.info()["passes"] > 1
    vec_ok = c["lin"] != "stack" and c["lout"] == "frame" and c["x"] == c["plan"] and C % (2 if c["plan"] == "f64" else 4) == 0 \
        and (rs * npdt().itemsize) % 16 == 0
    if wide and not (vec_ok or c["staging"]):
        with pytest.raises(_lib.ZmlsError):          # a forced wide tile needs vector-aligned data on both sides
            p.impulse_device(d_x.ptr, CODE[c["x"]], R * L, C, rs, cs, d_h.ptr, ors, ocs)
        p.close()
        return
    try:
        p.impulse_device(d_x.ptr, CODE[c["x"]], R * L, C, rs, cs, d_h.ptr, ors, ocs)
    except _lib.ZmlsError:
        if wide:                                     # staged on one side only, element path on the other: also refused
            p.close()
            return
        raise
    h = d_h.download((L, C) if c["lout"] == "frame" else (C, L), npdt)
    h = h if c["lout"] == "frame" else h.T
    tol = 1e-12 if c["plan"] == "f64" else 1e-5
    if c["x"] == "f32" and c["plan"] == "f64":
        tol = 1e-12                                  # float32 samples are exact in float64: same bar
    assert rel_err(h, want) <= tol
    p.close()
