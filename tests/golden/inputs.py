"""Deterministic capture generators shared by make_golden.py (which feeds them
to the real reference) and by the parity tests (which feed the same arrays to
the oracle and to the HIP library).

Nothing here comes from the reference: these are the *inputs* the build owns
(SURVEY.md §8(c) item 3). Every generator takes the one-period stimulus `seq`
(+1.0/-1.0 float64, shape (L,)) from whoever calls it, so the reference, the
oracle and the device sequence can each be plugged in.
"""
import hashlib

import numpy as np

KINDS = ("loopback", "noise", "system")


def system_taps(seed, ntaps=32):
    """Short decaying FIR 'room' used by the 'system' kind."""
    rng = np.random.default_rng(seed ^ 0x5A17)
    g = rng.standard_normal(ntaps) * np.exp(-np.arange(ntaps) / 6.0)
    g[0] = 1.0
    return g


def make_input(kind, seq, repeats, seed, shape="col", dtype="float64"):
    """Return a capture of `repeats` periods.

    kind     : 'loopback' -> the stimulus itself
               'noise'    -> standard normal noise (no MLS content at all)
               'system'   -> steady-state response of a 32-tap FIR with a
                             seed-dependent delay, plus 1e-3 noise per sample
    shape    : 'col' -> (R*L, 1), 'flat' -> (R*L,)
    dtype    : 'float64' | 'float32' | 'int16' (int16: scaled by 1024, rounded)
    """
    seq = np.asarray(seq, dtype=np.float64).reshape(-1)
    L = seq.shape[0]
    rng = np.random.default_rng(seed)
    if kind == "loopback":
        x = np.tile(seq, repeats)
    elif kind == "noise":
        x = rng.standard_normal(repeats * L)
    elif kind == "system":
        g = system_taps(seed)
        delay = int(seed * 9973 % L)
        period = np.zeros(L)
        for k, gk in enumerate(g):
            period += gk * np.roll(seq, delay + k)
        x = np.tile(period, repeats) + 1e-3 * rng.standard_normal(repeats * L)
    else:
        raise ValueError(kind)
    if dtype == "float64":
        pass
    elif dtype == "float32":
        x = x.astype(np.float32)
    elif dtype == "int16":
        x = np.clip(np.rint(x * 1024.0), -32768, 32767).astype(np.int16)
    else:
        raise ValueError(dtype)
    if shape == "col":
        x = x.reshape(-1, 1)
    elif shape != "flat":
        raise ValueError(shape)
    return np.ascontiguousarray(x)


def digest(a):
    """SHA-256 of the raw bytes of a C-contiguous array (dtype-sensitive)."""
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def pack_bits_from_samples(samples):
    """+1/-1 samples -> LSB-first packed bits, the KAT packing of SURVEY §8(c)."""
    s = np.asarray(samples, dtype=np.float64).reshape(-1)
    bits = ((1.0 - s) / 2.0).astype(np.uint8)
    return np.packbits(bits, bitorder="little")


def sparse_indices(L, seed, n=4096):
    rng = np.random.default_rng(seed ^ 0xC0FFEE)
    return np.sort(rng.choice(L, size=min(n, L), replace=False))
