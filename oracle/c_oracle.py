"""ctypes loader for oracle/_build/libmls_oracle.so (test infrastructure only)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libmls_oracle.so")
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = ctypes.CDLL(_SO)
        i32p = ctypes.POINTER(ctypes.c_int32)
        _lib.mls_oracle_bits.argtypes = [ctypes.c_int, i32p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64]
        _lib.mls_oracle_bits_packed.argtypes = _lib.mls_oracle_bits.argtypes
        _lib.mls_oracle_bits_packed_word.argtypes = _lib.mls_oracle_bits.argtypes
        _lib.mls_oracle_sequence.argtypes = [ctypes.c_int, i32p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int64]
        _lib.mls_oracle_average.argtypes = [ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                            ctypes.c_void_p]
        _lib.mls_oracle_xcorr_direct.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]
    return _lib


def _taps(taps):
    arr = (ctypes.c_int32 * len(taps))(*[int(t) for t in taps])
    return arr, len(taps)


def bits(N, taps, nbits):
    out = np.empty(nbits, dtype=np.uint8)
    t, n = _taps(taps)
    if lib().mls_oracle_bits(N, t, n, out.ctypes.data, nbits):
        raise ValueError("bad LFSR arguments")
    return out


def bits_packed(N, taps, nbits=None):
    nbits = (1 << N) - 1 if nbits is None else nbits
    out = np.empty((nbits + 7) // 8, dtype=np.uint8)
    t, n = _taps(taps)
    if lib().mls_oracle_bits_packed(N, t, n, out.ctypes.data, nbits):
        raise ValueError("bad LFSR arguments")
    return out


def bits_packed_word(N, taps, nbits=None):
    """bits_packed from the register held in one machine word (orders 31, 32 in reasonable time)"""
    nbits = (1 << N) - 1 if nbits is None else nbits
    out = np.empty((nbits + 7) // 8, dtype=np.uint8)
    t, n = _taps(taps)
    if lib().mls_oracle_bits_packed_word(N, t, n, out.ctypes.data, nbits):
        raise ValueError("bad LFSR arguments")
    return out


def sequence(N, taps, repeats=1):
    L = (1 << N) - 1
    out = np.empty((repeats * L, 1), dtype=np.float64)
    t, n = _taps(taps)
    if lib().mls_oracle_sequence(N, t, n, out.ctypes.data, repeats):
        raise ValueError("bad LFSR arguments")
    return out


def average(x, L):
    x = np.ascontiguousarray(x, dtype=np.float64).reshape(-1)
    avg = np.empty(L, dtype=np.float64)
    if lib().mls_oracle_average(x.ctypes.data, x.shape[0], 1, L, avg.ctypes.data):
        raise ValueError("length must be a multiple of L and > L")
    return avg


def xcorr_direct(avg, ref):
    a = np.ascontiguousarray(avg, dtype=np.float64).reshape(-1)
    r = np.ascontiguousarray(ref, dtype=np.float64).reshape(-1)
    h = np.empty_like(a)
    lib().mls_oracle_xcorr_direct(a.ctypes.data, r.ctypes.data, a.shape[0], h.ctypes.data)
    return h
