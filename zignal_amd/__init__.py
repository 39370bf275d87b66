"""zignal_amd — MI355X-native implementation of zignal's MLS measurement hot path.

Drop-in for the path `zignal.measure.mls` (zignal v0.6.0): same class and method
names, argument meaning and error behaviour; the sequence generator and the
impulse-response recovery run in hand-written HIP kernels (libzmls.so, C ABI in
include/zmls.h) reached through ctypes on numpy buffers. No PyTorch, no CPU
fallback: without the shared library or without a GPU the compute calls raise.

    from zignal_amd.measure.mls import MLS, MLS_simple, TAPS
"""
from .audio import Audio  # noqa: F401
from . import filters, measure  # noqa: F401

__version__ = "0.1.0"
__all__ = ["Audio", "filters", "measure"]
