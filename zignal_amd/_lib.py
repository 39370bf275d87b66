"""ctypes binding of libzmls.so (C ABI: include/zmls.h).

There is no CPU fallback: if the shared library is missing this module raises at
import of the first symbol, and if no GPU is usable every compute entry point
returns ZMLS_E_NODEVICE, surfaced here as RuntimeError.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# $ZMLS_LIBRARY: another build of the same library (tuning: A/B runs of two builds on one box)
LIB_PATH = os.environ.get("ZMLS_LIBRARY") or os.path.join(_HERE, "lib", "libzmls.so")

ZMLS_F64, ZMLS_F32, ZMLS_I16, ZMLS_I32 = 0, 1, 2, 3
E_ARG, E_HIP, E_NODEVICE, E_WORKSPACE, E_UNSUPPORTED, E_COMM = -1, -2, -3, -4, -5, -6
COMM_ID_BYTES = 128

_DT = {np.dtype(np.float64): ZMLS_F64, np.dtype(np.float32): ZMLS_F32,
       np.dtype(np.int16): ZMLS_I16, np.dtype(np.int32): ZMLS_I32}
NP_OF = {ZMLS_F64: np.float64, ZMLS_F32: np.float32, ZMLS_I16: np.int16, ZMLS_I32: np.int32}

c_int, c_i64, c_u64, c_vp, c_sz = ctypes.c_int, ctypes.c_int64, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_size_t
_i32p = ctypes.POINTER(ctypes.c_int32)
_vpp = ctypes.POINTER(c_vp)

# name -> argtypes; every function returns int except where noted
SIGNATURES = {
    "zmls_version": [],
    "zmls_device_count": [ctypes.POINTER(c_int)],
    "zmls_plan_create": [_vpp, c_int, _i32p, c_int, c_int, c_int],
    "zmls_plan_destroy": [c_vp],
    "zmls_plan_reserve": [c_vp, c_i64],
    "zmls_plan_reserve_staging": [c_vp, c_i64],
    "zmls_plan_force_tile": [c_vp, c_int],
    "zmls_plan_set_resident": [c_vp, c_int, c_sz, c_int],
    "zmls_plan_force_pseudo_channels": [c_vp, c_int],
    "zmls_plan_last_launches": [c_vp, c_vp],
    "zmls_iir_set_segment": [c_vp, c_sz],
    "zmls_iir_set_fused_form": [c_vp, c_int],
    "zmls_iir_set_one_sweep": [c_vp, c_int, c_int],
    "zmls_iir_last_fused_form": [c_vp],
    "zmls_plan_set_input_scale_log2": [c_vp, c_int],
    "zmls_plan_info": [c_vp, ctypes.POINTER(c_i64)],
    "zmls_sequence_bits": [c_vp, c_vp, c_i64],
    "zmls_sequence_bits_from": [c_vp, c_u64, c_vp, c_i64],
    "zmls_plan_kind": [c_vp, ctypes.POINTER(c_int), ctypes.POINTER(c_int)],
    "zmls_sequence_f64": [c_vp, c_vp, c_i64],
    "zmls_sequence_device": [c_vp, c_vp, c_int, c_i64, c_i64, c_vp],
    "zmls_impulse_device": [c_vp, c_vp, c_int, c_i64, c_i64, c_i64, c_i64, c_vp, c_i64, c_i64, c_vp],
    "zmls_impulse_device_static": [c_vp, c_vp, c_int, c_i64, c_i64, c_i64, c_i64, c_vp, c_i64, c_i64, c_vp],
    "zmls_impulse_device_timed": [c_vp, c_vp, c_int, c_i64, c_i64, c_i64, c_i64, c_vp, c_i64, c_i64, c_vp,
                                  ctypes.POINTER(ctypes.c_float)],
    "zmls_impulse_host": [c_vp, c_vp, c_int, c_i64, c_i64, c_i64, c_i64, c_vp, c_i64, c_i64],
    "zmls_capture_alloc": [c_vp, c_int, c_i64, c_i64, c_int, _vpp, ctypes.POINTER(ctypes.c_float)],
    "zmls_capture_alloc_stats": [c_vp, c_int, c_i64, c_i64, c_int, _vpp, ctypes.POINTER(ctypes.c_float)],
    "zmls_capture_free": [c_vp, c_vp],
    "zmls_average_host": [c_vp, c_vp, c_int, c_i64, c_i64, c_i64, c_i64, c_vp],
    "zmls_comm_unique_id": [c_vp],
    "zmls_comm_library_info": [ctypes.c_char_p, c_sz, ctypes.POINTER(c_int)],
    "zmls_comm_create": [_vpp, c_vp, c_int, c_int, c_int],
    "zmls_comm_destroy": [c_vp],
    "zmls_comm_loopback_hub_create": [_vpp, c_int],
    "zmls_comm_loopback_hub_destroy": [c_vp],
    "zmls_comm_create_loopback": [_vpp, c_vp, c_int, c_int],
    "zmls_comm_info": [c_vp, ctypes.POINTER(c_i64)],
    "zmls_comm_reserve": [c_vp, c_vp, c_i64],
    "zmls_comm_reduce_scatter_device": [c_vp, c_vp, c_vp, c_i64, c_int, c_vp],
    "zmls_comm_all_gather_device": [c_vp, c_vp, c_vp, c_i64, c_int, c_vp],
    "zmls_period_sum_device": [c_vp, c_vp, c_int, c_i64, c_int, c_i64, c_i64, c_i64, c_i64, c_i64, c_i64, c_vp, c_vp],
    "zmls_impulse_sums_device": [c_vp, c_vp, c_i64, c_i64, c_i64, c_i64, c_vp, c_i64, c_i64, c_vp],
    "zmls_impulse_period_sharded_device": [c_vp, c_vp, c_vp, c_int, c_i64, c_int, c_i64, c_i64, c_i64, c_i64, c_vp, c_i64,
                                           c_i64, c_vp],
    "zmls_iir_create": [_vpp, c_vp, c_int, c_vp, c_int, c_int],
    "zmls_iir_destroy": [c_vp],
    "zmls_iir_reserve": [c_vp, c_i64, c_i64],
    "zmls_iir_device": [c_vp, c_vp, c_int, c_i64, c_i64, c_i64, c_i64, c_vp, c_i64, c_i64, c_vp],
    "zmls_iir_reserve_fused": [c_vp, c_vp, c_i64, c_i64],
    "zmls_impulse_filtered_device": [c_vp, c_vp, c_vp, c_int, c_i64, c_i64, c_i64, c_i64, c_vp, c_i64, c_i64, c_vp],
    "zmls_sources_reserve": [_vpp, c_int, c_vp, c_i64, c_i64],
    "zmls_impulse_sources_device": [_vpp, c_int, c_vp, c_vp, c_int, c_i64, c_i64, c_i64, c_i64, c_vp, c_i64, c_i64, c_i64, c_vp],
    "zmls_iir_host": [c_vp, c_vp, c_int, c_i64, c_i64, c_i64, c_i64, c_vp],
    "zmls_spectrum_create": [_vpp, c_i64, c_i64, c_int],
    "zmls_spectrum_destroy": [c_vp],
    "zmls_spectrum_info": [c_vp, ctypes.POINTER(c_i64)],
    "zmls_spectrum_device": [c_vp, c_vp, c_i64, c_i64, c_vp, c_int, c_vp, c_vp],
    "zmls_spectrum_host": [c_vp, c_vp, c_vp, c_int, c_vp],
    "zmls_synth_device": [c_vp, c_vp, c_int, c_i64, c_i64, c_i64, c_i64, c_i64, c_u64, ctypes.c_double, c_vp],
    "zmls_synth_system": [c_vp, c_i64, c_u64, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(c_i64)],
    "zmls_dev_alloc": [c_int, _vpp, c_sz],
    "zmls_dev_free": [c_int, c_vp],
    "zmls_dev_upload": [c_int, c_vp, c_vp, c_sz],
    "zmls_dev_download": [c_int, c_vp, c_vp, c_sz],
    "zmls_dev_memset": [c_int, c_vp, c_int, c_sz],
    "zmls_dev_copy": [c_int, c_vp, c_vp, c_sz, c_vp],
    "zmls_dev_sync": [c_int],
    "zmls_dev_mem_info": [c_int, ctypes.POINTER(c_sz), ctypes.POINTER(c_sz)],
    "zmls_stream_create": [c_int, _vpp],
    "zmls_stream_destroy": [c_int, c_vp],
    "zmls_stream_sync": [c_int, c_vp],
    "zmls_event_create": [c_int, _vpp],
    "zmls_event_destroy": [c_int, c_vp],
    "zmls_event_record": [c_int, c_vp, c_vp],
    "zmls_event_elapsed_ms": [c_int, c_vp, c_vp, ctypes.POINTER(ctypes.c_float)],
    "zmls_dev_download_channel": [c_int, c_vp, c_vp, c_int, c_i64, c_i64, c_i64, c_i64],
    "zmls_plan_export_table": [c_vp, c_int, c_vp],
    "zmls_plan_import_table": [c_vp, c_int, c_vp],
    "zmls_host_pass_split": [c_int, ctypes.POINTER(c_int)],
    "zmls_host_lfsr_state": [c_int, _i32p, c_int, c_u64, ctypes.POINTER(ctypes.c_uint32)],
    "zmls_plan_pipeline_status": [c_vp, ctypes.POINTER(ctypes.c_int32)],
    "zmls_plan_set_pipeline_delay": [c_vp, ctypes.c_uint32],
    "zmls_host_geometry_selfcheck": [c_int, c_int, c_int],
    "zmls_host_fft_selfcheck": [c_int, c_i64],
    "zmls_host_placement_sim": [ctypes.POINTER(ctypes.c_float), c_int, c_int, c_int, ctypes.c_float, ctypes.POINTER(c_i64)],
    "zmls_dev_stream_probe": [c_int, c_vp, c_sz, c_int, c_vp],
}

_lib = None


class ZmlsError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libzmls error %d: %s" % (code, msg))
        self.code = code
        self.msg = msg


def lib():
    """Load libzmls.so (once). Fails loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("%s is missing: build it with `make -C zignal_amd/csrc` (or "
                               "`python -c 'import __graft_entry__ as g; g.build()'`). "
                               "zignal_amd has no CPU fallback." % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, args in SIGNATURES.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = c_int
        L.zmls_last_error.argtypes = []
        L.zmls_last_error.restype = ctypes.c_char_p
        _lib = L
    return _lib


def last_error():
    return lib().zmls_last_error().decode("utf-8", "replace")


def check(rc):
    if rc < 0:
        raise ZmlsError(rc, last_error())
    return rc


def dtype_code(dt):
    try:
        return _DT[np.dtype(dt)]
    except KeyError:
        return None


def taps_array(taps):
    arr = (ctypes.c_int32 * len(taps))(*[int(t) for t in taps])
    return arr, len(taps)


def device_count():
    n = c_int(0)
    lib().zmls_device_count(ctypes.byref(n))
    return n.value


def default_device():
    """Device ordinal for new plans: $ZMLS_DEVICE, else $LOCAL_RANK, else 0."""
    for var in ("ZMLS_DEVICE", "LOCAL_RANK"):
        v = os.environ.get(var)
        if v not in (None, ""):
            return int(v)
    return 0
