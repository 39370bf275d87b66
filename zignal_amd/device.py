"""Device-resident plans and buffers: the additive, batched face of the library
(no counterpart in the reference, whose MLS class is single-channel and
host-only, mls.py:338-343). `zignal_amd.measure.mls` is built on top of this.
"""
import ctypes

import numpy as np

from . import _lib as L


class DeviceBuffer:
    """hipMalloc'ed bytes owned by Python."""

    def __init__(self, nbytes, device=0, _ptr=None):
        self.device = device
        self.nbytes = int(nbytes)
        if _ptr is None:
            p = ctypes.c_void_p()
            L.check(L.lib().zmls_dev_alloc(device, ctypes.byref(p), self.nbytes))
            _ptr = p.value
        self.ptr = _ptr

    def upload(self, arr):
        a = np.ascontiguousarray(arr)
        assert a.nbytes <= self.nbytes
        L.check(L.lib().zmls_dev_upload(self.device, self.ptr, a.ctypes.data, a.nbytes))
        return self

    def download(self, shape, dtype):
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        L.check(L.lib().zmls_dev_download(self.device, out.ctypes.data, self.ptr, out.nbytes))
        return out

    def fill(self, byte=0):
        L.check(L.lib().zmls_dev_memset(self.device, self.ptr, byte, self.nbytes))

    def free(self):
        if self.ptr:
            L.lib().zmls_dev_free(self.device, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass


class Plan:
    """One (N, taps, dtype, device) plan: packed bit sequence, gather/scatter
    tables and the FWHT workspace live on the GPU for its lifetime."""

    def __init__(self, N, taps, dtype="f64", device=None):
        self.device = L.default_device() if device is None else int(device)
        self.dtype_code = {"f64": L.ZMLS_F64, "f32": L.ZMLS_F32}[dtype]
        self.np_dtype = L.NP_OF[self.dtype_code]
        self.N = int(N)
        self.L = (1 << self.N) - 1
        self.taps = tuple(int(t) for t in taps)
        arr, n = L.taps_array(self.taps)
        h = ctypes.c_void_p()
        L.check(L.lib().zmls_plan_create(ctypes.byref(h), self.N, arr, n, self.dtype_code, self.device))
        self._h = h

    # -- housekeeping
    def close(self):
        if getattr(self, "_h", None):
            L.lib().zmls_plan_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def info(self):
        buf = (ctypes.c_int64 * 16)()
        L.check(L.lib().zmls_plan_info(self._h, buf))
        v = list(buf)
        return {"N": v[0], "L": v[1], "dtype": v[2], "device": v[3], "passes": v[4],
                "pass_bits": [b for b in v[5:8] if b], "workspace_bytes": v[8], "channels_per_slab": v[9],
                "channels_per_slab_big": v[10], "channels_per_slab_mid": v[11], "staging_bytes": v[12],
                "tall_passes": v[13], "tall_pass_bits": [b for b in v[14:16] if b]}

    def force_tile(self, cfg):
        """-1 auto, 0 small, 1 big, 2 mid, 3 tall tile (tests / tuning)"""
        L.check(L.lib().zmls_plan_force_tile(self._h, int(cfg)))

    def force_pseudo_channels(self, mode=-1):
        """pseudo-channel layout of few-channel batches (tests / tuning): -1 the library decides, 0 never, 1 whenever the order allows"""
        L.check(L.lib().zmls_plan_force_pseudo_channels(self._h, int(mode)))

    def last_launches(self):
        """pass-kernel launches of the most recent impulse call: [first (gather) pass, middle passes, last (scatter) pass]"""
        out = (ctypes.c_int32 * 3)()
        L.check(L.lib().zmls_plan_last_launches(self._h, out))
        return list(out)

    def set_resident(self, mode=-1, nbytes=0, flags=0):
        """cache-resident schedule (tests / tuning): -1 the library decides, 0 off, 1 on; chunk bytes (0 = built-in); flags"""
        L.check(L.lib().zmls_plan_set_resident(self._h, int(mode), int(nbytes), int(flags)))

    def reserve(self, n_ch, staging=False):
        """Workspace for n_ch channels per launch; staging=True also reserves the layout-adapter arrays for captures
        that are not aligned (frames, channels) data of the plan's dtype (zmls_plan_reserve_staging)."""
        L.check(L.lib().zmls_plan_reserve(self._h, int(n_ch)))
        if staging:
            L.check(L.lib().zmls_plan_reserve_staging(self._h, int(n_ch)))

    def alloc_capture(self, n_rows, n_ch, dtype_code=None, tries=0, stats=False):
        """DeviceBuffer for an (n_rows, n_ch) C-contiguous capture, placed where this plan's gather pass reads it
        fastest (zmls_capture_alloc). Returns (buffer, gather-pass milliseconds of the chosen placement or 0); with
        stats=True the second item is a dict instead: candidates_tried (0: nothing to choose, plainly allocated),
        plain_alloc_gather_ms (the first candidate = what one hipMalloc gives in this process), chosen_ms, model_ms
        (what the pass should take on a good placement), rounds, memory_full (an allocation failed during the search)."""
        code = self.dtype_code if dtype_code is None else dtype_code
        p, st = ctypes.c_void_p(), (ctypes.c_float * 8)()
        L.check(L.lib().zmls_capture_alloc_stats(self._h, code, int(n_rows), int(n_ch), int(tries), ctypes.byref(p), st))
        nbytes = int(n_rows) * int(n_ch) * np.dtype(L.NP_OF[code]).itemsize
        buf = DeviceBuffer(nbytes, self.device, _ptr=p.value)
        if not stats:
            return buf, st[2]
        return buf, {"candidates_tried": int(st[0]), "plain_alloc_gather_ms": st[1], "chosen_ms": st[2], "model_ms": st[3],
                     "rounds": int(st[4]), "memory_full": bool(st[5])}

    # -- sequence
    def bits_packed(self, nbits=None):
        nbits = self.L if nbits is None else int(nbits)
        out = np.zeros((nbits + 7) // 8, dtype=np.uint8)
        L.check(L.lib().zmls_sequence_bits(self._h, out.ctypes.data, nbits))
        return out

    def bits_from(self, start, nbits):
        """bits start..start+nbits-1 of the endless generator (the register keeps running; zmls_sequence_bits_from), one per element"""
        out = np.zeros((int(nbits) + 7) // 8, dtype=np.uint8)
        L.check(L.lib().zmls_sequence_bits_from(self._h, int(start), out.ctypes.data, int(nbits)))
        return np.unpackbits(out, bitorder="little")[:int(nbits)]

    def kind(self):
        """(maximal, transform): is the register maximum-length; do the impulse-response entry points work on this plan"""
        m, t = ctypes.c_int(), ctypes.c_int()
        L.check(L.lib().zmls_plan_kind(self._h, ctypes.byref(m), ctypes.byref(t)))
        return bool(m.value), bool(t.value)

    def sequence(self, repeats=1):
        out = np.empty((repeats * self.L, 1), dtype=np.float64)
        L.check(L.lib().zmls_sequence_f64(self._h, out.ctypes.data, int(repeats)))
        return out

    def tables(self):
        n = 1 << self.N
        q = np.empty(n, dtype=np.uint32)
        m = np.empty(n, dtype=np.uint32)
        L.check(L.lib().zmls_plan_export_table(self._h, 0, q.ctypes.data))
        L.check(L.lib().zmls_plan_export_table(self._h, 1, m.ctypes.data))
        return q, m

    # -- impulse responses
    def set_input_scale_log2(self, k):
        """Captures handed to the impulse entry points are raw integers with full scale 2^k: responses come out
        divided by 2^k (zmls_plan_set_input_scale_log2). 0: samples at face value."""
        L.check(L.lib().zmls_plan_set_input_scale_log2(self._h, int(k)))

    def impulse_host(self, x, pcm=False):
        """x: ndarray (R*L,) or (R*L, C), any strides; returns (L, C) of the plan dtype.
        pcm=True: x holds raw int16/int32 frames; the result is that of Audio.convert_to_float (audio.py:582)
        followed by the same call, with the conversion done on the device."""
        x2, code, n_rows, n_ch, rs, cs = self._as_capture(x)
        h = np.empty((self.L, n_ch), dtype=self.np_dtype)
        if pcm:
            if code not in (L.ZMLS_I16, L.ZMLS_I32):
                raise TypeError("raw PCM captures are int16 or int32, not %s" % x2.dtype)
            self.set_input_scale_log2(8 * x2.dtype.itemsize - 1)
        try:
            L.check(L.lib().zmls_impulse_host(self._h, x2.ctypes.data, code, n_rows, n_ch, rs, cs,
                                              h.ctypes.data, n_ch, 1))
        finally:
            if pcm:
                self.set_input_scale_log2(0)
        return h

    @staticmethod
    def _as_capture(x):
        """ndarray -> (array kept alive, dtype code, frames, channels, row stride, channel stride)"""
        x2 = x.reshape(-1, 1) if x.ndim == 1 else x
        if x2.ndim != 2:
            raise ValueError("captures are (frames,) or (frames, channels)")
        code = L.dtype_code(x2.dtype)
        if code is None:
            x2 = np.ascontiguousarray(x2, dtype=np.float64)
            code = L.ZMLS_F64
        es = x2.dtype.itemsize
        # ctypes.data of a positive-stride view is its lowest address; anything else is copied
        if x2.strides[0] % es or x2.strides[1] % es or x2.strides[0] <= 0 or x2.strides[1] <= 0:
            x2 = np.ascontiguousarray(x2)
        return x2, code, x2.shape[0], x2.shape[1], x2.strides[0] // es, x2.strides[1] // es

    def average_host(self, x):
        """Periods 1..R-1 of x averaged -> (L, C); float32 in float32, everything else in float64."""
        x2, code, n_rows, n_ch, rs, cs = self._as_capture(x)
        out = np.empty((self.L, n_ch), dtype=np.float32 if code == L.ZMLS_F32 else np.float64)
        L.check(L.lib().zmls_average_host(self._h, x2.ctypes.data, code, n_rows, n_ch, rs, cs, out.ctypes.data))
        return out

    def impulse_device(self, d_x, in_dtype, n_rows, n_ch, row_stride, ch_stride, d_h,
                       out_row_stride, out_ch_stride, stream=None):
        L.check(L.lib().zmls_impulse_device(self._h, d_x, in_dtype, n_rows, n_ch, row_stride, ch_stride,
                                            d_h, out_row_stride, out_ch_stride, stream))

    def impulse_filtered_device(self, iir, d_x, in_dtype, n_rows, n_ch, row_stride, ch_stride, d_h,
                                out_row_stride, out_ch_stride, stream=None):
        """get_impulse(lfilter(b, a, x)) in one: MLS.apply_deemphasis + get_impulse (mls.py:401-414) without the filtered
        capture ever being written (zmls_impulse_filtered_device). `iir.reserve_fused(plan, n_rows, n_ch)` first."""
        L.check(L.lib().zmls_impulse_filtered_device(self._h, iir._h, d_x, in_dtype, int(n_rows), int(n_ch), int(row_stride),
                                                     int(ch_stride), d_h, int(out_row_stride), int(out_ch_stride), stream))

    def impulse_device_timed(self, d_x, in_dtype, n_rows, n_ch, row_stride, ch_stride, d_h,
                             out_row_stride, out_ch_stride, stream=None):
        """Same launches with HIP events around every pass kernel; returns [ms per pass]."""
        ms = (ctypes.c_float * 4)()
        L.check(L.lib().zmls_impulse_device_timed(self._h, d_x, in_dtype, n_rows, n_ch, row_stride, ch_stride,
                                                  d_h, out_row_stride, out_ch_stride, stream, ms))
        return list(ms)[:self.info()["passes"]]

    def synth_device(self, d_x, dtype_code, repeats, n_ch, row_stride, ch_stride, seed=0x5A17,
                     sigma=1e-3, ch_first=0, stream=None):
        L.check(L.lib().zmls_synth_device(self._h, d_x, dtype_code, repeats, n_ch, ch_first, row_stride,
                                          ch_stride, seed, sigma, stream))

    def synth_system(self, channel, seed=0x5A17):
        taps = (ctypes.c_double * 16)()
        d = ctypes.c_int64()
        L.check(L.lib().zmls_synth_system(self._h, channel, seed, taps, ctypes.byref(d)))
        return np.array(list(taps)), d.value


class Sources:
    """Several MLS sources of one order played at once and recorded together: the multichannel measurement the reference
    describes and leaves to the caller (mls.py:338-343). `plans`: one Plan per source (same N, dtype, device), in source
    order. One call extracts the responses of every (source, recorded channel) pair; the periods are added up once for all
    sources when that moves fewer bytes, and the plans share the first plan's workspace (zmls_impulse_sources_device)."""

    def __init__(self, plans):
        self.plans = list(plans)
        if not self.plans:
            raise ValueError("at least one source plan")
        p0 = self.plans[0]
        self.device, self.np_dtype, self.L = p0.device, p0.np_dtype, p0.L
        self._arr = (ctypes.c_void_p * len(self.plans))(*[p._h for p in self.plans])

    def __len__(self):
        return len(self.plans)

    def reserve(self, n_rows, n_ch, iir=None):
        L.check(L.lib().zmls_sources_reserve(self._arr, len(self.plans), None if iir is None else iir._h, int(n_rows), int(n_ch)))

    def impulse_device(self, d_x, in_dtype, n_rows, n_ch, row_stride, ch_stride, d_h, source_stride, out_row_stride, out_ch_stride,
                       iir=None, stream=None):
        L.check(L.lib().zmls_impulse_sources_device(self._arr, len(self.plans), None if iir is None else iir._h, d_x, in_dtype,
                                                    int(n_rows), int(n_ch), int(row_stride), int(ch_stride), d_h, int(source_stride),
                                                    int(out_row_stride), int(out_ch_stride), stream))

    def impulse_host(self, x, iir=None):
        """x: (R*L,) or (R*L, C) capture -> (K, L, C) of the plans' dtype: [k, :, c] is what plan k's get_impulse returns
        for column c (after de-emphasis with `iir`, when given)."""
        x2, code, n_rows, n_ch, rs, cs = Plan._as_capture(np.asarray(x))
        K, es = len(self.plans), np.dtype(self.np_dtype).itemsize
        self.reserve(n_rows, n_ch, iir)
        d_x = DeviceBuffer(((n_rows - 1) * rs + (n_ch - 1) * cs + 1) * x2.itemsize, self.device)
        d_h = DeviceBuffer(K * self.L * n_ch * es, self.device)
        try:
            L.check(L.lib().zmls_dev_upload(self.device, d_x.ptr, x2.ctypes.data, d_x.nbytes))
            self.impulse_device(d_x.ptr, code, n_rows, n_ch, rs, cs, d_h.ptr, self.L * n_ch, n_ch, 1, iir=iir)
            sync(self.device)
            return d_h.download((K, self.L, n_ch), self.np_dtype)
        finally:
            d_x.free()
            d_h.free()


class IirFilter:
    """lfilter(B, A, x, axis=0) on the GPU (zmls_iir_*): what MLS.apply_emphasis / apply_deemphasis run
    (zignal/measure/mls.py:393-407, zignal/filters/linearfilter.py:62-63). Orders up to 8."""

    MAX_ORDER = 8

    def __init__(self, B, A, device=None):
        self.device = L.default_device() if device is None else int(device)
        b = np.ascontiguousarray(np.atleast_1d(B), dtype=np.float64)
        a = np.ascontiguousarray(np.atleast_1d(A), dtype=np.float64)
        h = ctypes.c_void_p()
        L.check(L.lib().zmls_iir_create(ctypes.byref(h), b.ctypes.data, b.size, a.ctypes.data, a.size, self.device))
        self._h = h

    def close(self):
        if getattr(self, "_h", None):
            L.lib().zmls_iir_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def reserve(self, n_rows, n_ch):
        L.check(L.lib().zmls_iir_reserve(self._h, int(n_rows), int(n_ch)))

    def set_segment(self, nbytes=None):
        """bytes of signal per segment of the time axis (tests / tuning): None the library's own size, 0 one segment"""
        L.check(L.lib().zmls_iir_set_segment(self._h, ctypes.c_size_t(-1).value if nbytes is None else int(nbytes)))

    def set_fused_form(self, form=-1):
        """which form Plan.impulse_filtered_device takes (tests / tuning): 0 one sweep in time order, 1 the filter after the
        transform, -1 the library's choice (zmls_iir_set_fused_form)"""
        L.check(L.lib().zmls_iir_set_fused_form(self._h, int(form)))

    def set_one_sweep(self, mode=-1, chunks_per_workgroup=0):
        """the one-sweep form of the filter for short-memory filters (tests / tuning): -1 the library's choice, 0 never, 1 forced
        with that many chunks per workgroup (zmls_iir_set_one_sweep)"""
        L.check(L.lib().zmls_iir_set_one_sweep(self._h, int(mode), int(chunks_per_workgroup)))

    def last_fused_form(self):
        return int(L.lib().zmls_iir_last_fused_form(self._h))

    def reserve_fused(self, plan, n_rows, n_ch):
        """scratch for Plan.impulse_filtered_device on captures of n_rows frames x n_ch channels (zmls_iir_reserve_fused)"""
        L.check(L.lib().zmls_iir_reserve_fused(self._h, plan._h, int(n_rows), int(n_ch)))

    def filter_host(self, x):
        """x: (n,) or (n, C) ndarray -> float64 array of the same shape, filtered along axis 0."""
        if np.size(x) == 0:
            return np.zeros(np.shape(x), dtype=np.float64)      # lfilter of nothing is nothing
        x2, code, n_rows, n_ch, rs, cs = Plan._as_capture(np.asarray(x))
        y = np.empty((n_rows, n_ch), dtype=np.float64)
        L.check(L.lib().zmls_iir_host(self._h, x2.ctypes.data, code, n_rows, n_ch, rs, cs, y.ctypes.data))
        return y.reshape(np.shape(x))

    def filter_device(self, d_x, in_dtype, n_rows, n_ch, row_stride, ch_stride, d_y, out_row_stride, out_ch_stride, stream=None):
        L.check(L.lib().zmls_iir_device(self._h, d_x, in_dtype, int(n_rows), int(n_ch), int(row_stride), int(ch_stride),
                                        d_y, int(out_row_stride), int(out_ch_stride), stream))


class Spectrum:
    """Audio.fft on the GPU (zignal/audio.py:678-704): dB magnitude of (n_samples, n_ch) float64 data, bins
    0..int(fftsize/2)-1, with the reference's Mersenne-prime size bump."""

    def __init__(self, n_samples, n_ch=1, device=None):
        self.device = L.default_device() if device is None else int(device)
        h = ctypes.c_void_p()
        L.check(L.lib().zmls_spectrum_create(ctypes.byref(h), int(n_samples), int(n_ch), self.device))
        self._h = h
        info = (ctypes.c_int64 * 4)()
        L.check(L.lib().zmls_spectrum_info(self._h, info))
        self.n_samples, self.n_ch, self.fftsize, self.n_bins = list(info)

    def close(self):
        if getattr(self, "_h", None):
            L.lib().zmls_spectrum_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def magnitude_db_host(self, samples, window=None, normalise=True):
        """samples: (n_samples, n_ch); window: n_samples values or None (rectangular) -> (bins, n_ch) dB"""
        x = np.ascontiguousarray(samples, dtype=np.float64).reshape(self.n_samples, self.n_ch)
        w = None if window is None else np.ascontiguousarray(window, dtype=np.float64).reshape(self.n_samples)
        mag = np.empty((self.n_bins, self.n_ch), dtype=np.float64)
        L.check(L.lib().zmls_spectrum_host(self._h, x.ctypes.data, None if w is None else w.ctypes.data, int(bool(normalise)),
                                           mag.ctypes.data))
        return mag

    def magnitude_db_device(self, d_h, row_stride, ch_stride, d_window, normalise, d_mag, stream=None):
        L.check(L.lib().zmls_spectrum_device(self._h, d_h, int(row_stride), int(ch_stride), d_window, int(bool(normalise)), d_mag, stream))


_spectra = {}                  # (n_samples, n_ch, device) -> Spectrum, most recently used last
_SPECTRA_KEPT = 4


def spectrum_for(n_samples, n_ch=1, device=None):
    """The Spectrum (device buffers + the chirp filter's spectrum, itself one forward transform) for this shape, kept between calls:
    setting one up for L = 2^N - 1 points costs more than the transform it then runs (Audio.fft used to build and destroy one per call). The few most recently
    used shapes stay; the one that falls out is closed."""
    dev = L.default_device() if device is None else int(device)
    key = (int(n_samples), int(n_ch), dev)
    sp = _spectra.pop(key, None)
    if sp is None or getattr(sp, "_h", None) is None:
        sp = Spectrum(key[0], key[1], device=dev)
    _spectra[key] = sp
    while len(_spectra) > _SPECTRA_KEPT:
        _spectra.pop(next(iter(_spectra))).close()
    return sp


def sync(device=0):
    L.check(L.lib().zmls_dev_sync(device))


def download_channel(device, d_src, dtype_code, n_rows, row_stride, ch_stride, channel):
    out = np.empty(n_rows, dtype=L.NP_OF[dtype_code])
    L.check(L.lib().zmls_dev_download_channel(device, out.ctypes.data, d_src, dtype_code, n_rows,
                                              row_stride, ch_stride, channel))
    return out


class Timer:
    """HIP events on one stream (bench.py)."""

    def __init__(self, device=0):
        self.device = device
        self.a, self.b = ctypes.c_void_p(), ctypes.c_void_p()
        L.check(L.lib().zmls_event_create(device, ctypes.byref(self.a)))
        L.check(L.lib().zmls_event_create(device, ctypes.byref(self.b)))

    def start(self, stream=None):
        L.check(L.lib().zmls_event_record(self.device, self.a, stream))

    def stop(self, stream=None):
        L.check(L.lib().zmls_event_record(self.device, self.b, stream))

    def elapsed_ms(self):
        ms = ctypes.c_float()
        L.check(L.lib().zmls_event_elapsed_ms(self.device, self.a, self.b, ctypes.byref(ms)))
        return ms.value
