/* zmls — MI355X (gfx950) implementation of zignal's MLS measurement hot path.
 *
 * C ABI of libzmls.so. The reference (zignal v0.6.0) is pure Python and has no
 * FFI boundary of its own; each entry point below names the reference method
 * whose work it takes over (file:line under the reference root). Plain pointers
 * and sizes only; no exceptions cross the boundary.
 *
 * Conventions
 *   - every function returns 0 on success or a negative zmls_status; the text of
 *     the last error on the calling thread is zmls_last_error().
 *   - "frames" are sample rows, "channels" are columns: element (j, c) of a
 *     capture lives at  base + (j*row_stride + c*ch_stride) * sizeof(element).
 *     (R*L, C) C-contiguous (zignal's Audio layout, audio.py:34) is
 *     row_stride=C, ch_stride=1; a stack of single-channel captures (C, R*L) is
 *     row_stride=1, ch_stride=R*L.
 *   - a plan owns its device buffers (bit sequence, permutation tables,
 *     workspace). Launch entry points (`*_device`) never allocate, free or
 *     synchronise; they enqueue on the caller's HIP stream.
 *   - one plan is not thread-safe; different plans are independent.
 *
 * Everything here replaces a piece of the reference (cited per entry point) or is needed to use those pieces on a
 * GPU. Hooks for the tests, the benchmark and tuning (synthetic captures, device memory for ctypes callers, per-pass
 * timing, plan internals) are declared in zmls_testing.h.
 */
#ifndef ZMLS_H
#define ZMLS_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    ZMLS_OK = 0,
    ZMLS_E_ARG = -1,        /* bad argument (Python shim raises AssertionError/ValueError as the reference does) */
    ZMLS_E_HIP = -2,        /* a HIP runtime call failed */
    ZMLS_E_NODEVICE = -3,   /* no usable GPU: there is NO CPU fallback */
    ZMLS_E_WORKSPACE = -4,  /* plan workspace too small for even one channel slab */
    ZMLS_E_UNSUPPORTED = -5,
    ZMLS_E_COMM = -6        /* librccl could not be loaded, or an RCCL call failed */
} zmls_status;

typedef enum { ZMLS_F64 = 0, ZMLS_F32 = 1, ZMLS_I16 = 2, ZMLS_I32 = 3 } zmls_dtype;

typedef struct zmls_plan zmls_plan;
typedef struct zmls_iir zmls_iir;     /* emphasis / de-emphasis filter, below */

const char *zmls_last_error(void);
int zmls_version(void);
int zmls_device_count(int *n);

/* ---- plan: replaces _MLS_base.__init__ (zignal/measure/mls.py:64-76) plus the
 * work the reference redoes on every call: the per-bit LFSR loop of
 * generator_bit/generator_chunk (mls.py:91-128) and the regeneration of the
 * stimulus inside xcorr_circular (mls.py:207).
 * taps: 1-indexed feedback delays as in zignal/measure/mlstaps.py:16-17. Like the reference, which checks nothing
 * (mls.py:64-68), every tap set its register can index is accepted: taps 1..N in any combination, maximum-length or
 * not, with or without N, and N+1..2N, which Python's negative deque index (mls.py:107) maps onto taps 1..N; anything
 * else is the reference's IndexError (ZMLS_E_ARG here). The SEQUENCE entry points work for all of them (and for
 * orders 31, 32). The impulse-response entry points need a maximum-length sequence of order <= 30 (zmls_plan_kind says
 * which a plan is) and return ZMLS_E_UNSUPPORTED otherwise.
 * dtype: arithmetic type of the impulse-response path, ZMLS_F64 (reference) or ZMLS_F32.
 * device: HIP device ordinal. 2 <= N <= 32. */
int zmls_plan_create(zmls_plan **out, int N, const int32_t *taps, int ntaps, int dtype, int device);
int zmls_plan_destroy(zmls_plan *p);
/* Make sure the workspace can hold `n_ch` channels in one go (allocates; call outside timed/captured regions). */
int zmls_plan_reserve(zmls_plan *p, int64_t n_ch);
/* Staging for captures / responses that are not 16-byte-aligned (frames, channels) arrays of the plan's dtype: a
 * stack of single-channel captures (channels, frames) - what a batch of the reference's own inputs looks like -,
 * int16/int32/float32 frames, odd channel counts. With staging for `n_ch` channels (2 arrays of L x n_ch elements;
 * fewer: the batch runs in channel chunks) the device entry points first add the periods up into a frame-major
 * array with coalesced reads and transpose the responses back at the end; without it they fall back to one
 * element per lane, an order of magnitude slower on large batches. zmls_impulse_host reserves it itself.
 * Allocates; call outside timed/captured regions. */
int zmls_plan_reserve_staging(zmls_plan *p, int64_t n_ch);
/* info[0]=N info[1]=L info[2]=dtype info[3]=device info[4]=number of FWHT passes
 * info[5..7]=bits per pass info[8]=workspace bytes info[9]/info[10]/info[11]=channels per slab of the small/big/mid tile
 * info[12]=bytes of each layout staging array
 * info[13]=passes of the tall-tile schedule (2 for N = 21, 22: wide aligned batches then take 11-bit passes, two
 * instead of info[4]'s three; 0 elsewhere) info[14..15]=its bits per pass */
int zmls_plan_info(const zmls_plan *p, int64_t info[16]);
/* Raw PCM captures: replaces Audio.convert_to_float (zignal/audio.py:565-589, samples / 2^(sourcebits-1)) on the
 * way from sndcard.PA.play_rec / a wav file (zignal/sndcard.py:296-318, audio.py:1108-1123) to get_impulse.
 * With k = sourcebits-1 set, the impulse entry points take the int16/int32 frames as they are (ZMLS_I16/ZMLS_I32,
 * a quarter / half of the float64 bytes over PCIe and out of HBM) and return the responses of the converted
 * capture: the division by 2^k is folded into the final 1/L scale, which is exact (a power of two commutes
 * with every rounding on the path). k = 0 (default): samples are taken at face value, as numpy does when the
 * reference is handed an integer array. State of the plan; 0 <= k <= 63. */
int zmls_plan_set_input_scale_log2(zmls_plan *p, int k);
/* *maximal: 1 when the register's period is 2^N - 1; *transform: 1 when the impulse-response entry points work on this
 * plan (maximal and N <= 30). Either pointer may be null. */
int zmls_plan_kind(const zmls_plan *p, int *maximal, int *transform);

/* ---- sequence: replaces generator_bit / generator_chunk / generator_samples /
 * get_full_sequence (mls.py:91-167). */
/* first `nbits` bits of the generator (it runs on across the period boundary,
 * mls.py:117-128), LSB-first packed, unused bits of the last byte zero;
 * nbits <= 2^N + 32. */
int zmls_sequence_bits(zmls_plan *p, uint8_t *out_packed, int64_t nbits);
/* `nbits` bits of the same endless generator from bit `start` on (generator_bit / generator_chunk keep their register
 * running for as long as they are asked, mls.py:104-128): the register's state at `start` is found by jump-ahead, the
 * bits are generated on the device. A maximum-length register repeats after L bits; any other runs on as the
 * reference's does. Same packing; nbits <= 2^31 per call. */
int zmls_sequence_bits_from(zmls_plan *p, uint64_t start, uint8_t *out_packed, int64_t nbits);
/* repeats*L samples, bit 0 -> +1.0, bit 1 -> -1.0 (mls.py:150), host buffer */
int zmls_sequence_f64(zmls_plan *p, double *out, int64_t repeats);
/* same on the device: d_out holds repeats*L elements of `dtype` (F64/F32), element stride `stride` */
int zmls_sequence_device(zmls_plan *p, void *d_out, int dtype, int64_t repeats, int64_t stride, void *hip_stream);

/* ---- impulse response: replaces trim_and_avgerage (mls.py:258-313) +
 * xcorr_circular (mls.py:169-256) = _MLS_base.get_impulse (mls.py:315-328),
 * for n_ch independent channels at once.
 *   x      : n_rows = R*L frames (R >= 2), n_ch channels, element type in_dtype
 *   h      : L frames, n_ch channels, element type = the plan's dtype
 * Period 0 is discarded, periods 1..R-1 are averaged, the average is circularly
 * cross-correlated with the plan's sequence and divided by L. */
int zmls_impulse_device(zmls_plan *p, const void *d_x, int in_dtype, int64_t n_rows, int64_t n_ch,
                        int64_t row_stride, int64_t ch_stride,
                        void *d_h, int64_t out_row_stride, int64_t out_ch_stride, void *hip_stream);
/* The same launch sequence for callers that cannot allocate. No counterpart in the reference: it is for code under
 * HIP graph capture, tight serving loops and harnesses that forbid hipMalloc. The intermediate between the FWHT
 * passes lives in a static device array of ZMLS_STATIC_WORKSPACE_BYTES inside the library (every GPU the library
 * runs on holds it until the process ends) instead of the workspace zmls_plan_reserve allocates, so a plan that
 * never reserved anything can be used; the batch runs in chunks of as many channel slabs as the array holds.
 * Orders above ZMLS_STATIC_MAX_ORDER (one slab of the intermediate no longer fits) are refused with
 * ZMLS_E_WORKSPACE - except a single channel, the reference's own call shape, which takes the sixteen times smaller
 * slabs of the pseudo-channel layout and is served up to order 24. Captures in any layout and input type are accepted; those that are not 16-byte-aligned
 * (frames, channels) arrays of the plan's dtype go one element per lane (the layout adapters need allocated
 * staging). Calls neither hipMalloc/hipFree nor synchronises. All calls on one GPU share the array: calls that
 * overlap on one GPU (two streams, two threads) race; calls in stream order do not. */
#define ZMLS_STATIC_WORKSPACE_BYTES (144u << 20)
#define ZMLS_STATIC_MAX_ORDER 21
int zmls_impulse_device_static(zmls_plan *p, const void *d_x, int in_dtype, int64_t n_rows, int64_t n_ch,
                               int64_t row_stride, int64_t ch_stride,
                               void *d_h, int64_t out_row_stride, int64_t out_ch_stride, void *hip_stream);
/* host buffers: uploads, runs, downloads, synchronous. h element type = plan dtype. */
int zmls_impulse_host(zmls_plan *p, const void *x, int in_dtype, int64_t n_rows, int64_t n_ch,
                      int64_t row_stride, int64_t ch_stride,
                      void *h, int64_t out_row_stride, int64_t out_ch_stride);

/* Device buffer for an (n_rows, n_ch) C-contiguous capture of `in_dtype`, placed for this plan: where a large
 * buffer lies in physical memory changes the speed of the gather pass by 5-25 % (measured; api.hip), so up to
 * `tries` candidate allocations (0 = default 4) are rated with the plan's own gather pass and the fastest is kept;
 * *probe_ms (may be null) receives its gather-pass time. Captures under 1 GiB are simply allocated. Free with
 * zmls_capture_free (it is plain device memory: hipFree works as well). Allocates, launches and synchronises:
 * plan-time only. No counterpart in the reference.
 * stats (may be null): [0] candidate allocations rated, [1] gather-pass ms of the first candidate (= what one plain
 * hipMalloc would have given in this process), [2] ms of the buffer returned, [3] ms the pass should take on a good
 * placement (the model the search stops at), [4] search rounds. */
int zmls_capture_alloc(zmls_plan *p, int in_dtype, int64_t n_rows, int64_t n_ch, int tries, void **d_x, float *probe_ms);
int zmls_capture_alloc_stats(zmls_plan *p, int in_dtype, int64_t n_rows, int64_t n_ch, int tries, void **d_x, float stats[8]);
int zmls_capture_free(zmls_plan *p, void *d_x);

/* ---- several sources at once: the multichannel measurement zignal's MLS class describes and leaves to the caller
 * (mls.py:338-343: one MLS per output channel, different taps, played together; "keep track of which channel is paired
 * with which MLS" when extracting). Replaces the loop of _MLS_base.get_impulse calls (mls.py:315-328), one per
 * (recorded channel, source) pair, that a zignal user writes today. `plans`: one plan per source, same order, dtype and
 * device, in the order the sources were appended. Response of source k, channel c, lag m:
 *     d_h[k*source_stride + m*out_row_stride + c*out_ch_stride]   (elements of the plans' dtype).
 * The periods are added up once for all sources when that moves fewer bytes than reading the capture once per source
 * (the average does not depend on the taps); `f` non-null: the capture is de-emphasised first (MLS.apply_deemphasis,
 * mls.py:401-407) through the single sweep of zmls_impulse_filtered_device, again once for all sources. The plans run
 * one after the other and share plan 0's workspace and staging: zmls_sources_reserve(plans, n, f, n_rows, n_ch)
 * allocates them (outside timed regions); the launch call never allocates or synchronises. */
int zmls_sources_reserve(zmls_plan *const *plans, int n_plans, zmls_iir *f, int64_t n_rows, int64_t n_ch);
int zmls_impulse_sources_device(zmls_plan *const *plans, int n_plans, zmls_iir *f, const void *d_x, int in_dtype,
                                int64_t n_rows, int64_t n_ch, int64_t row_stride, int64_t ch_stride,
                                void *d_h, int64_t source_stride, int64_t out_row_stride, int64_t out_ch_stride,
                                void *hip_stream);

/* ---- the averaging step alone: replaces trim_and_avgerage (mls.py:258-313) when it is called
 * on its own. out: (L, n_ch) C-contiguous HOST buffer, float32 when in_dtype is ZMLS_F32 (numpy
 * averages float32 in float32), float64 otherwise. */
int zmls_average_host(zmls_plan *p, const void *x, int in_dtype, int64_t n_rows, int64_t n_ch,
                      int64_t row_stride, int64_t ch_stride, void *out);

/* ---- several GPUs. Channels are independent (the reference handles one channel per call,
 * mls.py:338-343), so the normal multi-GPU use is one process per GPU, each calling
 * zmls_impulse_device on its own channels: no exchange, nothing below is needed.
 *
 * The entry points below are for captures whose PERIODS are spread over the GPUs (every rank holds some
 * of the repeats of every channel). trim_and_avgerage (mls.py:284-306) is a sum over periods then a
 * division, and xcorr_circular (mls.py:169-256) is linear, so: every rank adds up its own periods, the
 * partial sums are reduce-scattered over the channel axis with RCCL (rank q receives the complete sums
 * of channels [q*n_ch/P, (q+1)*n_ch/P)), and rank q divides by the total period count, transforms and
 * scatters those channels. librccl is loaded on first use. One process per GPU. */
typedef struct zmls_comm zmls_comm;
#define ZMLS_COMM_ID_BYTES 128
/* rank 0 makes the id (ncclGetUniqueId) and hands the bytes to the other ranks by whatever means the
 * host program has (torch.distributed/gloo broadcast, MPI, a file); every rank then creates its end. */
int zmls_comm_unique_id(uint8_t *id /* ZMLS_COMM_ID_BYTES */);
int zmls_comm_create(zmls_comm **out, const uint8_t *id, int rank, int n_ranks, int device);
int zmls_comm_destroy(zmls_comm *c);
/* which librccl answered (the file the loader resolved; the ROCm installation's own is tried first, by absolute path,
 * $ZMLS_RCCL_LIBRARY overrides) and its ncclGetVersion code. Either pointer may be null. Loads the library. */
int zmls_comm_library_info(char *path, size_t path_bytes, int *version);
/* info[0]=rank info[1]=ranks info[2]=device info[3]=chunk channels info[4]=staging bytes */
int zmls_comm_info(const zmls_comm *c, int64_t info[8]);
/* Staging for zmls_impulse_period_sharded_device: the exchange handles `chunk_ch` channels of every
 * rank's share at a time, double-buffered: 2*(P+1)*L*chunk_ch elements of the plan's dtype, plus the
 * plan's workspace for chunk_ch channels. Allocates; call outside timed regions. */
int zmls_comm_reserve(zmls_comm *c, zmls_plan *p, int64_t chunk_ch);
/* ncclReduceScatter(sum) / ncclAllGather on device buffers of ZMLS_F64 / ZMLS_F32, enqueued on hip_stream */
int zmls_comm_reduce_scatter_device(zmls_comm *c, const void *d_send, void *d_recv, int64_t recv_count, int dtype, void *hip_stream);
int zmls_comm_all_gather_device(zmls_comm *c, const void *d_send, void *d_recv, int64_t send_count, int dtype, void *hip_stream);

/* the three steps on their own (the fused entry point below chains them; tests use them one by one) */
/* out[q][j][c'] = sum over periods skip_periods..n_rows/L-1 of x[(period*L + j), q*block_ch_stride + ch_first + c'],
 * q < n_blocks, j < L, c' < block_ch; element type = the plan's dtype; this is the reduce-scatter send layout. */
int zmls_period_sum_device(zmls_plan *p, const void *d_x, int in_dtype, int64_t n_rows, int skip_periods,
                           int64_t row_stride, int64_t ch_stride, int64_t n_blocks, int64_t block_ch_stride,
                           int64_t ch_first, int64_t block_ch, void *d_out, void *hip_stream);
/* impulse responses from ONE period of sums (L frames, plan dtype): h = xcorr_circular(s / total_periods) */
int zmls_impulse_sums_device(zmls_plan *p, const void *d_s, int64_t total_periods, int64_t n_ch,
                             int64_t row_stride, int64_t ch_stride,
                             void *d_h, int64_t out_row_stride, int64_t out_ch_stride, void *hip_stream);
/* Fused, chunked and overlapped. Collective: every rank of the communicator calls it with the same N, taps,
 * dtype, n_ch and total_periods.
 *   d_x           this rank's periods: n_rows = R_local*L frames x n_ch channels (R_local may be 0)
 *   skip_periods  leading periods of d_x that are not averaged (1 on the rank that holds the period the
 *                 reference throws away, mls.py:284; else 0)
 *   total_periods averaged periods over ALL ranks (the divisor)
 *   d_h           L frames x n_ch/P channels: this rank's channel block [rank*n_ch/P, (rank+1)*n_ch/P)
 * Enqueues on hip_stream and on the communicator's own stream; never allocates or synchronises. */
int zmls_impulse_period_sharded_device(zmls_plan *p, zmls_comm *c, const void *d_x, int in_dtype, int64_t n_rows,
                                       int skip_periods, int64_t total_periods, int64_t n_ch,
                                       int64_t row_stride, int64_t ch_stride,
                                       void *d_h, int64_t out_row_stride, int64_t out_ch_stride, void *hip_stream);

/* ---- emphasis / de-emphasis: replaces Filter.filter_samples = scipy.signal.lfilter(B, A, x, axis=0)
 * (zignal/filters/linearfilter.py:62-63) as MLS.apply_emphasis / apply_deemphasis use it (mls.py:393-407) on the
 * whole stimulus / capture. Same transposed-direct-form-II arithmetic per sample, float64, zero initial state; the
 * recurrence is cut into blocks whose start states are put together with powers of the state matrix (iir.hip): two
 * sweeps over the signal; filters whose memory is short (A^n below 1e-22 within a few blocks) take one sweep with a short
 * warm-up per workgroup instead, when x and y do not overlap.
 * b, a: numerator / denominator as lfilter takes them (normalised by a[0] here); order max(nb, na)-1 <= 8. */
int zmls_iir_create(zmls_iir **out, const double *b, int nb, const double *a, int na, int device);
int zmls_iir_destroy(zmls_iir *f);
/* scratch for signals of up to n_rows frames x n_ch channels (allocates; outside timed regions) */
int zmls_iir_reserve(zmls_iir *f, int64_t n_rows, int64_t n_ch);
/* y (float64, any strides; may be x itself when x is float64 with the same strides) = lfilter(b, a, x) along frames.
 * Enqueues three to five kernels on hip_stream (two sweeps over the signal and the start-state assembly between them:
 * one kernel up to 16 chunks, three beyond); never allocates or synchronises. */
int zmls_iir_device(zmls_iir *f, const void *d_x, int in_dtype, int64_t n_rows, int64_t n_ch, int64_t row_stride,
                    int64_t ch_stride, double *d_y, int64_t out_row_stride, int64_t out_ch_stride, void *hip_stream);
/* De-emphasis and impulse response in one: replaces MLS.apply_deemphasis (mls.py:401-407) followed by get_impulse
 * (mls.py:409-414), the chain MLS_simple.get_impulse runs (mls.py:454-458): h = get_impulse(lfilter(b, a, x)) for n_ch
 * channels. Filter and average are linear, so the capture is swept ONCE (periods added up, the filter's state at
 * every period boundary taken beside it) and the filter itself runs over one period of sums from the summed states
 * (iir.hip): nothing of the capture's size is written. Where the filter's memory is a small part of a period (and the
 * plan is float64) the filter runs AFTER the transform instead - it commutes with the circular cross-correlation up to
 * a rank-S term that the last frames of every period determine - so the capture takes the ordinary gather pass
 * (impulse_filtered_post in iir.hip; same result to rounding). x as in zmls_impulse_device (n_rows = R*L frames, period 0
 * filtered but not averaged), h in the plan's dtype. zmls_iir_reserve_fused allocates (outside timed regions);
 * the launch call never allocates or synchronises. */
int zmls_iir_reserve_fused(zmls_iir *f, zmls_plan *p, int64_t n_rows, int64_t n_ch);
int zmls_impulse_filtered_device(zmls_plan *p, zmls_iir *f, const void *d_x, int in_dtype, int64_t n_rows, int64_t n_ch,
                                 int64_t row_stride, int64_t ch_stride,
                                 void *d_h, int64_t out_row_stride, int64_t out_ch_stride, void *hip_stream);
/* host buffers: y is (n_rows, n_ch) C-contiguous float64; uploads, filters, downloads, synchronous */
int zmls_iir_host(zmls_iir *f, const void *x, int in_dtype, int64_t n_rows, int64_t n_ch, int64_t row_stride,
                  int64_t ch_stride, double *y);

/* ---- magnitude response: replaces Audio.fft (zignal/audio.py:678-704) as MLS_simple.plot_fft / examples/ex_mls.py:80
 * call it on the recovered response: window, DFT of fftsize points (n_samples, or n_samples+1 for the Mersenne-prime
 * sizes of audio.py:682-685), optional /fftsize, 20 log10 |Y|, bins 0..int(fftsize/2)-1. The DFT is this library's own (csrc/fft.hip:
 * power-of-two sizes directly, every other size as a Bluestein convolution fused into four to six passes; no FFT library is
 * loaded); float64. */
typedef struct zmls_spectrum zmls_spectrum;
int zmls_spectrum_create(zmls_spectrum **out, int64_t n_samples, int64_t n_ch, int device);
int zmls_spectrum_destroy(zmls_spectrum *s);
/* info[0]=n_samples info[1]=n_ch info[2]=fftsize info[3]=bins returned */
int zmls_spectrum_info(const zmls_spectrum *s, int64_t info[4]);
/* d_h: (n_samples, n_ch) float64 with the given strides; d_window: n_samples float64 or null (rectangular);
 * d_mag: (bins, n_ch) C-contiguous float64, in dB. Enqueues on hip_stream; never allocates. */
int zmls_spectrum_device(zmls_spectrum *s, const double *d_h, int64_t row_stride, int64_t ch_stride,
                         const double *d_window, int normalise, double *d_mag, void *hip_stream);
/* host buffers: h (n_samples, n_ch) C-contiguous, window n_samples or null, mag (bins, n_ch); synchronous */
int zmls_spectrum_host(zmls_spectrum *s, const double *h, const double *window, int normalise, double *mag);

#ifdef __cplusplus
}
#endif
#endif /* ZMLS_H */
