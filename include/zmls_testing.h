/* zmls_testing.h — hooks of libzmls.so for its tests, its benchmark and tuning.
 *
 * None of these has a counterpart in the reference (zignal v0.6.0) and none is needed to use the library: a
 * program that binds libzmls in place of zignal/measure/mls.py needs include/zmls.h only. They exist because the
 * Python tests and bench.py reach the GPU through ctypes alone (no HIP binding of their own), and because the
 * benchmark has to make its captures on the device (BASELINE config 3 is 73 GB per GPU).
 */
#ifndef ZMLS_TESTING_H
#define ZMLS_TESTING_H

#include "zmls.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- tuning */
/* Tile configuration used by the impulse entry points: -1 = chosen per call (default), 0 = small tile
 * (64 KiB, 64-byte row segments, any layout), 1 = big tile (256 KiB, 256-byte row segments), 2 = mid tile
 * (128 KiB, 128-byte row segments), 3 = tall tile (256 KiB, 2^11 rows of 128 bytes; orders 21 and 22 only);
 * 1, 2 and 3 need contiguous, 16-byte aligned channels. Tuning bits: +256
 * switches the XCD-aware block mapping off, +1024 lays the workspace out row-major. */
int zmls_plan_force_tile(zmls_plan *p, int cfg);

/* ---- per-pass timing */
/* profiling variant (bench.py): same launches with a HIP event either side of every pass kernel on
 * `hip_stream`; synchronises and returns the milliseconds of pass i in pass_ms[i] (summed over channel
 * chunks). Creates and destroys its events: not for timed or captured regions. */
int zmls_impulse_device_timed(zmls_plan *p, const void *d_x, int in_dtype, int64_t n_rows, int64_t n_ch,
                              int64_t row_stride, int64_t ch_stride,
                              void *d_h, int64_t out_row_stride, int64_t out_ch_stride, void *hip_stream,
                              float pass_ms[4]);

/* ---- synthetic captures for benchmarking/tests (no reference counterpart; the
 * shape follows examples/ex_mls.py:16-28: stimulus through a short filter with a
 * delay, plus noise). Channel c: 16-tap decaying FIR g_c, delay d_c =
 * (c*9973) mod L, sigma*noise per sample, `repeats` periods, steady state.
 * Writes element type `dtype` with the given strides. */
int zmls_synth_device(zmls_plan *p, void *d_x, int dtype, int64_t repeats, int64_t n_ch, int64_t ch_first,
                      int64_t row_stride, int64_t ch_stride, uint64_t seed, double sigma, void *hip_stream);
/* the FIR the synthetic channel c uses (16 doubles) and its delay */
int zmls_synth_system(const zmls_plan *p, int64_t channel, uint64_t seed, double *taps16, int64_t *delay);

/* ---- device memory and streams for ctypes callers that have no HIP binding of
 * their own (bench.py, tests). Thin wrappers over hipMalloc, hipFree, hipMemcpy and the
 * hipStream / hipEvent calls, on the given device. */
int zmls_dev_alloc(int device, void **d_ptr, size_t bytes);
int zmls_dev_free(int device, void *d_ptr);
int zmls_dev_upload(int device, void *d_dst, const void *src, size_t bytes);
int zmls_dev_download(int device, void *dst, const void *d_src, size_t bytes);
int zmls_dev_memset(int device, void *d_ptr, int value, size_t bytes);
int zmls_dev_copy(int device, void *d_dst, const void *d_src, size_t bytes, void *hip_stream);   /* device to device, asynchronous */
int zmls_dev_sync(int device);
int zmls_dev_mem_info(int device, size_t *free_bytes, size_t *total_bytes);
int zmls_stream_create(int device, void **stream);
int zmls_stream_destroy(int device, void *stream);
int zmls_stream_sync(int device, void *stream);
int zmls_event_create(int device, void **event);
int zmls_event_destroy(int device, void *event);
int zmls_event_record(int device, void *event, void *stream);
int zmls_event_elapsed_ms(int device, void *start, void *stop, float *ms); /* synchronises on `stop` */
/* strided 2-D gather of one channel into a contiguous device/host buffer (tests: pull column c of a big capture) */
int zmls_dev_download_channel(int device, void *dst, const void *d_src, int dtype, int64_t n_rows,
                              int64_t row_stride, int64_t ch_stride, int64_t channel);

/* ---- plan internals for tests (tables in natural order, host buffers of 2^N uint32):
 * which=0: gather table   Q[d] = j with Ps[j] = d   (Q[0] = 0xFFFFFFFF)
 * which=1: scatter table  M[d] = m with Pl[m] = d   (M[0] = 0xFFFFFFFF)
 * Ps / Pl as defined in SURVEY.md §A.2. */
int zmls_plan_export_table(zmls_plan *p, int which, uint32_t *out);

/* tuning: overwrite the gather table (which = 0) with 2^N entries of the caller's, e.g. a sorted one, to time the gather
 * pass on rows that are not scattered (tools/gather_order.py). The plan's responses are meaningless afterwards. */
int zmls_plan_import_table(zmls_plan *p, int which, const uint32_t *in);

/* Tuning / tests: the cache-resident schedule of zmls_impulse_device (csrc/api.hip::run_passes). Batches of one
 * or two averaged periods run in chunks whose intermediate fits the 256 MiB Infinity Cache, with the intermediate's stores
 * and loads on the default cache policy. mode -1: the library decides (default), 0: never, 1: whatever the period count.
 * bytes: chunk size (0 = the built-in 256 MiB). flags: 1 = the in-place second pass of a three-pass order keeps its stores in
 * the cache too, 2 = non-temporal loads of the resident intermediate. The results do not depend on any of it (same
 * arithmetic, another launch order). No reference counterpart. */
int zmls_plan_set_resident(zmls_plan *p, int mode, size_t bytes, int flags);

/* Tuning / tests: the pseudo-channel layout that one channel from N = 18 and up to four at N = 20..22 take by default (csrc/fwht_kernels.inc,
 * one channel per 128-byte-row slab, the low 4 / 5 transform bits inside the row). mode -1: the library decides
 * (default), 0: never (such batches go through the ordinary channel slabs and layout adapters), 1: whenever the order allows
 * (N >= 6 for f64 plans, 7 for f32), whatever the channel count. No reference counterpart. */
int zmls_plan_force_pseudo_channels(zmls_plan *p, int mode);

/* Tuning / tests: the filter's sweeps run over segments of the time axis of about `bytes` bytes of signal (all channels), so
 * that the second sweep of a segment finds in the 256 MiB Infinity Cache what the first sweep read; the filter state is carried
 * from segment to segment on the device. 0: one segment; (size_t)-1: the library's own choice (default) - which is one segment: segments measured 14-28 % slower
 * (profiles/r03_iir_segments.txt), the option stays for tuning on other parts.
 * Same chunks and per-sample arithmetic either way; the chunk start states are put together segment by segment instead of in one
 * scan over the whole signal, so outputs of different segment sizes agree to rounding, and one size gives the same bits every time.
 * No reference counterpart. */
int zmls_iir_set_segment(zmls_iir *f, size_t bytes);

/* Tests / tuning: which form zmls_impulse_filtered_device takes (csrc/iir.hip). 0: one sweep over the capture in time order
 * (period sums + filter states), then the filter over the sums, then the transform. 1: the transform on the unfiltered capture,
 * then the filter over the responses (circular, plus a rank-S term for the filter's state at the period boundaries, which comes
 * from the last chunks of every period) - wherever it can run at all: float64 plans, filters whose memory fits a period.
 * -1 (default): form 1 when the filter's memory is at most a quarter of a period, else form 0. Both are regroupings of the same
 * linear map (mls.py:401-414: apply_deemphasis, then get_impulse): results agree to rounding. zmls_iir_last_fused_form: the form
 * the most recent call took (0 / 1; -1 for a null filter). No reference counterpart. */
int zmls_iir_set_fused_form(zmls_iir *f, int form);

/* Tests / tuning: the one-sweep form of the filter (csrc/iir.hip, iir_once_kernel). For filters whose memory is short (A^n below
 * 1e-22 after a few chunks) a workgroup walks `chunks_per_workgroup` chunks in time order carrying the state, after a warm-up over
 * the chunks before its first: the signal is read 1 + warm-up / chunks_per_workgroup times instead of twice. Never in place (the
 * library checks for overlapping buffers) and never with zmls_iir_set_segment. -1 (default): where the warm-up is at most a
 * quarter of a workgroup's work and 512 workgroups remain; 0: never; 1: wherever it can run, with the given chunk count.
 * Same per-sample arithmetic; start states equal to 1e-22 of the state's size. No reference counterpart. */
int zmls_iir_set_one_sweep(zmls_iir *f, int mode, int chunks_per_workgroup);
int zmls_iir_last_fused_form(const zmls_iir *f);

/* Profiling: how many kernels the most recent impulse call on this plan launched, by kind - out[0] first (gather) pass, out[1]
 * middle passes, out[2] last (scatter) pass. The cache-resident schedule runs a step as many chunk launches; a profile's
 * per-launch counters times these counts are the step's (tools/summarize_profile.py). No reference counterpart. */
int zmls_plan_last_launches(const zmls_plan *p, int32_t out[3]);
/* one-launch schedule of two-pass orders (tuning flag 8 of zmls_plan_set_resident): out[0] = tickets taken by the last launch (twice
 * the tiles of a pass when all is well), out[1] = 1 if a workgroup gave up waiting for its slab's gather tiles (its tile was skipped:
 * the responses are then incomplete). Synchronises the device. zmls_plan_set_pipeline_delay: tiles by which the scatter tiles trail the
 * gather tiles beyond one slab (0: default 768). No reference counterpart. */
int zmls_plan_pipeline_status(zmls_plan *p, int32_t out[2]);
int zmls_plan_set_pipeline_delay(zmls_plan *p, uint32_t extra_tiles);

/* ---- host-only helpers (no GPU needed; used by the CPU test tier) */
/* how a 2^N-point transform is split into LDS-resident passes: bits[i] for i < returned count */
int zmls_host_pass_split(int N, int bits[4]);
/* state of the N-bit Fibonacci register after `steps` steps from the reference's initial state, computed
 * with the same GF(2) jump-ahead matrices the device kernels use; bit i = sequence bit (steps + i). */
int zmls_host_lfsr_state(int N, const int32_t *taps, int ntaps, uint64_t steps, uint32_t *state);
/* enumerate the tile geometry of one pass and verify every element is covered exactly once per round and
 * every LDS slot / workspace row is in range; returns 0 if consistent. */
int zmls_host_geometry_selfcheck(int K, int tile_cfg, int first_pass);
/* the transform passes behind the magnitude response (csrc/fft_tile.h, fft.hip) run thread by thread on the CPU - the very
 * code of the kernels - and held to a direct DFT. mode 0: forward passes of two transforms of `size` = 2^m points read back
 * through the passes' own output order, then the inverse passes; mode 1: the whole Bluestein chain (chirp, forward, filter
 * spectrum, inverse, chirp) for any `size` >= 2. size <= 2^23. Returns 0 when within 1e-11 of the largest bin. */
int zmls_host_fft_selfcheck(int mode, int64_t size);

/* the capture placement search of zmls_capture_alloc (zmls_host.cpp::placement_search) on a made-up machine: the
 * i-th buffer ever allocated is gathered in ms_of_alloc[i % n_ms] milliseconds and at most max_live buffers fit in
 * memory at once. out: [0] index of the allocation kept (-1 none) [1] candidates rated [2] first candidate's
 * microseconds [3] the kept buffer's microseconds [4] rounds [5] 1 if an allocation failed [6] buffers still
 * allocated at the end [7] most buffers allocated at once. */
int zmls_host_placement_sim(const float *ms_of_alloc, int n_ms, int max_live, int tries, float model_ms, int64_t out[8]);

/* ---- loopback transport for the period-sharded entry points: P "ranks" that are THREADS of one process on one GPU
 * (own plan, communicator and stream each). RCCL refuses two ranks on one device, and a test box has one GPU: this is
 * how zmls_impulse_period_sharded_device runs with P > 1 there. The exchange keeps a collective's stream semantics
 * (comm.hip); every rank must call from its own thread with its own non-null stream, like ranks of a real job.
 * A hub holds 1..16 ranks; destroy the communicators before the hub. */
int zmls_comm_loopback_hub_create(void **hub, int n_ranks);
int zmls_comm_loopback_hub_destroy(void *hub);
int zmls_comm_create_loopback(zmls_comm **out, void *hub, int rank, int device);

/* ---- what this GPU streams, measured in the same run as the benchmark (bench.py's roofline calibration): hand-written
 * 16-byte-per-lane kernels over `bytes` bytes at d_buf (a multiple of 4096; the buffer's contents are destroyed by mode 1).
 * mode 0: read only (every lane adds up what it loads; nothing is written). mode 1: copy of the first half onto the
 * second half. Enqueues one kernel on hip_stream. */
int zmls_dev_stream_probe(int device, void *d_buf, size_t bytes, int mode, void *hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* ZMLS_TESTING_H */
