// Emphasis / de-emphasis on the device: a linear recursive filter along time, SURVEY.md §8(f) row 1.
//
// Reference: MLS.apply_emphasis / apply_deemphasis (zignal/measure/mls.py:393-407) call
// Filter.filter_samples = scipy.signal.lfilter(B, A, x, axis=0) (zignal/filters/linearfilter.py:62-63) on the
// whole (R*L, 1) stimulus / capture right before the hot path. lfilter runs the transposed direct form II
//     y[n] = b0 x[n] + z0 ;  z_i = b_{i+1} x[n] + z_{i+1} - a_{i+1} y[n] ;  z_{S-1} = b_S x[n] - a_S y[n]
// one sample after the other. The recurrence is linear in the state: z' = A z + B x, so it is cut into
// sub-blocks that run in parallel from a zero state, and the true start state of every sub-block is put
// together from the zero-state end states with powers of A:
//   chunk kernel (pass 1): a workgroup loads a chunk (8192 samples: 256 sub-blocks of 32) into LDS with coalesced
//       reads, every thread runs its sub-block from zero, an in-LDS scan with A^32, A^64, ... gives the chunk's
//       zero-state end state;
//   combine: the chunks' start states, s_{c+1} = A^chunk s_c + end_c. Up to 16 chunks: one thread per channel walks
//       them. More: the same scan once more, one level up - a workgroup scans 256 chunks of one channel in LDS with
//       A^chunk, A^(2 chunk), ... (group scan kernel), one thread per channel walks the GROUPS (256 times fewer steps),
//       and a fix-up kernel adds (A^chunk)^t x (group start) to chunk t of every group but the first;
//   chunk kernel (pass 2): same load and scan, now with the chunk's start state, every thread reruns its sub-block
//       from its true start state and the chunk is written back coalesced.
// Same arithmetic per sample as lfilter (same operation order inside a sub-block); the start states differ from
// the sequential ones by rounding only (~1e-16 relative for the stable filters MLS accepts: it asserts minimum
// phase, mls.py:365-369). All arithmetic is float64, as lfilter's is for float64 coefficients.
// Streaming: the signal is read twice and written once; integer VALU and LDS are far from limits; bound: HBM.
// Filters with a short memory read it once (iir_once_kernel, round 3): a workgroup walks several chunks carrying the state.
//
// De-emphasis FUSED with the impulse response (zmls_impulse_filtered_device = MLS.apply_deemphasis + get_impulse,
// mls.py:401-414, as MLS_simple.get_impulse chains them, mls.py:454-458). Filtering R periods and then averaging R-1 of
// them reads and writes the whole capture (three sweeps); but filter and average are both linear, so with
//     z_r = filter state just before period r,   x_r = the samples of period r
// the filtered period is  y_r = ZSR(x_r) + ZIR(z_r)  (zero-state + zero-input response) and the sum over the averaged periods is
//     sum_r y_r = ZSR(sum_r x_r) + ZIR(sum_r z_r) = lfilter(S, zi = Z),   S = sum_{r>=1} x_r,  Z = sum_{r>=1} z_r.
// So ONE sweep over the capture in time order adds the periods up (S, one period long) and, beside it, takes the
// zero-state end state of every chunk of every period (period 0 included: it is thrown away by the average but its
// samples still ring in period 1); the end states fold into the period end states E_r = sum_q (A^chunk)^(last-q) e_rq,
// the chain z_1 = E_0, z_{r+1} = A^L z_r + E_r gives Z, and the filter then runs over ONE period of sums from the
// initial state Z. Exact (every step is the same linear map, regrouped), the capture is read once and nothing of its
// size is written: 19 instead of 54 ms for N=20, 512 channels x 17 periods.
// Round 3, the post-filter form (impulse_filtered_post, further down): where the filter's memory is a small part of a period the
// filter runs over the RESPONSES instead, and the capture takes the ordinary gather pass: 15.3 ms.
#include <algorithm>
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include "zmls_internal.h"

namespace zmls {

constexpr int kIirSub = 32;        // samples per thread
constexpr int kIirMaxS = 8;        // largest state size (filter order) built
constexpr int kIirLevels = 17;     // P[d] = A^(32 * 2^d), d = 0..16: runs (d < 8), chunks of 2^4 or 2^8 runs, groups of 2^8 chunks
constexpr int kIirGroup = 256;     // chunks per group of the second-level scan
constexpr int kIirSerialChunks = 16;   // up to here one thread per channel walks the chunks
constexpr int kIirPitch = kIirSub + 1;
constexpr size_t kIirHostKeepBytes = (size_t)256 << 20;   // zmls_iir_host keeps its device staging up to this size (in + out)
constexpr int kIirDotFrames = 512;   // frames of a frame-major chunk tile (256 / 16 runs of kIirSub): the dot-product form of the fused sweep
constexpr size_t kIirSegmentBytes = 0;   // signal per segment of the time axis (launch_iir_s); 0 = one segment: segments of 64-192 MiB measured 14-28 % SLOWER (profiles/r03_iir_segments.txt)

struct IirDev {
    double b[kIirMaxS + 1], a[kIirMaxS + 1];       // normalised: a[0] = 1
    double P[kIirLevels][kIirMaxS * kIirMaxS];     // row-major, row stride kIirMaxS
    // zero-state end state of a 512-frame chunk as a weighted sum of its samples: E = sum_i W[i] x[i], W[i] = A^(511-i) beta,
    // beta_k = b[k+1] - a[k+1] b[0] (the input vector of the transposed direct form II); [frame][state], state stride = the order S
    double W16[kIirDotFrames * kIirMaxS];
};

namespace {

__device__ __forceinline__ double iir_load(const void *base, int64_t idx, int dtype)
{
    switch (dtype) {
    case ZMLS_F64: return ((const double *)base)[idx];
    case ZMLS_F32: return (double)((const float *)base)[idx];
    case ZMLS_I16: return (double)((const int16_t *)base)[idx];
    default: return (double)((const int32_t *)base)[idx];
    }
}

template <int S>
__device__ __forceinline__ void matvec_acc(const double *__restrict__ M, const double (&u)[S], double (&v)[S])
{
#pragma unroll
    for (int i = 0; i < S; ++i) {
        double s = v[i];
#pragma unroll
        for (int j = 0; j < S; ++j) s = __builtin_fma(M[i * kIirMaxS + j], u[j], s);
        v[i] = s;
    }
}

// one sample of lfilter's transposed direct form II, in its operation order
template <int S>
__device__ __forceinline__ double df2t_step(const IirDev *__restrict__ f, double x, double (&z)[S])
{
#pragma clang fp contract(off)      // separate multiplies and adds, as lfilter's compiled C has them
    const double y = f->b[0] * x + z[0];
#pragma unroll
    for (int k = 0; k + 1 < S; ++k) z[k] = f->b[k + 1] * x + z[k + 1] - f->a[k + 1] * y;
    z[S - 1] = f->b[S] * x - f->a[S] * y;
    return y;
}

// CC = channels per chunk tile: 1 for time-contiguous signals (8192 frames x 1 channel), 16 for frame-major ones
// (512 frames x 16 channels); either way a tile is 8192 samples and a thread owns 32 consecutive frames of one channel.
// FINAL: 0 first sweep (chunk end states), 1 second sweep (outputs), 2 second sweep with the rank-S term of the post-filter form
template <int S, int CC, int FINAL>
__global__ void __launch_bounds__(256) iir_chunk_kernel(const IirDev *__restrict__ f, const void *__restrict__ x, int in_dtype, int64_t n_rows,
                                                        int64_t n_ch, int64_t rs, int64_t cs, uint32_t n_cgrp, double *__restrict__ zc,
                                                        const double *__restrict__ sin, double *__restrict__ y, int64_t ors, int64_t ocs,
                                                        int keep /* first sweep of a segment: the second sweep wants these lines in the Infinity Cache */,
                                                        const double *__restrict__ psi, const double *__restrict__ zeta /* FINAL: y[t][c] += sum_k zeta[c][k] psi[t][k], or null */)
{
    constexpr int NB = 256 / CC;
    constexpr int LOGNB = (CC == 1) ? 8 : 4;
    constexpr int TT = NB * kIirSub;
    static_assert(CC == 1 || CC == 16, "tile shapes built");
    __shared__ double tile[256 * kIirPitch];
    __shared__ double sc[256 * S];
    // frame-major tiles of the final sweep keep the tile's rows of psi here (TT x S, one coalesced read; every row is used by 16 channels)
    __shared__ double psl[(FINAL == 2 && CC == 16) ? TT * S : 1];
    const int tid = threadIdx.x;
    // frame-major tiles: the channel groups of one chunk are pieces of the same rows; renumbered so that they run on one XCD,
    // next to each other in time (workgroups are dealt to the 8 XCDs in turn)
    const uint32_t bid = (CC > 1 && gridDim.x % 8 == 0) ? (blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8 : blockIdx.x;
    const int64_t chunk = bid / n_cgrp;
    const int64_t t0 = chunk * TT, c0 = (int64_t)(bid % n_cgrp) * CC;

    constexpr int PER = TT * CC / 256;             // elements per thread in the coalesced load / store sweeps (32)
    if constexpr (FINAL == 2 && CC == 16) {
        {
            const int64_t have = (n_rows - t0 < TT ? n_rows - t0 : TT) * S;      // psi holds n_rows rows
#pragma unroll
            for (int i = 0; i < TT * S / 256; ++i) psl[tid + i * 256] = (tid + i * 256 < have) ? psi[t0 * S + tid + i * 256] : 0.0;
        }
    }
    if (in_dtype == ZMLS_F64 && t0 + TT <= n_rows && c0 + CC <= n_ch) {
        // interior tile of float64 data: all loads of a thread issued back to back (the kernel is bound by memory
        // latency otherwise: two workgroups per CU and one 8-byte load in flight per lane do not fill the HBM pipe)
        double r[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int e = tid + i * 256, tt = e / CC, cc = e % CC;
            const double *q = (const double *)x + (t0 + tt) * rs + (c0 + cc) * cs;
            r[i] = keep ? *q : __builtin_nontemporal_load(q);      // streaming, unless the other sweep of this segment comes right behind
        }
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int e = tid + i * 256, tt = e / CC, cc = e % CC;
            tile[((tt / kIirSub) * CC + cc) * kIirPitch + (tt % kIirSub)] = r[i];
        }
    } else {
        for (int e = tid; e < TT * CC; e += 256) {
            const int tt = e / CC, cc = e % CC;
            const int64_t t = t0 + tt, c = c0 + cc;
            const double v = (t < n_rows && c < n_ch) ? iir_load(x, t * rs + c * cs, in_dtype) : 0.0;
            tile[((tt / kIirSub) * CC + cc) * kIirPitch + (tt % kIirSub)] = v;
        }
    }
    __syncthreads();

    const int sb = tid / CC;                       // which 32-frame run of its channel this thread owns
    const int64_t c = c0 + tid % CC;
    double v[S];
#pragma unroll
    for (int k = 0; k < S; ++k) v[k] = 0.0;
#pragma unroll 4
    for (int i = 0; i < kIirSub; ++i) (void)df2t_step<S>(f, tile[tid * kIirPitch + i], v);

    // inclusive scan over the runs of one channel: v_k = sum_{i<=k} (A^32)^(k-i) end_i
#pragma unroll
    for (int d = 0; d < LOGNB; ++d) {
#pragma unroll
        for (int k = 0; k < S; ++k) sc[tid * S + k] = v[k];
        __syncthreads();
        if (sb >= (1 << d)) {
            double u[S];
#pragma unroll
            for (int k = 0; k < S; ++k) u[k] = sc[(tid - (CC << d)) * S + k];
            matvec_acc<S>(f->P[d], u, v);
        }
        __syncthreads();
    }
    if constexpr (!FINAL) {
        if (sb == NB - 1 && c < n_ch) {
#pragma unroll
            for (int k = 0; k < S; ++k) zc[(chunk * n_ch + c) * S + k] = v[k];
        }
        return;
    } else {
#pragma unroll
        for (int k = 0; k < S; ++k) sc[tid * S + k] = v[k];
        __syncthreads();
        double z[S], w[S];
#pragma unroll
        for (int k = 0; k < S; ++k) {
            z[k] = sb > 0 ? sc[(tid - CC) * S + k] : 0.0;
            w[k] = c < n_ch ? sin[(chunk * n_ch + c) * S + k] : 0.0;
        }
        // (A^32)^sb applied to the chunk's start state
#pragma unroll
        for (int d = 0; d < LOGNB; ++d) {
            if ((sb >> d) & 1) {
                double r[S];
#pragma unroll
                for (int k = 0; k < S; ++k) r[k] = 0.0;
                matvec_acc<S>(f->P[d], w, r);
#pragma unroll
                for (int k = 0; k < S; ++k) w[k] = r[k];
            }
        }
#pragma unroll
        for (int k = 0; k < S; ++k) z[k] += w[k];
#pragma unroll 4
        for (int i = 0; i < kIirSub; ++i) tile[tid * kIirPitch + i] = df2t_step<S>(f, tile[tid * kIirPitch + i], z);
        __syncthreads();
        // the post-filter form of the fused path (impulse_filtered_post) adds a rank-S term on the way out: the thread's channel in
        // the store sweep is fixed (256 is a multiple of CC), so its S coefficients sit in registers and psi's rows are shared reads
        double zt[S];
#pragma unroll
        for (int k = 0; k < S; ++k) zt[k] = (FINAL == 2 && c < n_ch) ? zeta[c * S + k] : 0.0;
        for (int e = tid; e < TT * CC; e += 256) {
            const int tt = e / CC, cc = e % CC;
            const int64_t t = t0 + tt, ch = c0 + cc;
            if (t < n_rows && ch < n_ch) {
                double v = tile[((tt / kIirSub) * CC + cc) * kIirPitch + (tt % kIirSub)];
                if constexpr (FINAL == 2) {
                    const double *ps = (CC == 16) ? psl + tt * S : psi + t * S;
#pragma unroll
                    for (int k = 0; k < S; ++k) v = __builtin_fma(zt[k], ps[k], v);
                }
                __builtin_nontemporal_store(v, y + t * ors + ch * ocs);
            }
        }
    }
}

template <int S>
__global__ void __launch_bounds__(64) iir_combine_kernel(const IirDev *__restrict__ f, int level, int64_t n_chunks, int64_t n_ch,
                                                         const double *__restrict__ zc, double *__restrict__ sin,
                                                         const double *__restrict__ zi /* initial state per channel, or null: at rest */)
{
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_ch) return;
    double s[S];
#pragma unroll
    for (int k = 0; k < S; ++k) s[k] = zi ? zi[c * S + k] : 0.0;        // lfilter starts at rest unless told otherwise
    // the end states do not depend on the walk: fetch a batch of them ahead of the dependent chain
    constexpr int AHEAD = 8;
    for (int64_t q0 = 0; q0 < n_chunks; q0 += AHEAD) {
        double e[AHEAD][S];
#pragma unroll
        for (int i = 0; i < AHEAD; ++i)
#pragma unroll
            for (int k = 0; k < S; ++k) e[i][k] = (q0 + i < n_chunks) ? zc[((q0 + i) * n_ch + c) * S + k] : 0.0;
#pragma unroll
        for (int i = 0; i < AHEAD; ++i) {
            if (q0 + i < n_chunks) {
#pragma unroll
                for (int k = 0; k < S; ++k) sin[((q0 + i) * n_ch + c) * S + k] = s[k];
                matvec_acc<S>(f->P[level], s, e[i]);          // e = A^chunk s + end_q
#pragma unroll
                for (int k = 0; k < S; ++k) s[k] = e[i][k];
            }
        }
    }
}

// second level of the start-state assembly: blockIdx = group * n_ch + channel; thread t holds chunk group*256 + t.
// Writes the chunks' start states as if the group started at rest (exclusive scan) and the group's own end state.
template <int S>
__global__ void __launch_bounds__(kIirGroup) iir_group_scan_kernel(const IirDev *__restrict__ f, int level, int64_t n_chunks, int64_t n_ch,
                                                                   const double *__restrict__ zc, double *__restrict__ sin, double *__restrict__ zg)
{
    __shared__ double sc[kIirGroup * S];
    const int tid = threadIdx.x;
    const int64_t c = blockIdx.x % n_ch, g = blockIdx.x / n_ch;
    const int64_t q = g * kIirGroup + tid;
    double v[S];
#pragma unroll
    for (int k = 0; k < S; ++k) v[k] = q < n_chunks ? zc[(q * n_ch + c) * S + k] : 0.0;
#pragma unroll
    for (int d = 0; d < 8; ++d) {
#pragma unroll
        for (int k = 0; k < S; ++k) sc[tid * S + k] = v[k];
        __syncthreads();
        if (tid >= (1 << d)) {
            double u[S];
#pragma unroll
            for (int k = 0; k < S; ++k) u[k] = sc[(tid - (1 << d)) * S + k];
            matvec_acc<S>(f->P[level + d], u, v);
        }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < S; ++k) sc[tid * S + k] = v[k];
    __syncthreads();
    if (q < n_chunks) {
#pragma unroll
        for (int k = 0; k < S; ++k) sin[(q * n_ch + c) * S + k] = tid ? sc[(tid - 1) * S + k] : 0.0;
    }
    if (tid == kIirGroup - 1) {
#pragma unroll
        for (int k = 0; k < S; ++k) zg[(g * n_ch + c) * S + k] = v[k];
    }
}

// chunk t of group g >= 1 starts from (A^chunk)^t x (start state of the group) on top of what the group scan wrote
template <int S>
__global__ void __launch_bounds__(256) iir_group_fix_kernel(const IirDev *__restrict__ f, int level, int64_t n_chunks, int64_t n_ch,
                                                            const double *__restrict__ sg, double *__restrict__ sin, int first_group)
{
    // group 0 starts at rest (first_group = 1) unless the filter was given an initial state
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x + (int64_t)first_group * kIirGroup * n_ch;
    if (idx >= n_chunks * n_ch) return;
    const int64_t q = idx / n_ch, c = idx % n_ch, g = q / kIirGroup;
    const int t = (int)(q % kIirGroup);
    double w[S];
#pragma unroll
    for (int k = 0; k < S; ++k) w[k] = sg[(g * n_ch + c) * S + k];
#pragma unroll
    for (int d = 0; d < 8; ++d) {
        if ((t >> d) & 1) {
            double r[S];
#pragma unroll
            for (int k = 0; k < S; ++k) r[k] = 0.0;
            matvec_acc<S>(f->P[level + d], w, r);
#pragma unroll
            for (int k = 0; k < S; ++k) w[k] = r[k];
        }
    }
#pragma unroll
    for (int k = 0; k < S; ++k) sin[idx * S + k] += w[k];
}

// the filter state after the last chunk of a segment = the start state of the next segment's first chunk:
// A^chunk x (start state of the last chunk) + its zero-state end state. One thread per channel.
template <int S>
__global__ void __launch_bounds__(64) iir_carry_kernel(const IirDev *__restrict__ f, int level, int64_t n_chunks, int64_t n_ch,
                                                       const double *__restrict__ zc, const double *__restrict__ sin, double *__restrict__ carry)
{
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_ch) return;
    double s[S], e[S];
#pragma unroll
    for (int k = 0; k < S; ++k) {
        s[k] = sin[((n_chunks - 1) * n_ch + c) * S + k];
        e[k] = zc[((n_chunks - 1) * n_ch + c) * S + k];
    }
    matvec_acc<S>(f->P[level], s, e);
#pragma unroll
    for (int k = 0; k < S; ++k) carry[c * S + k] = e[k];
}


// ---- one sweep for filters with a short memory (round 3). When A^n is below 1e-22 after nt chunks (tail_chunks), the filter state at
// a chunk boundary is known from the nt chunks before it alone: a workgroup walks M chunks in time order carrying the state in LDS,
// after a warm-up over the nt chunks before its first (read again, nothing written; the first workgroup of the signal starts from the
// initial state instead). The signal is read 1 + nt/M times and written once, instead of read twice. Per chunk the same steps as
// iir_chunk_kernel's second sweep: zero-state runs, scan, true start states, rerun, coalesced store. Not in place: the warm-up reads
// what the workgroup before may already have overwritten.
template <int S, int CC>
__device__ __forceinline__ void chunk_end_state(const IirDev *__restrict__ f, const double *tile, double *sc, int tid, double (&z)[S]);   // further down

template <int S, int CC, int MODE /* 1: outputs, 2: with the rank-S term of the post-filter form */>
__global__ void __launch_bounds__(256) iir_once_kernel(const IirDev *__restrict__ f, const void *__restrict__ x, int in_dtype, int64_t n_rows, int64_t n_ch,
                                                       int64_t rs, int64_t cs, uint32_t n_cgrp, int M, int nt, const double *__restrict__ zi,
                                                       double *__restrict__ y, int64_t ors, int64_t ocs, const double *__restrict__ psi,
                                                       const double *__restrict__ zeta)
{
    constexpr int NB = 256 / CC;
    constexpr int LOGNB = (CC == 1) ? 8 : 4;
    constexpr int TT = NB * kIirSub;
    constexpr int PER = TT * CC / 256;
    __shared__ double tile[256 * kIirPitch];
    __shared__ double sc[256 * S];
    __shared__ double psl[(MODE == 2 && CC == 16) ? TT * S : 1];
    __shared__ double carry[CC * S];
    const int tid = threadIdx.x;
    const uint32_t bid = (CC > 1 && gridDim.x % 8 == 0) ? (blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8 : blockIdx.x;
    const int64_t sup = bid / n_cgrp, c0 = (int64_t)(bid % n_cgrp) * CC;
    const int64_t n_chunks = (n_rows + TT - 1) / TT;
    const int64_t q_first = sup * M;
    const int64_t q_begin = q_first - nt > 0 ? q_first - nt : 0, q_end = q_first + M < n_chunks ? q_first + M : n_chunks;
    const int sb = tid / CC, ccl = tid % CC;
    const int64_t c = c0 + ccl;
    if (tid < CC * S) {
        const int64_t ch = c0 + tid / S;
        carry[tid] = (q_begin == 0 && zi && ch < n_ch) ? zi[ch * S + tid % S] : 0.0;
    }
    double zt[S];
#pragma unroll
    for (int k = 0; k < S; ++k) zt[k] = (MODE == 2 && c < n_ch) ? zeta[c * S + k] : 0.0;
    // interior chunks of float64 data: the next chunk's 32 loads per thread are issued before this chunk's arithmetic and fly through
    // it (the barriers below wait for LDS only), so the workgroup does not alternate between a memory phase and an LDS phase
    const bool f64_cols = in_dtype == ZMLS_F64 && c0 + CC <= n_ch;
    double r[PER];
    bool pre = q_begin < q_end && f64_cols && (q_begin + 1) * TT <= n_rows;
    if (pre) {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int e = tid + i * 256, tt = e / CC, cc = e % CC;
            r[i] = __builtin_nontemporal_load((const double *)x + (q_begin * TT + tt) * rs + (c0 + cc) * cs);
        }
    }
    for (int64_t q = q_begin; q < q_end; ++q) {
        const int64_t t0 = q * TT;
        const bool out = q >= q_first;
        if constexpr (MODE == 2 && CC == 16) {
            if (out) {
                const int64_t have = (n_rows - t0 < TT ? n_rows - t0 : TT) * S;
#pragma unroll
                for (int i = 0; i < TT * S / 256; ++i) psl[tid + i * 256] = (tid + i * 256 < have) ? psi[t0 * S + tid + i * 256] : 0.0;
            }
        }
        if (pre) {
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int e = tid + i * 256, tt = e / CC, cc = e % CC;
                tile[((tt / kIirSub) * CC + cc) * kIirPitch + (tt % kIirSub)] = r[i];
            }
        } else {
            for (int e = tid; e < TT * CC; e += 256) {
                const int tt = e / CC, cc = e % CC;
                const int64_t t = t0 + tt, ch = c0 + cc;
                tile[((tt / kIirSub) * CC + cc) * kIirPitch + (tt % kIirSub)] = (t < n_rows && ch < n_ch) ? iir_load(x, t * rs + ch * cs, in_dtype) : 0.0;
            }
        }
        pre = q + 1 < q_end && f64_cols && (q + 2) * TT <= n_rows;
        if (pre) {
#pragma unroll
            for (int i = 0; i < PER; ++i) {
                const int e = tid + i * 256, tt = e / CC, cc = e % CC;
                r[i] = __builtin_nontemporal_load((const double *)x + (t0 + TT + tt) * rs + (c0 + cc) * cs);
            }
        }
        __syncthreads();                          // the tile, psl and the carried state are in LDS
        double v[S], w[S];
        chunk_end_state<S, CC>(f, tile, sc, tid, v);       // inclusive scan of the runs' zero-state end states (ends with a barrier)
#pragma unroll
        for (int k = 0; k < S; ++k) {
            sc[tid * S + k] = v[k];
            w[k] = carry[ccl * S + k];
        }
        __syncthreads();
        double z[S];
#pragma unroll
        for (int k = 0; k < S; ++k) z[k] = sb > 0 ? sc[(tid - CC) * S + k] : 0.0;
#pragma unroll
        for (int d = 0; d < LOGNB; ++d) {
            if ((sb >> d) & 1) {
                double r[S];
#pragma unroll
                for (int k = 0; k < S; ++k) r[k] = 0.0;
                matvec_acc<S>(f->P[d], w, r);
#pragma unroll
                for (int k = 0; k < S; ++k) w[k] = r[k];
            }
        }
#pragma unroll
        for (int k = 0; k < S; ++k) z[k] += w[k];
#pragma unroll 4
        for (int i = 0; i < kIirSub; ++i) tile[tid * kIirPitch + i] = df2t_step<S>(f, tile[tid * kIirPitch + i], z);
        if (sb == NB - 1) {                       // the last run's state after its rerun is the state at the end of the chunk
#pragma unroll
            for (int k = 0; k < S; ++k) carry[ccl * S + k] = z[k];
        }
        __syncthreads();
        if (out) {
            for (int e = tid; e < TT * CC; e += 256) {
                const int tt = e / CC, cc = e % CC;
                const int64_t t = t0 + tt, ch = c0 + cc;
                if (t < n_rows && ch < n_ch) {
                    double o = tile[((tt / kIirSub) * CC + cc) * kIirPitch + (tt % kIirSub)];
                    if constexpr (MODE == 2) {
                        const double *ps = (CC == 16) ? psl + tt * S : psi + t * S;
#pragma unroll
                        for (int k = 0; k < S; ++k) o = __builtin_fma(zt[k], ps[k], o);
                    }
                    __builtin_nontemporal_store(o, y + t * ors + ch * ocs);
                }
            }
            __syncthreads();                      // before the next chunk's loads overwrite the tile
        }
    }
}

// ZMLS_IIR_ONCE=0 (tuning): always the two sweeps
static bool kOnceSweep()
{
    static const bool on = [] { const char *v = getenv("ZMLS_IIR_ONCE"); return !(v && *v == '0'); }();
    return on;
}

template <int S, int CC>
int launch_iir_s(const IirDev *d_f, const void *d_x, int in_dtype, int64_t n_rows, int64_t n_ch, int64_t rs, int64_t cs, double *d_zc, double *d_sin,
                 double *d_zg, double *d_sg, double *d_y, int64_t ors, int64_t ocs, const double *d_zi, double *d_carry, size_t segment_bytes,
                 hipStream_t st, const double *d_psi = nullptr, const double *d_zeta = nullptr, int nt_once = 0, int once_M = 0)
{
    constexpr int NB = 256 / CC, TT = NB * kIirSub, LOGNB = (CC == 1) ? 8 : 4;
    const uint64_t n_cgrp = (uint64_t)((n_ch + CC - 1) / CC);
    if (n_cgrp * (uint64_t)((n_rows + TT - 1) / TT) == 0 || n_cgrp * (uint64_t)((n_rows + TT - 1) / TT) > 0x7FFFFFFFull) { set_error("filter grid out of range"); return ZMLS_E_ARG; }
    if (nt_once > 0 && !(segment_bytes && d_carry) && kOnceSweep()) {
        // one sweep (iir_once_kernel): not in place, and only when a workgroup's warm-up is a small part of its work and there are
        // still workgroups enough to fill the GPU twice over
        const size_t is0 = dtype_size(in_dtype);
        const char *xb = (const char *)d_x, *xe = xb + (size_t)((n_rows - 1) * rs + (n_ch - 1) * cs + 1) * is0;
        const char *yb = (const char *)d_y, *ye = yb + (size_t)((n_rows - 1) * ors + (n_ch - 1) * ocs + 1) * 8;
        const uint64_t n_chunks = (uint64_t)((n_rows + TT - 1) / TT);
        int M = once_M > 0 ? once_M : 0;              // forced (tests): any number of chunks per workgroup, any grid
        for (int mult : {8, 4}) {
            if (M) break;
            const uint64_t m = (uint64_t)std::max(8, mult * nt_once);
            if (n_cgrp * ((n_chunks + m - 1) / m) >= 512) { M = (int)m; break; }       // 512: two workgroups on every CU at once
        }
        if (M && (xe <= yb || ye <= xb)) {
            const uint64_t blocks = n_cgrp * ((n_chunks + M - 1) / M);
            if (d_psi)
                hipLaunchKernelGGL((iir_once_kernel<S, CC, 2>), dim3((uint32_t)blocks), dim3(256), 0, st, d_f, d_x, in_dtype, n_rows, n_ch, rs, cs, (uint32_t)n_cgrp, M,
                                   nt_once, d_zi, d_y, ors, ocs, d_psi, d_zeta);
            else
                hipLaunchKernelGGL((iir_once_kernel<S, CC, 1>), dim3((uint32_t)blocks), dim3(256), 0, st, d_f, d_x, in_dtype, n_rows, n_ch, rs, cs, (uint32_t)n_cgrp, M,
                                   nt_once, d_zi, d_y, ors, ocs, (const double *)nullptr, (const double *)nullptr);
            ZHIP(hipGetLastError());
            return ZMLS_OK;
        }
    }
    // Segments of the time axis (round 3, a tuning option that is OFF by default): the signal is swept twice, and a segment
    // small enough to stay in the 256 MiB Infinity Cache between its two sweeps could be read from HBM once. Whole chunks per
    // segment, all channels; the state after a segment's last chunk (iir_carry_kernel) is the next segment's initial state.
    // Measured (profiles/r03_iir_segments.txt, (2L, 512) frames x channels): 4.5 ms in one segment, 5.75 ms in segments of
    // 64-128 MiB, 5.1 ms at 192 MiB, with streaming or cached first-sweep loads alike: the cache does not hand the second
    // sweep its lines faster than HBM streams them, and the segments' launch ramps cost more than is saved.
    int64_t seg_rows = n_rows;
    if (segment_bytes && d_carry) {
        const int64_t per_row = n_ch * 8;
        seg_rows = std::max<int64_t>(1, (int64_t)(segment_bytes / (size_t)per_row) / TT) * TT;
        // the grid of a segment should still fill the GPU a few times over
        const int64_t min_chunks = (int64_t)((2048 + n_cgrp - 1) / n_cgrp);
        if (seg_rows < min_chunks * TT) seg_rows = min_chunks * TT;
    }
    const size_t is = dtype_size(in_dtype);
    // ZMLS_IIR_KEEP=0/1 (tuning): the first sweep's loads streaming / cached, whatever the segment count
    static const int keep_env = [] { const char *v = getenv("ZMLS_IIR_KEEP"); return (v && *v) ? atoi(v) : -1; }();
    const int keep_first = keep_env >= 0 ? keep_env : (seg_rows < n_rows ? 1 : 0);
    for (int64_t r0 = 0; r0 < n_rows; r0 += seg_rows) {
        const int64_t rows = std::min<int64_t>(seg_rows, n_rows - r0);
        const void *x_seg = (const char *)d_x + (size_t)(r0 * rs) * is;
        double *y_seg = d_y + r0 * ors;
        const double *zi_seg = r0 == 0 ? d_zi : d_carry;
        const bool more = r0 + rows < n_rows;
        const uint64_t n_chunks = (uint64_t)((rows + TT - 1) / TT);
        const uint64_t blocks = n_cgrp * n_chunks;
        hipLaunchKernelGGL((iir_chunk_kernel<S, CC, 0>), dim3((uint32_t)blocks), dim3(256), 0, st, d_f, x_seg, in_dtype, rows, n_ch, rs, cs,
                           (uint32_t)n_cgrp, d_zc, (const double *)nullptr, (double *)nullptr, ors, ocs, keep_first, (const double *)nullptr,
                           (const double *)nullptr);
        if (n_chunks <= (uint64_t)kIirSerialChunks) {
            hipLaunchKernelGGL(iir_combine_kernel<S>, dim3((uint32_t)((n_ch + 63) / 64)), dim3(64), 0, st, d_f, LOGNB, (int64_t)n_chunks, n_ch,
                               (const double *)d_zc, d_sin, zi_seg);
        } else {
            const uint64_t n_groups = (n_chunks + kIirGroup - 1) / kIirGroup;
            if (n_groups * (uint64_t)n_ch > 0x7FFFFFFFull) { set_error("filter grid out of range"); return ZMLS_E_ARG; }
            hipLaunchKernelGGL(iir_group_scan_kernel<S>, dim3((uint32_t)(n_groups * n_ch)), dim3(kIirGroup), 0, st, d_f, LOGNB, (int64_t)n_chunks, n_ch,
                               (const double *)d_zc, d_sin, d_zg);
            if (n_groups > 1 || zi_seg) {
                hipLaunchKernelGGL(iir_combine_kernel<S>, dim3((uint32_t)((n_ch + 63) / 64)), dim3(64), 0, st, d_f, LOGNB + 8, (int64_t)n_groups, n_ch,
                                   (const double *)d_zg, d_sg, zi_seg);
                const int first_group = zi_seg ? 0 : 1;
                const uint64_t rest = (n_chunks - (uint64_t)first_group * kIirGroup) * (uint64_t)n_ch;
                hipLaunchKernelGGL(iir_group_fix_kernel<S>, dim3((uint32_t)((rest + 255) / 256)), dim3(256), 0, st, d_f, LOGNB, (int64_t)n_chunks, n_ch,
                                   (const double *)d_sg, d_sin, first_group);
            }
        }
        // (before the second sweep: in-place filtering overwrites nothing the carry needs, but the next segment's
        // combine must not start before this one's start states have been read - stream order takes care of both)
        if (more)
            hipLaunchKernelGGL(iir_carry_kernel<S>, dim3((uint32_t)((n_ch + 63) / 64)), dim3(64), 0, st, d_f, LOGNB, (int64_t)n_chunks, n_ch,
                               (const double *)d_zc, (const double *)d_sin, d_carry);
        if (d_psi)
            hipLaunchKernelGGL((iir_chunk_kernel<S, CC, 2>), dim3((uint32_t)blocks), dim3(256), 0, st, d_f, x_seg, in_dtype, rows, n_ch, rs, cs,
                               (uint32_t)n_cgrp, (double *)nullptr, (const double *)d_sin, y_seg, ors, ocs, 0, d_psi + r0 * S, d_zeta);
        else
            hipLaunchKernelGGL((iir_chunk_kernel<S, CC, 1>), dim3((uint32_t)blocks), dim3(256), 0, st, d_f, x_seg, in_dtype, rows, n_ch, rs, cs,
                               (uint32_t)n_cgrp, (double *)nullptr, (const double *)d_sin, y_seg, ors, ocs, 0, (const double *)nullptr, (const double *)nullptr);
    }
    ZHIP(hipGetLastError());
    return ZMLS_OK;
}

template <int S>
int launch_iir_cc(bool time_contiguous, const IirDev *d_f, const void *d_x, int in_dtype, int64_t n_rows, int64_t n_ch, int64_t rs, int64_t cs,
                  double *d_zc, double *d_sin, double *d_zg, double *d_sg, double *d_y, int64_t ors, int64_t ocs, const double *d_zi, double *d_carry,
                  size_t segment_bytes, hipStream_t st, const double *d_psi = nullptr, const double *d_zeta = nullptr, const int *nt_once = nullptr, int once_M = 0)
{
    // stacked single-channel signals (time-contiguous, several channels) are not cut: a segment of all channels is no contiguous
    // piece of memory there and the channels are far apart anyway
    if (time_contiguous) return launch_iir_s<S, 1>(d_f, d_x, in_dtype, n_rows, n_ch, rs, cs, d_zc, d_sin, d_zg, d_sg, d_y, ors, ocs, d_zi, d_carry,
                                                   n_ch == 1 ? segment_bytes : 0, st, d_psi, d_zeta, nt_once ? nt_once[1] : 0, once_M);
    return launch_iir_s<S, 16>(d_f, d_x, in_dtype, n_rows, n_ch, rs, cs, d_zc, d_sin, d_zg, d_sg, d_y, ors, ocs, d_zi, d_carry, segment_bytes, st, d_psi,
                               d_zeta, nt_once ? nt_once[0] : 0, once_M);
}

// ---- the fused path: one sweep over the capture (period sums + chunk end states), then two small kernels
// A chunk is TT frames of CC channels, as in iir_chunk_kernel, but the chunks of a period are aligned to its END: chunk q
// covers frames [L - (n_chunks - q) TT, L - (n_chunks - q - 1) TT); the first one starts before the period does and
// reads zeros there, which leaves a filter at rest at rest - so every end state is exact and every chunk boundary is
// one A^TT away from the next.
// zero-state end state of the chunk in `tile` (run-major, as iir_chunk_kernel lays it out): every thread runs its 32-frame
// run from rest, then the in-LDS scan with A^32, A^64, ...; the threads of the last run hold the chunk's end state in z
template <int S, int CC>
__device__ __forceinline__ void chunk_end_state(const IirDev *__restrict__ f, const double *tile, double *sc, int tid, double (&z)[S])
{
    constexpr int LOGNB = (CC == 1) ? 8 : 4;
    const int sb = tid / CC;
#pragma unroll
    for (int k = 0; k < S; ++k) z[k] = 0.0;
#pragma unroll 4
    for (int i = 0; i < kIirSub; ++i) (void)df2t_step<S>(f, tile[tid * kIirPitch + i], z);
#pragma unroll
    for (int d = 0; d < LOGNB; ++d) {
#pragma unroll
        for (int k = 0; k < S; ++k) sc[tid * S + k] = z[k];
        __syncthreads();
        if (sb >= (1 << d)) {
            double u[S];
#pragma unroll
            for (int k = 0; k < S; ++k) u[k] = sc[(tid - (CC << d)) * S + k];
            matvec_acc<S>(f->P[d], u, z);
        }
        __syncthreads();
    }
}

template <int S, int CC>
__global__ void __launch_bounds__(256, (S <= 5 ? 2 : 1)) iir_period_kernel(const IirDev *__restrict__ f, const void *__restrict__ x, int in_dtype, int64_t L, int R,
                                                         int64_t n_ch, int64_t rs, int64_t cs, uint32_t n_cgrp, int64_t n_chunks,
                                                         double *__restrict__ e, double *__restrict__ sum, int64_t srs, int64_t scs,
                                                         int vec /* 16-byte accesses are aligned on both sides (frame-major tiles only) */)
{
    constexpr int NB = 256 / CC;
    constexpr int TT = NB * kIirSub;
    constexpr int PER = TT * CC / 256;
    __shared__ double tile[256 * kIirPitch];
    __shared__ double sc[256 * S];
    const int tid = threadIdx.x;
    // workgroups are dealt to the 8 XCDs in turn: renumber them so that the channel groups of one chunk - the pieces of the
    // same capture rows - run on one XCD, next to each other in time
    const uint32_t bid = (gridDim.x % 8 == 0) ? (blockIdx.x % 8) * (gridDim.x / 8) + blockIdx.x / 8 : blockIdx.x;
    const int64_t chunk = bid / n_cgrp;
    const int64_t j0 = L - (n_chunks - chunk) * TT, c0 = (int64_t)(bid % n_cgrp) * CC;
    const bool interior = in_dtype == ZMLS_F64 && j0 >= 0 && c0 + CC <= n_ch;
    const int sb = tid / CC;
    const int64_t c = c0 + tid % CC;
    double z[S];

    if constexpr (CC == 16) {
        if (vec == 2 && interior) {
            // The same tiles with the chunk's end state taken as a WEIGHTED SUM of its samples (IirDev::W16) instead of running the
            // recurrence: the state after the chunk is linear in the chunk, E = sum_i A^(TT-1-i) beta x[i], so every thread multiplies
            // the 16 vectors it has just loaded by their frames' weights, the 8 lanes of a wave that share a channel pair add up
            // (three xor shuffles), and the four waves' partial sums wait in LDS until a batch of periods is full. No tile in LDS, no
            // 32-step recurrence, no nine barriers per period: between two flushes the sweep is a plain stream of 16-byte loads.
            // (Another order of the same additions as the recurrence: equal to rounding, 1e-16 of the state's size.)
            typedef double D2 __attribute__((ext_vector_type(2)));
            constexpr int NV = PER / 2;
            constexpr int kPartAt = kIirDotFrames * kIirMaxS;                          // the partial sums' place behind the weights
            constexpr int PB = (256 * kIirPitch - kPartAt) / (64 * S);                // periods per flush
            static_assert(TT == kIirDotFrames && PB >= 1, "dot-product form: tile shape");
            double *Wl = tile, *part = tile + kPartAt;
            for (int i = tid; i < TT * S; i += 256) Wl[i] = f->W16[i];
            __syncthreads();
            const int fr = tid >> 3, cpi = tid & 7, cp = cpi * 2, wave = tid >> 6;
            const int64_t vstep = 32 * rs;
            D2 acc2[NV];
#pragma unroll
            for (int i = 0; i < NV; ++i) acc2[i] = (D2)0.0;
            int slot = 0, r_first = 0;
            // (the weights of a thread's 16 frames kept in registers - 64 of them at order 2 - spill; they are read from LDS per period)
            // (issuing period r + 1's loads ahead of the lane sums of period r measured 15.9 against 12.4 ms for this sweep)
            for (int r = 0; r < R; ++r) {
                const double *q = (const double *)x + ((int64_t)r * L + j0 + fr) * rs + c0 + cp;
                D2 v2[NV];
#pragma unroll
                for (int i = 0; i < NV; ++i, q += vstep) v2[i] = __builtin_nontemporal_load((const D2 *)q);
                if (r >= 1) {
#pragma unroll
                    for (int i = 0; i < NV; ++i) acc2[i] += v2[i];       // period order, as numpy adds the rows (mls.py:298-306)
                }
                if (r + 1 == R) break;                                   // the last period rings into nothing that is kept
                D2 p[S];
#pragma unroll
                for (int k = 0; k < S; ++k) p[k] = (D2)0.0;
#pragma unroll
                for (int i = 0; i < NV; ++i)
#pragma unroll
                    for (int k = 0; k < S; ++k) p[k] += Wl[(fr + 32 * i) * S + k] * v2[i];
#pragma unroll
                for (int k = 0; k < S; ++k)
#pragma unroll
                    for (int ch = 0; ch < 2; ++ch) {
                        double t = p[k][ch];
                        t += __shfl_xor(t, 8, 64);
                        t += __shfl_xor(t, 16, 64);
                        t += __shfl_xor(t, 32, 64);
                        if ((tid & 63) < 8) part[((slot * 4 + wave) * 8 + cpi) * (2 * S) + k * 2 + ch] = t;
                    }
                ++slot;
                if (slot == PB || r + 2 == R) {
                    __syncthreads();
                    for (int o = tid; o < slot * 16 * S; o += 256) {
                        const int pr = o / (16 * S), rem = o - pr * (16 * S), cc = rem / S, k = rem - cc * S;
                        const int ci = cc >> 1, ch = cc & 1;
                        double t = 0.0;
#pragma unroll
                        for (int w = 0; w < 4; ++w) t += part[((pr * 4 + w) * 8 + ci) * (2 * S) + k * 2 + ch];
                        e[(((int64_t)(r_first + pr) * n_chunks + chunk) * n_ch + c0 + cc) * S + k] = t;
                    }
                    __syncthreads();
                    r_first += slot;
                    slot = 0;
                }
            }
            double *o = sum + (j0 + fr) * srs + c0 + cp;
#pragma unroll
            for (int i = 0; i < NV; ++i, o += 32 * srs) __builtin_nontemporal_store(acc2[i], (D2 *)o);
            return;
        }
        if (vec && interior) {
            // interior tiles of aligned float64 frames: 16 loads of 16 bytes (two adjacent channels) in flight per lane.
            // Vector i of this thread is frame tid/8 + 32 i of the tile, channels 2 (tid%8) and 2 (tid%8) + 1: run i, offset tid/8.
            typedef double D2 __attribute__((ext_vector_type(2)));
            constexpr int NV = PER / 2;
            const int fr = tid >> 3, cp = (tid & 7) * 2;
            const int64_t vstep = 32 * rs;
            D2 acc2[NV];
#pragma unroll
            for (int i = 0; i < NV; ++i) acc2[i] = (D2)0.0;
            for (int r = 0; r < R; ++r) {
                const double *q = (const double *)x + ((int64_t)r * L + j0 + fr) * rs + c0 + cp;
                D2 v2[NV];
#pragma unroll
                for (int i = 0; i < NV; ++i, q += vstep) v2[i] = __builtin_nontemporal_load((const D2 *)q);
                if (r >= 1) {
#pragma unroll
                    for (int i = 0; i < NV; ++i) acc2[i] += v2[i];       // period order, as numpy adds the rows (mls.py:298-306)
                }
                if (r + 1 == R) break;                                   // the last period rings into nothing that is kept
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    tile[(i * CC + cp) * kIirPitch + fr] = v2[i][0];
                    tile[(i * CC + cp + 1) * kIirPitch + fr] = v2[i][1];
                }
                __syncthreads();
                chunk_end_state<S, CC>(f, tile, sc, tid, z);
                if (sb == NB - 1) {
#pragma unroll
                    for (int k = 0; k < S; ++k) e[(((int64_t)r * n_chunks + chunk) * n_ch + c) * S + k] = z[k];
                }
            }
            double *o = sum + (j0 + fr) * srs + c0 + cp;
#pragma unroll
            for (int i = 0; i < NV; ++i, o += 32 * srs) __builtin_nontemporal_store(acc2[i], (D2 *)o);
            return;
        }
    }

    // element i of this thread in the coalesced sweeps: frame tt0 + i * FSTEP of the tile, channel cc0
    constexpr int FSTEP = 256 / CC;
    constexpr int HALF = PER / 2;                       // loads in flight per lane (two halves keep the kernel at two workgroups per CU)
    const int tt0 = tid / CC, cc0 = tid % CC;
    const int64_t step = (int64_t)FSTEP * rs;
    double acc[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) acc[i] = 0.0;
    for (int r = 0; r < R; ++r) {
        const bool keep_state = r + 1 < R;              // the last period rings into nothing that is kept
#pragma unroll
        for (int h0 = 0; h0 < PER; h0 += HALF) {
            double v[HALF];
            if (interior) {
                const double *q = (const double *)x + ((int64_t)r * L + j0 + tt0 + (int64_t)h0 * FSTEP) * rs + (c0 + cc0) * cs;
#pragma unroll
                for (int i = 0; i < HALF; ++i, q += step) v[i] = __builtin_nontemporal_load(q);
            } else {
#pragma unroll
                for (int i = 0; i < HALF; ++i) {
                    const int64_t j = j0 + tt0 + (int64_t)(h0 + i) * FSTEP, ch = c0 + cc0;
                    v[i] = (j >= 0 && ch < n_ch) ? iir_load(x, ((int64_t)r * L + j) * rs + ch * cs, in_dtype) : 0.0;
                }
            }
            if (r >= 1) {
#pragma unroll
                for (int i = 0; i < HALF; ++i) acc[h0 + i] += v[i];        // period order, as numpy adds the rows (mls.py:298-306)
            }
            if (keep_state) {
#pragma unroll
                for (int i = 0; i < HALF; ++i) {
                    const int tt = tt0 + (h0 + i) * FSTEP;
                    tile[((tt / kIirSub) * CC + cc0) * kIirPitch + (tt % kIirSub)] = v[i];
                }
            }
        }
        if (!keep_state) break;
        __syncthreads();
        chunk_end_state<S, CC>(f, tile, sc, tid, z);
        if (sb == NB - 1 && c < n_ch) {
#pragma unroll
            for (int k = 0; k < S; ++k) e[(((int64_t)r * n_chunks + chunk) * n_ch + c) * S + k] = z[k];
        }
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int64_t j = j0 + tt0 + (int64_t)i * FSTEP, ch = c0 + cc0;
        if (j >= 0 && ch < n_ch) sum[j * srs + ch * scs] = acc[i];
    }
}

// E[r][c] = zero-state end state of period r = sum_q (A^TT)^(n_chunks-1-q) e[r][q][c]; one thread per (r, c)
template <int S>
__global__ void __launch_bounds__(64) iir_period_end_kernel(const IirDev *__restrict__ f, int level, int64_t n_chunks, int64_t n_ch, int64_t n_rc,
                                                            const double *__restrict__ e, double *__restrict__ E)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_rc) return;
    const int64_t r = idx / n_ch, c = idx % n_ch;
    const double *er = e + (r * n_chunks * n_ch + c) * S;
    double s[S];
#pragma unroll
    for (int k = 0; k < S; ++k) s[k] = 0.0;
    constexpr int AHEAD = 8;
    for (int64_t q0 = 0; q0 < n_chunks; q0 += AHEAD) {
        double t[AHEAD][S];
#pragma unroll
        for (int i = 0; i < AHEAD; ++i)
#pragma unroll
            for (int k = 0; k < S; ++k) t[i][k] = (q0 + i < n_chunks) ? er[(q0 + i) * n_ch * S + k] : 0.0;
#pragma unroll
        for (int i = 0; i < AHEAD; ++i) {
            if (q0 + i < n_chunks) {
                matvec_acc<S>(f->P[level], s, t[i]);          // t = A^TT s + e_q
#pragma unroll
                for (int k = 0; k < S; ++k) s[k] = t[i][k];
            }
        }
    }
#pragma unroll
    for (int k = 0; k < S; ++k) E[idx * S + k] = s[k];
}

// z_1 = E_0, z_{r+1} = A^L z_r + E_r; zi = sum of the states at the start of the averaged periods 1..R-1
template <int S>
__global__ void __launch_bounds__(64) iir_period_chain_kernel(const double *__restrict__ phi /* A^L, row stride kIirMaxS */, int R, int64_t n_ch,
                                                              const double *__restrict__ E, double *__restrict__ zi)
{
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_ch) return;
    double z[S], Z[S];
#pragma unroll
    for (int k = 0; k < S; ++k) z[k] = Z[k] = 0.0;
    for (int r = 0; r + 1 < R; ++r) {
        double t[S];
#pragma unroll
        for (int k = 0; k < S; ++k) t[k] = E[((int64_t)r * n_ch + c) * S + k];
        matvec_acc<S>(phi, z, t);
#pragma unroll
        for (int k = 0; k < S; ++k) { z[k] = t[k]; Z[k] += t[k]; }
    }
#pragma unroll
    for (int k = 0; k < S; ++k) zi[c * S + k] = Z[k];
}

// ---- the post-filter form of the fused path (impulse_filtered_post below): only the END of every period is read in time order
// zero-state end state of tail chunk q (of nt, aligned to the period's end like iir_period_kernel's) of period r; r = 0..R-1.
template <int S, int CC>
__global__ void __launch_bounds__(256) iir_tail_kernel(const IirDev *__restrict__ f, const void *__restrict__ x, int in_dtype, int64_t L, int64_t n_ch,
                                                       int64_t rs, int64_t cs, uint32_t n_cgrp, int nt, double *__restrict__ e)
{
    constexpr int NB = 256 / CC;
    constexpr int TT = NB * kIirSub;
    constexpr int PER = TT * CC / 256;
    __shared__ double tile[256 * kIirPitch];
    __shared__ double sc[256 * S];
    const int tid = threadIdx.x;
    const uint32_t cg = blockIdx.x % n_cgrp, rq = blockIdx.x / n_cgrp;
    const int q = (int)(rq % (uint32_t)nt);
    const int64_t r = rq / (uint32_t)nt;
    const int64_t j0 = L - (int64_t)(nt - q) * TT, c0 = (int64_t)cg * CC;
    double v[PER];
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int el = tid + i * 256, tt = el / CC, cc = el % CC;
        const int64_t j = j0 + tt, ch = c0 + cc;
        v[i] = (j >= 0 && ch < n_ch) ? iir_load(x, (r * L + j) * rs + ch * cs, in_dtype) : 0.0;
    }
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        const int el = tid + i * 256, tt = el / CC, cc = el % CC;
        tile[((tt / kIirSub) * CC + cc) * kIirPitch + (tt % kIirSub)] = v[i];
    }
    __syncthreads();
    double z[S];
    chunk_end_state<S, CC>(f, tile, sc, tid, z);
    const int64_t c = c0 + tid % CC;
    if (tid / CC == NB - 1 && c < n_ch) {
#pragma unroll
        for (int k = 0; k < S; ++k) e[((r * nt + q) * n_ch + c) * S + k] = z[k];
    }
}

// One thread per channel, from the period end states E[0..R-1] of the capture and E[R] of the unfiltered response:
//   z_0 = 0, z_{r+1} = A^L z_r + E_r;  Zm = mean_{r=1..R-1} z_r (the filter's state at the start of the averaged periods),
//   Em = mean_{r=1..R-1} E_r (the zero-state end state of the averaged period, by linearity),
//   zeta = scale (Zm - (I - A^L)^-1 Em)   what the true start state has over the periodic steady state of the averaged period,
//   zi   = (I - A^L)^-1 E_h               the start state that makes the filter over the response circular.
template <int S>
__global__ void __launch_bounds__(64) iir_post_state_kernel(const double *__restrict__ phi /* A^L */, const double *__restrict__ minv /* (I - A^L)^-1 */,
                                                            int R, int64_t n_ch, const double *__restrict__ E, double scale,
                                                            double *__restrict__ zeta, double *__restrict__ zi)
{
    const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_ch) return;
    double z[S], Z[S], Es[S];
#pragma unroll
    for (int k = 0; k < S; ++k) z[k] = Z[k] = Es[k] = 0.0;
    for (int r = 0; r < R; ++r) {
        double t[S];
#pragma unroll
        for (int k = 0; k < S; ++k) t[k] = E[((int64_t)r * n_ch + c) * S + k];
        if (r >= 1) {
#pragma unroll
            for (int k = 0; k < S; ++k) { Z[k] += z[k]; Es[k] += t[k]; }
        }
        matvec_acc<S>(phi, z, t);
#pragma unroll
        for (int k = 0; k < S; ++k) z[k] = t[k];
    }
    const double inv = 1.0 / (double)(R - 1);
    double em[S], zs[S], eh[S], z0[S];
#pragma unroll
    for (int k = 0; k < S; ++k) { em[k] = Es[k] * inv; zs[k] = 0.0; z0[k] = 0.0; eh[k] = E[((int64_t)R * n_ch + c) * S + k]; }
    matvec_acc<S>(minv, em, zs);
    matvec_acc<S>(minv, eh, z0);
#pragma unroll
    for (int k = 0; k < S; ++k) {
        zeta[c * S + k] = scale * (Z[k] * inv - zs[k]);
        zi[c * S + k] = z0[k];
    }
}

template <int S>
int launch_tail_s(bool tc, const IirDev *d_f, const void *d_x, int in_dtype, int64_t L, int64_t R, int64_t n_ch, int64_t rs, int64_t cs, int nt, double *d_e,
                  double *d_E, hipStream_t st)
{
    const int CC = tc ? 1 : 16;
    const uint64_t n_cgrp = (uint64_t)((n_ch + CC - 1) / CC);
    const uint64_t blocks = n_cgrp * (uint64_t)nt * (uint64_t)R, n_rc = (uint64_t)R * (uint64_t)n_ch;
    if (blocks == 0 || blocks > 0x7FFFFFFFull || n_rc > 0x7FFFFFFFull * 64) { set_error("filter grid out of range"); return ZMLS_E_ARG; }
    if (tc)
        hipLaunchKernelGGL((iir_tail_kernel<S, 1>), dim3((uint32_t)blocks), dim3(256), 0, st, d_f, d_x, in_dtype, L, n_ch, rs, cs, (uint32_t)n_cgrp, nt, d_e);
    else
        hipLaunchKernelGGL((iir_tail_kernel<S, 16>), dim3((uint32_t)blocks), dim3(256), 0, st, d_f, d_x, in_dtype, L, n_ch, rs, cs, (uint32_t)n_cgrp, nt, d_e);
    hipLaunchKernelGGL(iir_period_end_kernel<S>, dim3((uint32_t)((n_rc + 63) / 64)), dim3(64), 0, st, d_f, tc ? 8 : 4, (int64_t)nt, n_ch, (int64_t)n_rc,
                       (const double *)d_e, d_E);
    ZHIP(hipGetLastError());
    return ZMLS_OK;
}

// ZMLS_IIR_DOT=0 (tuning): chunk end states of the fused sweep by the recurrence in LDS instead of the weighted sums
static bool kDotSweep()
{
    static const bool on = [] { const char *v = getenv("ZMLS_IIR_DOT"); return !(v && *v == '0'); }();
    return on;
}

template <int S, int CC>
int launch_period_s(const IirDev *d_f, const void *d_x, int in_dtype, int64_t L, int64_t R, int64_t n_ch, int64_t rs, int64_t cs, double *d_e, double *d_E,
                    const double *d_phi, double *d_zi, double *d_sum, int64_t srs, int64_t scs, hipStream_t st)
{
    constexpr int NB = 256 / CC, TT = NB * kIirSub, LOGNB = (CC == 1) ? 8 : 4;
    const uint64_t n_cgrp = (uint64_t)((n_ch + CC - 1) / CC), n_chunks = (uint64_t)((L + TT - 1) / TT);
    const uint64_t blocks = n_cgrp * n_chunks, n_rc = (uint64_t)(R - 1) * (uint64_t)n_ch;
    if (blocks == 0 || blocks > 0x7FFFFFFFull || n_rc > 0x7FFFFFFFull * 64) { set_error("filter grid out of range"); return ZMLS_E_ARG; }
    hipLaunchKernelGGL((iir_period_kernel<S, CC>), dim3((uint32_t)blocks), dim3(256), 0, st, d_f, d_x, in_dtype, L, (int)R, n_ch, rs, cs, (uint32_t)n_cgrp,
                       (int64_t)n_chunks, d_e, d_sum, srs, scs,
                       (CC == 16 && cs == 1 && scs == 1 && rs % 2 == 0 && srs % 2 == 0 && ((uintptr_t)d_x & 15) == 0 && ((uintptr_t)d_sum & 15) == 0) ? (kDotSweep() ? 2 : 1) : 0);
    hipLaunchKernelGGL(iir_period_end_kernel<S>, dim3((uint32_t)((n_rc + 63) / 64)), dim3(64), 0, st, d_f, LOGNB, (int64_t)n_chunks, n_ch, (int64_t)n_rc,
                       (const double *)d_e, d_E);
    hipLaunchKernelGGL(iir_period_chain_kernel<S>, dim3((uint32_t)((n_ch + 63) / 64)), dim3(64), 0, st, d_phi, (int)R, n_ch, (const double *)d_E, d_zi);
    ZHIP(hipGetLastError());
    return ZMLS_OK;
}

// A^k of the SxS state matrix by squaring, in long double
void mat_mul(int S, const long double *X, const long double *Y, long double *Z)
{
    long double T[kIirMaxS * kIirMaxS];
    for (int i = 0; i < S; ++i)
        for (int j = 0; j < S; ++j) {
            long double s = 0;
            for (int k = 0; k < S; ++k) s += X[i * S + k] * Y[k * S + j];
            T[i * S + j] = s;
        }
    memcpy(Z, T, sizeof(long double) * S * S);
}

} // namespace
} // namespace zmls

struct zmls_iir {
    int device;
    int S;
    zmls::IirDev *d_f;
    double *d_zc, *d_sin;      // zero-state end state / start state of every chunk
    double *d_zg, *d_sg;       // the same of every group of 256 chunks (1/256 of the above)
    size_t scratch_elems;      // of d_zc, d_sin each
    size_t group_elems;        // of d_zg, d_sg each: not monotonic in scratch_elems (few long channels against many short ones)
    double *d_carry;           // filter state handed from one segment of the time axis to the next: n_ch x S
    size_t carry_elems;
    size_t segment_bytes;      // signal bytes per segment (0: one segment)
    // the fused de-emphasis + impulse response path (zmls_iir_reserve_fused)
    long double A[zmls::kIirMaxS * zmls::kIirMaxS];   // state matrix, S x S (host)
    double *d_e, *d_E, *d_zi, *d_sum, *d_phi;          // chunk end states of every period, period end states, sum of start states, period sums, A^L
    size_t e_elems, E_elems, zi_elems, sum_elems;
    int64_t phi_L;             // the period length d_phi was built for
    void *d_host[2];           // zmls_iir_host: signal in / out on the device, kept between calls while small
    size_t host_bytes[2];
    // the post-filter form of the fused path (impulse_filtered_post)
    int post_nt[2];            // chunks at the end of a period that hold all of its end state: [0] frame-major tiles (512 frames), [1] time-contiguous (8192); 0: the filter rings too long
    int post_form;             // -1: the post-filter form where it applies, 0: never, 1: wherever it can run at all (tests)
    int once_mode, once_M;     // the one-sweep filter (iir_once_kernel): -1 where it pays, 0 never, 1 wherever it can run, once_M chunks per workgroup (tests)
    int last_form;             // which form the most recent zmls_impulse_filtered_device call took: 0 sweep, 1 post-filter
    double *d_zeta, *d_minv, *d_psi;   // per-channel coefficients of the rank-S term, (I - A^L)^-1, responses to the S zero-input basis sequences (L x S)
    size_t zeta_elems, psi_elems;
    int64_t psi_L;             // what d_psi was built for: period, register
    uint32_t psi_G;
};

using namespace zmls;

namespace zmls {
namespace {
// elements of the chunk arrays (d_zc, d_sin) and of the group arrays (d_zg, d_sg) a signal of this shape needs when its
// chunks are tt frames long. The two are sized and checked apart: ceil(chunks / 256) * channels is not monotonic in
// chunks * channels (a long one-channel signal needs more chunk states and FEWER group states than a shorter one of a
// thousand channels).
void iir_scratch_need(const zmls_iir *f, int64_t n_rows, int64_t n_ch, int64_t tt, size_t *need, size_t *need_g)
{
    const size_t chunks = (size_t)((n_rows + tt - 1) / tt);
    *need = chunks * (size_t)n_ch * (size_t)f->S;
    *need_g = chunks > (size_t)kIirSerialChunks ? ((chunks + kIirGroup - 1) / kIirGroup) * (size_t)n_ch * (size_t)f->S : 0;
}
int iir_scratch_check(const zmls_iir *f, int64_t n_rows, int64_t n_ch, int64_t tt)
{
    size_t need = 0, need_g = 0;
    iir_scratch_need(f, n_rows, n_ch, tt, &need, &need_g);
    if (need > f->scratch_elems || need_g > f->group_elems) {
        set_error("filter scratch holds %zu chunk and %zu group states, %zu and %zu needed: call zmls_iir_reserve first", f->scratch_elems,
                  f->group_elems, need, need_g);
        return ZMLS_E_WORKSPACE;
    }
    return ZMLS_OK;
}
} // namespace
} // namespace zmls


// The capture swept once: periods 1..R-1 added up, the filter's state at every period boundary beside it; then the
// filter over the one period of sums from the summed states. Leaves the filtered sums in the filter's own array.
int zmls::filtered_period_sums(zmls_plan *p, zmls_iir *f, const void *d_x, int in_dtype, int64_t n_rows, int64_t n_ch, int64_t row_stride,
                               int64_t ch_stride, const double **sums, int64_t *out_srs, int64_t *out_scs, hipStream_t st)
{
    if (!p || !f || !d_x) { set_error("null argument"); return ZMLS_E_ARG; }
    if (p->device != f->device) { set_error("plan is on device %d, filter on device %d", p->device, f->device); return ZMLS_E_ARG; }
    if (!dtype_size(in_dtype)) { set_error("unknown input dtype %d", in_dtype); return ZMLS_E_ARG; }
    if (n_ch < 1 || row_stride < 1 || ch_stride < 1) { set_error("bad shape/strides"); return ZMLS_E_ARG; }
    int64_t periods = 0;
    int rc = capture_periods(p, n_rows, &periods);
    if (rc) return rc;
    if (int rc0 = sequence_only(p)) return rc0;
    ZDEVICE(f->device);
    const int64_t L = p->L, R = periods + 1;
    const bool tc = row_stride == 1 && (n_ch == 1 || ch_stride != 1);
    const int64_t tt = (256 / (tc ? 1 : 16)) * kIirSub;
    const size_t n_chunks = (size_t)((L + tt - 1) / tt);
    if ((size_t)(R - 1) * n_chunks * (size_t)n_ch * (size_t)f->S > f->e_elems || (size_t)L * (size_t)n_ch > f->sum_elems ||
        (size_t)n_ch * (size_t)f->S > f->zi_elems || f->phi_L != L) {
        set_error("call zmls_iir_reserve_fused(filter, plan, n_rows, n_ch) first");
        return ZMLS_E_WORKSPACE;
    }
    rc = iir_scratch_check(f, L, n_ch, tt);                    // the filter over the one period of sums
    if (rc) return rc;
    // the sums keep the layout class of the capture: (L, n_ch) frames x channels, or n_ch runs of L frames
    const int64_t srs = tc ? 1 : n_ch, scs = tc ? L : 1;
#define ZPER(SS) case SS: rc = tc ? launch_period_s<SS, 1>(f->d_f, d_x, in_dtype, L, R, n_ch, row_stride, ch_stride, f->d_e, f->d_E, f->d_phi, f->d_zi, f->d_sum, srs, scs, st) \
                                  : launch_period_s<SS, 16>(f->d_f, d_x, in_dtype, L, R, n_ch, row_stride, ch_stride, f->d_e, f->d_E, f->d_phi, f->d_zi, f->d_sum, srs, scs, st); break;
#define ZFIL(SS) case SS: rc = launch_iir_cc<SS>(tc, f->d_f, f->d_sum, ZMLS_F64, L, n_ch, srs, scs, f->d_zc, f->d_sin, f->d_zg, f->d_sg, f->d_sum, srs, scs, f->d_zi, (size_t)n_ch * (size_t)f->S <= f->carry_elems ? f->d_carry : nullptr, f->segment_bytes, st); break;
    switch (f->S) {
        ZPER(1) ZPER(2) ZPER(3) ZPER(4) ZPER(5) ZPER(6) ZPER(7) ZPER(8)
    default: set_error("no filter kernel for order %d", f->S); return ZMLS_E_UNSUPPORTED;
    }
    if (rc) return rc;
    switch (f->S) {
        ZFIL(1) ZFIL(2) ZFIL(3) ZFIL(4) ZFIL(5) ZFIL(6) ZFIL(7) ZFIL(8)
    default: break;
    }
#undef ZPER
#undef ZFIL
    if (rc) return rc;
    *sums = f->d_sum;
    *out_srs = srs;
    *out_scs = scs;
    return ZMLS_OK;
}

namespace zmls {
namespace {
// smallest k with |(A^TT)^k| <= 1e-22 in every entry: samples more than k chunks before a period's end leave nothing of
// themselves in its end state (0: more than 4096 chunks, the post-filter form is not for this filter)
int tail_chunks(int S, const long double *M)
{
    long double Mk[kIirMaxS * kIirMaxS];
    memcpy(Mk, M, sizeof Mk);
    for (int k = 1; k <= 4096; ++k) {
        long double mx = 0;
        for (int i = 0; i < S * S; ++i) {
            if (!(Mk[i] == Mk[i])) return 0;
            mx = std::max(mx, fabsl(Mk[i]));
        }
        if (mx <= 1e-22L) return k;
        mat_mul(S, Mk, M, Mk);
    }
    return 0;
}

// X <- X^-1 (S x S, long double, Gauss-Jordan with partial pivoting); false: singular
bool mat_inv(int S, long double *X)
{
    long double W[kIirMaxS][2 * kIirMaxS];
    for (int i = 0; i < S; ++i)
        for (int j = 0; j < S; ++j) { W[i][j] = X[i * S + j]; W[i][S + j] = i == j ? 1.0L : 0.0L; }
    for (int c = 0; c < S; ++c) {
        int piv = c;
        for (int i = c + 1; i < S; ++i)
            if (fabsl(W[i][c]) > fabsl(W[piv][c])) piv = i;
        if (!(fabsl(W[piv][c]) > 1e-12L)) return false;
        if (piv != c)
            for (int j = 0; j < 2 * S; ++j) std::swap(W[piv][j], W[c][j]);
        const long double d = 1.0L / W[c][c];
        for (int j = 0; j < 2 * S; ++j) W[c][j] *= d;
        for (int i = 0; i < S; ++i) {
            if (i == c) continue;
            const long double m = W[i][c];
            if (m == 0.0L) continue;
            for (int j = 0; j < 2 * S; ++j) W[i][j] -= m * W[c][j];
        }
    }
    for (int i = 0; i < S; ++i)
        for (int j = 0; j < S; ++j) X[i * S + j] = W[i][S + j];
    return true;
}

// ZMLS_FUSED_POST=0 (tuning): the fused call always sweeps the capture in time order
bool kPostForm()
{
    static const bool on = [] { const char *v = getenv("ZMLS_FUSED_POST"); return !(v && *v == '0'); }();
    return on;
}
} // namespace
} // namespace zmls

extern "C" {

int zmls_iir_create(zmls_iir **out, const double *b, int nb, const double *a, int na, int device)
{
    if (!out || !b || !a || nb < 1 || na < 1) { set_error("null or empty filter coefficients"); return ZMLS_E_ARG; }
    *out = nullptr;
    if (a[0] == 0.0 || !isfinite(a[0])) { set_error("a[0] must be a finite non-zero number"); return ZMLS_E_ARG; }
    int S = (nb > na ? nb : na) - 1;
    if (S > kIirMaxS) { set_error("filter order %d is beyond the %d built for the device", S, kIirMaxS); return ZMLS_E_UNSUPPORTED; }
    if (S < 1) S = 1;           // a pure gain runs as a first-order filter with zero taps
    ZDEVICE(device);
    int rc = ZMLS_OK;
    IirDev h;
    memset(&h, 0, sizeof h);
    for (int i = 0; i < nb; ++i) h.b[i] = b[i] / a[0];     // lfilter normalises by a[0]
    for (int i = 0; i < na; ++i) h.a[i] = a[i] / a[0];
    // state matrix of the transposed direct form II: z' = A z + (...) x with A[i][0] = -a[i+1], A[i][i+1] = 1
    long double A[kIirMaxS * kIirMaxS] = {0}, Pw[kIirMaxS * kIirMaxS];
    for (int i = 0; i < S; ++i) {
        A[i * S + 0] = -(long double)h.a[i + 1];
        if (i + 1 < S) A[i * S + i + 1] += 1.0L;
    }
    memcpy(Pw, A, sizeof(long double) * S * S);
    long double A_keep[kIirMaxS * kIirMaxS];
    memcpy(A_keep, A, sizeof A_keep);
    for (int k = 0; k < 5; ++k) mat_mul(S, Pw, Pw, Pw);      // A^32
    int post_nt[2] = {0, 0};
    for (int d = 0; d < kIirLevels; ++d) {
        for (int i = 0; i < S; ++i)
            for (int j = 0; j < S; ++j) h.P[d][i * kIirMaxS + j] = (double)Pw[i * S + j];
        if (d == 4) post_nt[0] = tail_chunks(S, Pw);         // A^512: a frame-major chunk
        if (d == 8) post_nt[1] = tail_chunks(S, Pw);         // A^8192: a time-contiguous chunk
        mat_mul(S, Pw, Pw, Pw);
    }
    {
        // weights of the dot-product form of the fused sweep: W[i] = A^(TT-1-i) beta, built backwards from the chunk's last frame
        long double w[kIirMaxS] = {0}, t[kIirMaxS];
        for (int k = 0; k < S; ++k) w[k] = (long double)h.b[k + 1] - (long double)h.a[k + 1] * (long double)h.b[0];
        for (int i = kIirDotFrames - 1; i >= 0; --i) {
            for (int k = 0; k < S; ++k) h.W16[i * S + k] = (double)w[k];
            for (int r = 0; r < S; ++r) {
                long double acc = 0;
                for (int c = 0; c < S; ++c) acc += A_keep[r * S + c] * w[c];
                t[r] = acc;
            }
            memcpy(w, t, sizeof(long double) * S);
        }
    }
    zmls_iir *f = (zmls_iir *)calloc(1, sizeof(zmls_iir));
    if (!f) { set_error("out of host memory"); return ZMLS_E_ARG; }
    f->device = device;
    f->S = S;
    f->segment_bytes = kIirSegmentBytes;
    f->post_nt[0] = post_nt[0];
    f->post_nt[1] = post_nt[1];
    f->post_form = -1;
    f->once_mode = -1;
    memcpy(f->A, A_keep, sizeof A_keep);
    hipError_t e = hipMalloc((void **)&f->d_f, sizeof(IirDev));
    if (e == hipSuccess) e = hipMemcpy(f->d_f, &h, sizeof h, hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        rc = hip_fail(e, "filter upload");
        if (f->d_f) (void)hipFree(f->d_f);
        free(f);
        return rc;
    }
    *out = f;
    return ZMLS_OK;
}

int zmls_iir_destroy(zmls_iir *f)
{
    if (!f) return ZMLS_OK;
    DeviceScope on_device(f->device);
    if (!on_device.rc) {
        if (f->d_f) (void)hipFree(f->d_f);
        if (f->d_zc) (void)hipFree(f->d_zc);
        if (f->d_sin) (void)hipFree(f->d_sin);
        if (f->d_zg) (void)hipFree(f->d_zg);
        if (f->d_sg) (void)hipFree(f->d_sg);
        if (f->d_carry) (void)hipFree(f->d_carry);
        for (void *q : f->d_host)
            if (q) (void)hipFree(q);
        double *fused[8] = {f->d_e, f->d_E, f->d_zi, f->d_sum, f->d_phi, f->d_zeta, f->d_minv, f->d_psi};
        for (double *q : fused)
            if (q) (void)hipFree(q);
    }
    free(f);
    return ZMLS_OK;
}

// chunk states for signals of up to n_rows frames x n_ch channels in either layout
int zmls_iir_reserve(zmls_iir *f, int64_t n_rows, int64_t n_ch)
{
    if (!f || n_rows < 1 || n_ch < 1) { set_error("bad reserve arguments"); return ZMLS_E_ARG; }
    ZDEVICE(f->device);
    int rc = ZMLS_OK;
    const int64_t tt = (256 / 16) * kIirSub;                                   // the shorter chunk (frame-major tiles)
    size_t need = 0, need_g = 0;
    iir_scratch_need(f, n_rows, n_ch, tt, &need, &need_g);
    if (need > f->scratch_elems) {
        if (f->d_zc) { ZHIP(hipFree(f->d_zc)); f->d_zc = nullptr; }
        if (f->d_sin) { ZHIP(hipFree(f->d_sin)); f->d_sin = nullptr; }
        f->scratch_elems = 0;
        ZHIP(hipMalloc((void **)&f->d_zc, need * 8));
        ZHIP(hipMalloc((void **)&f->d_sin, need * 8));
        f->scratch_elems = need;
    }
    if ((size_t)n_ch * (size_t)f->S > f->carry_elems) {
        if (f->d_carry) { ZHIP(hipFree(f->d_carry)); f->d_carry = nullptr; }
        f->carry_elems = 0;
        ZHIP(hipMalloc((void **)&f->d_carry, (size_t)n_ch * (size_t)f->S * 8));
        f->carry_elems = (size_t)n_ch * (size_t)f->S;
    }
    if (need_g > f->group_elems) {
        if (f->d_zg) { ZHIP(hipFree(f->d_zg)); f->d_zg = nullptr; }
        if (f->d_sg) { ZHIP(hipFree(f->d_sg)); f->d_sg = nullptr; }
        f->group_elems = 0;
        ZHIP(hipMalloc((void **)&f->d_zg, need_g * 8));
        ZHIP(hipMalloc((void **)&f->d_sg, need_g * 8));
        f->group_elems = need_g;
    }
    return ZMLS_OK;
}

int zmls_iir_set_fused_form(zmls_iir *f, int form)
{
    if (!f || form < -1 || form > 1) { set_error("fused form must be -1 (auto), 0 (one sweep in time order) or 1 (filter after the transform)"); return ZMLS_E_ARG; }
    f->post_form = form;
    return ZMLS_OK;
}

int zmls_iir_set_one_sweep(zmls_iir *f, int mode, int chunks_per_workgroup)
{
    if (!f || mode < -1 || mode > 1 || (mode == 1 && (chunks_per_workgroup < 1 || chunks_per_workgroup > 65536))) {
        set_error("one-sweep mode must be -1 (auto), 0 (never) or 1 (forced, with 1..65536 chunks per workgroup)");
        return ZMLS_E_ARG;
    }
    f->once_mode = mode;
    f->once_M = mode == 1 ? chunks_per_workgroup : 0;
    return ZMLS_OK;
}

int zmls_iir_last_fused_form(const zmls_iir *f)
{
    return f ? f->last_form : -1;
}

int zmls_iir_set_segment(zmls_iir *f, size_t bytes)
{
    if (!f) { set_error("null argument"); return ZMLS_E_ARG; }
    f->segment_bytes = bytes == (size_t)-1 ? kIirSegmentBytes : bytes;
    return ZMLS_OK;
}

int zmls_iir_device(zmls_iir *f, const void *d_x, int in_dtype, int64_t n_rows, int64_t n_ch, int64_t row_stride, int64_t ch_stride,
                    double *d_y, int64_t out_row_stride, int64_t out_ch_stride, void *stream)
{
    if (!f || !d_x || !d_y) { set_error("null argument"); return ZMLS_E_ARG; }
    if (!dtype_size(in_dtype)) { set_error("unknown input dtype %d", in_dtype); return ZMLS_E_ARG; }
    if (n_rows < 1 || n_ch < 1 || row_stride < 1 || ch_stride < 1 || out_row_stride < 1 || out_ch_stride < 1) { set_error("bad shape/strides"); return ZMLS_E_ARG; }
    ZDEVICE(f->device);
    int rc = ZMLS_OK;
    const bool tc = row_stride == 1 && (n_ch == 1 || ch_stride != 1);
    const int64_t tt = (256 / (tc ? 1 : 16)) * kIirSub;
    rc = iir_scratch_check(f, n_rows, n_ch, tt);
    if (rc) return rc;
    hipStream_t st = (hipStream_t)stream;
#define ZIIR(SS) case SS: return launch_iir_cc<SS>(tc, f->d_f, d_x, in_dtype, n_rows, n_ch, row_stride, ch_stride, f->d_zc, f->d_sin, f->d_zg, f->d_sg, d_y, out_row_stride, out_ch_stride, nullptr, (size_t)n_ch * (size_t)f->S <= f->carry_elems ? f->d_carry : nullptr, f->segment_bytes, st, nullptr, nullptr, f->once_mode ? f->post_nt : nullptr, f->once_mode > 0 ? f->once_M : 0);
    switch (f->S) {
        ZIIR(1) ZIIR(2) ZIIR(3) ZIIR(4) ZIIR(5) ZIIR(6) ZIIR(7) ZIIR(8)
    default: break;
    }
#undef ZIIR
    set_error("no filter kernel for order %d", f->S);
    return ZMLS_E_UNSUPPORTED;
}

static int grow(double **ptr, size_t *have, size_t need)
{
    if (need <= *have) return ZMLS_OK;
    if (*ptr) { ZHIP(hipFree(*ptr)); *ptr = nullptr; *have = 0; }
    ZHIP(hipMalloc((void **)ptr, need * 8));
    *have = need;
    return ZMLS_OK;
}

int zmls_iir_reserve_fused(zmls_iir *f, zmls_plan *p, int64_t n_rows, int64_t n_ch)
{
    if (!f || !p || n_ch < 1) { set_error("bad reserve arguments"); return ZMLS_E_ARG; }
    if (p->device != f->device) { set_error("plan is on device %d, filter on device %d", p->device, f->device); return ZMLS_E_ARG; }
    int64_t periods = 0;
    int rc = capture_periods(p, n_rows, &periods);
    if (rc) return rc;
    ZDEVICE(f->device);
    const int64_t L = p->L, R = periods + 1;
    rc = zmls_iir_reserve(f, L, n_ch);                          // the filter proper runs over one period of sums
    if (rc) return rc;
    const int64_t tt = (256 / 16) * kIirSub;                   // the shorter chunk (frame-major tiles)
    const size_t n_chunks = (size_t)((L + tt - 1) / tt);
    const size_t nt_max = (size_t)std::max(f->post_nt[0], f->post_nt[1]);
    rc = grow(&f->d_e, &f->e_elems, std::max((size_t)(R - 1) * n_chunks, (size_t)(R + 1) * nt_max) * (size_t)n_ch * (size_t)f->S);
    if (!rc) rc = grow(&f->d_E, &f->E_elems, (size_t)(R + 1) * (size_t)n_ch * (size_t)f->S);     // the post-filter form: all R periods and the response
    if (!rc) rc = grow(&f->d_zi, &f->zi_elems, (size_t)n_ch * (size_t)f->S);
    if (!rc) rc = grow(&f->d_zeta, &f->zeta_elems, (size_t)n_ch * (size_t)f->S);
    if (!rc) rc = grow(&f->d_sum, &f->sum_elems, (size_t)L * (size_t)n_ch);
    if (rc) return rc;
    if (!f->d_phi) ZHIP(hipMalloc((void **)&f->d_phi, sizeof(double) * kIirMaxS * kIirMaxS));
    if (!f->d_minv) ZHIP(hipMalloc((void **)&f->d_minv, sizeof(double) * kIirMaxS * kIirMaxS));
    if (f->phi_L != L) {
        // A^L by binary powers, in long double like the other powers
        const int S = f->S;
        long double Rm[kIirMaxS * kIirMaxS] = {0}, Bm[kIirMaxS * kIirMaxS];
        for (int i = 0; i < S; ++i) Rm[i * S + i] = 1.0L;
        memcpy(Bm, f->A, sizeof Bm);
        for (uint64_t k = (uint64_t)L; k; k >>= 1) {
            if (k & 1) mat_mul(S, Bm, Rm, Rm);
            mat_mul(S, Bm, Bm, Bm);
        }
        double phi[kIirMaxS * kIirMaxS] = {0}, minv[kIirMaxS * kIirMaxS] = {0};
        for (int i = 0; i < S; ++i)
            for (int j = 0; j < S; ++j) phi[i * kIirMaxS + j] = (double)Rm[i * S + j];
        // (I - A^L)^-1: the periodic steady state of a period with zero-state end state E starts from this times E
        for (int i = 0; i < S * S; ++i) Rm[i] = -Rm[i];
        for (int i = 0; i < S; ++i) Rm[i * S + i] += 1.0L;
        if (!mat_inv(S, Rm)) f->post_nt[0] = f->post_nt[1] = 0;       // a pole on the unit circle: no such state (and no tail: both were 0 already)
        else
            for (int i = 0; i < S; ++i)
                for (int j = 0; j < S; ++j) minv[i * kIirMaxS + j] = (double)Rm[i * S + j];
        ZHIP(hipMemcpy(f->d_phi, phi, sizeof phi, hipMemcpyHostToDevice));
        ZHIP(hipMemcpy(f->d_minv, minv, sizeof minv, hipMemcpyHostToDevice));
        f->phi_L = L;
        f->psi_L = 0;
    }
    // the transform reads the filtered sums as float64 frames: a float32 plan, an odd channel count or a
    // time-contiguous capture take the layout adapters
    rc = zmls_plan_reserve_staging(p, n_ch);
    if (rc) return rc;
    // the post-filter form: psi = the plan's responses to the filter's S zero-input sequences, u_k[n] = (A^n e_k)[0], made by the
    // filter itself (zeros in, the unit states as initial states) and the plan itself (one period, no scaling)
    if (p->dtype == ZMLS_F64 && (f->post_nt[0] || f->post_nt[1]) && (f->psi_L != L || f->psi_G != p->G)) {
        const int S = f->S;
        rc = zmls_plan_reserve_staging(p, S);
        if (!rc) rc = zmls_iir_reserve(f, L, std::max<int64_t>(n_ch, S));
        if (!rc) rc = grow(&f->d_psi, &f->psi_elems, (size_t)L * (size_t)S);
        if (rc) return rc;
        double *d_u = nullptr, *d_unit = nullptr;
        double unit[kIirMaxS * kIirMaxS] = {0};
        for (int i = 0; i < S; ++i) unit[i * S + i] = 1.0;
        hipError_t e = hipMalloc((void **)&d_u, (size_t)L * S * 8);
        if (e == hipSuccess) e = hipMalloc((void **)&d_unit, sizeof unit);
        if (e == hipSuccess) e = hipMemset(d_u, 0, (size_t)L * S * 8);
        if (e == hipSuccess) e = hipMemcpy(d_unit, unit, sizeof unit, hipMemcpyHostToDevice);
        if (e != hipSuccess) rc = hip_fail(e, "zero-input responses");
#define ZZIR(SS) case SS: rc = launch_iir_cc<SS>(false, f->d_f, d_u, ZMLS_F64, L, S, S, 1, f->d_zc, f->d_sin, f->d_zg, f->d_sg, d_u, S, 1, d_unit, nullptr, 0, nullptr); break;
        if (!rc) switch (S) {
            ZZIR(1) ZZIR(2) ZZIR(3) ZZIR(4) ZZIR(5) ZZIR(6) ZZIR(7) ZZIR(8)
        default: break;
        }
#undef ZZIR
        if (!rc) {
            const int k = p->in_scale_log2;
            p->in_scale_log2 = 0;
            rc = impulse_device_core(p, d_u, ZMLS_F64, 1, 0, 1, S, S, 1, f->d_psi, S, 1, nullptr, nullptr, nullptr, 0);
            p->in_scale_log2 = k;
        }
        e = hipDeviceSynchronize();
        if (!rc && e != hipSuccess) rc = hip_fail(e, "zero-input responses");
        if (d_u) (void)hipFree(d_u);
        if (d_unit) (void)hipFree(d_unit);
        if (rc) return rc;
        f->psi_L = L;
        f->psi_G = p->G;
    }
    return ZMLS_OK;
}

// The post-filter form of the fused call: filter and transform commute up to a rank-S term. With S_x the averaged period, g_c the
// filter's impulse response folded modulo L and E(.) the zero-state end state of one period,
//     lfilter(S_x, zi = Zm) = g_c (*) S_x + ZIR(Zm - (I - A^L)^-1 E(S_x))              ((*) circular convolution),
// the circular cross-correlation with the sequence commutes with (*), and ZIR(zeta) = sum_k zeta_k u_k, so
//     h = g_c (*) h_plain + sum_k zeta_k psi_k,       g_c (*) h_plain = lfilter(h_plain, zi = (I - A^L)^-1 E(h_plain)).
// The capture goes through the ordinary gather pass (13.0 instead of 12.4 + 1.6 ms for the headline batch), the filter runs over
// the responses in place with the rank-S term added on its way out, and the end states E_r come from the last post_nt chunks of
// every period: 1e-22 of anything earlier is left there (tail_chunks).
static int impulse_filtered_post(zmls_plan *p, zmls_iir *f, const void *d_x, int in_dtype, int64_t n_rows, int64_t n_ch, int64_t rs, int64_t cs,
                                 double *d_h, int64_t ors, int64_t ocs, bool tc_in, bool tc_out, hipStream_t st)
{
    const int64_t L = p->L, R = n_rows / L;
    const int S = f->S;
    const int nt_in = f->post_nt[tc_in ? 1 : 0], nt_out = f->post_nt[tc_out ? 1 : 0];
    double *e_out = f->d_e + (size_t)R * (size_t)nt_in * (size_t)n_ch * (size_t)S;
    double *E_out = f->d_E + (size_t)R * (size_t)n_ch * (size_t)S;
    int rc = ZMLS_OK;
    ZDEVICE(f->device);
    // frame-major responses: the transform writes them into the filter's own period-sized array and the filter reads them from there,
    // so that it may take its one-sweep form (iir_once_kernel: not in place); else in place in d_h
    const bool via = !tc_out && kOnceSweep() && f->once_mode && !f->segment_bytes && (size_t)L * (size_t)n_ch <= f->sum_elems;
    double *h_mid = via ? f->d_sum : d_h;
    const int64_t mrs = via ? n_ch : ors, mcs = via ? 1 : ocs;
#define ZTAIL(SS) case SS: \
        rc = launch_tail_s<SS>(tc_in, f->d_f, d_x, in_dtype, L, R, n_ch, rs, cs, nt_in, f->d_e, f->d_E, st); \
        if (!rc) rc = impulse_device_core(p, d_x, in_dtype, R - 1, 1, R - 1, n_ch, rs, cs, h_mid, mrs, mcs, st, nullptr, nullptr, 0); \
        if (!rc) rc = launch_tail_s<SS>(tc_out, f->d_f, h_mid, ZMLS_F64, L, 1, n_ch, mrs, mcs, nt_out, e_out, E_out, st); \
        if (!rc) { \
            hipLaunchKernelGGL(iir_post_state_kernel<SS>, dim3((uint32_t)((n_ch + 63) / 64)), dim3(64), 0, st, (const double *)f->d_phi, (const double *)f->d_minv, \
                               (int)R, n_ch, (const double *)f->d_E, ldexp(1.0, -p->in_scale_log2), f->d_zeta, f->d_zi); \
            rc = launch_iir_cc<SS>(tc_out, f->d_f, h_mid, ZMLS_F64, L, n_ch, mrs, mcs, f->d_zc, f->d_sin, f->d_zg, f->d_sg, d_h, ors, ocs, f->d_zi, \
                                   (size_t)n_ch * (size_t)S <= f->carry_elems ? f->d_carry : nullptr, f->segment_bytes, st, f->d_psi, f->d_zeta, f->once_mode ? f->post_nt : nullptr, f->once_mode > 0 ? f->once_M : 0); \
        } \
        break;
    switch (S) {
        ZTAIL(1) ZTAIL(2) ZTAIL(3) ZTAIL(4) ZTAIL(5) ZTAIL(6) ZTAIL(7) ZTAIL(8)
    default: set_error("no filter kernel for order %d", S); return ZMLS_E_UNSUPPORTED;
    }
#undef ZTAIL
    return rc;
}

int zmls_impulse_filtered_device(zmls_plan *p, zmls_iir *f, const void *d_x, int in_dtype, int64_t n_rows, int64_t n_ch, int64_t row_stride,
                                 int64_t ch_stride, void *d_h, int64_t out_row_stride, int64_t out_ch_stride, void *stream)
{
    if (!d_h) { set_error("null argument"); return ZMLS_E_ARG; }
    if (out_row_stride < 1 || out_ch_stride < 1) { set_error("bad shape/strides"); return ZMLS_E_ARG; }
    const double *sums = nullptr;
    int64_t srs = 0, scs = 0;
    hipStream_t st = (hipStream_t)stream;
    if (p && f && d_x && p->device == f->device && p->dtype == ZMLS_F64 && dtype_size(in_dtype) && n_ch >= 1 && row_stride >= 1 && ch_stride >= 1 &&
        f->post_form != 0 && (kPostForm() || f->post_form == 1) && n_rows > p->L && n_rows % p->L == 0 && n_rows / p->L <= 0x7FFFFFFF && f->phi_L == p->L &&
        f->psi_L == p->L && f->psi_G == p->G) {
        // the post-filter form: when the filter's memory is a small part of a period (else the tails are most of the capture)
        const int64_t L = p->L, R = n_rows / L;
        const bool tc_in = row_stride == 1 && (n_ch == 1 || ch_stride != 1), tc_out = out_row_stride == 1 && (n_ch == 1 || out_ch_stride != 1);
        const int64_t nt_in = f->post_nt[tc_in ? 1 : 0], nt_out = f->post_nt[tc_out ? 1 : 0];
        const int64_t tt_in = tc_in ? 8192 : 512, tt_out = tc_out ? 8192 : 512;
        const int64_t share = f->post_form == 1 ? 1 : 4;            // auto: the tails are at most a quarter of the capture
        const bool short_memory = nt_in > 0 && nt_out > 0 && nt_in * tt_in * share <= L && nt_out * tt_out * share <= L;
        const bool room = (size_t)(R * nt_in + nt_out) * (size_t)n_ch * (size_t)f->S <= f->e_elems && (size_t)(R + 1) * (size_t)n_ch * (size_t)f->S <= f->E_elems &&
                          (size_t)n_ch * (size_t)f->S <= std::min(f->zi_elems, f->zeta_elems) && (size_t)L * (size_t)f->S <= f->psi_elems &&
                          !iir_scratch_check(f, L, n_ch, tt_out);
        f->last_form = (short_memory && room) ? 1 : 0;
        if (short_memory && room)
            return impulse_filtered_post(p, f, d_x, in_dtype, n_rows, n_ch, row_stride, ch_stride, (double *)d_h, out_row_stride, out_ch_stride, tc_in, tc_out, st);
    }
    if (f) f->last_form = 0;
    int rc = filtered_period_sums(p, f, d_x, in_dtype, n_rows, n_ch, row_stride, ch_stride, &sums, &srs, &scs, st);
    if (rc) return rc;
    // one period of filtered sums -> divided by the period count in the gather, transformed, scattered
    return impulse_device_core(p, sums, ZMLS_F64, 1, 0, n_rows / p->L - 1, n_ch, srs, scs, d_h, out_row_stride, out_ch_stride, st, nullptr, nullptr, 0);
}

int zmls_iir_host(zmls_iir *f, const void *x, int in_dtype, int64_t n_rows, int64_t n_ch, int64_t row_stride, int64_t ch_stride, double *y)
{
    if (!f || !x || !y) { set_error("null argument"); return ZMLS_E_ARG; }
    const size_t is = dtype_size(in_dtype);
    if (!is) { set_error("unknown input dtype %d", in_dtype); return ZMLS_E_ARG; }
    if (n_rows < 1 || n_ch < 1 || row_stride < 1 || ch_stride < 1) { set_error("bad shape/strides"); return ZMLS_E_ARG; }
    int rc = zmls_iir_reserve(f, n_rows, n_ch);
    if (rc) return rc;
    ZDEVICE(f->device);                 // the staging buffers and the copies below belong to the filter's device
    const size_t in_span = (size_t)((n_rows - 1) * row_stride + (n_ch - 1) * ch_stride + 1) * is;
    const size_t out_bytes = (size_t)n_rows * n_ch * 8;
    // staging on the device: kept with the filter between calls while small and large enough (grow-only), per call beyond that
    // (two hipMalloc / hipFree pairs cost more than filtering a one-channel stimulus, as in zmls_impulse_host)
    const bool keep = in_span + out_bytes <= kIirHostKeepBytes;
    void *d_x = nullptr;
    double *d_y = nullptr;
    hipError_t e = hipSuccess;
    if (keep) {
        if (f->host_bytes[0] < in_span) {
            if (f->d_host[0]) (void)hipFree(f->d_host[0]);
            f->d_host[0] = nullptr; f->host_bytes[0] = 0;
            e = hipMalloc(&f->d_host[0], in_span);
            if (e == hipSuccess) f->host_bytes[0] = in_span;
        }
        if (e == hipSuccess && f->host_bytes[1] < out_bytes) {
            if (f->d_host[1]) (void)hipFree(f->d_host[1]);
            f->d_host[1] = nullptr; f->host_bytes[1] = 0;
            e = hipMalloc(&f->d_host[1], out_bytes);
            if (e == hipSuccess) f->host_bytes[1] = out_bytes;
        }
        d_x = f->d_host[0]; d_y = (double *)f->d_host[1];
    } else {
        e = hipMalloc(&d_x, in_span);
        if (e == hipSuccess) e = hipMalloc((void **)&d_y, out_bytes);
    }
    if (e != hipSuccess) rc = hip_fail(e, "filter staging");
    if (!rc) rc = host_copy(f->device, d_x, (void *)x, in_span, true);
    // the result keeps the layout class of the input: (n_rows, n_ch) C-contiguous
    if (!rc) rc = zmls_iir_device(f, d_x, in_dtype, n_rows, n_ch, row_stride, ch_stride, d_y, n_ch, 1, nullptr);
    if (!rc) rc = host_copy(f->device, d_y, y, out_bytes, false);
    if (!keep) {
        if (d_x) (void)hipFree(d_x);
        if (d_y) (void)hipFree(d_y);
    }
    return rc;
}

} // extern "C"
