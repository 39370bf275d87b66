// Layout adapters either side of the transform, for captures that are not in the (frames, channels)
// vector layout the fused gather pass wants.
//
// The reference handles one channel per call, so a batch of its captures is naturally a STACK of
// single-channel arrays: (C, R*L), time contiguous ("capture-major"). Gathering a permuted row from that
// layout is one 8-byte element per channel at stride R*L: 13x slower than the frame-major path (measured,
// N=20, 512 captures x 16 periods: 192 ms against 14.4 ms). So such captures take two coalesced steps:
//   sum_transpose_kernel : periods added up in time order (trim_and_avgerage's sum, zignal/measure/mls.py:284-306;
//                          every sample read once, 512 contiguous bytes per wave) and transposed through
//                          LDS into a frame-major (L, C) array of sums
//   -> the normal passes on that one period of sums (division by the period count fused into the gather)
//   transpose_out_kernel : the (L, C) frame-major responses back to (C, L), if the caller wants that.
// The bound of all three is HBM; the extra traffic is 2(+2) x L x C elements, against R_avg x L x C read.
#include "zmls_internal.h"

namespace zmls {
namespace {

template <typename T> struct LVec;
template <> struct LVec<double> { typedef double type __attribute__((ext_vector_type(2))); static constexpr int E = 2; };
template <> struct LVec<float> { typedef float type __attribute__((ext_vector_type(4))); static constexpr int E = 4; };

constexpr int kTileRows = 64;       // frames per tile = lanes of a wave
constexpr int kTileBytes = 256;     // bytes of adjacent channels per frame of a tile

template <typename T, typename TIN>
__global__ void __launch_bounds__(256) sum_transpose_kernel(const TIN *__restrict__ x, int64_t L, int periods, int skip, int64_t ch_stride,
                                                            int64_t ch_first, int64_t n_ch_valid, T *__restrict__ S, int64_t Cs,
                                                            uint32_t n_cblk, int64_t j_first /* first frame of this launch */)
{
    using V = typename LVec<T>::type;
    constexpr int E = LVec<T>::E;
    constexpr int TC = kTileBytes / (int)sizeof(T);     // channels per tile
    constexpr int CPW = TC / 4;                          // channels per wave
    __shared__ T tile[kTileRows][TC + E];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // consecutive workgroups are the channel blocks of one run of frames: together they write whole rows of S
    const int64_t j0 = j_first + (int64_t)(blockIdx.x / n_cblk) * kTileRows;
    const int64_t cb = (int64_t)(blockIdx.x % n_cblk) * TC;   // first channel of the tile inside the staging array
    const int64_t j = j0 + lane;

    T acc[CPW];
    const TIN *src[CPW];
    bool ok[CPW];
#pragma unroll
    for (int i = 0; i < CPW; ++i) {
        const int64_t ch = ch_first + cb + wave * CPW + i;
        ok[i] = j < L && ch < n_ch_valid;
        src[i] = x + (ok[i] ? ch : 0) * ch_stride + (int64_t)skip * L + (ok[i] ? j : 0);
        acc[i] = (T)0;
    }
    // the capture is read exactly once: non-temporal loads, DEPTH periods of them in flight per lane
    constexpr int DEPTH = 4;
    int p = 0;
    for (; p + DEPTH <= periods; p += DEPTH) {
        TIN t[DEPTH][CPW];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
#pragma unroll
            for (int i = 0; i < CPW; ++i) t[d][i] = __builtin_nontemporal_load(src[i] + (int64_t)(p + d) * L);
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
#pragma unroll
            for (int i = 0; i < CPW; ++i) acc[i] += (T)t[d][i];     // period order, as numpy adds the rows
    }
    for (; p < periods; ++p) {
#pragma unroll
        for (int i = 0; i < CPW; ++i) acc[i] += (T)__builtin_nontemporal_load(src[i] + (int64_t)p * L);
    }
#pragma unroll
    for (int i = 0; i < CPW; ++i) tile[lane][wave * CPW + i] = ok[i] ? acc[i] : (T)0;
    __syncthreads();
    constexpr int VPR = TC / E;                          // 16-byte vectors per tile row
    constexpr int RPP = 256 / VPR;                       // rows per sweep of the workgroup
    const int vv = threadIdx.x % VPR, r0 = threadIdx.x / VPR;
#pragma unroll
    for (int r = r0; r < kTileRows; r += RPP) {
        if (j0 + r < L && cb + vv * E < Cs) {
            V o;
#pragma unroll
            for (int e = 0; e < E; ++e) o[e] = tile[r][vv * E + e];
            __builtin_nontemporal_store(o, (V *)(S + (j0 + r) * Cs + cb + vv * E));
        }
    }
}

// The same for float64 captures in 16-byte loads (round 3): a lane owns TWO consecutive frames, a tile is 128 frames. A period of an
// odd length L starts on an odd element every other time, so a lane's pair (j, j+1) is 16-byte aligned only for half of the (channel,
// period) runs; for the others the lane loads the aligned pair (j+1, j+2) instead and takes frame j from the lane below (lane 0: one
// 8-byte load). Which of the two is uniform over the wave. Frame j0 + 128 is touched, so only tiles with j0 + 129 <= L run here; the
// rows after them go through the kernel above. Same additions in the same order: same bits.
constexpr int kTileRows2 = 128;
__global__ void __launch_bounds__(256) sum_transpose_pairs_kernel(const double *__restrict__ x, int64_t L, int periods, int skip, int64_t ch_stride,
                                                                  int64_t ch_first, int64_t n_ch_valid, double *__restrict__ S, int64_t Cs,
                                                                  uint32_t n_cblk)
{
    typedef double D2 __attribute__((ext_vector_type(2)));
    constexpr int TC = kTileBytes / 8;                   // 32 channels per tile
    constexpr int CPW = TC / 4;                          // 8 per wave
    __shared__ double tile[kTileRows2][TC + 2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t j0 = (int64_t)(blockIdx.x / n_cblk) * kTileRows2;
    const int64_t cb = (int64_t)(blockIdx.x % n_cblk) * TC;

    D2 acc[CPW];
    int64_t e0[CPW];                                     // element offset of frame j0 of the first summed period, per channel
    bool ok[CPW];
#pragma unroll
    for (int i = 0; i < CPW; ++i) {
        const int64_t ch = ch_first + cb + wave * CPW + i;
        ok[i] = ch < n_ch_valid;
        e0[i] = (ok[i] ? ch : 0) * ch_stride + (int64_t)skip * L + j0;
        acc[i] = (D2)0.0;
    }
    constexpr int DEPTH = 2;
    for (int p = 0; p < periods; p += DEPTH) {
        D2 t[DEPTH][CPW];
        double lo[DEPTH][CPW];
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
#pragma unroll
            for (int i = 0; i < CPW; ++i) {
                const int64_t e = e0[i] + (int64_t)(p + d) * L;
                const int64_t par = e & 1;
                if (p + d < periods) {
                    t[d][i] = __builtin_nontemporal_load((const D2 *)(x + e + par + 2 * lane));
                    lo[d][i] = (par && lane == 0) ? __builtin_nontemporal_load(x + e) : 0.0;
                } else {
                    t[d][i] = (D2)0.0;
                    lo[d][i] = 0.0;
                }
            }
#pragma unroll
        for (int d = 0; d < DEPTH; ++d)
#pragma unroll
            for (int i = 0; i < CPW; ++i) {
                if (p + d >= periods) continue;
                const int64_t par = (e0[i] + (int64_t)(p + d) * L) & 1;
                if (par) {
                    double below = __shfl_up(t[d][i][1], 1, 64);
                    if (lane == 0) below = lo[d][i];
                    acc[i][0] += below;                   // period order, as numpy adds the rows
                    acc[i][1] += t[d][i][0];
                } else {
                    acc[i] += t[d][i];
                }
            }
    }
#pragma unroll
    for (int i = 0; i < CPW; ++i) {
        tile[2 * lane][wave * CPW + i] = ok[i] ? acc[i][0] : 0.0;
        tile[2 * lane + 1][wave * CPW + i] = ok[i] ? acc[i][1] : 0.0;
    }
    __syncthreads();
    constexpr int VPR = TC / 2;                          // 16-byte vectors per tile row
    constexpr int RPP = 256 / VPR;                       // rows per sweep of the workgroup
    const int vv = threadIdx.x % VPR, r0 = threadIdx.x / VPR;
#pragma unroll
    for (int r = r0; r < kTileRows2; r += RPP) {
        if (cb + vv * 2 < Cs) {
            D2 o;
            o[0] = tile[r][vv * 2];
            o[1] = tile[r][vv * 2 + 1];
            __builtin_nontemporal_store(o, (D2 *)(S + (j0 + r) * Cs + cb + vv * 2));
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(256) transpose_out_kernel(const T *__restrict__ H, int64_t L, int64_t Cs, int64_t n_ch, T *__restrict__ out,
                                                            int64_t out_ch_stride, uint32_t n_cblk, int64_t j_first /* first frame of this launch */)
{
    using V = typename LVec<T>::type;
    constexpr int E = LVec<T>::E;
    constexpr int TC = kTileBytes / (int)sizeof(T);
    constexpr int CPW = TC / 4;
    __shared__ T tile[kTileRows][TC + E];
    const int64_t j0 = j_first + (int64_t)(blockIdx.x / n_cblk) * kTileRows;
    const int64_t cb = (int64_t)(blockIdx.x % n_cblk) * TC;
    constexpr int VPR = TC / E;
    constexpr int RPP = 256 / VPR;
    const int vv = threadIdx.x % VPR, r0 = threadIdx.x / VPR;
#pragma unroll
    for (int r = r0; r < kTileRows; r += RPP) {
        V o = (V)(T)0;
        if (j0 + r < L && cb + vv * E < Cs) o = __builtin_nontemporal_load((const V *)(H + (j0 + r) * Cs + cb + vv * E));
#pragma unroll
        for (int e = 0; e < E; ++e) tile[r][vv * E + e] = o[e];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t j = j0 + lane;
#pragma unroll
    for (int i = 0; i < CPW; ++i) {
        const int64_t ch = cb + wave * CPW + i;
        if (j < L && ch < n_ch) __builtin_nontemporal_store(tile[lane][wave * CPW + i], out + ch * out_ch_stride + j);
    }
}

// The same for float64 responses in 16-byte stores (round 3), whole tiles of 128 frames only: a lane writes frames (j, j+1) of a
// channel whose run starts on an even element, else (j+1, j+2) - lanes 0..62 - with frames j0 and j0 + 127 as single stores;
// which of the two is uniform over the wave (channel c starts c * out_ch_stride elements in, and L is odd).
__global__ void __launch_bounds__(256) transpose_out_pairs_kernel(const double *__restrict__ H, int64_t Cs, int64_t n_ch, double *__restrict__ out,
                                                                  int64_t out_ch_stride, uint32_t n_cblk)
{
    typedef double D2 __attribute__((ext_vector_type(2)));
    constexpr int TC = kTileBytes / 8;
    constexpr int CPW = TC / 4;
    __shared__ double tile[kTileRows2][TC + 2];
    const int64_t j0 = (int64_t)(blockIdx.x / n_cblk) * kTileRows2;
    const int64_t cb = (int64_t)(blockIdx.x % n_cblk) * TC;
    constexpr int VPR = TC / 2;
    constexpr int RPP = 256 / VPR;
    const int vv = threadIdx.x % VPR, r0 = threadIdx.x / VPR;
#pragma unroll
    for (int r = r0; r < kTileRows2; r += RPP) {
        D2 o = (D2)0.0;
        if (cb + vv * 2 < Cs) o = __builtin_nontemporal_load((const D2 *)(H + (j0 + r) * Cs + cb + vv * 2));
        tile[r][vv * 2] = o[0];
        tile[r][vv * 2 + 1] = o[1];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < CPW; ++i) {
        const int64_t ch = cb + wave * CPW + i;
        if (ch >= n_ch) continue;
        const int col = wave * CPW + i;
        double *dst = out + ch * out_ch_stride + j0;
        if ((((uintptr_t)dst) & 15) == 0) {
            D2 o;
            o[0] = tile[2 * lane][col];
            o[1] = tile[2 * lane + 1][col];
            __builtin_nontemporal_store(o, (D2 *)(dst + 2 * lane));
        } else {
            if (lane < 63) {
                D2 o;
                o[0] = tile[2 * lane + 1][col];
                o[1] = tile[2 * lane + 2][col];
                __builtin_nontemporal_store(o, (D2 *)(dst + 2 * lane + 1));
            } else {
                __builtin_nontemporal_store(tile[127][col], dst + 127);
            }
            if (lane == 0) __builtin_nontemporal_store(tile[0][col], dst);
        }
    }
}

template <typename T, typename TIN>
int launch_sum_transpose_t(const void *d_x, int64_t L, int64_t periods, int skip, int64_t ch_stride, int64_t ch_first, int64_t n_ch_valid,
                           void *d_S, int64_t Cs, hipStream_t st)
{
    constexpr int TC = kTileBytes / (int)sizeof(T);
    const uint64_t n_cblk = (uint64_t)((Cs + TC - 1) / TC);
    int64_t j_first = 0;
    if constexpr (__is_same(T, double) && __is_same(TIN, double)) {
        // float64 in and out, 16-byte aligned capture: the tiles of 128 frames that lie wholly before the period's last frame in pairs
        static const bool pairs_on = [] { const char *v = getenv("ZMLS_SUM_PAIRS"); return !(v && *v == '0'); }();
        const int64_t n_full = (L - 1) / kTileRows2;
        if (pairs_on && n_full > 0 && ((uintptr_t)d_x & 15) == 0 && (uint64_t)n_full * n_cblk <= 0x7FFFFFFFull) {
            hipLaunchKernelGGL(sum_transpose_pairs_kernel, dim3((uint32_t)((uint64_t)n_full * n_cblk)), dim3(256), 0, st, (const double *)d_x, L, (int)periods, skip,
                               ch_stride, ch_first, n_ch_valid, (double *)d_S, Cs, (uint32_t)n_cblk);
            j_first = n_full * kTileRows2;
        }
    }
    const uint64_t blocks = (uint64_t)((L - j_first + kTileRows - 1) / kTileRows) * n_cblk;
    if (blocks > 0x7FFFFFFFull) { set_error("layout grid out of range"); return ZMLS_E_ARG; }
    hipLaunchKernelGGL((sum_transpose_kernel<T, TIN>), dim3((uint32_t)blocks), dim3(256), 0, st, (const TIN *)d_x, L, (int)periods, skip, ch_stride,
                       ch_first, n_ch_valid, (T *)d_S, Cs, (uint32_t)n_cblk, j_first);
    ZHIP(hipGetLastError());
    return ZMLS_OK;
}

template <typename T>
int launch_sum_transpose_in(const void *d_x, int in_dtype, int64_t L, int64_t periods, int skip, int64_t ch_stride, int64_t ch_first,
                            int64_t n_ch_valid, void *d_S, int64_t Cs, hipStream_t st)
{
    switch (in_dtype) {
    case ZMLS_F64: return launch_sum_transpose_t<T, double>(d_x, L, periods, skip, ch_stride, ch_first, n_ch_valid, d_S, Cs, st);
    case ZMLS_F32: return launch_sum_transpose_t<T, float>(d_x, L, periods, skip, ch_stride, ch_first, n_ch_valid, d_S, Cs, st);
    case ZMLS_I16: return launch_sum_transpose_t<T, int16_t>(d_x, L, periods, skip, ch_stride, ch_first, n_ch_valid, d_S, Cs, st);
    case ZMLS_I32: return launch_sum_transpose_t<T, int32_t>(d_x, L, periods, skip, ch_stride, ch_first, n_ch_valid, d_S, Cs, st);
    default: set_error("bad dtype"); return ZMLS_E_ARG;
    }
}

} // namespace

// x: channel `ch` at d_x + ch*ch_stride, time contiguous. S: (L, Cs) frame-major sums of the plan's dtype for channels
// ch_first .. ch_first+Cs-1 (columns beyond n_ch_valid are zero). Cs must be a multiple of the 16-byte vector.
int launch_sum_transpose(const void *d_x, int in_dtype, int out_dtype, int64_t L, int64_t periods, int skip_periods, int64_t ch_stride,
                         int64_t ch_first, int64_t n_ch_valid, void *d_S, int64_t Cs, hipStream_t st)
{
    if (periods < 1 || periods > 0x7FFFFFFF) { set_error("period count out of range"); return ZMLS_E_ARG; }
    if (out_dtype == ZMLS_F64) return launch_sum_transpose_in<double>(d_x, in_dtype, L, periods, skip_periods, ch_stride, ch_first, n_ch_valid, d_S, Cs, st);
    if (out_dtype == ZMLS_F32) return launch_sum_transpose_in<float>(d_x, in_dtype, L, periods, skip_periods, ch_stride, ch_first, n_ch_valid, d_S, Cs, st);
    set_error("sums are f64 or f32");
    return ZMLS_E_ARG;
}

// H: (L, Cs) frame-major of `dtype`; out: channel c (< n_ch) at d_out + c*out_ch_stride, time contiguous.
int launch_transpose_out(const void *d_H, int dtype, int64_t L, int64_t Cs, int64_t n_ch, void *d_out, int64_t out_ch_stride, hipStream_t st)
{
    if (dtype == ZMLS_F64) {
        constexpr int TC = kTileBytes / 8;
        const uint64_t n_cblk = (uint64_t)((n_ch + TC - 1) / TC);
        int64_t j_first = 0;
        static const bool pairs_on = [] { const char *v = getenv("ZMLS_SUM_PAIRS"); return !(v && *v == '0'); }();
        const int64_t n_full = L / kTileRows2;
        if (pairs_on && n_full > 0 && ((uintptr_t)d_out & 7) == 0 && (uint64_t)n_full * n_cblk <= 0x7FFFFFFFull) {
            hipLaunchKernelGGL(transpose_out_pairs_kernel, dim3((uint32_t)((uint64_t)n_full * n_cblk)), dim3(256), 0, st, (const double *)d_H, Cs, n_ch,
                               (double *)d_out, out_ch_stride, (uint32_t)n_cblk);
            j_first = n_full * kTileRows2;
        }
        const uint64_t blocks = (uint64_t)((L - j_first + kTileRows - 1) / kTileRows) * n_cblk;
        if (blocks > 0x7FFFFFFFull) { set_error("layout grid out of range"); return ZMLS_E_ARG; }
        if (blocks)
            hipLaunchKernelGGL(transpose_out_kernel<double>, dim3((uint32_t)blocks), dim3(256), 0, st, (const double *)d_H, L, Cs, n_ch, (double *)d_out,
                               out_ch_stride, (uint32_t)n_cblk, j_first);
    } else if (dtype == ZMLS_F32) {
        constexpr int TC = kTileBytes / 4;
        const uint64_t n_cblk = (uint64_t)((n_ch + TC - 1) / TC), blocks = (uint64_t)((L + kTileRows - 1) / kTileRows) * n_cblk;
        if (blocks > 0x7FFFFFFFull) { set_error("layout grid out of range"); return ZMLS_E_ARG; }
        hipLaunchKernelGGL(transpose_out_kernel<float>, dim3((uint32_t)blocks), dim3(256), 0, st, (const float *)d_H, L, Cs, n_ch, (float *)d_out,
                           out_ch_stride, (uint32_t)n_cblk, (int64_t)0);
    } else { set_error("responses are f64 or f32"); return ZMLS_E_ARG; }
    ZHIP(hipGetLastError());
    return ZMLS_OK;
}

// ---- what this GPU streams (bench.py's same-run calibration of the roofline): 16 bytes per lane, 8 loads in flight
// per lane, every workgroup sweeping the buffer with the whole grid's stride (consecutive workgroups read
// consecutive 4 KiB)
namespace {
typedef uint32_t U4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) stream_read_kernel(const U4 *src, uint64_t n_vec, uint32_t *sink)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    U4 acc = (U4)0u;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 7 * stride < n_vec; i += 8 * stride) {
        U4 t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = __builtin_nontemporal_load(src + i + k * stride);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc ^= t[k];
    }
    for (; i < n_vec; i += stride) acc ^= src[i];
    // never true for the data the benchmark holds; keeps the loads alive without a store per lane (the sink is a
    // kernel argument: a store to a file-scope variable nobody reads is removed, and the loads with it)
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9E3779B9u && acc.x == 0x7F4A7C15u) *sink = acc.y;
}

__global__ void __launch_bounds__(256) stream_copy_kernel(const U4 *__restrict__ src, U4 *__restrict__ dst, uint64_t n_vec)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 7 * stride < n_vec; i += 8 * stride) {
        U4 t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = __builtin_nontemporal_load(src + i + k * stride);
#pragma unroll
        for (int k = 0; k < 8; ++k) __builtin_nontemporal_store(t[k], dst + i + k * stride);
    }
    for (; i < n_vec; i += stride) dst[i] = src[i];
}
} // namespace

} // namespace zmls

extern "C" int zmls_dev_stream_probe(int device, void *d_buf, size_t bytes, int mode, void *stream)
{
    using namespace zmls;
    if (!d_buf || bytes < 8192 || bytes % 4096 || ((uintptr_t)d_buf & 15) || mode < 0 || mode > 1) { set_error("bad stream probe arguments"); return ZMLS_E_ARG; }
    ZDEVICE(device);
    int rc = ZMLS_OK;
    const uint32_t grid = 16384;            // measured best of 1024 / 2048 / 4096 / 16384 workgroups for both shapes (tools/stream_probe.hip)
    if (mode == 0) hipLaunchKernelGGL(stream_read_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const U4 *)d_buf, (uint64_t)(bytes / 16), (uint32_t *)d_buf);
    else hipLaunchKernelGGL(stream_copy_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, (const U4 *)d_buf,
                            (U4 *)((char *)d_buf + bytes / 2), (uint64_t)(bytes / 32));
    ZHIP(hipGetLastError());
    return ZMLS_OK;
}
