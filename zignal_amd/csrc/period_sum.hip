// Partial period sums in time order, for captures whose periods are spread over several GPUs.
//
// trim_and_avgerage (zignal/measure/mls.py:284-306) is a sum over periods followed by one division;
// the sum is linear, so every GPU adds up the periods it holds and the partial sums are added across
// GPUs (RCCL reduce-scatter, comm.hip) before the transform. The output is laid out for that exchange:
//     out[block q][frame j][channel c']  =  sum over local periods of  x[(skip + p) L + j][q*block_ch_stride + ch_first + c']
// block q is what rank q of the communicator receives. Pure streaming: every sample is read once,
// in address order inside a frame; the bound is HBM.
#include "zmls_internal.h"

namespace zmls {
namespace {

template <typename T> struct SumVec;
template <> struct SumVec<double> { typedef double type __attribute__((ext_vector_type(2))); static constexpr int E = 2; };
template <> struct SumVec<float> { typedef float type __attribute__((ext_vector_type(4))); static constexpr int E = 4; };

template <typename T>
__device__ __forceinline__ T sum_load(const void *base, int64_t idx, int dtype)
{
    switch (dtype) {
    case ZMLS_F64: return (T)((const double *)base)[idx];
    case ZMLS_F32: return (T)((const float *)base)[idx];
    case ZMLS_I16: return (T)((const int16_t *)base)[idx];
    default: return (T)((const int32_t *)base)[idx];
    }
}

// contiguous, 16-byte aligned channels of the plan's own element type: one 16-byte vector per lane
template <typename T>
__global__ void __launch_bounds__(256) period_sum_vec_kernel(const T *__restrict__ x, int64_t L, int periods, int skip, int64_t row_stride,
                                                             int64_t n_blocks, int64_t block_ch_stride, int64_t ch_first,
                                                             int64_t vec_per_row, T *__restrict__ out)
{
    using V = typename SumVec<T>::type;
    constexpr int E = SumVec<T>::E;
    constexpr int PU = 4;                                   // periods loaded back to back
    const int64_t per_frame = n_blocks * vec_per_row;
    const int64_t total = L * per_frame;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t j = idx / per_frame;
        const int64_t r = idx - j * per_frame;
        const int64_t q = r / vec_per_row;
        const int64_t vv = r - q * vec_per_row;
        const T *src = x + ((int64_t)skip * L + j) * row_stride + q * block_ch_stride + ch_first + vv * E;
        const int64_t pstep = L * row_stride;
        V acc = (V)(T)0;
        int p = 0;
        for (; p + PU <= periods; p += PU) {
            V t[PU];
#pragma unroll
            for (int i = 0; i < PU; ++i) t[i] = __builtin_nontemporal_load((const V *)(src + (int64_t)(p + i) * pstep));     // read once
#pragma unroll
            for (int i = 0; i < PU; ++i) acc += t[i];       // period order, as numpy adds the rows
        }
        for (; p < periods; ++p) acc += __builtin_nontemporal_load((const V *)(src + (int64_t)p * pstep));
        __builtin_nontemporal_store(acc, (V *)(out + ((q * L + j) * vec_per_row + vv) * E));
    }
}

// contiguous, 16-byte aligned channels of ANOTHER sample type (raw PCM int16 / int32, float32 into a float64 plan, float64 into a
// float32 plan): every lane owns W = max(16/sizeof(TIN), 16/sizeof(T)) adjacent channels, reads them as whole 16-byte vectors,
// converts, adds in period order (the same additions as the element kernel below) and the W sums leave as 16-byte vectors.
// TR: a wave's 64 lanes own one contiguous run of the output (the launcher checks), so the sums are transposed through LDS and
// every store instruction of the wave writes 1 KiB in one piece instead of 16 bytes at a stride of W elements.
template <typename T, typename TIN, bool TR>
__global__ void __launch_bounds__(256) period_sum_cvt_kernel(const TIN *__restrict__ x, int64_t L, int periods, int skip, int64_t row_stride,
                                                             int64_t n_blocks, int64_t block_ch_stride, int64_t ch_first,
                                                             int64_t lanes_per_row, T *__restrict__ out)
{
    constexpr int NE = (int)(kVecBytes / sizeof(TIN)), E = SumVec<T>::E;
    constexpr int W = NE > E ? NE : E, NL = W / NE, NS = W / E;
    typedef TIN VI __attribute__((ext_vector_type(NE)));
    using VO = typename SumVec<T>::type;
    constexpr int PU = (NL == 1) ? 4 : 2;                   // periods loaded back to back
    __shared__ VO lds[TR ? 256 * NS : 1];
    const int64_t per_frame = n_blocks * lanes_per_row;
    const int64_t total = L * per_frame;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // the same trip count for the whole workgroup (barriers inside)
    for (int64_t base = (int64_t)blockIdx.x * blockDim.x; base < total; base += (int64_t)gridDim.x * blockDim.x) {
        const int64_t idx = base + threadIdx.x;
        const bool active = idx < total;
        const int64_t j = (active ? idx : 0) / per_frame;
        const int64_t r = (active ? idx : 0) - j * per_frame;
        const int64_t q = r / lanes_per_row;
        const int64_t vv = r - q * lanes_per_row;
        const TIN *src = x + ((int64_t)skip * L + j) * row_stride + q * block_ch_stride + ch_first + vv * W;
        const int64_t pstep = L * row_stride;
        T acc[W];
#pragma unroll
        for (int e = 0; e < W; ++e) acc[e] = (T)0;
        int p = 0;
        for (; p + PU <= (active ? periods : 0); p += PU) {
            VI t[PU][NL];
#pragma unroll
            for (int i = 0; i < PU; ++i)
#pragma unroll
                for (int l = 0; l < NL; ++l) t[i][l] = __builtin_nontemporal_load((const VI *)(src + (int64_t)(p + i) * pstep) + l);     // read once
#pragma unroll
            for (int i = 0; i < PU; ++i)
#pragma unroll
                for (int l = 0; l < NL; ++l)
#pragma unroll
                    for (int e = 0; e < NE; ++e) acc[l * NE + e] += (T)t[i][l][e];       // period order, as numpy adds the rows
        }
        for (; p < (active ? periods : 0); ++p)
#pragma unroll
            for (int l = 0; l < NL; ++l) {
                const VI t = __builtin_nontemporal_load((const VI *)(src + (int64_t)p * pstep) + l);
#pragma unroll
                for (int e = 0; e < NE; ++e) acc[l * NE + e] += (T)t[e];
            }
        T *dst = out + ((q * L + j) * lanes_per_row + vv) * W;
        if constexpr (!TR || NS == 1) {
            if (active) {
#pragma unroll
                for (int k = 0; k < NS; ++k) {
                    VO o;
#pragma unroll
                    for (int e = 0; e < E; ++e) o[e] = acc[k * E + e];
                    __builtin_nontemporal_store(o, (VO *)dst + k);
                }
            }
        } else {
            VO *mine = lds + wave * 64 * NS;
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                VO o;
#pragma unroll
                for (int e = 0; e < E; ++e) o[e] = acc[k * E + e];
                mine[lane * NS + k] = o;
            }
            __syncthreads();
            // where lane 0 of this wave writes: all 64 lanes of the wave lie in one run of the output
            const int64_t idx0 = base + wave * 64;
            const int64_t j0 = (idx0 < total ? idx0 : 0) / per_frame, r0 = (idx0 < total ? idx0 : 0) - j0 * per_frame;
            const int64_t q0 = r0 / lanes_per_row, v0 = r0 - q0 * lanes_per_row;
            VO *wdst = (VO *)(out + ((q0 * L + j0) * lanes_per_row + v0) * W);
#pragma unroll
            for (int k = 0; k < NS; ++k) {
                const int v = k * 64 + lane;                 // vector v of the wave's run belongs to lane v / NS
                if (idx0 + v / NS < total) __builtin_nontemporal_store(mine[v], wdst + v);
            }
            __syncthreads();
        }
    }
}

// any strides, any input type
template <typename T>
__global__ void __launch_bounds__(256) period_sum_elem_kernel(const void *__restrict__ x, int in_dtype, int64_t L, int periods, int skip,
                                                              int64_t row_stride, int64_t ch_stride, int64_t n_blocks,
                                                              int64_t block_ch_stride, int64_t ch_first, int64_t block_ch,
                                                              int64_t ch_limit, T *__restrict__ out)
{
    const int64_t per_frame = n_blocks * block_ch;
    const int64_t total = L * per_frame;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t j = idx / per_frame;
        const int64_t r = idx - j * per_frame;
        const int64_t q = r / block_ch;
        const int64_t c = r - q * block_ch;
        const int64_t ch = q * block_ch_stride + ch_first + c;
        T acc = (T)0;
        const int np = ch < ch_limit ? periods : 0;
        const int64_t first = ((int64_t)skip * L + j) * row_stride + ch * ch_stride, pstep = L * row_stride;
        constexpr int PU = 4;                               // periods loaded back to back (one 8-byte load in flight per lane is latency-bound)
        int p = 0;
        for (; p + PU <= np; p += PU) {
            T t[PU];
#pragma unroll
            for (int i = 0; i < PU; ++i) t[i] = sum_load<T>(x, first + (int64_t)(p + i) * pstep, in_dtype);
#pragma unroll
            for (int i = 0; i < PU; ++i) acc += t[i];       // period order, as numpy adds the rows
        }
        for (; p < np; ++p) acc += sum_load<T>(x, first + (int64_t)p * pstep, in_dtype);
        out[(q * L + j) * block_ch + c] = acc;
    }
}

} // namespace

template <typename T, typename TIN>
static int launch_cvt(const void *d_x, int64_t L, int64_t periods, int skip, int64_t row_stride, int64_t n_blocks, int64_t block_ch_stride,
                      int64_t ch_first, int64_t block_ch, void *d_out, hipStream_t st)
{
    constexpr int NE = (int)(kVecBytes / sizeof(TIN)), E = SumVec<T>::E, W = NE > E ? NE : E;
    const int64_t work = L * n_blocks * (block_ch / W);
    int64_t grid = (work + 255) / 256;
    if (grid < 1) grid = 1;
    if (grid > 65536) grid = 65536;
    // whole waves inside one contiguous run of the output: one block of channels (the run is the whole array), or rows of whole waves
    const bool runs = n_blocks == 1 || (block_ch / W) % 64 == 0;
    if (runs)
        hipLaunchKernelGGL((period_sum_cvt_kernel<T, TIN, true>), dim3((uint32_t)grid), dim3(256), 0, st, (const TIN *)d_x, L, (int)periods, skip,
                           row_stride, n_blocks, block_ch_stride, ch_first, block_ch / W, (T *)d_out);
    else
        hipLaunchKernelGGL((period_sum_cvt_kernel<T, TIN, false>), dim3((uint32_t)grid), dim3(256), 0, st, (const TIN *)d_x, L, (int)periods, skip,
                           row_stride, n_blocks, block_ch_stride, ch_first, block_ch / W, (T *)d_out);
    ZHIP(hipGetLastError());
    return ZMLS_OK;
}

int launch_period_sum(const void *d_x, int in_dtype, int out_dtype, int64_t L, int64_t periods, int skip_periods,
                      int64_t row_stride, int64_t ch_stride, int64_t n_blocks, int64_t block_ch_stride, int64_t ch_first,
                      int64_t block_ch, int64_t ch_limit, void *d_out, hipStream_t st)
{
    const int64_t last = (n_blocks - 1) * block_ch_stride + ch_first + block_ch;     // one past the last channel touched
    if (ch_limit < 0) ch_limit = last;
    if (out_dtype != ZMLS_F64 && out_dtype != ZMLS_F32) { set_error("period sums are f64 or f32"); return ZMLS_E_ARG; }
    if (periods < 1 || periods > 0x7FFFFFFF) { set_error("period count out of range"); return ZMLS_E_ARG; }
    const size_t ts = dtype_size(out_dtype);
    const int E = (int)(kVecBytes / ts);
    const int64_t total_elems = L * n_blocks * block_ch;
    const bool vec = last <= ch_limit && in_dtype == out_dtype && ch_stride == 1 && block_ch % E == 0 && ch_first % E == 0 && block_ch_stride % E == 0 &&
                     (row_stride * (int64_t)ts) % kVecBytes == 0 && ((uintptr_t)d_x % kVecBytes) == 0 && ((uintptr_t)d_out % kVecBytes) == 0;
    // the same alignment in units of the wider of the two vectors when the sample type differs from the plan's
    const size_t is = dtype_size(in_dtype);
    const int64_t W = (int64_t)(kVecBytes / (is < ts ? is : ts));
    const bool cvt = last <= ch_limit && in_dtype != out_dtype && ch_stride == 1 && block_ch % W == 0 && ch_first % W == 0 && block_ch_stride % W == 0 &&
                     (row_stride * (int64_t)is) % kVecBytes == 0 && ((uintptr_t)d_x % kVecBytes) == 0 && ((uintptr_t)d_out % kVecBytes) == 0;
    if (cvt) {
#define ZCVT(TT, CODE, TI) if (in_dtype == CODE) return launch_cvt<TT, TI>(d_x, L, periods, skip_periods, row_stride, n_blocks, block_ch_stride, ch_first, block_ch, d_out, st)
        if (out_dtype == ZMLS_F64) { ZCVT(double, ZMLS_F32, float); ZCVT(double, ZMLS_I16, int16_t); ZCVT(double, ZMLS_I32, int32_t); }
        else { ZCVT(float, ZMLS_F64, double); ZCVT(float, ZMLS_I16, int16_t); ZCVT(float, ZMLS_I32, int32_t); }
#undef ZCVT
    }
    const int64_t work = vec ? total_elems / E : total_elems;
    int64_t grid = (work + 255) / 256;
    if (grid < 1) grid = 1;
    if (grid > 65536) grid = 65536;
    if (vec) {
        if (out_dtype == ZMLS_F64)
            hipLaunchKernelGGL(period_sum_vec_kernel<double>, dim3((uint32_t)grid), dim3(256), 0, st, (const double *)d_x, L, (int)periods,
                               skip_periods, row_stride, n_blocks, block_ch_stride, ch_first, block_ch / E, (double *)d_out);
        else
            hipLaunchKernelGGL(period_sum_vec_kernel<float>, dim3((uint32_t)grid), dim3(256), 0, st, (const float *)d_x, L, (int)periods,
                               skip_periods, row_stride, n_blocks, block_ch_stride, ch_first, block_ch / E, (float *)d_out);
    } else {
        if (out_dtype == ZMLS_F64)
            hipLaunchKernelGGL(period_sum_elem_kernel<double>, dim3((uint32_t)grid), dim3(256), 0, st, d_x, in_dtype, L, (int)periods,
                               skip_periods, row_stride, ch_stride, n_blocks, block_ch_stride, ch_first, block_ch, ch_limit, (double *)d_out);
        else
            hipLaunchKernelGGL(period_sum_elem_kernel<float>, dim3((uint32_t)grid), dim3(256), 0, st, d_x, in_dtype, L, (int)periods,
                               skip_periods, row_stride, ch_stride, n_blocks, block_ch_stride, ch_first, block_ch, ch_limit, (float *)d_out);
    }
    ZHIP(hipGetLastError());
    return ZMLS_OK;
}

} // namespace zmls
