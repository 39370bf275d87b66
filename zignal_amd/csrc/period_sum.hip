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
        for (int p = 0; p < (ch < ch_limit ? periods : 0); ++p)
            acc += sum_load<T>(x, ((int64_t)(skip + p) * L + j) * row_stride + ch * ch_stride, in_dtype);
        out[(q * L + j) * block_ch + c] = acc;
    }
}

} // namespace

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
