// Plan-time integer kernels: the LFSR bit sequence, the two permutation tables, the +1/-1
// stimulus, synthetic captures.
//
// Sequence definition (bit-exact contract, zignal/measure/mls.py:101-115): register of N
// cells, newest cell initialised to 1, all others 0; every step emits
//     a[n] = XOR over taps t of a[n - t].
// Here the register is kept as the N-bit WINDOW  s(n) = sum_i a[n+i] << i  (a Fibonacci LFSR):
//     s(n+1) = (s(n) >> 1) | parity(s(n) & G) << (N-1),   G = XOR_t 1 << (N-t),   a[n] = s(n) & 1.
// s(n) is exactly Ps[n] of SURVEY.md §A.2, so the same walk that emits the bits also fills the
// gather table Q[s(n)] = n. The dual (Galois) register C[k+1] = x*C[k] mod p(x), C[0] = 1, gives
// Pl[m] = C[(L-m) mod L] and with it the scatter table M[C[k]] = (L-k) mod L.
//
// Parallelism: thread t owns bits [64 t, 64 t + 64). It jumps to its start state with
// GF(2) matrix powers F^(2^k) (built on the host, zmls_host.cpp; any start costs the same ~2N
// matrix-vector products, so short runs are free) and then steps 32 bits per output word: N=20 is
// 16384 threads in 256 one-wave workgroups (every CU busy) instead of the 1024 threads a 1024-bit run gave. Integer VALU only; the
// output is tiny (2 MiB of bits, 2 x 64 MiB of tables at N = 24) and is written once per plan.
// Orders 31 and 32 (the end of zignal's tap table) get the bit sequence only: gather / scatter_nat
// are null there (zmls_plan_create explains why).
#include "zmls_internal.h"

namespace zmls {
namespace {

constexpr int kBitsPerThread = 64;
constexpr int kLfsrBlock = 64;      // one wave per workgroup: N=20 is 256 workgroups, one per CU

__global__ void __launch_bounds__(kLfsrBlock) lfsr_tables_kernel(int N, uint32_t G, uint32_t s0, const JumpTables *__restrict__ jt,
                                                          uint32_t *__restrict__ bits, uint64_t n_words,
                                                          uint32_t *__restrict__ gather, uint32_t *__restrict__ scatter_nat, int wrap)
{
    const uint64_t t = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t n0 = t * kBitsPerThread;
    if (n0 >= n_words * 32) return;
    const uint64_t L = (1ull << N) - 1;

    uint32_t s = s0;      // window at n = 0
    uint32_t c = 1u;      // C[0]
    // maximum-length registers have period L; the bit buffer runs a little past it (and past 2^32 at N = 32). Any other
    // register keeps running as the reference's does (mls.py:104-115): no wrap, and F^(2^32) is F^(2^31) twice.
    const uint64_t jump = wrap ? n0 % L : n0;
    for (int k = 0; k < 32; ++k)
        if ((jump >> k) & 1ull) {
            s = gf2_apply(jt->fib[k], s, N);
            c = gf2_apply(jt->gal[k], c, N);
        }
    if (jump >> 32) s = gf2_apply(jt->fib[31], gf2_apply(jt->fib[31], s, N), N);       // (no tables there: c is not used)
    for (int w = 0; w < kBitsPerThread / 32; ++w) {
        uint32_t word = 0;
        for (int j = 0; j < 32; ++j) {
            const uint64_t n = n0 + (uint64_t)w * 32 + j;
            if (n < L && gather) {
                gather[s] = (uint32_t)n;
                scatter_nat[c] = (n == 0) ? 0u : (uint32_t)(L - n);
            }
            word |= (s & 1u) << j;
            s = fib_step(s, G, N);
            c = gal_step(c, G, N);
        }
        const uint64_t wi = t * (kBitsPerThread / 32) + w;
        if (wi < n_words) bits[wi] = word;
    }
}

// M in the order the last pass's threads want it: entry ((macro*2^RB + reg) << TLB) + tlog is the
// response row of the element thread-logical-id tlog holds in register reg (TLB = thread logical bits)
__global__ void __launch_bounds__(256) scatter_perm_kernel(int N, int K, TileCfg cfg, int b, const uint32_t *__restrict__ nat,
                                                           uint32_t *__restrict__ perm)
{
    const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (1ull << N)) return;
    const int LT = log2_tiles(K, cfg);
    const RoundGeo g = make_round(K, cfg, false, num_rounds(K, cfg) - 1);
    const uint32_t per_macro_bits = cfg_logical_bits(cfg);
    const uint32_t tlb = cfg_thread_logical_bits(cfg);
    const uint64_t macro = idx >> per_macro_bits;
    const uint32_t within = (uint32_t)(idx & ((1u << per_macro_bits) - 1u));
    const uint32_t reg = within >> tlb;
    const uint32_t tlog = within & ((1u << tlb) - 1u);
    const uint32_t l = logical_of_thread(g, tlog) | logical_of_reg(g, reg, cfg.rb);
    perm[idx] = nat[row_of_logical(macro, l, b, K, LT)];
}

__device__ __forceinline__ uint32_t seq_bit(const uint32_t *bits, uint64_t n) { return (bits[n >> 5] >> (n & 31)) & 1u; }

template <typename T>
__global__ void __launch_bounds__(256) sequence_expand_kernel(const uint32_t *__restrict__ bits, int64_t L, T *__restrict__ out,
                                                              int64_t repeats, int64_t stride)
{
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < L; j += (int64_t)gridDim.x * blockDim.x) {
        const T v = seq_bit(bits, (uint64_t)j) ? (T)-1 : (T)1;      // bit 0 -> +1.0, bit 1 -> -1.0 (mls.py:150)
        for (int64_t r = 0; r < repeats; ++r) out[(r * L + j) * stride] = v;
    }
}

// ---- synthetic captures -----------------------------------------------------------------
ZHD uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
ZHD double unit_uniform(uint64_t h) { return (double)(h >> 11) * (1.0 / 9007199254740992.0); }   // [0,1)
ZHD double synth_tap(int64_t channel, int k, uint64_t seed)
{
    if (k == 0) return 1.0;
    double decay = 1.0;
    for (int i = 0; i < k; ++i) decay *= 0.75;
    return (2.0 * unit_uniform(splitmix64(seed ^ splitmix64((uint64_t)channel * 131u + (uint64_t)k))) - 1.0) * decay;
}
ZHD int64_t synth_delay(int64_t channel, int64_t L) { return (channel * 9973) % L; }
ZHD double synth_noise(uint64_t seed, int64_t channel, int64_t sample)
{
    // sum of four 16-bit uniforms: zero mean, variance 1/3, cheap and reproducible everywhere
    const uint64_t h = splitmix64(seed ^ splitmix64((uint64_t)channel) ^ ((uint64_t)sample * 0xD1342543DE82EF95ull));
    const double s = (double)(h & 0xFFFF) + (double)((h >> 16) & 0xFFFF) + (double)((h >> 32) & 0xFFFF) + (double)(h >> 48);
    return (s - 2.0 * 65535.0) * (1.0 / 65536.0);
}

template <typename T>
__global__ void __launch_bounds__(256) synth_kernel(const uint32_t *__restrict__ bits, int64_t L, T *__restrict__ x, int64_t repeats,
                                                    int64_t n_ch, int64_t ch_first, int64_t row_stride, int64_t ch_stride,
                                                    uint64_t seed, double sigma, int ch_fastest)
{
    const int64_t total = L * n_ch;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t j = ch_fastest ? idx / n_ch : idx % L;
        const int64_t cl = ch_fastest ? idx % n_ch : idx / L;
        const int64_t c = ch_first + cl;
        const int64_t d = synth_delay(c, L);
        double y = 0.0;
        for (int k = 0; k < kSynthTaps; ++k) {
            int64_t n = j - d - k;
            n %= L;
            if (n < 0) n += L;
            y += synth_tap(c, k, seed) * (seq_bit(bits, (uint64_t)n) ? -1.0 : 1.0);
        }
        for (int64_t r = 0; r < repeats; ++r) {
            const int64_t row = r * L + j;
            x[row * row_stride + cl * ch_stride] = (T)(y + sigma * synth_noise(seed, c, row));
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(256) extract_channel_kernel(T *__restrict__ dst, const T *__restrict__ src, int64_t n_rows,
                                                              int64_t row_stride, int64_t offset)
{
    for (int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; j < n_rows; j += (int64_t)gridDim.x * blockDim.x)
        dst[j] = src[j * row_stride + offset];
}

// trim_and_avgerage on its own (mls.py:284-306): drop period 0, add periods 1..R-1 in order,
// divide by their count. ACC is float for float32 captures (numpy averages those in float32).
template <typename TIN, typename ACC>
__global__ void __launch_bounds__(256) average_kernel(const TIN *__restrict__ x, int64_t L, int64_t periods, int64_t n_ch,
                                                      int64_t row_stride, int64_t ch_stride, ACC *__restrict__ out)
{
    const int64_t total = L * n_ch;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t j = idx / n_ch, c = idx % n_ch;
        ACC s = (ACC)x[(L + j) * row_stride + c * ch_stride];
        for (int64_t p = 2; p <= periods; ++p) s += (ACC)x[(p * L + j) * row_stride + c * ch_stride];
        out[idx] = s / (ACC)periods;
    }
}

inline uint32_t grid_for(int64_t n, int cap = 8192)
{
    int64_t g = (n + 255) / 256;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (uint32_t)g;
}

} // namespace

int launch_lfsr_tables(int N, uint32_t G, uint32_t s0, const JumpTables *d_jt, uint32_t *d_bits, uint64_t n_words,
                       uint32_t *d_gather, uint32_t *d_scatter_nat, int wrap, hipStream_t st)
{
    const uint64_t threads = (n_words * 32 + kBitsPerThread - 1) / kBitsPerThread;
    const uint32_t blocks = (uint32_t)((threads + kLfsrBlock - 1) / kLfsrBlock);
    hipLaunchKernelGGL(lfsr_tables_kernel, dim3(blocks), dim3(kLfsrBlock), 0, st, N, G, s0, d_jt, d_bits, n_words, d_gather, d_scatter_nat, wrap);
    ZHIP(hipGetLastError());
    return ZMLS_OK;
}

int launch_scatter_perm(int N, int K, TileCfg cfg, int b, const uint32_t *d_scatter_nat, uint32_t *d_scatter_perm, hipStream_t st)
{
    const uint64_t n = 1ull << N;
    hipLaunchKernelGGL(scatter_perm_kernel, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, st, N, K, cfg, b, d_scatter_nat, d_scatter_perm);
    ZHIP(hipGetLastError());
    return ZMLS_OK;
}

int launch_sequence_expand(const uint32_t *d_bits, int64_t L, void *d_out, int dtype, int64_t repeats, int64_t stride, hipStream_t st)
{
    if (dtype == ZMLS_F64)
        hipLaunchKernelGGL(sequence_expand_kernel<double>, dim3(grid_for(L)), dim3(256), 0, st, d_bits, L, (double *)d_out, repeats, stride);
    else if (dtype == ZMLS_F32)
        hipLaunchKernelGGL(sequence_expand_kernel<float>, dim3(grid_for(L)), dim3(256), 0, st, d_bits, L, (float *)d_out, repeats, stride);
    else { set_error("sequence dtype must be f64 or f32"); return ZMLS_E_ARG; }
    ZHIP(hipGetLastError());
    return ZMLS_OK;
}

int launch_synth(const uint32_t *d_bits, int64_t L, void *d_x, int dtype, int64_t repeats, int64_t n_ch, int64_t ch_first,
                 int64_t row_stride, int64_t ch_stride, uint64_t seed, double sigma, hipStream_t st)
{
    const int ch_fastest = (ch_stride <= row_stride) ? 1 : 0;
    const uint32_t grid = grid_for(L * n_ch, 65536);
    if (dtype == ZMLS_F64)
        hipLaunchKernelGGL(synth_kernel<double>, dim3(grid), dim3(256), 0, st, d_bits, L, (double *)d_x, repeats, n_ch, ch_first,
                           row_stride, ch_stride, seed, sigma, ch_fastest);
    else if (dtype == ZMLS_F32)
        hipLaunchKernelGGL(synth_kernel<float>, dim3(grid), dim3(256), 0, st, d_bits, L, (float *)d_x, repeats, n_ch, ch_first,
                           row_stride, ch_stride, seed, sigma, ch_fastest);
    else { set_error("synthetic captures are f64 or f32"); return ZMLS_E_ARG; }
    ZHIP(hipGetLastError());
    return ZMLS_OK;
}

int launch_average(const void *d_x, int in_dtype, int64_t L, int64_t periods, int64_t n_ch, int64_t row_stride,
                   int64_t ch_stride, void *d_out, hipStream_t st)
{
    const uint32_t grid = grid_for(L * n_ch, 65536);
    switch (in_dtype) {
    case ZMLS_F64: hipLaunchKernelGGL((average_kernel<double, double>), dim3(grid), dim3(256), 0, st, (const double *)d_x, L, periods, n_ch, row_stride, ch_stride, (double *)d_out); break;
    case ZMLS_F32: hipLaunchKernelGGL((average_kernel<float, float>), dim3(grid), dim3(256), 0, st, (const float *)d_x, L, periods, n_ch, row_stride, ch_stride, (float *)d_out); break;
    case ZMLS_I16: hipLaunchKernelGGL((average_kernel<int16_t, double>), dim3(grid), dim3(256), 0, st, (const int16_t *)d_x, L, periods, n_ch, row_stride, ch_stride, (double *)d_out); break;
    case ZMLS_I32: hipLaunchKernelGGL((average_kernel<int32_t, double>), dim3(grid), dim3(256), 0, st, (const int32_t *)d_x, L, periods, n_ch, row_stride, ch_stride, (double *)d_out); break;
    default: set_error("bad dtype"); return ZMLS_E_ARG;
    }
    ZHIP(hipGetLastError());
    return ZMLS_OK;
}

void synth_system(int64_t L, int64_t channel, uint64_t seed, double *taps, int64_t *delay)
{
    for (int k = 0; k < kSynthTaps; ++k) taps[k] = synth_tap(channel, k, seed);
    *delay = synth_delay(channel, L);
}

int launch_extract_channel(void *d_dst, const void *d_src, int dtype, int64_t n_rows, int64_t row_stride, int64_t ch_stride,
                           int64_t channel, hipStream_t st)
{
    const int64_t off = channel * ch_stride;
    switch (dtype) {
    case ZMLS_F64: hipLaunchKernelGGL(extract_channel_kernel<double>, dim3(grid_for(n_rows)), dim3(256), 0, st, (double *)d_dst, (const double *)d_src, n_rows, row_stride, off); break;
    case ZMLS_F32: hipLaunchKernelGGL(extract_channel_kernel<float>, dim3(grid_for(n_rows)), dim3(256), 0, st, (float *)d_dst, (const float *)d_src, n_rows, row_stride, off); break;
    case ZMLS_I16: hipLaunchKernelGGL(extract_channel_kernel<int16_t>, dim3(grid_for(n_rows)), dim3(256), 0, st, (int16_t *)d_dst, (const int16_t *)d_src, n_rows, row_stride, off); break;
    case ZMLS_I32: hipLaunchKernelGGL(extract_channel_kernel<int32_t>, dim3(grid_for(n_rows)), dim3(256), 0, st, (int32_t *)d_dst, (const int32_t *)d_src, n_rows, row_stride, off); break;
    default: set_error("bad dtype"); return ZMLS_E_ARG;
    }
    ZHIP(hipGetLastError());
    return ZMLS_OK;
}

} // namespace zmls
