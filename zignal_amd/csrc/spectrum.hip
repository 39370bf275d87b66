// Magnitude response of an impulse response on the device — SURVEY.md §8(f) row 2.
//
// Reference: Audio.fft (zignal/audio.py:678-704) as MLS_simple.plot_fft / ex_mls.py:80 use it on the recovered
// response: window the L samples, np.fft.fft with n = fftsize (L, or L+1 when L is one of the Mersenne primes
// 2^13-1, 2^17-1, 2^19-1, 2^31-1, audio.py:682-685), optionally divide by fftsize, 20 log10 |Y|, keep bins
// 0 .. int(fftsize/2)-1. Only the spectrum (half the response's size) then has to cross PCIe.
//
// The DFT is this library's own (fft.hip, fft_tile.h; round 3 - rounds 1 and 2 called hipFFT, which ran every L = 2^N - 1 as a
// Bluestein convolution of thirteen kernels): power-of-two sizes (the bumped Mersenne sizes) as forward passes whose output
// order the dB kernel gathers from, every other size as a Bluestein convolution fused into four passes. Around it two
// streaming kernels: window + transpose into pair-major staging, and |Y| -> dB back into (bins, channels).
// Two real channels ride in one complex transform, z = x1 + i x2, and come apart afterwards as
// X1[k] = (Z[k] + conj Z[F-k]) / 2, X2[k] = (Z[k] - conj Z[F-k]) / 2i - half the transforms (a lone last channel rides with zeros).
// The rounding error of one transform is relative to the larger of its two channels, so the second channel of a pair is first
// brought to the level of the first by a power of two taken from the channels' peaks (exact, undone after the transform): a quiet
// channel beside a loud one keeps the accuracy a transform of its own would give it (tests: channels 200 dB apart).
#include <algorithm>
#include <stdlib.h>
#include <string.h>
#include "zmls_internal.h"
#include "fft_tile.h"

namespace zmls {
namespace fft {
int forward(const Plan &p, PassArgs a, int load_mode, int store_mode, hipStream_t st, bool device);
int inverse(const Plan &p, PassArgs a, int store_mode, hipStream_t st, bool device);
}
namespace {

// where bin k of a transform lies: natural order (Bluestein's output), or the forward passes' own order (power-of-two sizes)
struct BinOrder {
    int permuted;
    fft::Plan plan;
};
__device__ __forceinline__ int64_t bin_at(const BinOrder &o, int64_t k) { return o.permuted ? (int64_t)fft::position_of(o.plan, (uint64_t)k) : k; }

// packed mode: peak[c] = max_t |h[t][c] * window[t]| as the bit pattern of a non-negative double (ordered like the integers)
__global__ void __launch_bounds__(256) spectrum_peak_kernel(const double *__restrict__ h, int64_t n, int64_t n_ch, int64_t row_stride,
                                                            int64_t ch_stride, const double *__restrict__ window, unsigned long long *__restrict__ peak)
{
    __shared__ unsigned long long red[256];
    for (int64_t c = blockIdx.y; c < n_ch; c += gridDim.y) {
        double m = 0.0;
        for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < n; t += (int64_t)gridDim.x * blockDim.x) {
            const double v = fabs(h[t * row_stride + c * ch_stride] * (window ? window[t] : 1.0));
            if (v > m && v <= 1.7976931348623157e308) m = v;        // NaN and infinity are left to the transform
        }
        red[threadIdx.x] = (unsigned long long)__double_as_longlong(m);
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s && red[threadIdx.x + s] > red[threadIdx.x]) red[threadIdx.x] = red[threadIdx.x + s];
            __syncthreads();
        }
        if (threadIdx.x == 0 && red[0]) atomicMax(peak + c, red[0]);
        __syncthreads();
    }
}

// the power of two that brings the second channel of pair p to the level of the first (0 when either is silent or alone)
__device__ __forceinline__ int pair_shift(const unsigned long long *__restrict__ peak, int64_t p, int64_t n_ch)
{
    if (2 * p + 1 >= n_ch) return 0;
    const double a = __longlong_as_double((long long)peak[2 * p]), b = __longlong_as_double((long long)peak[2 * p + 1]);
    if (!(a > 0.0) || !(b > 0.0)) return 0;
    return ilogb(a) - ilogb(b);
}

// packed mode: stage[p][t] = (h[t][2p], 2^shift h[t][2p+1]) * window[t]; a missing second channel is zero
__global__ void __launch_bounds__(256) spectrum_prepare_pairs_kernel(const double *__restrict__ h, int64_t n, int64_t n_ch, int64_t row_stride,
                                                                     int64_t ch_stride, const double *__restrict__ window, int64_t n_pairs,
                                                                     const unsigned long long *__restrict__ peak, double2 *__restrict__ stage)
{
    const int64_t total = n * n_pairs;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = idx / n, t = idx - p * n;
        const double w = window ? window[t] : 1.0;
        double2 v;
        v.x = h[t * row_stride + (2 * p) * ch_stride] * w;
        v.y = (2 * p + 1 < n_ch) ? ldexp(h[t * row_stride + (2 * p + 1) * ch_stride] * w, pair_shift(peak, p, n_ch)) : 0.0;
        stage[idx] = v;
    }
}

// 20 log10(g sqrt(re^2 + im^2)) as 10 log10(re^2 + im^2) + 20 log10(g): one logarithm and no hypot (its scaling and square root
// are a third of this kernel's arithmetic); sums of squares near the ends of the double range take the long way
__device__ __forceinline__ double db_of(double re, double im, double gain_db, double gain)
{
    const double ss = re * re + im * im;
    if (ss > 1e-290 && ss < 1e290) return 10.0 * log10(ss) + gain_db;
    return 20.0 * log10(hypot(re, im) * gain);
}

// packed mode: the two spectra of a pair from Z[k] and Z[F-k]
__global__ void __launch_bounds__(256) spectrum_db_pairs_kernel(const double2 *__restrict__ Y, int64_t F, int64_t n_bins, int64_t n_ch, int64_t n_pairs,
                                                                double scale, const unsigned long long *__restrict__ peak, double *__restrict__ mag,
                                                                const BinOrder order)
{
    const int64_t total = n_bins * n_pairs;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = idx / n_bins, k = idx - p * n_bins;
        const double2 a = Y[p * F + bin_at(order, k)], b = Y[p * F + bin_at(order, k ? F - k : 0)];
        // conj(b) = (b.x, -b.y):  X1 = (a + conj b) / 2,  |X2| = |a - conj b| / 2
        const double g1 = 0.5 * scale, g2 = ldexp(0.5 * scale, -pair_shift(peak, p, n_ch));
        mag[k * n_ch + 2 * p] = db_of(a.x + b.x, a.y - b.y, 20.0 * log10(g1), g1);
        if (2 * p + 1 < n_ch) mag[k * n_ch + 2 * p + 1] = db_of(a.x - b.x, a.y + b.y, 20.0 * log10(g2), g2);
    }
}

// ---- the same three steps of the packed mode for frame-major responses (channels adjacent in memory, the layout the
// impulse entry points write): adjacent lanes walk CHANNELS on the response side and FRAMES / BINS on the transform side, so a
// 64 x 64 tile crosses LDS in between and both sides move whole 512-byte rows. (One element per lane with frames on the lanes
// read 8 bytes out of every 512: the three kernels took 36 % of the call, profiles/r03_spectrum_kernel_stats.csv.)
constexpr int kSpT = 64;            // frames (bins) and channels per tile
constexpr size_t kSpectrumKeepBytes = (size_t)256 << 20;    // host-call staging kept with the object up to this size
// (measured, profiles/r03_spectrum_own_fft.txt: 256 MiB best of 32 ... 512 MiB and all at once)
constexpr size_t kSpectrumChunkBytes = (size_t)256 << 20;

__global__ void __launch_bounds__(256) spectrum_peak_rows_kernel(const double *__restrict__ h, int64_t n, int64_t n_ch, int64_t row_stride,
                                                                 const double *__restrict__ window, int64_t rows_per_block,
                                                                 unsigned long long *__restrict__ peak)
{
    __shared__ double red[256];
    const int lane_c = threadIdx.x & (kSpT - 1), rg = threadIdx.x >> 6;           // channel within the tile, row group 0..3
    const int64_t c = (int64_t)blockIdx.y * kSpT + lane_c;
    const int64_t t_begin = (int64_t)blockIdx.x * rows_per_block, t_end = t_begin + rows_per_block < n ? t_begin + rows_per_block : n;
    double m = 0.0;
    if (c < n_ch) {
        for (int64_t t = t_begin + rg; t < t_end; t += 4) {
            const double v = fabs(h[t * row_stride + c] * (window ? window[t] : 1.0));
            if (v > m && v <= 1.7976931348623157e308) m = v;
        }
    }
    red[threadIdx.x] = m;
    __syncthreads();
    if (rg == 0 && c < n_ch) {
        m = fmax(fmax(red[lane_c], red[64 + lane_c]), fmax(red[128 + lane_c], red[192 + lane_c]));
        if (m > 0.0) atomicMax(peak + c, (unsigned long long)__double_as_longlong(m));
    }
}

// ... with 16-byte loads (even channel counts on aligned rows): lane pr holds channels 2 pr and 2 pr + 1, eight row groups
__global__ void __launch_bounds__(256) spectrum_peak_rows16_kernel(const double *__restrict__ h, int64_t n, int64_t n_ch, int64_t row_stride,
                                                                   const double *__restrict__ window, int64_t rows_per_block,
                                                                   unsigned long long *__restrict__ peak)
{
    typedef double D2 __attribute__((ext_vector_type(2)));
    __shared__ double red[2][256];
    const int pr = threadIdx.x & 31, rg = threadIdx.x >> 5;
    const int64_t c = (int64_t)blockIdx.y * kSpT + 2 * pr;
    const int64_t t_begin = (int64_t)blockIdx.x * rows_per_block, t_end = t_begin + rows_per_block < n ? t_begin + rows_per_block : n;
    double m0 = 0.0, m1 = 0.0;
    if (c < n_ch) {
        for (int64_t t = t_begin + rg; t < t_end; t += 8) {
            const D2 v = __builtin_nontemporal_load((const D2 *)(h + t * row_stride + c));
            const double w = window ? window[t] : 1.0, a = fabs(v[0] * w), b = fabs(v[1] * w);
            if (a > m0 && a <= 1.7976931348623157e308) m0 = a;
            if (b > m1 && b <= 1.7976931348623157e308) m1 = b;
        }
    }
    red[0][threadIdx.x] = m0;
    red[1][threadIdx.x] = m1;
    __syncthreads();
    if (rg == 0 && c < n_ch) {
#pragma unroll
        for (int g = 1; g < 8; ++g) { m0 = fmax(m0, red[0][g * 32 + pr]); m1 = fmax(m1, red[1][g * 32 + pr]); }
        if (m0 > 0.0) atomicMax(peak + c, (unsigned long long)__double_as_longlong(m0));
        if (m1 > 0.0) atomicMax(peak + c + 1, (unsigned long long)__double_as_longlong(m1));
    }
}

__global__ void __launch_bounds__(256) spectrum_prepare_pairs_rows_kernel(const double *__restrict__ h, int64_t n, int64_t n_ch, int64_t row_stride,
                                                                          const double *__restrict__ window, int64_t n_pairs,
                                                                          const unsigned long long *__restrict__ peak, double2 *__restrict__ stage)
{
    __shared__ double tile[kSpT][kSpT + 1];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t t0 = (int64_t)blockIdx.x * kSpT, c0 = (int64_t)blockIdx.y * kSpT;
    for (int r = wv; r < kSpT; r += 4) {
        const int64_t t = t0 + r, c = c0 + lane;
        tile[r][lane] = (t < n && c < n_ch) ? h[t * row_stride + c] * (window ? window[t] : 1.0) : 0.0;
    }
    __syncthreads();
    // pair pl of the tile, frame lane: 16 bytes per lane, 1 KiB per wave, contiguous in stage[p][t0 ...]
    for (int pl = wv; pl < kSpT / 2; pl += 4) {
        const int64_t p = c0 / 2 + pl, t = t0 + lane;
        if (p < n_pairs && t < n) {
            double2 v;
            v.x = tile[lane][2 * pl];
            v.y = ldexp(tile[lane][2 * pl + 1], pair_shift(peak, p, n_ch));
            stage[p * n + t] = v;
        }
    }
}

// the same for even channel counts on 16-byte-aligned rows: a pair IS 16 adjacent bytes of a frame, so the response side moves
// double2 as well (32 lanes x 16 bytes per row, eight rows per load instruction, all eight loads of a thread in flight; 8-byte
// loads run at 0.5-0.7 of the 16-byte rate)
__global__ void __launch_bounds__(256) spectrum_prepare_pairs_rows16_kernel(const double *__restrict__ h, int64_t n, int64_t n_ch, int64_t row_stride,
                                                                            const double *__restrict__ window, int64_t n_pairs,
                                                                            const unsigned long long *__restrict__ peak, double2 *__restrict__ stage)
{
    typedef double D2 __attribute__((ext_vector_type(2)));
    __shared__ D2 tile[kSpT][kSpT / 2 + 1];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int pr = threadIdx.x & 31, r0 = threadIdx.x >> 5;            // pair of the tile, first of this thread's eight rows
    const int64_t t0 = (int64_t)blockIdx.x * kSpT, c0 = (int64_t)blockIdx.y * kSpT;
    const bool have = c0 + 2 * pr < n_ch;
    D2 v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int64_t t = t0 + r0 + 8 * i;
        v[i] = (have && t < n) ? __builtin_nontemporal_load((const D2 *)(h + t * row_stride + c0 + 2 * pr)) : (D2)0.0;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int64_t t = t0 + r0 + 8 * i;
        tile[r0 + 8 * i][pr] = (window && t < n) ? v[i] * window[t] : v[i];
    }
    __syncthreads();
    for (int pl = wv; pl < kSpT / 2; pl += 4) {
        const int64_t p = c0 / 2 + pl, t = t0 + lane;
        if (p < n_pairs && t < n) {
            const D2 u = tile[lane][pl];
            double2 o;
            o.x = u[0];
            o.y = ldexp(u[1], pair_shift(peak, p, n_ch));
            stage[p * n + t] = o;
        }
    }
}

__global__ void __launch_bounds__(256) spectrum_db_pairs_rows_kernel(const double2 *__restrict__ Y, int64_t F, int64_t n_bins, int64_t n_ch, int64_t n_pairs,
                                                                     double scale, const unsigned long long *__restrict__ peak, double *__restrict__ mag,
                                                                     const BinOrder order)
{
    __shared__ double tile[kSpT][kSpT + 1];
    __shared__ double gain[kSpT], gain_db[kSpT];        // per channel of the tile: 1/2 x scale x the undone level shift, and its dB
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t k0 = (int64_t)blockIdx.x * kSpT, c0 = (int64_t)blockIdx.y * kSpT;
    if (threadIdx.x < kSpT) {
        const int64_t c = c0 + threadIdx.x;
        const double g = (c & 1) && c / 2 < n_pairs ? ldexp(0.5 * scale, -pair_shift(peak, c / 2, n_ch)) : 0.5 * scale;
        gain[threadIdx.x] = g;
        gain_db[threadIdx.x] = 20.0 * log10(g);
    }
    __syncthreads();
    for (int pl = wv; pl < kSpT / 2; pl += 4) {
        const int64_t p = c0 / 2 + pl, k = k0 + lane;
        double m1 = 0.0, m2 = 0.0;
        if (p < n_pairs && k < n_bins) {
            const double2 a = Y[p * F + bin_at(order, k)], b = Y[p * F + bin_at(order, k ? F - k : 0)];
            m1 = db_of(a.x + b.x, a.y - b.y, gain_db[2 * pl], gain[2 * pl]);
            m2 = db_of(a.x - b.x, a.y + b.y, gain_db[2 * pl + 1], gain[2 * pl + 1]);
        }
        tile[lane][2 * pl] = m1;
        tile[lane][2 * pl + 1] = m2;
    }
    __syncthreads();
    for (int r = wv; r < kSpT; r += 4) {
        const int64_t k = k0 + r, c = c0 + lane;
        if (k < n_bins && c < n_ch) mag[k * n_ch + c] = tile[r][lane];
    }
}

bool mersenne_prime_size(int64_t n)
{
    return n == ((int64_t)1 << 13) - 1 || n == ((int64_t)1 << 17) - 1 || n == ((int64_t)1 << 19) - 1 || n == ((int64_t)1 << 31) - 1;
}

} // namespace
} // namespace zmls

struct zmls_spectrum {
    int device;
    int64_t n, n_ch, fftsize, n_bins;
    int64_t n_pairs;        // two channels per complex transform (the last one alone when n_ch is odd)
    int64_t chunk_pairs;    // Bluestein: pairs that go through the passes together (the work array holds that many transforms)
    int bluestein;          // 0: fftsize is a power of two, forward passes only; 1: chirp-z on M = 2^m >= 2 fftsize - 1 points
    zmls::fft::Plan plan;        // passes of the 2^m-point transform
    double2 *d_stage;       // (n_pairs, n) windowed pairs
    double2 *d_work;        // (n_pairs, 2^m) the transforms' array; Bluestein: (chunk_pairs, 2^m)
    double2 *d_Y;           // Bluestein: (n_pairs, fftsize) spectra in natural order (power of two: d_work in the passes' order)
    double2 *d_table;       // Bluestein: spectrum of the chirp filter, 2^m points in the passes' order
    unsigned long long *d_peak;   // per-channel peak of the windowed samples (bit pattern of a double)
    double *d_host[3];      // zmls_spectrum_host: response, spectrum and window on the device, kept between calls while small
};

using namespace zmls;

extern "C" {

int zmls_spectrum_create(zmls_spectrum **out, int64_t n_samples, int64_t n_ch, int device)
{
    if (!out) { set_error("null argument"); return ZMLS_E_ARG; }
    *out = nullptr;
    if (n_samples < 2 || n_ch < 1 || n_samples >= ((int64_t)1 << 31) - 1) { set_error("bad spectrum size"); return ZMLS_E_ARG; }
    ZDEVICE(device);
    int rc = ZMLS_OK;
    zmls_spectrum *s = (zmls_spectrum *)calloc(1, sizeof(zmls_spectrum));
    if (!s) { set_error("out of host memory"); return ZMLS_E_ARG; }
    s->device = device;
    s->n = n_samples; s->n_ch = n_ch;
    s->fftsize = n_samples + (mersenne_prime_size(n_samples) ? 1 : 0);      // audio.py:682-685
    s->n_bins = s->fftsize / 2;                                             // int(fftsize/2), audio.py:701-702
    s->n_pairs = (n_ch + 1) / 2;
    s->bluestein = (s->fftsize & (s->fftsize - 1)) != 0;
    int m = 1;
    while (((int64_t)1 << m) < (s->bluestein ? 2 * s->fftsize - 1 : s->fftsize)) ++m;
    s->plan = fft::make_plan(m);
    const size_t M = (size_t)1 << m;
    if ((((uint64_t)s->n_pairs << m) >> fft::kTileLog) > 0x7FFFFFFFull) {
        free(s);
        set_error("spectrum of %lld channels x %lld points: more than 2^31 tiles in one pass", (long long)n_ch, (long long)s->fftsize);
        return ZMLS_E_UNSUPPORTED;
    }
    // Bluestein runs a chunk of pairs through its four passes before the next chunk: work array of kSpectrumChunkBytes (one pair
    // at least) instead of n_pairs x 2^m points (N=20 x 64 channels: 256 MiB instead of 1 GiB)
    static const size_t chunk_bytes = [] {
        const char *v = getenv("ZMLS_SPECTRUM_CHUNK_MIB");          // tuning
        return (v && *v) ? (size_t)strtoull(v, nullptr, 10) << 20 : kSpectrumChunkBytes;
    }();
    s->chunk_pairs = s->bluestein ? std::max<int64_t>(1, std::min<int64_t>(s->n_pairs, (int64_t)(chunk_bytes / (M * 16)))) : s->n_pairs;
    hipError_t e = hipMalloc((void **)&s->d_stage, (size_t)s->n * s->n_pairs * 16);
    if (e == hipSuccess) e = hipMalloc((void **)&s->d_work, M * (size_t)s->chunk_pairs * 16);
    if (e == hipSuccess && s->bluestein) e = hipMalloc((void **)&s->d_Y, (size_t)s->fftsize * s->n_pairs * 16);
    if (e == hipSuccess && s->bluestein) e = hipMalloc((void **)&s->d_table, M * 16);
    if (e == hipSuccess) e = hipMalloc((void **)&s->d_peak, (size_t)n_ch * 8);
    if (e != hipSuccess) {
        rc = hip_fail(e, "spectrum buffers");
        (void)zmls_spectrum_destroy(s);
        return rc;
    }
    if (s->bluestein) {
        // the chirp filter b[u] = exp(+i pi u^2 / F), |u| < F, wrapped around M: its spectrum, once per size, by the same passes
        fft::PassArgs b{};
        b.work = (fft::cd *)s->d_table; b.n_transforms = 1; b.F = (uint64_t)s->fftsize;
        rc = fft::forward(s->plan, b, fft::LOAD_BSEQ, fft::STORE_WORK, nullptr, true);
        if (!rc) { hipError_t es = hipStreamSynchronize(nullptr); if (es != hipSuccess) rc = hip_fail(es, "chirp filter spectrum"); }
        if (rc) { (void)zmls_spectrum_destroy(s); return rc; }
    }
    *out = s;
    return ZMLS_OK;
}

int zmls_spectrum_destroy(zmls_spectrum *s)
{
    if (!s) return ZMLS_OK;
    DeviceScope on_device(s->device);
    if (!on_device.rc) {
        if (s->d_stage) (void)hipFree(s->d_stage);
        if (s->d_work) (void)hipFree(s->d_work);
        if (s->d_Y) (void)hipFree(s->d_Y);
        if (s->d_table) (void)hipFree(s->d_table);
        if (s->d_peak) (void)hipFree(s->d_peak);
        for (double *q : s->d_host)
            if (q) (void)hipFree(q);
    }
    free(s);
    return ZMLS_OK;
}

int zmls_spectrum_info(const zmls_spectrum *s, int64_t info[4])
{
    if (!s || !info) { set_error("null argument"); return ZMLS_E_ARG; }
    info[0] = s->n; info[1] = s->n_ch; info[2] = s->fftsize; info[3] = s->n_bins;
    return ZMLS_OK;
}

int zmls_spectrum_device(zmls_spectrum *s, const double *d_h, int64_t row_stride, int64_t ch_stride, const double *d_window, int normalise,
                         double *d_mag, void *stream)
{
    if (!s || !d_h || !d_mag) { set_error("null argument"); return ZMLS_E_ARG; }
    if (row_stride < 1 || ch_stride < 1) { set_error("strides must be positive"); return ZMLS_E_ARG; }
    ZDEVICE(s->device);
    int rc = ZMLS_OK;
    hipStream_t st = (hipStream_t)stream;
    auto grid = [](int64_t n) { int64_t g = (n + 255) / 256; return (uint32_t)(g < 1 ? 1 : g > 65536 ? 65536 : g); };
    ZHIP(hipMemsetAsync(s->d_peak, 0, (size_t)s->n_ch * 8, st));
    const uint32_t c_tiles = (uint32_t)((s->n_ch + kSpT - 1) / kSpT);
    // frame-major responses: whole rows of adjacent channels on the response side, whole runs of frames / bins on the other.
    // Below eight channels a row is shorter than 64 bytes and the tile kernels would keep one lane in eight or fewer busy (one
    // channel of 2^20 points: 253 us in them against 76 us in the four transform passes); frames on the lanes then.
    const bool rows = ch_stride == 1 && s->n_ch >= 8 && c_tiles <= 65535u;
    if (rows) {
        const int64_t rows_per_block = std::max<int64_t>(kSpT, (s->n + 4095) / 4096);
        const bool wide = s->n_ch % 2 == 0 && row_stride % 2 == 0 && ((uintptr_t)d_h & 15) == 0;
        if (wide)
            hipLaunchKernelGGL(spectrum_peak_rows16_kernel, dim3((uint32_t)((s->n + rows_per_block - 1) / rows_per_block), c_tiles), dim3(256), 0, st, d_h, s->n,
                               s->n_ch, row_stride, d_window, rows_per_block, s->d_peak);
        else
            hipLaunchKernelGGL(spectrum_peak_rows_kernel, dim3((uint32_t)((s->n + rows_per_block - 1) / rows_per_block), c_tiles), dim3(256), 0, st, d_h, s->n,
                               s->n_ch, row_stride, d_window, rows_per_block, s->d_peak);
        if (wide)
            hipLaunchKernelGGL(spectrum_prepare_pairs_rows16_kernel, dim3((uint32_t)((s->n + kSpT - 1) / kSpT), c_tiles), dim3(256), 0, st, d_h, s->n, s->n_ch,
                               row_stride, d_window, s->n_pairs, s->d_peak, s->d_stage);
        else
            hipLaunchKernelGGL(spectrum_prepare_pairs_rows_kernel, dim3((uint32_t)((s->n + kSpT - 1) / kSpT), c_tiles), dim3(256), 0, st, d_h, s->n, s->n_ch,
                               row_stride, d_window, s->n_pairs, s->d_peak, s->d_stage);
    } else {
        hipLaunchKernelGGL(spectrum_peak_kernel, dim3(std::min<uint32_t>(grid(s->n), 1024u), (uint32_t)std::min<int64_t>(s->n_ch, 65535)), dim3(256), 0, st, d_h, s->n, s->n_ch,
                           row_stride, ch_stride, d_window, s->d_peak);
        hipLaunchKernelGGL(spectrum_prepare_pairs_kernel, dim3(grid(s->n * s->n_pairs)), dim3(256), 0, st, d_h, s->n, s->n_ch, row_stride, ch_stride,
                           d_window, s->n_pairs, s->d_peak, s->d_stage);
    }
    ZHIP(hipGetLastError());
    // the transforms: samples beyond n (the size bump) and beyond fftsize (Bluestein's padding) are zeros the first pass makes up
    fft::PassArgs a{};
    a.work = (fft::cd *)s->d_work; a.n_transforms = (uint64_t)s->n_pairs;
    a.stage = (const fft::cd *)s->d_stage; a.n_stage = (uint64_t)s->n; a.F = (uint64_t)s->fftsize;
    BinOrder order{};
    const double2 *Y = s->d_work;
    if (s->bluestein) {
        // a few pairs at a time through all four passes: the work array is only that large, and what one pass writes the next
        // reads back from the Infinity Cache (zmls_spectrum_create)
        a.table = (const fft::cd *)s->d_table; a.scale = ldexp(1.0, -s->plan.m);
        for (int64_t p0 = 0; p0 < s->n_pairs && !rc; p0 += s->chunk_pairs) {
            a.n_transforms = (uint64_t)std::min(s->chunk_pairs, s->n_pairs - p0);
            a.stage = (const fft::cd *)s->d_stage + p0 * s->n;
            a.Y = (fft::cd *)s->d_Y + p0 * s->fftsize;
            rc = fft::forward(s->plan, a, fft::LOAD_STAGE_CHIRP, fft::STORE_WORK_TIMES_TABLE, st, true);
            if (!rc) rc = fft::inverse(s->plan, a, fft::STORE_FINAL_CHIRP, st, true);
        }
        Y = s->d_Y;
    } else {
        rc = fft::forward(s->plan, a, fft::LOAD_STAGE, fft::STORE_WORK, st, true);
        order.permuted = 1; order.plan = s->plan;
    }
    if (rc) return rc;
    const double scale = normalise ? 1.0 / (double)s->fftsize : 1.0;
    if (rows)
        hipLaunchKernelGGL(spectrum_db_pairs_rows_kernel, dim3((uint32_t)((s->n_bins + kSpT - 1) / kSpT), c_tiles), dim3(256), 0, st, Y, s->fftsize,
                           s->n_bins, s->n_ch, s->n_pairs, scale, s->d_peak, d_mag, order);
    else
        hipLaunchKernelGGL(spectrum_db_pairs_kernel, dim3(grid(s->n_bins * s->n_pairs)), dim3(256), 0, st, Y, s->fftsize, s->n_bins, s->n_ch, s->n_pairs,
                           scale, s->d_peak, d_mag, order);
    ZHIP(hipGetLastError());
    return ZMLS_OK;
}

int zmls_spectrum_host(zmls_spectrum *s, const double *h, const double *window, int normalise, double *mag)
{
    if (!s || !h || !mag) { set_error("null argument"); return ZMLS_E_ARG; }
    ZDEVICE(s->device);
    int rc = ZMLS_OK;
    // staging for the host call: made on the first one and kept with the object (three hipMalloc / hipFree pairs per call cost more
    // than the transform of a single response); held while small, per call beyond kSpectrumKeepBytes
    const size_t hb = (size_t)s->n * s->n_ch * 8, mb = (size_t)s->n_bins * s->n_ch * 8, wb = (size_t)s->n * 8;
    const bool keep = hb + mb + wb <= kSpectrumKeepBytes;
    double *d_h = s->d_host[0], *d_m = s->d_host[1], *d_w = s->d_host[2];
    hipError_t e = hipSuccess;
    if (!d_h) e = hipMalloc((void **)&d_h, hb);
    if (e == hipSuccess && !d_m) e = hipMalloc((void **)&d_m, mb);
    if (e == hipSuccess && !d_w) e = hipMalloc((void **)&d_w, wb);
    if (e != hipSuccess) rc = hip_fail(e, "spectrum staging");
    if (keep && !rc) { s->d_host[0] = d_h; s->d_host[1] = d_m; s->d_host[2] = d_w; }
    if (!rc) rc = host_copy(s->device, d_h, (void *)h, hb, true);
    if (!rc && window) rc = host_copy(s->device, d_w, (void *)window, wb, true);
    if (!rc) rc = zmls_spectrum_device(s, d_h, s->n_ch, 1, window ? d_w : nullptr, normalise, d_m, nullptr);
    if (!rc) rc = host_copy(s->device, d_m, mag, mb, false);
    if (!keep || rc) {
        // (after an error nothing is kept either: the next call starts from scratch)
        if (d_h) (void)hipFree(d_h);
        if (d_m) (void)hipFree(d_m);
        if (d_w) (void)hipFree(d_w);
        s->d_host[0] = s->d_host[1] = s->d_host[2] = nullptr;
    }
    return rc;
}

} // extern "C"
