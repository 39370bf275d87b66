// Magnitude response of an impulse response on the device — SURVEY.md §8(f) row 2.
//
// Reference: Audio.fft (zignal/audio.py:678-704) as MLS_simple.plot_fft / ex_mls.py:80 use it on the recovered
// response: window the L samples, np.fft.fft with n = fftsize (L, or L+1 when L is one of the Mersenne primes
// 2^13-1, 2^17-1, 2^19-1, 2^31-1, audio.py:682-685), optionally divide by fftsize, 20 log10 |Y|, keep bins
// 0 .. int(fftsize/2)-1. Only the spectrum (half the response's size) then has to cross PCIe.
//
// The DFT itself is a plain library transform: hipFFT, loaded on first use. Around it two streaming kernels: window +
// zero-pad + transpose into channel-major staging, and |Y| -> dB back into (bins, channels).
// Even sizes (the bumped Mersenne sizes are powers of two): one double real-to-complex transform per channel.
// Odd sizes (every other L = 2^N - 1: the library runs them as Bluestein convolutions on complex data, a real input
// buys nothing): two channels ride in one complex transform, z = x1 + i x2, and come apart afterwards as
// X1[k] = (Z[k] + conj Z[F-k]) / 2, X2[k] = (Z[k] - conj Z[F-k]) / 2i - half the transforms (N=20 x 64 channels: 12.6 -> 6.x ms).
// The rounding error of one transform is relative to the larger of its two channels, so the second channel of a pair is first
// brought to the level of the first by a power of two taken from the channels' peaks (exact, undone after the transform): a quiet
// channel beside a loud one keeps the accuracy a transform of its own would give it (tests: channels 200 dB apart).
#include <dlfcn.h>
#include <algorithm>
#include <stdlib.h>
#include <string.h>
#include <hipfft/hipfft.h>
#include "zmls_internal.h"

namespace zmls {
namespace {

struct HipFft {
    void *handle = nullptr;
    hipfftResult (*PlanMany)(hipfftHandle *, int, int *, int *, int, int, int *, int, int, hipfftType, int) = nullptr;
    hipfftResult (*SetStream)(hipfftHandle, hipStream_t) = nullptr;
    hipfftResult (*ExecD2Z)(hipfftHandle, hipfftDoubleReal *, hipfftDoubleComplex *) = nullptr;
    hipfftResult (*ExecZ2Z)(hipfftHandle, hipfftDoubleComplex *, hipfftDoubleComplex *, int) = nullptr;
    hipfftResult (*Destroy)(hipfftHandle) = nullptr;
    char why[256] = "";
    bool ok = false;
    HipFft()
    {
        const char *names[] = {"libhipfft.so.0", "libhipfft.so", "/opt/rocm/lib/libhipfft.so.0"};
        for (const char *n : names) {
            handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (handle) break;
        }
        if (!handle) { snprintf(why, sizeof why, "cannot load libhipfft: %s", dlerror()); return; }
#define ZSYM(field, name)                                                                      \
    field = (decltype(field))dlsym(handle, name);                                              \
    if (!field) { snprintf(why, sizeof why, "libhipfft has no symbol %s", name); return; }
        ZSYM(PlanMany, "hipfftPlanMany")
        ZSYM(SetStream, "hipfftSetStream")
        ZSYM(ExecD2Z, "hipfftExecD2Z")
        ZSYM(ExecZ2Z, "hipfftExecZ2Z")
        ZSYM(Destroy, "hipfftDestroy")
#undef ZSYM
        ok = true;
    }
};

HipFft *hipfft()
{
    static HipFft f;
    if (!f.ok) { set_error("%s", f.why); return nullptr; }
    return &f;
}

// stage[c][t] = h[t][c] * window[t] for t < n, 0 for n <= t < fftsize
__global__ void __launch_bounds__(256) spectrum_prepare_kernel(const double *__restrict__ h, int64_t n, int64_t n_ch, int64_t row_stride,
                                                               int64_t ch_stride, const double *__restrict__ window, int64_t fftsize,
                                                               double *__restrict__ stage)
{
    const int64_t total = fftsize * n_ch;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t c = idx / fftsize, t = idx - c * fftsize;
        double v = 0.0;
        if (t < n) {
            v = h[t * row_stride + c * ch_stride];
            if (window) v *= window[t];
        }
        stage[idx] = v;
    }
}

// mag[k][c] = 20 log10(|Y[c][k]| * scale), k < n_bins
__global__ void __launch_bounds__(256) spectrum_db_kernel(const double2 *__restrict__ Y, int64_t n_cplx, int64_t n_bins, int64_t n_ch, double scale,
                                                          double *__restrict__ mag)
{
    const int64_t total = n_bins * n_ch;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t c = idx / n_bins, k = idx - c * n_bins;
        const double2 y = Y[c * n_cplx + k];
        mag[k * n_ch + c] = 20.0 * log10(hypot(y.x, y.y) * scale);
    }
}

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

// packed mode: the two spectra of a pair from Z[k] and Z[F-k]
__global__ void __launch_bounds__(256) spectrum_db_pairs_kernel(const double2 *__restrict__ Y, int64_t F, int64_t n_bins, int64_t n_ch, int64_t n_pairs,
                                                                double scale, const unsigned long long *__restrict__ peak, double *__restrict__ mag)
{
    const int64_t total = n_bins * n_pairs;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = idx / n_bins, k = idx - p * n_bins;
        const double2 a = Y[p * F + k], b = Y[p * F + (k ? F - k : 0)];
        // conj(b) = (b.x, -b.y):  X1 = (a + conj b) / 2,  |X2| = |a - conj b| / 2
        const double m1 = 0.5 * hypot(a.x + b.x, a.y - b.y), m2 = 0.5 * hypot(a.x - b.x, a.y + b.y);
        mag[k * n_ch + 2 * p] = 20.0 * log10(m1 * scale);
        if (2 * p + 1 < n_ch) mag[k * n_ch + 2 * p + 1] = 20.0 * log10(ldexp(m2, -pair_shift(peak, p, n_ch)) * scale);
    }
}

// ---- the same three steps of the packed mode for frame-major responses (channels adjacent in memory, the layout the
// impulse entry points write): adjacent lanes walk CHANNELS on the response side and FRAMES / BINS on the transform side, so a
// 64 x 64 tile crosses LDS in between and both sides move whole 512-byte rows. (One element per lane with frames on the lanes
// read 8 bytes out of every 512: the three kernels took 36 % of the call, profiles/r03_spectrum_kernel_stats.csv.)
constexpr int kSpT = 64;            // frames (bins) and channels per tile

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

__global__ void __launch_bounds__(256) spectrum_db_pairs_rows_kernel(const double2 *__restrict__ Y, int64_t F, int64_t n_bins, int64_t n_ch, int64_t n_pairs,
                                                                     double scale, const unsigned long long *__restrict__ peak, double *__restrict__ mag)
{
    __shared__ double tile[kSpT][kSpT + 1];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int64_t k0 = (int64_t)blockIdx.x * kSpT, c0 = (int64_t)blockIdx.y * kSpT;
    for (int pl = wv; pl < kSpT / 2; pl += 4) {
        const int64_t p = c0 / 2 + pl, k = k0 + lane;
        double m1 = 0.0, m2 = 0.0;
        if (p < n_pairs && k < n_bins) {
            const double2 a = Y[p * F + k], b = Y[p * F + (k ? F - k : 0)];
            m1 = 0.5 * hypot(a.x + b.x, a.y - b.y);
            m2 = ldexp(0.5 * hypot(a.x - b.x, a.y + b.y), -pair_shift(peak, p, n_ch));
        }
        tile[lane][2 * pl] = m1;
        tile[lane][2 * pl + 1] = m2;
    }
    __syncthreads();
    for (int r = wv; r < kSpT; r += 4) {
        const int64_t k = k0 + r, c = c0 + lane;
        if (k < n_bins && c < n_ch) mag[k * n_ch + c] = 20.0 * log10(tile[r][lane] * scale);
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
    int64_t n, n_ch, fftsize, n_bins, n_cplx;
    hipfftHandle plan;
    int64_t n_pairs;        // > 0: packed mode (odd fftsize, two channels per complex transform)
    double *d_stage;        // (n_ch, fftsize) real, or (n_pairs, fftsize) complex
    double2 *d_Y;           // (n_ch, fftsize/2 + 1), or (n_pairs, fftsize)
    unsigned long long *d_peak;   // packed mode: per-channel peak of the windowed samples (bit pattern of a double)
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
    HipFft *F = hipfft();
    if (!F) return ZMLS_E_UNSUPPORTED;
    zmls_spectrum *s = (zmls_spectrum *)calloc(1, sizeof(zmls_spectrum));
    if (!s) { set_error("out of host memory"); return ZMLS_E_ARG; }
    s->device = device;
    s->n = n_samples; s->n_ch = n_ch;
    s->fftsize = n_samples + (mersenne_prime_size(n_samples) ? 1 : 0);      // audio.py:682-685
    s->n_bins = s->fftsize / 2;                                             // int(fftsize/2), audio.py:701-702
    s->n_cplx = s->fftsize / 2 + 1;
    s->n_pairs = ((s->fftsize & 1) && n_ch >= 2) ? (n_ch + 1) / 2 : 0;
    hipError_t e = hipMalloc((void **)&s->d_stage, s->n_pairs ? (size_t)s->fftsize * s->n_pairs * 16 : (size_t)s->fftsize * n_ch * 8);
    if (e == hipSuccess) e = hipMalloc((void **)&s->d_Y, s->n_pairs ? (size_t)s->fftsize * s->n_pairs * 16 : (size_t)s->n_cplx * n_ch * 16);
    if (e == hipSuccess && s->n_pairs) e = hipMalloc((void **)&s->d_peak, (size_t)n_ch * 8);
    if (e != hipSuccess) {
        rc = hip_fail(e, "spectrum buffers");
        (void)zmls_spectrum_destroy(s);
        return rc;
    }
    int len[1] = {(int)s->fftsize};
    hipfftResult r = s->n_pairs ? F->PlanMany(&s->plan, 1, len, nullptr, 1, (int)s->fftsize, nullptr, 1, (int)s->fftsize, HIPFFT_Z2Z, (int)s->n_pairs)
                                : F->PlanMany(&s->plan, 1, len, nullptr, 1, (int)s->fftsize, nullptr, 1, (int)s->n_cplx, HIPFFT_D2Z, (int)n_ch);
    if (r != HIPFFT_SUCCESS) {
        set_error("hipfftPlanMany(%lld points x %lld) failed with %d", (long long)s->fftsize, (long long)n_ch, (int)r);
        s->plan = 0;
        (void)zmls_spectrum_destroy(s);
        return ZMLS_E_HIP;
    }
    *out = s;
    return ZMLS_OK;
}

int zmls_spectrum_destroy(zmls_spectrum *s)
{
    if (!s) return ZMLS_OK;
    DeviceScope on_device(s->device);
    if (!on_device.rc) {
        HipFft *F = hipfft();
        if (F && s->plan) (void)F->Destroy(s->plan);
        if (s->d_stage) (void)hipFree(s->d_stage);
        if (s->d_Y) (void)hipFree(s->d_Y);
        if (s->d_peak) (void)hipFree(s->d_peak);
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
    HipFft *F = hipfft();
    if (!F) return ZMLS_E_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    auto grid = [](int64_t n) { int64_t g = (n + 255) / 256; return (uint32_t)(g < 1 ? 1 : g > 65536 ? 65536 : g); };
    if (s->n_pairs) {
        // odd size: fftsize == n, nothing to pad
        ZHIP(hipMemsetAsync(s->d_peak, 0, (size_t)s->n_ch * 8, st));
        const uint32_t c_tiles = (uint32_t)((s->n_ch + kSpT - 1) / kSpT);
        if (ch_stride == 1 && c_tiles <= 65535u) {
            // frame-major responses: whole rows of adjacent channels on the response side, whole runs of frames / bins on the other
            const int64_t rows_per_block = std::max<int64_t>(kSpT, (s->n + 4095) / 4096);
            hipLaunchKernelGGL(spectrum_peak_rows_kernel, dim3((uint32_t)((s->n + rows_per_block - 1) / rows_per_block), c_tiles), dim3(256), 0, st, d_h, s->n,
                               s->n_ch, row_stride, d_window, rows_per_block, s->d_peak);
            hipLaunchKernelGGL(spectrum_prepare_pairs_rows_kernel, dim3((uint32_t)((s->n + kSpT - 1) / kSpT), c_tiles), dim3(256), 0, st, d_h, s->n, s->n_ch,
                               row_stride, d_window, s->n_pairs, s->d_peak, (double2 *)s->d_stage);
            ZHIP(hipGetLastError());
            hipfftResult rq = F->SetStream(s->plan, st);
            if (rq == HIPFFT_SUCCESS) rq = F->ExecZ2Z(s->plan, (hipfftDoubleComplex *)s->d_stage, (hipfftDoubleComplex *)s->d_Y, HIPFFT_FORWARD);
            if (rq != HIPFFT_SUCCESS) { set_error("hipfftExecZ2Z failed with %d", (int)rq); return ZMLS_E_HIP; }
            hipLaunchKernelGGL(spectrum_db_pairs_rows_kernel, dim3((uint32_t)((s->n_bins + kSpT - 1) / kSpT), c_tiles), dim3(256), 0, st, s->d_Y, s->fftsize,
                               s->n_bins, s->n_ch, s->n_pairs, normalise ? 1.0 / (double)s->fftsize : 1.0, s->d_peak, d_mag);
            ZHIP(hipGetLastError());
            return ZMLS_OK;
        }
        hipLaunchKernelGGL(spectrum_peak_kernel, dim3(std::min<uint32_t>(grid(s->n), 1024u), (uint32_t)std::min<int64_t>(s->n_ch, 65535)), dim3(256), 0, st, d_h, s->n, s->n_ch,
                           row_stride, ch_stride, d_window, s->d_peak);
        hipLaunchKernelGGL(spectrum_prepare_pairs_kernel, dim3(grid(s->n * s->n_pairs)), dim3(256), 0, st, d_h, s->n, s->n_ch, row_stride, ch_stride,
                           d_window, s->n_pairs, s->d_peak, (double2 *)s->d_stage);
        ZHIP(hipGetLastError());
        hipfftResult rp = F->SetStream(s->plan, st);
        if (rp == HIPFFT_SUCCESS) rp = F->ExecZ2Z(s->plan, (hipfftDoubleComplex *)s->d_stage, (hipfftDoubleComplex *)s->d_Y, HIPFFT_FORWARD);
        if (rp != HIPFFT_SUCCESS) { set_error("hipfftExecZ2Z failed with %d", (int)rp); return ZMLS_E_HIP; }
        hipLaunchKernelGGL(spectrum_db_pairs_kernel, dim3(grid(s->n_bins * s->n_pairs)), dim3(256), 0, st, s->d_Y, s->fftsize, s->n_bins, s->n_ch, s->n_pairs,
                           normalise ? 1.0 / (double)s->fftsize : 1.0, s->d_peak, d_mag);
        ZHIP(hipGetLastError());
        return ZMLS_OK;
    }
    hipLaunchKernelGGL(spectrum_prepare_kernel, dim3(grid(s->fftsize * s->n_ch)), dim3(256), 0, st, d_h, s->n, s->n_ch, row_stride, ch_stride,
                       d_window, s->fftsize, s->d_stage);
    ZHIP(hipGetLastError());
    hipfftResult r = F->SetStream(s->plan, st);
    if (r == HIPFFT_SUCCESS) r = F->ExecD2Z(s->plan, s->d_stage, (hipfftDoubleComplex *)s->d_Y);
    if (r != HIPFFT_SUCCESS) { set_error("hipfftExecD2Z failed with %d", (int)r); return ZMLS_E_HIP; }
    hipLaunchKernelGGL(spectrum_db_kernel, dim3(grid(s->n_bins * s->n_ch)), dim3(256), 0, st, s->d_Y, s->n_cplx, s->n_bins, s->n_ch,
                       normalise ? 1.0 / (double)s->fftsize : 1.0, d_mag);
    ZHIP(hipGetLastError());
    return ZMLS_OK;
}

int zmls_spectrum_host(zmls_spectrum *s, const double *h, const double *window, int normalise, double *mag)
{
    if (!s || !h || !mag) { set_error("null argument"); return ZMLS_E_ARG; }
    ZDEVICE(s->device);
    int rc = ZMLS_OK;
    double *d_h = nullptr, *d_w = nullptr, *d_m = nullptr;
    const size_t hb = (size_t)s->n * s->n_ch * 8, mb = (size_t)s->n_bins * s->n_ch * 8;
    hipError_t e = hipMalloc((void **)&d_h, hb);
    if (e == hipSuccess) e = hipMalloc((void **)&d_m, mb);
    if (e == hipSuccess && window) e = hipMalloc((void **)&d_w, (size_t)s->n * 8);
    if (e == hipSuccess && host_copy(s->device, d_h, (void *)h, hb, true) != ZMLS_OK) e = hipErrorUnknown;
    if (e == hipSuccess && window && host_copy(s->device, d_w, (void *)window, (size_t)s->n * 8, true) != ZMLS_OK) e = hipErrorUnknown;
    if (e != hipSuccess) rc = hip_fail(e, "spectrum staging");
    if (!rc) rc = zmls_spectrum_device(s, d_h, s->n_ch, 1, d_w, normalise, d_m, nullptr);
    if (!rc) {
        rc = host_copy(s->device, d_m, mag, mb, false);
    }
    if (d_h) (void)hipFree(d_h);
    if (d_w) (void)hipFree(d_w);
    if (d_m) (void)hipFree(d_m);
    return rc;
}

} // extern "C"
