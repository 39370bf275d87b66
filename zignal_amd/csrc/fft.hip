// Hand-written complex DFT (double) for the magnitude response - SURVEY.md §8(f) row 2; reference: np.fft.fft inside Audio.fft
// (zignal/audio.py:694). Round 3 replaces the library transform (hipFFT ran every L = 2^N - 1 as a Bluestein convolution of
// thirteen kernels) by passes of this library's own, built like the FWHT passes: 64 KiB tiles, radix-16 rounds on registers,
// LDS exchanges, whole 128-byte rows on the HBM side (fft_tile.h has the geometry and the arithmetic).
//   power-of-two sizes: forward passes; the spectrum stays in the passes' own order and the dB kernel gathers from it;
//   every other size F: Bluestein on M = 2^m >= 2F - 1 points with everything fused into the passes -
//       first forward pass   loads the staged samples times the chirp exp(-i pi t^2 / F), zero beyond;
//       last forward pass    stores times the chirp filter's spectrum (built once per size by the same passes);
//       last inverse pass    stores chirp x value / M for the F bins that exist.
//     Four (N <= 20: 9 + 12 bits each way) passes over the batch instead of thirteen kernels; chirps and twiddles are computed
//     (sincospi), not read.
#include <stdlib.h>
#include <vector>
#include "zmls_internal.h"
#include "fft_tile.h"

namespace zmls {
namespace fft {
namespace {

template <int K, bool STRIDED>
__global__ void __launch_bounds__(kThreads, 2) fft_pass_kernel(const PassArgs a)
{
    __shared__ cd lds[kLdsSlots];
    const uint32_t tid = threadIdx.x;
    const TileAt at = tile_at(blockIdx.x, a.n_transforms, a.m, a.lv);
    phase_load<K, STRIDED>(a, at, tid, lds);
    __syncthreads();
    cd v[kPerThread];
    constexpr int NR = num_rounds(K);
    if constexpr (NR >= 2) {
        round_read_compute<K, STRIDED, 0>(lds, tid, v);
        round_write<K, STRIDED, 0>(lds, tid, v);
        __syncthreads();
    }
    if constexpr (NR >= 3) {
        round_read_compute<K, STRIDED, 1>(lds, tid, v);
        round_write<K, STRIDED, 1>(lds, tid, v);
        __syncthreads();
    }
    round_read_compute<K, STRIDED, NR - 1>(lds, tid, v);
    __syncthreads();                        // the last round writes natural order: other threads' positions
    round_write<K, STRIDED, NR - 1>(lds, tid, v);
    __syncthreads();
    phase_store<K, STRIDED>(a, at, tid, lds);
}

template <int K, bool STRIDED>
void emulate_pass_tiles(const PassArgs &a, uint64_t n_tiles)
{
    std::vector<cd> lds(kLdsSlots);
    std::vector<cd> regs((size_t)kThreads * kPerThread);
    auto V = [&](uint32_t tid) -> cd (&)[kPerThread] { return *reinterpret_cast<cd (*)[kPerThread]>(&regs[(size_t)tid * kPerThread]); };
    constexpr int NR = num_rounds(K);
    for (uint64_t tile = 0; tile < n_tiles; ++tile) {
        const TileAt at = tile_at(tile, a.n_transforms, a.m, a.lv);
        for (uint32_t t = 0; t < kThreads; ++t) phase_load<K, STRIDED>(a, at, t, lds.data());
        if constexpr (NR >= 2)
            for (uint32_t t = 0; t < kThreads; ++t) { round_read_compute<K, STRIDED, 0>(lds.data(), t, V(t)); round_write<K, STRIDED, 0>(lds.data(), t, V(t)); }
        if constexpr (NR >= 3)
            for (uint32_t t = 0; t < kThreads; ++t) { round_read_compute<K, STRIDED, 1>(lds.data(), t, V(t)); round_write<K, STRIDED, 1>(lds.data(), t, V(t)); }
        for (uint32_t t = 0; t < kThreads; ++t) round_read_compute<K, STRIDED, NR - 1>(lds.data(), t, V(t));
        for (uint32_t t = 0; t < kThreads; ++t) round_write<K, STRIDED, NR - 1>(lds.data(), t, V(t));
        for (uint32_t t = 0; t < kThreads; ++t) phase_store<K, STRIDED>(a, at, t, lds.data());
    }
}

uint64_t tiles_of(const PassArgs &a) { return (a.total + kTile - 1) >> kTileLog; }

// exp(-i pi i^2 d^2 / F), i = 0..15, for the distance d between a thread's points in this pass (fft_tile.h::ThreadPoints)
void chirp_pass_constants(PassArgs &a)
{
    const uint64_t d = a.lv.strided ? (uint64_t)(kThreads >> (kTileLog - a.lv.K)) << a.lv.logC : (uint64_t)kThreads;
    for (int i = 0; i < kPerThread; ++i) {
        const unsigned __int128 x = (unsigned __int128)d * d * (unsigned)(i * i);
        const uint64_t r = (uint64_t)(x % (unsigned __int128)(2 * a.F));
        const long double ang = -3.14159265358979323846264338327950288L * (long double)r / (long double)a.F;
        a.chirp_i2[i] = cd{(double)cosl(ang), (double)sinl(ang)};
    }
}

// device == false: the same passes on the host (zmls_host_fft_selfcheck)
int run_pass(const PassArgs &a, hipStream_t st, bool device)
{
    const uint64_t n_tiles = tiles_of(a);
    if (n_tiles == 0 || n_tiles > 0x7FFFFFFFull) { set_error("transform pass grid out of range (%llu tiles)", (unsigned long long)n_tiles); return ZMLS_E_ARG; }
// -DZMLS_FFT_NO_DEVICE (the sanitizer build of the host logic, tests/test_sanitizers_cpu.py): no kernel is instantiated
#ifdef ZMLS_FFT_NO_DEVICE
#define ZFFT_LAUNCH(KK, ST) do { set_error("built without the transform kernels"); return ZMLS_E_UNSUPPORTED; } while (0)
#else
#define ZFFT_LAUNCH(KK, ST) hipLaunchKernelGGL((fft_pass_kernel<KK, ST>), dim3((uint32_t)n_tiles), dim3(kThreads), 0, st, a)
#endif
#define ZFFT_CASE(KK, ST)                                                                                              \
    case KK:                                                                                                           \
        if (device) ZFFT_LAUNCH(KK, ST);                                                                               \
        else emulate_pass_tiles<KK, ST>(a, n_tiles);                                                                   \
        break;
    if (a.lv.strided) {
        switch (a.lv.K) {
            ZFFT_CASE(4, true) ZFFT_CASE(5, true) ZFFT_CASE(6, true) ZFFT_CASE(7, true) ZFFT_CASE(8, true) ZFFT_CASE(9, true)
        default: set_error("no strided transform pass of %d bits", a.lv.K); return ZMLS_E_UNSUPPORTED;
        }
    } else {
        switch (a.lv.K) {
            ZFFT_CASE(1, false) ZFFT_CASE(2, false) ZFFT_CASE(3, false) ZFFT_CASE(4, false) ZFFT_CASE(5, false) ZFFT_CASE(6, false)
            ZFFT_CASE(7, false) ZFFT_CASE(8, false) ZFFT_CASE(9, false) ZFFT_CASE(10, false) ZFFT_CASE(11, false) ZFFT_CASE(12, false)
        default: set_error("no contiguous transform pass of %d bits", a.lv.K); return ZMLS_E_UNSUPPORTED;
        }
    }
#undef ZFFT_CASE
#undef ZFFT_LAUNCH
    if (device) ZHIP(hipGetLastError());
    return ZMLS_OK;
}

} // namespace

// forward passes over `work` (n_transforms x 2^m points, in place); the first pass loads by `load_mode`, the last stores by
// `store_mode` (fft_tile.h). The spectrum is left in the passes' own order (position_of).
int forward(const Plan &p, PassArgs a, int load_mode, int store_mode, hipStream_t st, bool device)
{
    a.m = p.m; a.total = a.n_transforms << p.m; a.inv = 0;
    for (int i = 0; i < p.n_levels; ++i) {
        a.lv = p.level[i];
        a.load_mode = i == 0 ? load_mode : LOAD_WORK;
        a.store_mode = i == p.n_levels - 1 ? store_mode : STORE_WORK;
        if (a.load_mode == LOAD_STAGE_CHIRP) chirp_pass_constants(a);
        if (int rc = run_pass(a, st, device)) return rc;
    }
    return ZMLS_OK;
}
// inverse passes (unnormalised), from the passes' own order back to natural order
int inverse(const Plan &p, PassArgs a, int store_mode, hipStream_t st, bool device)
{
    a.m = p.m; a.total = a.n_transforms << p.m; a.inv = 1;
    for (int i = p.n_levels - 1; i >= 0; --i) {
        a.lv = p.level[i];
        a.load_mode = LOAD_WORK;
        a.store_mode = i == 0 ? store_mode : STORE_WORK;
        if (a.store_mode == STORE_FINAL_CHIRP) chirp_pass_constants(a);
        if (int rc = run_pass(a, st, device)) return rc;
    }
    return ZMLS_OK;
}

} // namespace fft
} // namespace zmls

using namespace zmls;

extern "C" {

// Host logic of the transform passes, no GPU: the same tile code run thread by thread on the CPU and held to a direct DFT.
//   mode 0: forward passes of 2 transforms of 2^m points, spectrum read through position_of; then the inverse passes back
//   mode 1: the whole Bluestein chain for F = `size` points (any F >= 2), 2 transforms
// Returns 0 when every value is within 1e-11 of the largest magnitude (the emulation's sin / cos are libm's, not sincospi).
int zmls_host_fft_selfcheck(int mode, int64_t size)
{
    using namespace zmls::fft;
    if (mode < 0 || mode > 1 || size < 2 || size > (1 << 23)) { set_error("fft selfcheck: mode 0 or 1, size 2 .. 2^23"); return ZMLS_E_ARG; }
    const uint64_t nt = 2;
    uint64_t rng = 0x9E3779B97F4A7C15ull ^ (uint64_t)size;
    auto rnd = [&]() { rng = rng * 6364136223846793005ull + 1442695040888963407ull; return (double)(int64_t)(rng >> 11) / 9007199254740992.0 - 0.5; };
    // direct DFT of a few bins (all of them for small sizes)
    auto direct = [](const std::vector<cd> &x, uint64_t F, uint64_t k) {
        long double re = 0, im = 0;
        for (uint64_t t = 0; t < F; ++t) {
            const uint64_t j = (t * k) % F;
            const long double ang = -2.0L * 3.14159265358979323846264338327950288L * (long double)j / (long double)F;
            re += x[t].x * cosl(ang) - x[t].y * sinl(ang);
            im += x[t].x * sinl(ang) + x[t].y * cosl(ang);
        }
        return cd{(double)re, (double)im};
    };
    if (mode == 0) {
        int m = 0;
        while ((1ll << m) < size) ++m;
        if ((1ll << m) != size) { set_error("fft selfcheck: mode 0 needs a power of two"); return ZMLS_E_ARG; }
        const uint64_t M = 1ull << m;
        const Plan p = make_plan(m);
        std::vector<cd> x(nt * M), w;
        for (auto &v : x) v = cd{rnd(), rnd()};
        w = x;
        fft::PassArgs a{};
        a.work = w.data(); a.n_transforms = nt;
        if (int rc = forward(p, a, LOAD_WORK, STORE_WORK, nullptr, false)) return rc;
        // reference: textbook radix-2 (bit reversal, then log2 M butterfly sweeps) in double, every bin; a direct DFT for tiny sizes
        double peak = 0, worst = 0;
        for (uint64_t tr = 0; tr < nt; ++tr) {
            std::vector<cd> ref(x.begin() + tr * M, x.begin() + (tr + 1) * M);
            if (M <= 64) {
                const std::vector<cd> one = ref;
                for (uint64_t k = 0; k < M; ++k) ref[k] = direct(one, M, k);
            } else {
                for (uint64_t i = 0; i < M; ++i) {
                    const uint64_t j = bit_reverse((uint32_t)i, m);
                    if (j > i) std::swap(ref[i], ref[j]);
                }
                for (uint64_t half = 1; half < M; half <<= 1)
                    for (uint64_t k = 0; k < half; ++k) {
                        const long double ang = -3.14159265358979323846264338327950288L * (long double)k / (long double)half;
                        const cd w{(double)cosl(ang), (double)sinl(ang)};
                        for (uint64_t b0 = k; b0 < M; b0 += 2 * half) {
                            const cd u = ref[b0], t = cmul(ref[b0 + half], w);
                            ref[b0] = u + t; ref[b0 + half] = u - t;
                        }
                    }
            }
            for (uint64_t k = 0; k < M; ++k) {
                const cd want = ref[k], got = w[tr * M + position_of(p, k)];
                peak = fmax(peak, hypot(want.x, want.y));
                worst = fmax(worst, hypot(got.x - want.x, got.y - want.y));
            }
        }
        if (!(worst <= 1e-11 * peak)) { set_error("fft selfcheck: forward passes of 2^%d points off by %.3g (peak %.3g)", m, worst, peak); return 1; }
        if (int rc = inverse(p, a, STORE_WORK, nullptr, false)) return rc;
        worst = 0;
        for (uint64_t i = 0; i < nt * M; ++i) worst = fmax(worst, hypot(w[i].x / (double)M - x[i].x, w[i].y / (double)M - x[i].y));
        if (!(worst <= 1e-11)) { set_error("fft selfcheck: inverse passes of 2^%d points off by %.3g", m, worst); return 1; }
        return ZMLS_OK;
    }
    const uint64_t F = (uint64_t)size;
    int m = 1;
    while ((1ull << m) < 2 * F - 1) ++m;
    const uint64_t M = 1ull << m;
    const Plan p = make_plan(m);
    std::vector<cd> x(nt * F), table(M), w(nt * M), Y(nt * F);
    for (auto &v : x) v = cd{rnd(), rnd()};
    fft::PassArgs b{};
    b.work = table.data(); b.n_transforms = 1; b.F = F;
    if (int rc = forward(p, b, LOAD_BSEQ, STORE_WORK, nullptr, false)) return rc;
    fft::PassArgs a{};
    a.work = w.data(); a.n_transforms = nt; a.stage = x.data(); a.n_stage = F; a.F = F; a.table = table.data(); a.Y = Y.data();
    a.scale = 1.0 / (double)M;
    if (int rc = forward(p, a, LOAD_STAGE_CHIRP, STORE_WORK_TIMES_TABLE, nullptr, false)) return rc;
    if (int rc = inverse(p, a, STORE_FINAL_CHIRP, nullptr, false)) return rc;
    double peak = 0, worst = 0;
    const uint64_t step = F <= 512 ? 1 : F <= 65536 ? F / 257 : F / 29;
    for (uint64_t tr = 0; tr < nt; ++tr) {
        std::vector<cd> one(x.begin() + tr * F, x.begin() + (tr + 1) * F);
        for (uint64_t k = 0; k < F; k += step) {
            const cd want = direct(one, F, k), got = Y[tr * F + k];
            peak = fmax(peak, hypot(want.x, want.y));
            worst = fmax(worst, hypot(got.x - want.x, got.y - want.y));
        }
    }
    if (!(worst <= 1e-11 * peak)) { set_error("fft selfcheck: Bluestein chain for %llu points (2^%d) off by %.3g (peak %.3g)", (unsigned long long)F, m, worst, peak); return 1; }
    return ZMLS_OK;
}

} // extern "C"
