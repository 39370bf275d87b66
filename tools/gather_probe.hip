// Probe: how fast can MI355X gather random row segments of S bytes and sum P periods of them?
// (design input for the first FWHT pass). hipcc --offload-arch=gfx950 -O3 tools/gather_probe.hip -Iinclude -Lzignal_amd/lib -lzmls
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "zmls.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
typedef double v2 __attribute__((ext_vector_type(2)));

// LPR lanes per row (each lane 16 B) -> segment = 16*LPR bytes. Each thread handles RPT rows.
// rows of this block: a contiguous run of the table, as a tile of the FWHT would.
template <int LPR, int RPT, int PF>
__global__ void __launch_bounds__(256, 2) gather_sum(const char *x, const uint32_t *q, uint64_t row_bytes, uint64_t period_bytes,
                                                   int periods, uint64_t rows_per_slab, uint32_t n_seg, v2 *out)
{
    const uint32_t tid = threadIdx.x;
    const uint32_t u = tid % LPR, rl = tid / LPR;
    constexpr int ROWS_PER_BLOCK = 256 / LPR * RPT;
    const uint64_t blk = blockIdx.x;
    const uint32_t seg = blk % n_seg;               // which column segment
    const uint64_t tile = blk / n_seg;
    uint64_t off[RPT];
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const uint64_t d = tile * ROWS_PER_BLOCK + (uint64_t)i * (256 / LPR) + rl;
        uint32_t j = q[d + 1];                       // skip sentinel at 0
        off[i] = (uint64_t)j * row_bytes + (uint64_t)seg * (16 * LPR) + u * 16;
    }
    v2 acc[RPT];
#pragma unroll
    for (int i = 0; i < RPT; ++i) acc[i] = (v2)0.0;
    const char *xp = x + period_bytes;
    if (PF == 0 || PF == 2) {
        for (int p = 0; p < periods; ++p, xp += period_bytes) {
            v2 t[RPT];
#pragma unroll
            for (int i = 0; i < RPT; ++i) t[i] = PF == 2 ? __builtin_nontemporal_load((const v2 *)(xp + off[i])) : *(const v2 *)(xp + off[i]);
#pragma unroll
            for (int i = 0; i < RPT; ++i) acc[i] += t[i];
        }
    } else {
        v2 t0[RPT], t1[RPT];
#pragma unroll
        for (int i = 0; i < RPT; ++i) t0[i] = *(const v2 *)(xp + off[i]);
        for (int p = 0; p + 1 < periods; p += 2) {
            const char *x1 = xp + period_bytes;
#pragma unroll
            for (int i = 0; i < RPT; ++i) t1[i] = *(const v2 *)(x1 + off[i]);
#pragma unroll
            for (int i = 0; i < RPT; ++i) acc[i] += t0[i];
            xp = x1 + period_bytes;
            if (p + 2 < periods) {
#pragma unroll
                for (int i = 0; i < RPT; ++i) t0[i] = *(const v2 *)(xp + off[i]);
            }
#pragma unroll
            for (int i = 0; i < RPT; ++i) acc[i] += t1[i];
        }
    }
    v2 s = (v2)0.0;
#pragma unroll
    for (int i = 0; i < RPT; ++i) s += acc[i];
    out[blk * 256 + tid] = s;
}

// The same gather through LDS-DMA: every wave-instruction moves 64 x 16 B from per-lane global addresses into 1 KiB of
// LDS (no VGPR destination); the wave then adds its own 16 bytes of every slot to its accumulators.
template <int LPR, int RPT, int AUX>
__global__ void __launch_bounds__(256, 2) gather_lds(const char *x, const uint32_t *q, uint64_t row_bytes, uint64_t period_bytes,
                                                   int periods, uint64_t rows_per_slab, uint32_t n_seg, v2 *out)
{
    __shared__ v2 ring[4][RPT][64];                 // [wave][slot][lane]: 4 * RPT KiB
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t u = tid % LPR, rl = tid / LPR;
    constexpr int ROWS_PER_BLOCK = 256 / LPR * RPT;
    const uint64_t blk = blockIdx.x;
    const uint32_t seg = blk % n_seg;
    const uint64_t tile = blk / n_seg;
    uint64_t off[RPT];
#pragma unroll
    for (int i = 0; i < RPT; ++i) {
        const uint64_t d = tile * ROWS_PER_BLOCK + (uint64_t)i * (256 / LPR) + rl;
        uint32_t j = q[d + 1];
        off[i] = (uint64_t)j * row_bytes + (uint64_t)seg * (16 * LPR) + u * 16;
    }
    v2 acc[RPT];
#pragma unroll
    for (int i = 0; i < RPT; ++i) acc[i] = (v2)0.0;
    const char *xp = x + period_bytes;
    for (int p = 0; p < periods; ++p, xp += period_bytes) {
#pragma unroll
        for (int i = 0; i < RPT; ++i)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(xp + off[i]),
                                             (__attribute__((address_space(3))) void *)&ring[wave][i][0], 16, 0, AUX);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < RPT; ++i) acc[i] += ring[wave][i][lane];
        asm volatile("" ::: "memory");
    }
    v2 s = (v2)0.0;
#pragma unroll
    for (int i = 0; i < RPT; ++i) s += acc[i];
    out[blk * 256 + tid] = s;
}

__global__ void stream_sum(const v2 *x, uint64_t n, v2 *out)
{
    v2 s = (v2)0.0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) s += x[i];
    out[(uint64_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void fill(v2 *x, uint64_t n)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) x[i] = (v2)(double)(i & 1023);
}

template <int LPR, int RPT, int AUX>
void run_lds(const char *name, const char *x, const uint32_t *q, int C, int periods, uint64_t L, v2 *out, double *check)
{
    const uint64_t row_bytes = (uint64_t)C * 8, period_bytes = L * row_bytes;
    constexpr int ROWS_PER_BLOCK = 256 / LPR * RPT;
    const uint32_t n_seg = (uint32_t)(row_bytes / (16 * LPR));
    const uint64_t tiles = (L - 1) / ROWS_PER_BLOCK;
    const uint64_t blocks = tiles * n_seg;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e30f;
    for (int it = 0; it < 4; ++it) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((gather_lds<LPR, RPT, AUX>), dim3((uint32_t)blocks), dim3(256), 0, 0, x, q, row_bytes, period_bytes, periods, L, n_seg, out);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (it && ms < best) best = ms;
    }
    CK(hipMemcpy(check, out, 2 * sizeof(double), hipMemcpyDeviceToHost));
    const double bytes = (double)tiles * ROWS_PER_BLOCK * row_bytes * periods;
    printf("%-34s seg %4d B rows/thr %2d aux %d : %8.3f ms  %7.1f GB/s  (out[0] %.1f)\n", name, 16 * LPR, RPT, AUX, best, bytes / best / 1e6, check[0]);
}

template <int LPR, int RPT, int PF>
void run(const char *name, const char *x, const uint32_t *q, int C, int periods, uint64_t L, v2 *out)
{
    const uint64_t row_bytes = (uint64_t)C * 8, period_bytes = L * row_bytes;
    constexpr int ROWS_PER_BLOCK = 256 / LPR * RPT;
    const uint32_t n_seg = (uint32_t)(row_bytes / (16 * LPR));
    const uint64_t tiles = (L - 1) / ROWS_PER_BLOCK;     // drop the ragged end
    const uint64_t blocks = tiles * n_seg;
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e30f;
    for (int it = 0; it < 4; ++it) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL((gather_sum<LPR, RPT, PF>), dim3((uint32_t)blocks), dim3(256), 0, 0, x, q, row_bytes, period_bytes, periods, L, n_seg, out);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (it && ms < best) best = ms;
    }
    const double bytes = (double)tiles * ROWS_PER_BLOCK * row_bytes * periods;
    printf("%-34s seg %4d B rows/thr %2d pf %d : %8.3f ms  %7.1f GB/s\n", name, 16 * LPR, RPT, PF, best, bytes / best / 1e6);
}

int main(int argc, char **argv)
{
    const int N = 20, C = argc > 1 ? atoi(argv[1]) : 512, periods = argc > 2 ? atoi(argv[2]) : 16;
    const uint64_t L = (1ull << N) - 1;
    int32_t taps[2] = {20, 17};
    zmls_plan *p;
    if (zmls_plan_create(&p, N, taps, 2, 0, 0)) { printf("plan: %s\n", zmls_last_error()); return 1; }
    std::vector<uint32_t> q(1u << N);
    zmls_plan_export_table(p, 0, q.data());
    uint32_t *dq; CK(hipMalloc(&dq, q.size() * 4)); CK(hipMemcpy(dq, q.data(), q.size() * 4, hipMemcpyHostToDevice));
    const uint64_t bytes = (uint64_t)(periods + 1) * L * C * 8;
    char *x; CK(hipMalloc(&x, bytes));
    v2 *out; CK(hipMalloc(&out, (size_t)1 << 30));
    hipLaunchKernelGGL(fill, dim3(8192), dim3(256), 0, 0, (v2 *)x, bytes / 16);
    CK(hipDeviceSynchronize());
    printf("C=%d periods=%d buffer %.1f GB\n", C, periods, bytes / 1e9);
    {
        hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        for (int g : {2048, 8192, 65536}) {
            float best = 1e30f;
            for (int it = 0; it < 3; ++it) {
                CK(hipEventRecord(a));
                hipLaunchKernelGGL(stream_sum, dim3(g), dim3(256), 0, 0, (const v2 *)x, bytes / 16, out);
                CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
                float ms; CK(hipEventElapsedTime(&ms, a, b)); if (it && ms < best) best = ms;
            }
            printf("stream read grid %6d: %8.3f ms %7.1f GB/s\n", g, best, bytes / best / 1e6);
        }
    }
    run<4, 16, 0>("64B x16rows", x, dq, C, periods, L, out);
    run<8, 16, 0>("128B x16rows", x, dq, C, periods, L, out);
    {
        double chk[2];
        run<8, 16, 0>("128B x16rows (again)", x, dq, C, periods, L, out);
        CK(hipMemcpy(chk, out, 16, hipMemcpyDeviceToHost)); printf("   register gather out[0] %.1f\n", chk[0]);
        run<8, 16, 2>("128B x16rows nt", x, dq, C, periods, L, out);
        run<16, 16, 2>("256B x16rows nt", x, dq, C, periods, L, out);
        run<64, 16, 2>("1024B x16rows nt", x, dq, C, periods, L, out);
        run_lds<8, 16, 0>("128B x16rows LDS-DMA", x, dq, C, periods, L, out, chk);
        run_lds<8, 16, 2>("128B x16rows LDS-DMA nt", x, dq, C, periods, L, out, chk);
        run_lds<16, 16, 0>("256B x16rows LDS-DMA", x, dq, C, periods, L, out, chk);
        run_lds<16, 16, 2>("256B x16rows LDS-DMA nt", x, dq, C, periods, L, out, chk);
        run_lds<64, 16, 0>("1024B x16rows LDS-DMA", x, dq, C, periods, L, out, chk);
        run_lds<64, 16, 2>("1024B x16rows LDS-DMA nt", x, dq, C, periods, L, out, chk);
    }
    run<16, 16, 0>("256B x16rows", x, dq, C, periods, L, out);
    run<32, 16, 0>("512B x16rows", x, dq, C, periods, L, out);
    run<64, 16, 0>("1024B x16rows", x, dq, C, periods, L, out);
    run<64, 8, 0>("1024B x8rows", x, dq, C, periods, L, out);
    run<256, 16, 0>("4096B x16rows", x, dq, C, periods, L, out);
    return 0;
}
