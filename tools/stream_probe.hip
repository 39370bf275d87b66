// What one MI355X streams: read-only, write-only and copy kernels in a few shapes (16 B per lane).
// Build: hipcc -O3 --offload-arch=gfx950 tools/stream_probe.hip -o tools/stream_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef uint32_t U4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <int UNR, bool NT>
__global__ void __launch_bounds__(256) rd(const U4 *src, uint64_t n, uint32_t *sink)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    U4 acc = (U4)0u;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (UNR - 1) * stride < n; i += UNR * stride) {
        U4 t[UNR];
#pragma unroll
        for (int k = 0; k < UNR; ++k) t[k] = NT ? __builtin_nontemporal_load(src + i + k * stride) : src[i + k * stride];
#pragma unroll
        for (int k = 0; k < UNR; ++k) acc ^= t[k];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9E3779B9u && acc.x == 0x7F4A7C15u) *sink = acc.y;
}
// cache-policy bits by hand: POL 0 plain, 1 nt, 2 sc1, 3 sc1 nt, 4 sc0 sc1, 5 sc0 sc1 nt, 6 sc0, 7 sc0 nt
template <int POL>
__device__ __forceinline__ U4 ld_pol(const U4 *p)
{
    U4 v;
    if constexpr (POL == 0) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory");
    if constexpr (POL == 1) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v) : "v"(p) : "memory");
    if constexpr (POL == 2) asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=v"(v) : "v"(p) : "memory");
    if constexpr (POL == 3) asm volatile("global_load_dwordx4 %0, %1, off sc1 nt" : "=v"(v) : "v"(p) : "memory");
    if constexpr (POL == 4) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v) : "v"(p) : "memory");
    if constexpr (POL == 5) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt" : "=v"(v) : "v"(p) : "memory");
    if constexpr (POL == 6) asm volatile("global_load_dwordx4 %0, %1, off sc0" : "=v"(v) : "v"(p) : "memory");
    if constexpr (POL == 7) asm volatile("global_load_dwordx4 %0, %1, off sc0 nt" : "=v"(v) : "v"(p) : "memory");
    return v;
}
template <int POL>
__global__ void __launch_bounds__(256) rd_pol(const U4 *src, uint64_t n, uint32_t *sink)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    U4 acc = (U4)0u;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 7 * stride < n; i += 8 * stride) {
        U4 t[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) t[k] = ld_pol<POL>(src + i + k * stride);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int k = 0; k < 8; ++k) acc ^= t[k];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x9E3779B9u && acc.x == 0x7F4A7C15u) *sink = acc.y;
}

template <int UNR, bool NT>
__global__ void __launch_bounds__(256) cp(const U4 *src, U4 *dst, uint64_t n)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (UNR - 1) * stride < n; i += UNR * stride) {
        U4 t[UNR];
#pragma unroll
        for (int k = 0; k < UNR; ++k) t[k] = NT ? __builtin_nontemporal_load(src + i + k * stride) : src[i + k * stride];
#pragma unroll
        for (int k = 0; k < UNR; ++k) { if (NT) __builtin_nontemporal_store(t[k], dst + i + k * stride); else dst[i + k * stride] = t[k]; }
    }
}
// one contiguous chunk per workgroup (blocked, not grid-strided)
template <int UNR>
__global__ void __launch_bounds__(256) cp_blocked(const U4 *src, U4 *dst, uint64_t per_block)
{
    const U4 *s = src + (uint64_t)blockIdx.x * per_block;
    U4 *d = dst + (uint64_t)blockIdx.x * per_block;
    for (uint64_t i = threadIdx.x; i + (UNR - 1) * 256 < per_block; i += UNR * 256) {
        U4 t[UNR];
#pragma unroll
        for (int k = 0; k < UNR; ++k) t[k] = s[i + k * 256];
#pragma unroll
        for (int k = 0; k < UNR; ++k) d[i + k * 256] = t[k];
    }
}
template <bool NT>
__global__ void __launch_bounds__(256) wr(U4 *dst, uint64_t n)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    const U4 v = {1u, 2u, 3u, (uint32_t)threadIdx.x};
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) { if (NT) __builtin_nontemporal_store(v, dst + i); else dst[i] = v; }
}

template <typename F> float timeit(F f)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e30f;
    for (int r = 0; r < 4; ++r) {
        CK(hipEventRecord(a)); f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b));
        if (r && ms < best) best = ms;
    }
    return best;
}

int main(int argc, char **argv)
{
    const size_t gb = argc > 1 ? atoi(argv[1]) : 16;
    const size_t bytes = gb << 30, n = bytes / 16;
    U4 *a, *b;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 2, bytes));
    for (int grid : {1024, 2048, 4096, 16384}) {
        printf("grid %5d: ", grid);
        printf("read u8 %.0f  u8nt %.0f  u4 %.0f  u2 %.0f | ", bytes / 1e6 / timeit([&] { rd<8, false><<<grid, 256>>>(a, n, (uint32_t *)b); }),
               bytes / 1e6 / timeit([&] { rd<8, true><<<grid, 256>>>(a, n, (uint32_t *)b); }), bytes / 1e6 / timeit([&] { rd<4, false><<<grid, 256>>>(a, n, (uint32_t *)b); }),
               bytes / 1e6 / timeit([&] { rd<2, false><<<grid, 256>>>(a, n, (uint32_t *)b); }));
        printf("copy(r+w) u8 %.0f  u8nt %.0f  u4 %.0f  u4nt %.0f  u1 %.0f | ", 2 * bytes / 1e6 / timeit([&] { cp<8, false><<<grid, 256>>>(a, b, n); }),
               2 * bytes / 1e6 / timeit([&] { cp<8, true><<<grid, 256>>>(a, b, n); }), 2 * bytes / 1e6 / timeit([&] { cp<4, false><<<grid, 256>>>(a, b, n); }),
               2 * bytes / 1e6 / timeit([&] { cp<4, true><<<grid, 256>>>(a, b, n); }), 2 * bytes / 1e6 / timeit([&] { cp<1, false><<<grid, 256>>>(a, b, n); }));
        printf("write %.0f  nt %.0f\n", bytes / 1e6 / timeit([&] { wr<false><<<grid, 256>>>(b, n); }), bytes / 1e6 / timeit([&] { wr<true><<<grid, 256>>>(b, n); }));
    }
    for (int grid : {2048, 16384}) {
        printf("read by cache policy, grid %5d: plain %.0f  nt %.0f  sc1 %.0f  sc1 nt %.0f  sc0 sc1 %.0f  sc0 sc1 nt %.0f  sc0 %.0f  sc0 nt %.0f\n", grid,
               bytes / 1e6 / timeit([&] { rd_pol<0><<<grid, 256>>>(a, n, (uint32_t *)b); }), bytes / 1e6 / timeit([&] { rd_pol<1><<<grid, 256>>>(a, n, (uint32_t *)b); }),
               bytes / 1e6 / timeit([&] { rd_pol<2><<<grid, 256>>>(a, n, (uint32_t *)b); }), bytes / 1e6 / timeit([&] { rd_pol<3><<<grid, 256>>>(a, n, (uint32_t *)b); }),
               bytes / 1e6 / timeit([&] { rd_pol<4><<<grid, 256>>>(a, n, (uint32_t *)b); }), bytes / 1e6 / timeit([&] { rd_pol<5><<<grid, 256>>>(a, n, (uint32_t *)b); }),
               bytes / 1e6 / timeit([&] { rd_pol<6><<<grid, 256>>>(a, n, (uint32_t *)b); }), bytes / 1e6 / timeit([&] { rd_pol<7><<<grid, 256>>>(a, n, (uint32_t *)b); }));
    }
    for (int grid : {2048, 16384, 131072}) {
        const uint64_t per = n / grid;
        printf("blocked copy, grid %6d (%.1f MiB per workgroup): u4 %.0f  u8 %.0f\n", grid, per * 16 / 1048576.0,
               2 * bytes / 1e6 / timeit([&] { cp_blocked<4><<<grid, 256>>>(a, b, per); }), 2 * bytes / 1e6 / timeit([&] { cp_blocked<8><<<grid, 256>>>(a, b, per); }));
    }
    // Does the Infinity Cache (256 MiB) serve lines back that a kernel has just WRITTEN? write S bytes, then read them,
    // for S inside and beyond the cache; beside it, the read of S bytes that were read just before (warm) and of S cold bytes
    for (size_t mb : {32, 64, 128, 192, 384, 1024}) {
        const size_t sb = mb << 20, sn = sb / 16;
        float t_rw = 1e30f, t_rr = 1e30f, t_cold = 1e30f, t_w = 1e30f;
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int r = 0; r < 5; ++r) {
            float ms;
            // flush: sweep 2 GiB of the other buffer
            rd<8, false><<<4096, 256>>>(a, (size_t)(2048u << 20) / 16 * 1, (uint32_t *)a);
            CK(hipEventRecord(e0)); wr<false><<<4096, 256>>>(b, sn); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < t_w) t_w = ms;
            CK(hipEventRecord(e0)); rd<1, false><<<2048, 256>>>(b, sn, (uint32_t *)a); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < t_rw) t_rw = ms;
            CK(hipEventRecord(e0)); rd<1, false><<<2048, 256>>>(b, sn, (uint32_t *)a); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < t_rr) t_rr = ms;
            rd<8, false><<<4096, 256>>>(a, (size_t)(2048u << 20) / 16 * 1, (uint32_t *)a);
            CK(hipEventRecord(e0)); rd<1, false><<<2048, 256>>>(b, sn, (uint32_t *)a); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < t_cold) t_cold = ms;
        }
        printf("%5zu MiB: write %.0f GB/s | read after write %.0f | read after read %.0f | read cold %.0f\n", mb, sb / 1e6 / t_w, sb / 1e6 / t_rw, sb / 1e6 / t_rr, sb / 1e6 / t_cold);
    }
    printf("hipMemcpyDtoD (r+w) %.0f\n", 2 * bytes / 1e6 / timeit([&] { CK(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0)); }));
    return 0;
}
