// Can the intermediate between two passes live in the 256 MiB Infinity Cache? The two-pass shape of a single-period
// batch, without the butterflies: per chunk of S bytes
//     A: read S bytes of a large source (once, non-temporal)  -> write S bytes of a small, REUSED workspace
//     B: read the workspace                                   -> write S bytes of a large destination (once, non-temporal)
// for S inside and beyond the residency rule of MI355X_MICROARCH.md ("the lines plus every byte loaded or stored between
// two uses fit in about 256 MiB"), with the workspace's stores and loads temporal or non-temporal, on one stream or with
// chunk k's B beside chunk k+1's A on two streams (two workspaces). Reported: source + destination bytes over time
// (= the algorithmic rate of such a batch), to hold against 2 * S / (copy time): what two plain copies take.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ic_probe.hip -o tools/ic_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
typedef uint32_t U4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

// LD / ST: 0 plain, 1 non-temporal. STRIDED: the reader walks the workspace in another order than the writer did
// (rows of 128 bytes at a stride of 1024 rows, the shape of a second FWHT pass)
template <int LD, int ST, bool STRIDED>
__global__ void __launch_bounds__(256) cp(const U4 *src, U4 *dst, uint64_t n)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i0 = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i0 < n; i0 += 4 * stride) {
        U4 t[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            uint64_t i = i0 + k * stride;
            if (i >= n) i = i0;
            uint64_t j = i;
            if (STRIDED) {
                // vector i of the sweep = (row r, lane u of 8): rows visited in bit-rotated order (low 10 bits <-> the rest)
                const uint64_t rows = n >> 3, r = i >> 3, u = i & 7;
                const uint64_t hi = rows >> 10;                    // rows = hi * 1024
                const uint64_t rr = (r % hi) * 1024 + (r / hi);
                j = rr * 8 + u;
            }
            t[k] = LD ? __builtin_nontemporal_load(src + j) : src[j];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint64_t i = i0 + k * stride;
            if (i < n) { if (ST) __builtin_nontemporal_store(t[k], dst + i); else dst[i] = t[k]; }
        }
    }
}

template <int WS_ST, int WS_LD>
static float run(const U4 *x, U4 *h, U4 *w[2], size_t total, size_t S, int streams, hipStream_t *st, bool strided)
{
    const uint64_t sn = S / 16;
    const int grid = (int)((sn / 4 + 255) / 256 < 8192 ? (sn / 4 + 255) / 256 : 8192);
    hipEvent_t e0, e1, done[2];
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&done[0])); CK(hipEventCreate(&done[1]));
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, st[0]));
        if (streams == 2) { CK(hipEventRecord(done[0], st[0])); CK(hipStreamWaitEvent(st[1], done[0], 0)); }
        size_t k = 0;
        for (size_t off = 0; off + S <= total; off += S, ++k) {
            hipStream_t s = st[streams == 2 ? (k & 1) : 0];
            U4 *ws = w[streams == 2 ? (k & 1) : 0];
            cp<1, WS_ST, false><<<grid, 256, 0, s>>>(x + off / 16, ws, sn);
            if (strided) cp<WS_LD, 1, true><<<grid, 256, 0, s>>>(ws, h + off / 16, sn);
            else cp<WS_LD, 1, false><<<grid, 256, 0, s>>>(ws, h + off / 16, sn);
        }
        if (streams == 2) { CK(hipEventRecord(done[1], st[1])); CK(hipStreamWaitEvent(st[0], done[1], 0)); }
        CK(hipEventRecord(e1, st[0]));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep && ms < best) best = ms;
    }
    return best;
}

int main(int argc, char **argv)
{
    const size_t total = (size_t)(argc > 1 ? atoi(argv[1]) : 8) << 30;     // bytes of source = bytes of destination
    U4 *x, *h, *w[2];
    CK(hipMalloc(&x, total)); CK(hipMalloc(&h, total));
    CK(hipMalloc(&w[0], (size_t)1 << 30)); CK(hipMalloc(&w[1], (size_t)1 << 30));
    CK(hipMemset(x, 1, total)); CK(hipMemset(h, 2, total)); CK(hipMemset(w[0], 0, (size_t)1 << 30)); CK(hipMemset(w[1], 0, (size_t)1 << 30));
    hipStream_t st[2];
    CK(hipStreamCreateWithFlags(&st[0], hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&st[1], hipStreamNonBlocking));
    // the yardstick: one plain pass source -> destination (what a single-pass transform would move)
    {
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        float best = 1e30f;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, st[0])); cp<1, 1, false><<<8192, 256, 0, st[0]>>>(x, h, total / 16); CK(hipEventRecord(e1, st[0])); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (rep && ms < best) best = ms;
        }
        printf("one copy of %zu GiB source -> destination (nt/nt): %.3f ms = %.0f GB/s of source + destination bytes\n", total >> 30, best, 2.0 * total / 1e6 / best);
    }
    printf("two-pass shape, GB/s of source + destination bytes (two plain copies would give half the line above)\n");
    printf("%8s %8s | %28s | %28s | %28s | %28s\n", "chunk", "streams", "ws st plain / ld plain", "ws st plain / ld nt", "ws st nt / ld plain", "ws st nt / ld nt");
    for (int strided = 0; strided < 2; ++strided) {
        printf("-- pass 2 reads the workspace %s\n", strided ? "in 128-byte rows at a stride of 1024 rows" : "in the order it was written");
        for (size_t mb : {16, 32, 48, 64, 96, 128, 192, 256, 512, 1024}) {
            const size_t S = mb << 20;
            for (int streams = 1; streams <= 2; ++streams) {
                const float a = run<0, 0>(x, h, w, total, S, streams, st, strided), b = run<0, 1>(x, h, w, total, S, streams, st, strided);
                const float c = run<1, 0>(x, h, w, total, S, streams, st, strided), d = run<1, 1>(x, h, w, total, S, streams, st, strided);
                const size_t moved = total / S * S;
                printf("%5zu MiB %8d | %18.3f ms %6.0f | %18.3f ms %6.0f | %18.3f ms %6.0f | %18.3f ms %6.0f\n", mb, streams, a, 2.0 * moved / 1e6 / a, b,
                       2.0 * moved / 1e6 / b, c, 2.0 * moved / 1e6 / c, d, 2.0 * moved / 1e6 / d);
                fflush(stdout);
            }
        }
    }
    return 0;
}
