// Host-to-device upload of pageable memory: plain hipMemcpy against hipHostRegister + copy + unregister and against
// a pinned bounce buffer filled by memcpy. Decides how zmls_impulse_host should move large captures.
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main()
{
    for (size_t mb : {16, 128, 512}) {
        const size_t n = mb << 20;
        char *h = (char *)malloc(n);
        memset(h, 1, n);
        void *d;
        CK(hipMalloc(&d, n));
        CK(hipMemcpy(d, h, n, hipMemcpyHostToDevice));
        double t0 = now();
        for (int i = 0; i < 3; ++i) CK(hipMemcpy(d, h, n, hipMemcpyHostToDevice));
        double t_plain = (now() - t0) / 3;
        double t_reg = 0, t_pinned = 1;   // hipHostRegister on malloc memory: "invalid argument" on this pool
        // bounce: 4 threads memcpy into two 8 MiB pinned chunks each, async copies behind them
        const int T = 4;
        const size_t chunk = 8 << 20;
        char *pin;
        CK(hipHostMalloc((void **)&pin, chunk * 2 * T, hipHostMallocDefault));
        hipStream_t st[T];
        hipEvent_t ev[T][2];
        for (int t = 0; t < T; ++t) { CK(hipStreamCreateWithFlags(&st[t], hipStreamNonBlocking)); CK(hipEventCreate(&ev[t][0])); CK(hipEventCreate(&ev[t][1])); }
        t0 = now();
        for (int rep = 0; rep < 3; ++rep) {
            std::vector<std::thread> th;
            for (int t = 0; t < T; ++t)
                th.emplace_back([&, t] {
                    (void)hipSetDevice(0);
                    const size_t lo = n / T * t, hi = (t == T - 1) ? n : n / T * (t + 1);
                    int k = 0;
                    bool used[2] = {false, false};
                    for (size_t o = lo; o < hi; o += chunk, k ^= 1) {
                        const size_t len = hi - o < chunk ? hi - o : chunk;
                        char *b = pin + (size_t)(2 * t + k) * chunk;
                        if (used[k]) (void)hipEventSynchronize(ev[t][k]);
                        memcpy(b, h + o, len);
                        (void)hipMemcpyAsync((char *)d + o, b, len, hipMemcpyHostToDevice, st[t]);
                        (void)hipEventRecord(ev[t][k], st[t]);
                        used[k] = true;
                    }
                    (void)hipStreamSynchronize(st[t]);
                });
            for (auto &x : th) x.join();
        }
        double t_bounce = (now() - t0) / 3;
        printf("%4zu MiB: plain %.2f ms (%.1f GB/s) | register+copy+unregister %.2f ms | already pinned %.2f ms (%.1f GB/s) | 4-thread bounce %.2f ms (%.1f GB/s)\n",
               mb, t_plain * 1e3, n / t_plain / 1e9, t_reg * 1e3, t_pinned * 1e3, n / t_pinned / 1e9, t_bounce * 1e3, n / t_bounce / 1e9);
        (void)hipHostFree(pin);
        (void)hipFree(d);
        free(h);
    }
    return 0;
}
