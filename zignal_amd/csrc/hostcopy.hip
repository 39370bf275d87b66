// Host <-> device copies of the host-facing entry points (zmls_impulse_host, zmls_sequence_f64, the ctypes buffer helpers).
//
// The drop-in call of the reference - get_impulse(x) on a numpy array (zignal/measure/mls.py:315-328) - hands over pageable
// host memory, and from N = 18 up the call is bound by that copy, not by the device work (N = 20 x 17 periods: 134 MB up, 8 MB
// down, 0.08 ms of kernels). One pageable hipMemcpy moves 25-31 GB/s host to device and 15 GB/s back on this pool (the runtime
// stages it through its own pinned buffer with one thread), and registering the caller's pages is refused here
// (hipHostRegister: "invalid argument"). So large copies are cut into chunks that a few PERSISTENT threads - created on the
// first large copy, asleep on a condition variable in between, never started per call (per-call threads and per-call pinned
// allocations are what sank round 2's probe, tools/pin_probe.hip) - move through pinned bounce buffers of their own, two per
// thread, each on its own stream: while one chunk's DMA runs the thread copies the next into (or the previous out of) the other
// buffer. The calling thread works as one of them. Copies of 64 KiB ... 1 MiB go through the calling thread's own pinned buffers
// alone (no helper is woken up); only below 64 KiB does hipMemcpy stay. Why not leave the middle to hipMemcpy: on this pool a
// pageable copy of 64 KiB ... 1 MiB crawls at 2.5-2.8 GB/s (512 KiB: 0.19 ms against 0.03 here), and the same 2 MiB upload takes
// anything between 74 and 720 us from one call to the next (ZMLS_TRACE_HOST=1 prints where a host call's time goes).
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <stdlib.h>
#include <string.h>
#include <thread>
#include <stdio.h>
#include <time.h>
#include "zmls_internal.h"

namespace zmls {
namespace {

constexpr size_t kChunk = (size_t)4 << 20;          // bytes per bounce buffer (smaller copies are cut finer, Engine::run)
constexpr size_t kEngineFrom = (size_t)1 << 20;     // copies from this size on go through the engine (all threads)
constexpr size_t kSmallFrom = (size_t)64 << 10;     // ... and from this size on through the calling thread's own pinned buffers,
constexpr int kWorkers = 4;                         // the calling thread included

struct Worker {
    hipStream_t st = nullptr;
    void *pin[2] = {nullptr, nullptr};
    hipEvent_t ev[2] = {nullptr, nullptr};
};

struct Engine {
    int device = -1;
    bool ok = false;
    Worker w[kWorkers];
    std::mutex call;                                // one copy at a time
    std::mutex m;
    std::condition_variable wake, done_cv;
    uint64_t generation = 0;
    int running = 0;
    // the job
    char *dev = nullptr;
    char *host = nullptr;
    size_t bytes = 0, n_chunks = 0, chunk = kChunk;
    bool to_device = true;
    std::atomic<size_t> next{0};
    std::atomic<int> err{0};

    void work(int id)
    {
        Worker &me = w[id];
        bool used[2] = {false, false};
        size_t pending = (size_t)-1;                // download: the chunk whose DMA into `slot ^ 1` is in flight
        int slot = 0;
        for (;;) {
            const size_t i = next.fetch_add(1);
            if (i >= n_chunks || err.load()) break;
            const size_t off = i * chunk, len = (bytes - off < chunk) ? bytes - off : chunk;
            if (to_device) {
                if (used[slot] && hipEventSynchronize(me.ev[slot]) != hipSuccess) { err = 1; break; }
                memcpy(me.pin[slot], host + off, len);
                if (hipMemcpyAsync(dev + off, me.pin[slot], len, hipMemcpyHostToDevice, me.st) != hipSuccess ||
                    hipEventRecord(me.ev[slot], me.st) != hipSuccess) { err = 1; break; }
                used[slot] = true;
            } else {
                if (hipMemcpyAsync(me.pin[slot], dev + off, len, hipMemcpyDeviceToHost, me.st) != hipSuccess ||
                    hipEventRecord(me.ev[slot], me.st) != hipSuccess) { err = 1; break; }
                if (pending != (size_t)-1) {
                    const size_t poff = pending * chunk, plen = (bytes - poff < chunk) ? bytes - poff : chunk;
                    if (hipEventSynchronize(me.ev[slot ^ 1]) != hipSuccess) { err = 1; break; }
                    memcpy(host + poff, me.pin[slot ^ 1], plen);
                }
                pending = i;
            }
            slot ^= 1;
        }
        if (hipStreamSynchronize(me.st) != hipSuccess) err = 1;
        if (!to_device && pending != (size_t)-1 && !err.load()) {
            const size_t poff = pending * chunk, plen = (bytes - poff < chunk) ? bytes - poff : chunk;
            memcpy(host + poff, me.pin[slot ^ 1], plen);
        }
    }

    void helper(int id)
    {
        (void)hipSetDevice(device);
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(m);
                wake.wait(lk, [&] { return generation != seen; });
                seen = generation;
            }
            work(id);
            {
                std::lock_guard<std::mutex> lk(m);
                if (--running == 0) done_cv.notify_one();
            }
        }
    }

    // copies below the engine's threshold: the calling thread alone, through its own two pinned buffers and stream (no helper is
    // woken up). One pageable hipMemcpy of 64 KiB ... 1 MiB takes 0.03-0.35 ms on this pool (2.5-2.8 GB/s); this takes the time
    // of one host memcpy plus one DMA (512 KiB: 0.19 -> 0.03 ms, profiles/r03_hostcopy_rates.txt).
    int run_small(void *d, void *h, size_t n, bool up)
    {
        std::lock_guard<std::mutex> one(call);
        dev = (char *)d; host = (char *)h; bytes = n; to_device = up;
        chunk = (size_t)512 << 10;                  // both buffers in flight from 1 MiB up: the host memcpy of one piece beside the DMA of the other
        n_chunks = (n + chunk - 1) / chunk;
        next = 0; err = 0;
        work(0);
        if (err.load()) { (void)hipGetLastError(); set_error("host copy of %zu bytes through the pinned buffers failed", n); return ZMLS_E_HIP; }
        return ZMLS_OK;
    }

    int run(void *d, void *h, size_t n, bool up)
    {
        std::lock_guard<std::mutex> one(call);
        dev = (char *)d; host = (char *)h; bytes = n; to_device = up;
        // every thread should get at least two chunks (its two buffers in flight), of 512 KiB or more
        chunk = kChunk;
        while (chunk > ((size_t)512 << 10) && n / chunk < 2 * kWorkers) chunk >>= 1;
        n_chunks = (n + chunk - 1) / chunk;
        next = 0; err = 0;
        {
            std::lock_guard<std::mutex> lk(m);
            running = kWorkers - 1;
            ++generation;
        }
        wake.notify_all();
        work(0);
        {
            std::unique_lock<std::mutex> lk(m);
            done_cv.wait(lk, [&] { return running == 0; });
        }
        if (err.load()) { (void)hipGetLastError(); set_error("chunked host copy of %zu bytes failed", n); return ZMLS_E_HIP; }
        return ZMLS_OK;
    }
};

// one engine per device, made on first use and kept until the process ends (its threads sleep on a condition variable
// that must outlive them, so it is never destroyed)
Engine *engine_for(int device)
{
    static std::mutex guard;
    static Engine *engines[64];
    if (device < 0 || device >= 64) return nullptr;
    std::lock_guard<std::mutex> lk(guard);
    Engine *&e = engines[device];
    if (e) return e->ok ? e : nullptr;
    e = new Engine;
    e->device = device;
    bool ok = true;
    // one pinned allocation for all bounce buffers (eight pieces of one 32 MiB block rather than eight blocks of 4 MiB)
    char *block = nullptr;
    ok = hipHostMalloc((void **)&block, (size_t)kWorkers * 2 * kChunk, hipHostMallocDefault) == hipSuccess;
    for (int i = 0; i < kWorkers && ok; ++i) {
        ok = hipStreamCreateWithFlags(&e->w[i].st, hipStreamNonBlocking) == hipSuccess;
        for (int s = 0; s < 2 && ok; ++s) {
            e->w[i].pin[s] = block + ((size_t)i * 2 + s) * kChunk;
            ok = hipEventCreateWithFlags(&e->w[i].ev[s], hipEventDisableTiming) == hipSuccess;
        }
    }
    if (!ok) { (void)hipGetLastError(); return nullptr; }     // no pinned memory to be had: the plain copy stays
    for (int i = 1; i < kWorkers; ++i) std::thread(&Engine::helper, e, i).detach();
    e->ok = true;
    return e;
}

} // namespace

// dst/src: one of them device memory on `device` (current), the other pageable host memory. Ordered like hipMemcpy on the
// null stream: everything enqueued before is complete before the first byte moves, and the copy is complete on return.
static int host_copy_untraced(int device, void *d_ptr, void *h_ptr, size_t bytes, bool to_device);

int host_copy(int device, void *d_ptr, void *h_ptr, size_t bytes, bool to_device)
{
    // ZMLS_TRACE_HOST=1 (diagnostics): every host copy with its time, to stderr
    static const bool trace = [] { const char *v = getenv("ZMLS_TRACE_HOST"); return v && *v == '1'; }();
    if (!trace) return host_copy_untraced(device, d_ptr, h_ptr, bytes, to_device);
    timespec a, b;
    clock_gettime(CLOCK_MONOTONIC, &a);
    const int rc = host_copy_untraced(device, d_ptr, h_ptr, bytes, to_device);
    clock_gettime(CLOCK_MONOTONIC, &b);
    fprintf(stderr, "  host_copy %s %zu bytes: %.0f us\n", to_device ? "up" : "down", bytes, (b.tv_sec - a.tv_sec) * 1e6 + (b.tv_nsec - a.tv_nsec) * 1e-3);
    return rc;
}

static int host_copy_untraced(int device, void *d_ptr, void *h_ptr, size_t bytes, bool to_device)
{
    if (!bytes) return ZMLS_OK;
    // ZMLS_HOSTCOPY_MIN_BYTES: the size from which copies go through the engine (tuning; a huge value switches it off)
    static const size_t from = [] {
        const char *v = getenv("ZMLS_HOSTCOPY_MIN_BYTES");
        return (v && *v) ? (size_t)strtoull(v, nullptr, 10) : kEngineFrom;
    }();
    // ZMLS_HOSTCOPY_SMALL_MIN_BYTES: from this size up to the engine's threshold the calling thread copies through its pinned buffer
    static const size_t small_from = [] {
        const char *v = getenv("ZMLS_HOSTCOPY_SMALL_MIN_BYTES");
        return (v && *v) ? (size_t)strtoull(v, nullptr, 10) : kSmallFrom;
    }();
    // ZMLS_HOSTCOPY_SMALL_DIRS (tuning): 3 both directions (default), 1 uploads only, 2 downloads only, 0 neither
    static const int dirs = [] { const char *v = getenv("ZMLS_HOSTCOPY_SMALL_DIRS"); return (v && *v) ? atoi(v) : 3; }();
    auto small_dir = [&](bool up) { return (dirs & (up ? 1 : 2)) != 0; };
    Engine *e = (bytes >= from || (bytes >= small_from && small_dir(to_device))) ? engine_for(device) : nullptr;
    if (e && bytes < from) {
        static const bool trace = [] { const char *v = getenv("ZMLS_TRACE_HOST"); return v && *v == '1'; }();
        timespec a, b;
        if (trace) clock_gettime(CLOCK_MONOTONIC, &a);
        ZHIP(hipDeviceSynchronize());
        if (trace) {
            clock_gettime(CLOCK_MONOTONIC, &b);
            fprintf(stderr, "    (device synchronisation before it: %.0f us)\n", (b.tv_sec - a.tv_sec) * 1e6 + (b.tv_nsec - a.tv_nsec) * 1e-3);
        }
        return e->run_small(d_ptr, h_ptr, bytes, to_device);
    }
    if (!e) {
        ZHIP(to_device ? hipMemcpy(d_ptr, h_ptr, bytes, hipMemcpyHostToDevice) : hipMemcpy(h_ptr, d_ptr, bytes, hipMemcpyDeviceToHost));
        return ZMLS_OK;
    }
    ZHIP(hipDeviceSynchronize());       // the engine's streams do not wait for the null stream by themselves
    return e->run(d_ptr, h_ptr, bytes, to_device);
}

} // namespace zmls
