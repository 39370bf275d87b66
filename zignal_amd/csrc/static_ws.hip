// zmls_impulse_device_static: the impulse-response launch sequence for callers that cannot allocate
// (code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc). It is
// zmls_impulse_device with one difference: the intermediate between the FWHT passes does not live in a
// workspace that zmls_plan_reserve hipMalloc'ed, but in the file-scope device array below, which every GPU
// the library runs on holds from the moment the code object is loaded until the process ends.
//
// `scratch` is a host-side stand-in here, not a device address: host code never passes its name to a kernel
// or a memory call. static_workspace() asks the runtime for the array's address on the current device
// (hipGetSymbolAddress, once per device: each GPU has its own copy at its own address) and everything else
// uses that pointer.
//
// Contents: every call on a GPU shares that GPU's array and finds the last call's contents in it. Nothing
// has to start at zero: the first pass writes every row of every slab of its chunk (rows of channels past
// the end of the batch are written as zeros by the kernel) before a later pass reads them, and the padding
// rows between them are never read. Calls that overlap on one GPU (two streams, two threads) race on it.
#include <atomic>
#include <mutex>
#include "zmls_internal.h"

namespace zmls {
namespace {

constexpr size_t kStaticBytes = (size_t)ZMLS_STATIC_WORKSPACE_BYTES;
static __device__ __align__(256) unsigned char scratch[kStaticBytes];

static_assert(work_rows_total(ZMLS_STATIC_MAX_ORDER) * (kVecBytes << kCfg[CFG_SMALL].lu) <= kStaticBytes &&
              work_rows_total(ZMLS_STATIC_MAX_ORDER + 1) * (kVecBytes << kCfg[CFG_SMALL].lu) > kStaticBytes,
              "ZMLS_STATIC_MAX_ORDER does not match ZMLS_STATIC_WORKSPACE_BYTES");

constexpr int kMaxDevices = 64;
std::mutex g_mu;
std::atomic<void *> g_ptr[kMaxDevices];        // address of `scratch` on device d, once resolved (read without the lock)

} // namespace

// With `device` current: the address and size of this device's copy of the array.
int static_workspace(int device, void **ptr, size_t *bytes)
{
    if (device < 0 || device >= kMaxDevices) { set_error("device %d out of range for the static workspace", device); return ZMLS_E_ARG; }
    void *known = g_ptr[device].load(std::memory_order_acquire);
    if (known) {                                    // the launch path: resolved by zmls_plan_create, no lock taken
        *ptr = known;
        if (bytes) *bytes = kStaticBytes;
        return ZMLS_OK;
    }
    std::lock_guard<std::mutex> lock(g_mu);
    if (!g_ptr[device].load(std::memory_order_relaxed)) {
        int cur = -1;
        ZHIP(hipGetDevice(&cur));
        if (cur != device) { set_error("static workspace of device %d asked for while device %d is current", device, cur); return ZMLS_E_ARG; }
        void *q = nullptr;
        hipError_t e = hipGetSymbolAddress((void **)&q, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return hip_fail(e, "hipGetSymbolAddress(scratch)");
        if (!q) { set_error("hipGetSymbolAddress returned no address for the static workspace"); return ZMLS_E_HIP; }
        g_ptr[device].store(q, std::memory_order_release);
    }
    *ptr = g_ptr[device].load(std::memory_order_relaxed);
    if (bytes) *bytes = kStaticBytes;
    return ZMLS_OK;
}

} // namespace zmls

using namespace zmls;

extern "C" int zmls_impulse_device_static(zmls_plan *p, const void *d_x, int in_dtype, int64_t n_rows, int64_t n_ch, int64_t row_stride,
                                          int64_t ch_stride, void *d_h, int64_t out_row_stride, int64_t out_ch_stride, void *stream)
{
    int64_t periods = 0;
    int rc = capture_periods(p, n_rows, &periods);
    if (rc) return rc;
    ZDEVICE(p->device);
    void *ws = nullptr;
    size_t ws_bytes = 0;
    rc = static_workspace(p->device, &ws, &ws_bytes);       // resolved by zmls_plan_create; here it is a table look-up
    if (rc) return rc;
    // the limit, stated before anything is launched: one 64-byte-row slab (the narrowest tile) of the order must fit - or,
    // for the few channels that take the pseudo-channel layout, one of its sixteen times smaller slabs (one channel up to N = 24)
    const size_t narrowest = (size_t)work_rows_total(p->N) * (kVecBytes << kCfg[CFG_SMALL].lu);
    const bool pc_fits = n_ch >= 1 && pc_applies(p, n_ch) && (p->pc_npass == 1 || pc_slab_bytes(p) <= ws_bytes);
    if (p->npass > 1 && narrowest > ws_bytes && !pc_fits) {
        set_error("order %d is beyond zmls_impulse_device_static: one channel slab of its intermediate is %zu bytes, the static "
                  "workspace holds %zu (orders up to %d); use zmls_plan_reserve + zmls_impulse_device",
                  p->N, narrowest, ws_bytes, ZMLS_STATIC_MAX_ORDER);
        return ZMLS_E_WORKSPACE;
    }
    return run_passes(p, d_x, in_dtype, periods, 1, periods, n_ch, row_stride, ch_stride, d_h, out_row_stride, out_ch_stride,
                      (hipStream_t)stream, nullptr, nullptr, 0, ws, ws_bytes);
}
