// C ABI of libzmls.so (declarations and the reference methods they replace: include/zmls.h).
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <stdio.h>
#include <algorithm>
#include <atomic>
#include <vector>
#include <thread>
#include "zmls_internal.h"

namespace zmls {
int validate_taps(int N, const int32_t *taps, int ntaps, uint32_t *G, uint32_t *s0);

size_t dtype_size(int dt)
{
    switch (dt) {
    case ZMLS_F64: return 8;
    case ZMLS_F32: return 4;
    case ZMLS_I16: return 2;
    case ZMLS_I32: return 4;
    default: return 0;
    }
}

int use_device(int device)
{
    // the device count does not change while the process lives: one runtime call per launch less
    static std::atomic<int> n_cached{0};
    int n = n_cached.load(std::memory_order_relaxed);
    if (n <= 0) {
        hipError_t e = hipGetDeviceCount(&n);
        if (e != hipSuccess || n <= 0) {
            set_error("no HIP device available (%s); libzmls has no CPU fallback", e == hipSuccess ? "0 devices" : hipGetErrorString(e));
            return ZMLS_E_NODEVICE;
        }
        n_cached.store(n, std::memory_order_relaxed);
    }
    if (device < 0 || device >= n) { set_error("device %d out of range (0..%d)", device, n - 1); return ZMLS_E_NODEVICE; }
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != device) ZHIP(hipSetDevice(device));
    return ZMLS_OK;
}

// Entry points make their object's device current for the duration of the call and put the caller's device back on
// the way out: a host program that holds another device current (torch, a second plan) does not find it changed.
DeviceScope::DeviceScope(int device)
{
    if (hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); prev = -1; }
    rc = use_device(device);
    if (rc || prev == device) prev = -1;
}
DeviceScope::~DeviceScope()
{
    if (prev >= 0) (void)hipSetDevice(prev);
}

// Large device allocations are made twice. Measured on MI355X / ROCm 7.2: the first hipMalloc of a
// multi-GiB buffer in a process lands in a physical layout on which random row gathers run 12 % slower
// (14.6 vs 12.9 ms for the N=20, 512-channel, 16-period gather pass, reproducibly), while memory handed
// back by hipFree and allocated again does not show it. Untouched memory costs nothing to free.
hipError_t settled_malloc(void **ptr, size_t bytes)
{
    if (bytes >= ((size_t)64 << 20)) {
        void *first = nullptr;
        hipError_t e = hipMalloc(&first, bytes);
        if (e != hipSuccess) return e;
        e = hipFree(first);
        if (e != hipSuccess) return e;
    }
    return hipMalloc(ptr, bytes);
}

namespace {

__global__ void count_sentinels_kernel(const uint32_t *t, uint64_t n, unsigned long long *count)
{
    unsigned long long local = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x + 1; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        local += (t[i] == kSentinel);
    if (local) atomicAdd(count, local);
}

size_t slab_bytes(const zmls_plan *p, int cfg) { return (size_t)work_rows_total(p->N) * (kVecBytes << kCfg[cfg].lu); }
static inline uint64_t U0(const TileCfg &tc) { return 1ull << tc.lu; }
int chans_per_slab(const zmls_plan *p, int cfg) { return (int)((kVecBytes << kCfg[cfg].lu) / dtype_size(p->dtype)); }

void free_plan_buffers(zmls_plan *p)
{
    if (p->d_bits) (void)hipFree(p->d_bits);
    if (p->d_gather) (void)hipFree(p->d_gather);
    if (p->d_scatter_nat) (void)hipFree(p->d_scatter_nat);
    for (int c = 0; c < CFG_COUNT; ++c)
        if (p->d_scatter_perm[c]) (void)hipFree(p->d_scatter_perm[c]);
    if (p->d_jt) (void)hipFree(p->d_jt);
    if (p->d_work) (void)hipFree(p->d_work);
    if (p->aux_stream) { (void)hipStreamDestroy(p->aux_stream); (void)hipEventDestroy(p->ev_fork); (void)hipEventDestroy(p->ev_join); }
    if (p->d_pipe) (void)hipFree(p->d_pipe);
    for (int i = 0; i < 2; ++i) {
        if (p->d_stage[i]) (void)hipFree(p->d_stage[i]);
        if (p->d_host_call[i]) (void)hipFree(p->d_host_call[i]);
    }
}

// M in the order in which the last pass of tile configuration c holds its elements (2^N entries)
int build_scatter_perm(zmls_plan *p, int c)
{
    const int np = (c == CFG_TALL) ? p->tall_npass : p->npass;
    const int *bits = (c == CFG_TALL) ? p->tall_bits : p->bits;
    if (np < 2 || p->d_scatter_perm[c]) return ZMLS_OK;
    int b = 0;
    for (int i = 0; i + 1 < np; ++i) b += bits[i];
    ZHIP(hipMalloc((void **)&p->d_scatter_perm[c], ((size_t)1 << p->N) * 4));
    return launch_scatter_perm(p->N, bits[np - 1], kCfg[c], b, p->d_scatter_nat, p->d_scatter_perm[c], 0);
}

int build_plan(zmls_plan *p)
{
    const uint64_t n = 1ull << p->N;
    JumpTables jt;
    build_jump_tables(p->N, p->G, &jt);
    ZHIP(hipMalloc((void **)&p->d_jt, sizeof jt));
    ZHIP(hipMemcpy(p->d_jt, &jt, sizeof jt, hipMemcpyHostToDevice));
    // bits: one period plus a margin so N-bit windows and generator_chunk() can run past the
    // period boundary; rounded up to whole threads of the generator
    p->n_words = ((n + 64 + 1023) / 1024) * 32;
    ZHIP(hipMalloc((void **)&p->d_bits, p->n_words * 4));
    // Is the register maximum-length? The reference does not ask (mls.py:64-68, 101-108: any taps generate a sequence),
    // so the BITS are generated either way; the transform needs the Hadamard identity of a maximal sequence and is
    // refused otherwise (sequence_only). Decided on the host with powers of the step matrix: F^L = I and F^(L/q) != I
    // for the primes q of L (a register whose taps leave out N is singular: never).
    p->maximal = ((p->G & 1u) && period_is_maximal(p->N, jt)) ? 1 : 0;
    if (p->N > kMaxTransformOrder || !p->maximal) {
        // orders 31 and 32 (zmls_internal.h::kMaxTransformOrder) and registers that are not maximum-length: the bit
        // sequence only. A register that is not maximal does not wrap at L: the buffer's bits past L continue it.
        int rc = launch_lfsr_tables(p->N, p->G, p->s0, p->d_jt, p->d_bits, p->n_words, nullptr, nullptr, p->maximal, 0);
        if (rc) return rc;
        p->npass = 0;
        ZHIP(hipDeviceSynchronize());
        return ZMLS_OK;
    }
    ZHIP(hipMalloc((void **)&p->d_gather, n * 4));
    ZHIP(hipMalloc((void **)&p->d_scatter_nat, n * 4));
    ZHIP(hipMemsetAsync(p->d_gather, 0xFF, n * 4, 0));
    ZHIP(hipMemsetAsync(p->d_scatter_nat, 0xFF, n * 4, 0));
    int rc = launch_lfsr_tables(p->N, p->G, p->s0, p->d_jt, p->d_bits, p->n_words, p->d_gather, p->d_scatter_nat, 1, 0);
    if (rc) return rc;
    // every window value 1..L must have been visited exactly once, else the taps do not give a
    // maximum-length sequence and the Hadamard identity does not hold
    unsigned long long *d_cnt = nullptr, h_cnt[2] = {0, 0};
    ZHIP(hipMalloc((void **)&d_cnt, 16));
    ZHIP(hipMemsetAsync(d_cnt, 0, 16, 0));
    hipLaunchKernelGGL(count_sentinels_kernel, dim3(1024), dim3(256), 0, 0, p->d_gather, n, d_cnt);
    hipLaunchKernelGGL(count_sentinels_kernel, dim3(1024), dim3(256), 0, 0, p->d_scatter_nat, n, d_cnt + 1);
    ZHIP(hipGetLastError());
    ZHIP(hipMemcpy(h_cnt, d_cnt, 16, hipMemcpyDeviceToHost));
    (void)hipFree(d_cnt);
    if (h_cnt[0] || h_cnt[1]) {
        // cannot happen for a register the host found maximal: the kernel's walk and the host's matrix powers disagree
        set_error("internal error: order %d taps are maximum-length by F^L = I but the walk missed %llu of %llu register states",
                  p->N, h_cnt[0], (unsigned long long)p->L);
        return ZMLS_E_HIP;
    }
    p->npass = pass_split(p->N, p->bits);
    // the tall tile earns its single workgroup per CU only where its 11-bit passes save a whole pass over HBM: N = 21, 22
    p->tall_npass = pass_split(p->N, p->tall_bits, kTallPassBits);
    if (!(p->npass == 3 && p->tall_npass == 2)) p->tall_npass = 0;
    // the pseudo-channel layout of few-channel batches: the row bits N - PCB go through the passes (2 are the least a pass takes)
    p->pc_npass = (p->N - pc_bits_of(p->dtype) >= 2) ? pass_split(p->N - pc_bits_of(p->dtype), p->pc_bits) : 0;
    if (p->npass > 1) {
        for (int c = 0; c < CFG_COUNT; ++c) {
            if (c == CFG_BIG) continue;                   // only ever used when forced: built by zmls_plan_force_tile
            rc = build_scatter_perm(p, c);
            if (rc) return rc;
        }
    }
    ZHIP(hipDeviceSynchronize());
    return ZMLS_OK;
}

} // namespace
} // namespace zmls

using namespace zmls;

extern "C" {

int zmls_device_count(int *n)
{
    if (!n) { set_error("null argument"); return ZMLS_E_ARG; }
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) (void)hipGetLastError();
    *n = (e == hipSuccess) ? c : 0;
    return ZMLS_OK;
}

int zmls_plan_create(zmls_plan **out, int N, const int32_t *taps, int ntaps, int dtype, int device)
{
    if (!out) { set_error("null plan pointer"); return ZMLS_E_ARG; }
    *out = nullptr;
    if (dtype != ZMLS_F64 && dtype != ZMLS_F32) { set_error("plan dtype must be ZMLS_F64 or ZMLS_F32"); return ZMLS_E_ARG; }
    uint32_t G, s0;
    int rc = validate_taps(N, taps, ntaps, &G, &s0);
    if (rc) return rc;
    ZDEVICE(device);
    {
        // set-up for zmls_impulse_device_static: this device's address of the static workspace, so that the
        // launch path itself never has to ask the runtime for it
        void *ws = nullptr;
        rc = static_workspace(device, &ws, nullptr);
        if (rc) return rc;
    }
    zmls_plan *p = (zmls_plan *)calloc(1, sizeof(zmls_plan));
    if (!p) { set_error("out of host memory"); return ZMLS_E_ARG; }
    p->N = N;
    p->L = ((int64_t)1 << N) - 1;
    p->ntaps = ntaps;
    memcpy(p->taps, taps, sizeof(int32_t) * ntaps);
    p->G = G;
    p->s0 = s0;
    p->dtype = dtype;
    p->device = device;
    p->force_cfg = -1;
    p->swizzle = 1;
    p->resident_mode = -1;
    p->force_pc = -1;
    rc = build_plan(p);
    if (rc) { free_plan_buffers(p); free(p); return rc; }
    *out = p;
    return ZMLS_OK;
}

int zmls_plan_destroy(zmls_plan *p)
{
    if (!p) return ZMLS_OK;
    DeviceScope on_device(p->device);
    if (!on_device.rc) free_plan_buffers(p);
    free(p);
    return ZMLS_OK;
}

int zmls_plan_reserve(zmls_plan *p, int64_t n_ch)
{
    if (!p || n_ch < 1) { set_error("bad reserve arguments"); return ZMLS_E_ARG; }
    if (int rc = sequence_only(p)) return rc;
    if (p->npass == 1) return ZMLS_OK;   // single-pass plans keep everything in LDS
    ZDEVICE(p->device);
    int rc = ZMLS_OK;
    // the widest slab among the tile configurations a call with n_ch channels can end up on: the small tile always, a
    // wide one only when the batch fills at least one of its slabs or it is forced (a 256-byte slab of a 2^30-point
    // transform alone is 275 GB: single-channel plans of the largest orders must not be charged for it)
    size_t need = 0;
    // the few channels that take the pseudo-channel layout need one of its slabs each - a sixteenth of an ordinary one (one
    // channel of N = 28: 2 GiB instead of 17)
    if (pc_applies(p, n_ch)) need = p->pc_npass > 1 ? (size_t)n_ch * pc_slab_bytes(p) : 0;
    for (int c = 0; c < CFG_COUNT && !pc_applies(p, n_ch); ++c) {
        const int cs = chans_per_slab(p, c);
        if (c != CFG_SMALL && n_ch < cs && p->force_cfg != c) continue;
        if (c == CFG_BIG && p->force_cfg != CFG_BIG) continue;         // the big tile is only ever used when forced
        if (c == CFG_TALL && !p->tall_npass) continue;
        const size_t want = (size_t)((n_ch + cs - 1) / cs) * slab_bytes(p, c);
        if (want > need) need = want;
    }
    if (need <= p->work_bytes) return ZMLS_OK;
    if (p->d_work) { ZHIP(hipFree(p->d_work)); p->d_work = nullptr; p->work_bytes = 0; }
    ZHIP(settled_malloc(&p->d_work, need));
    p->work_bytes = need;
    return ZMLS_OK;
}

int zmls_plan_reserve_staging(zmls_plan *p, int64_t n_ch)
{
    if (!p || n_ch < 1) { set_error("bad reserve arguments"); return ZMLS_E_ARG; }
    if (int rc = sequence_only(p)) return rc;
    if (p->npass == 1) return ZMLS_OK;   // single-pass plans read and write any layout directly
    ZDEVICE(p->device);
    int rc = ZMLS_OK;
    const size_t ts = dtype_size(p->dtype);
    const int64_t E = (int64_t)(kVecBytes / ts);
    const size_t need = (size_t)p->L * (size_t)((n_ch + E - 1) / E * E) * ts;
    rc = zmls_plan_reserve(p, pc_applies(p, n_ch) ? n_ch : (n_ch + E - 1) / E * E);
    if (rc) return rc;
    if (need <= p->stage_bytes) return ZMLS_OK;
    for (int i = 0; i < 2; ++i) {
        if (p->d_stage[i]) { ZHIP(hipFree(p->d_stage[i])); p->d_stage[i] = nullptr; }
    }
    p->stage_bytes = 0;
    for (int i = 0; i < 2; ++i) ZHIP(settled_malloc(&p->d_stage[i], need));
    p->stage_bytes = need;
    return ZMLS_OK;
}

int zmls_plan_force_tile(zmls_plan *p, int cfg)
{
    if (!p || cfg < -1 || (cfg >= 0 && (cfg & 0xFF) >= CFG_COUNT)) { set_error("tile configuration must be -1 (auto), 0 (small), 1 (big), 2 (mid) or 3 (tall)"); return ZMLS_E_ARG; }
    if (cfg >= 0 && (cfg & 0xFF) == CFG_TALL && !p->tall_npass) { set_error("the tall tile exists for orders 21 and 22 only"); return ZMLS_E_UNSUPPORTED; }
    if (cfg >= 0 && ((cfg & 0xFF) == CFG_BIG || (cfg & 0xFF) == CFG_TALL) && p->npass == 1) {
        set_error("orders up to 10 run as one pass in the small or the mid tile; the big and tall tiles do not apply");
        return ZMLS_E_UNSUPPORTED;
    }
    if (cfg >= 0 && (cfg & 0xFF) == CFG_BIG) {
        ZDEVICE(p->device);
        int rc = build_scatter_perm(p, CFG_BIG);
        if (rc) return rc;
        ZHIP(hipDeviceSynchronize());                     // callers may launch on streams that do not wait for the null stream
    }
    p->swizzle = (cfg >= 0 && (cfg & 0x100)) ? 0 : 1;     // +256: switch the XCD-aware block mapping off (tuning)
    p->work_layout = (cfg >= 0 && (cfg & 0x400)) ? 1 : 0; // +1024: row-major workspace (tuning)
    p->force_cfg = cfg < 0 ? -1 : (cfg & 0xFF);
    return ZMLS_OK;
}

int zmls_plan_last_launches(const zmls_plan *p, int32_t out[3])
{
    if (!p || !out) { set_error("null argument"); return ZMLS_E_ARG; }
    for (int i = 0; i < 3; ++i) out[i] = p->last_launches[i];
    return ZMLS_OK;
}

int zmls_plan_force_pseudo_channels(zmls_plan *p, int mode)
{
    if (!p || mode < -1 || mode > 1) { set_error("pseudo-channel layout: mode must be -1 (auto), 0 (off) or 1 (on)"); return ZMLS_E_ARG; }
    if (mode > 0 && !p->pc_npass) { set_error("order %d is too small for the pseudo-channel layout", p->N); return ZMLS_E_UNSUPPORTED; }
    p->force_pc = mode;
    return ZMLS_OK;
}

int zmls_plan_set_resident(zmls_plan *p, int mode, size_t bytes, int flags)
{
    if (!p || mode < -1 || mode > 1) { set_error("cache-resident schedule: mode must be -1 (auto), 0 (off) or 1 (on)"); return ZMLS_E_ARG; }
    if ((flags & 4) && !p->aux_stream) {
        ZDEVICE(p->device);
        int rc = ZMLS_OK;
        ZHIP(hipStreamCreateWithFlags(&p->aux_stream, hipStreamNonBlocking));
        ZHIP(hipEventCreateWithFlags(&p->ev_fork, hipEventDisableTiming));
        ZHIP(hipEventCreateWithFlags(&p->ev_join, hipEventDisableTiming));
    }
    if ((flags & 8) && !p->d_pipe) {
        ZDEVICE(p->device);
        int rc = ZMLS_OK;
        ZHIP(hipMalloc((void **)&p->d_pipe, (size_t)(4 + kPipeMaxSlabs) * sizeof(uint32_t)));
    }
    p->resident_mode = mode;
    p->resident_bytes = bytes;
    p->resident_flags = flags;
    return ZMLS_OK;
}

int zmls_plan_set_pipeline_delay(zmls_plan *p, uint32_t extra_tiles)
{
    if (!p) { set_error("null argument"); return ZMLS_E_ARG; }
    p->pipe_delay = extra_tiles;
    return ZMLS_OK;
}

int zmls_plan_pipeline_status(zmls_plan *p, int32_t out[2])
{
    if (!p || !out) { set_error("null argument"); return ZMLS_E_ARG; }
    out[0] = out[1] = 0;
    if (!p->d_pipe) return ZMLS_OK;
    ZDEVICE(p->device);
    int rc = ZMLS_OK;
    uint32_t w[2];
    ZHIP(hipDeviceSynchronize());
    ZHIP(hipMemcpy(w, p->d_pipe, sizeof w, hipMemcpyDeviceToHost));
    out[0] = (int32_t)w[0]; out[1] = (int32_t)w[1];
    return ZMLS_OK;
}

int zmls_plan_set_input_scale_log2(zmls_plan *p, int k)
{
    if (!p || k < 0 || k > 63) { set_error("input scale must be 2^k with 0 <= k <= 63"); return ZMLS_E_ARG; }
    p->in_scale_log2 = k;
    return ZMLS_OK;
}

int zmls_plan_info(const zmls_plan *p, int64_t info[16])
{
    if (!p || !info) { set_error("null argument"); return ZMLS_E_ARG; }
    memset(info, 0, sizeof(int64_t) * 16);
    info[0] = p->N; info[1] = p->L; info[2] = p->dtype; info[3] = p->device; info[4] = p->npass;
    info[5] = p->bits[0]; info[6] = p->bits[1]; info[7] = p->bits[2];
    info[8] = (int64_t)p->work_bytes; info[9] = chans_per_slab(p, CFG_SMALL); info[10] = chans_per_slab(p, CFG_BIG);
    info[11] = chans_per_slab(p, CFG_MID); info[12] = (int64_t)p->stage_bytes;
    info[13] = p->tall_npass; info[14] = p->tall_bits[0]; info[15] = p->tall_bits[1];
    return ZMLS_OK;
}

int zmls_sequence_bits(zmls_plan *p, uint8_t *out, int64_t nbits)
{
    if (!p || !out || nbits < 0 || (uint64_t)nbits > (1ull << p->N) + 32) { set_error("bad sequence_bits arguments"); return ZMLS_E_ARG; }
    ZDEVICE(p->device);
    int rc = ZMLS_OK;
    const size_t nbytes = (size_t)((nbits + 7) / 8);
    if (!nbytes) return ZMLS_OK;
    rc = host_copy(p->device, p->d_bits, out, nbytes, false);
    if (rc) return rc;
    if (nbits & 7) out[nbytes - 1] &= (uint8_t)((1u << (nbits & 7)) - 1u);
    return ZMLS_OK;
}

int zmls_sequence_bits_from(zmls_plan *p, uint64_t start, uint8_t *out, int64_t nbits)
{
    if (!p || !out || nbits < 0 || nbits > ((int64_t)1 << 31)) { set_error("bad sequence_bits_from arguments (at most 2^31 bits per call)"); return ZMLS_E_ARG; }
    if (!nbits) return ZMLS_OK;
    ZDEVICE(p->device);
    int rc = ZMLS_OK;
    // where the register stands after `start` steps: jump-ahead bookkeeping on the host (matrix powers, as the plan's
    // own tables are built), then the kernel generates the bits from there. A maximum-length register wraps at L.
    JumpTables jt;
    build_jump_tables(p->N, p->G, &jt);
    const uint32_t s_at = p->maximal ? host_fib_state(p->N, p->G, jt, start % (uint64_t)p->L) : host_fib_state_at(p->N, p->G, jt, start);
    const uint64_t n_words = (uint64_t)((nbits + 1023) / 1024) * 32;
    uint32_t *d = nullptr;
    ZHIP(hipMalloc((void **)&d, n_words * 4));
    rc = launch_lfsr_tables(p->N, p->G, s_at, p->d_jt, d, n_words, nullptr, nullptr, p->maximal, 0);
    const size_t nbytes = (size_t)((nbits + 7) / 8);
    if (!rc) rc = host_copy(p->device, d, out, nbytes, false);
    (void)hipFree(d);
    if (!rc && (nbits & 7)) out[nbytes - 1] &= (uint8_t)((1u << (nbits & 7)) - 1u);
    return rc;
}

int zmls_plan_kind(const zmls_plan *p, int *maximal, int *transform)
{
    if (!p) { set_error("null argument"); return ZMLS_E_ARG; }
    if (maximal) *maximal = p->maximal;
    if (transform) *transform = p->npass > 0;
    return ZMLS_OK;
}

int zmls_sequence_device(zmls_plan *p, void *d_out, int dtype, int64_t repeats, int64_t stride, void *stream)
{
    if (!p || !d_out || repeats < 1 || stride < 1) { set_error("bad sequence_device arguments"); return ZMLS_E_ARG; }
    ZDEVICE(p->device);
    int rc = ZMLS_OK;
    return launch_sequence_expand(p->d_bits, p->L, d_out, dtype, repeats, stride, (hipStream_t)stream);
}

int zmls_sequence_f64(zmls_plan *p, double *out, int64_t repeats)
{
    if (!p || !out || repeats < 1) { set_error("bad sequence_f64 arguments"); return ZMLS_E_ARG; }
    ZDEVICE(p->device);
    int rc = ZMLS_OK;
    double *d = nullptr;
    ZHIP(hipMalloc((void **)&d, (size_t)p->L * 8));
    rc = launch_sequence_expand(p->d_bits, p->L, d, ZMLS_F64, 1, 1, 0);
    if (!rc) rc = host_copy(p->device, d, out, (size_t)p->L * 8, false);
    (void)hipFree(d);
    if (rc) return rc;
    // np.tile, mls.py:164. The caller's array is usually fresh (numpy.empty: every page faults on first touch), so long
    // stimuli are tiled by a few threads, each writing whole repeats of its own
    const int64_t copies = repeats - 1;
    const int nthr = (copies * p->L * 8 >= ((int64_t)32 << 20)) ? (int)std::min<int64_t>(4, copies) : 1;
    auto tile = [&](int t) {
        for (int64_t r = 1 + t; r < repeats; r += nthr) memcpy(out + r * p->L, out, (size_t)p->L * 8);
    };
    if (nthr == 1) tile(0);
    else {
        std::thread th[3];
        for (int t = 1; t < nthr; ++t) th[t - 1] = std::thread(tile, t);
        tile(0);
        for (int t = 1; t < nthr; ++t) th[t - 1].join();
    }
    return ZMLS_OK;
}

} // extern "C"

namespace zmls {
// does a batch of n_ch channels take the pseudo-channel layout on this plan (forced, or the measured rule pc_pays), and how
// large is one of its slabs (one channel: 2^(N-PCB) rows of 128 bytes plus the padding rows)
bool pc_applies(const zmls_plan *p, int64_t n_ch)
{
    return p->pc_npass > 0 && (p->force_pc > 0 || (p->force_pc < 0 && p->force_cfg < 0 && pc_pays(p->N, n_ch)));
}
size_t pc_slab_bytes(const zmls_plan *p)
{
    return p->pc_npass ? (size_t)work_rows_total(p->N - pc_bits_of(p->dtype)) * (kVecBytes << kCfg[CFG_MID].lu) : 0;
}

// plans of orders 31 and 32, and of taps that are not maximum-length, hold the bit sequence only
int sequence_only(const zmls_plan *p)
{
    if (p->npass > 0) return ZMLS_OK;
    if (!p->maximal)
        set_error("taps do not generate a maximum-length sequence of order %d: the sequence is generated as the reference generates it, "
                  "but the impulse response needs the two-valued autocorrelation of a maximum-length sequence (the reference would "
                  "correlate against a short-period sequence without saying so)", p->N);
    else
        set_error("order %d: the sequence is generated, impulse responses are supported up to order %d (one 64-byte-row slab of the "
                  "intermediate of a 2^%d-point transform alone is %.0f GB, beside %d GiB of tables and the capture)", p->N, kMaxTransformOrder, p->N,
                  (double)work_rows_total(p->N) * 64.0 / 1e9, (int)((2ull << p->N) * 4 >> 30));
    return ZMLS_E_UNSUPPORTED;
}
// the launch sequence of the passes on buffers as they are (no layout adapters)
// `work` / `work_bytes`: the intermediate between the passes; the plan's own workspace unless the caller brings another
// (static_ws.hip), in which case a tile configuration whose slab does not fit gives way to the next narrower one.
int run_passes(zmls_plan *p, const void *d_x, int in_dtype, int64_t periods, int skip_periods, int64_t divisor,
               int64_t n_ch, int64_t row_stride, int64_t ch_stride, void *d_h, int64_t out_row_stride,
               int64_t out_ch_stride, hipStream_t st, hipEvent_t *ev /* 2 per pass, or null */,
               float *pass_ms /* accumulated, or null */, int gather_pass_only /* placement probe: stop after pass 1 */,
               void *work, size_t work_bytes)
{
    const bool own_work = work == nullptr;
    if (own_work) { work = p ? p->d_work : nullptr; work_bytes = p ? p->work_bytes : 0; }
    if (!p || !d_x || (!d_h && !gather_pass_only)) { set_error("null argument"); return ZMLS_E_ARG; }
    if (int rc = sequence_only(p)) return rc;
    if (!dtype_size(in_dtype)) { set_error("unknown input dtype %d", in_dtype); return ZMLS_E_ARG; }
    if (n_ch < 1) { set_error("need at least one channel"); return ZMLS_E_ARG; }
    if (row_stride < 1 || ch_stride < 1 || out_row_stride < 1 || out_ch_stride < 1) { set_error("strides must be positive"); return ZMLS_E_ARG; }
    if (periods < 1 || periods > 0x7FFFFFFF || divisor < 1 || divisor > 0x7FFFFFFF) { set_error("period count out of range"); return ZMLS_E_ARG; }
    ZDEVICE(p->device);
    int rc = ZMLS_OK;

    const size_t ts = dtype_size(p->dtype);
    const int E = (int)(kVecBytes / ts);
    const bool single = p->npass == 1;
    const bool vec_in = in_dtype == p->dtype && ch_stride == 1 && (row_stride * (int64_t)ts) % kVecBytes == 0 &&
                        ((uintptr_t)d_x % kVecBytes) == 0 && n_ch % E == 0;
    const bool vec_out = out_ch_stride == 1 && (out_row_stride * (int64_t)ts) % kVecBytes == 0 &&
                         ((uintptr_t)d_h % kVecBytes) == 0 && n_ch % E == 0;
    auto launch_raw = (p->dtype == ZMLS_F64) ? launch_pass_f64 : launch_pass_f32;
    // (kernel launches of this call by kind - first / middle / last pass - for the profile summaries: zmls_plan_last_launches)
    p->last_launches[0] = p->last_launches[1] = p->last_launches[2] = 0;
    auto launch = [&](int cfg_, int K_, int inm_, int outm_, const PassArgs &a_, hipStream_t s_) {
        ++p->last_launches[inm_ != IN_WORK ? 0 : outm_ == OUT_WORK ? 1 : 2];
        return launch_raw(cfg_, K_, inm_, outm_, a_, s_);
    };
    // wide vector-aligned channel blocks stream through the mid tile (128-byte row segments, two
    // register-resident workgroups per CU); everything else through the small one
    // (measured on the N=20, 512-channel, 16-period batch: small 19.3 ms, big 15.9 ms, mid 14.9 ms per step).
    // Single-pass orders (N <= 10: the whole transform of 2^(10-N) slabs in one mid tile) follow the same rule.
    int cfg = CFG_SMALL;
    if (vec_in && vec_out && n_ch >= chans_per_slab(p, CFG_MID) && p->force_cfg != CFG_SMALL) {
        // ... unless the batch is so small that the mid tile's grid (2^(N-10) macro tiles per 128-byte slab) would leave
        // CUs idle: under 512 workgroups the small tile's four times finer grid is 10-20 % faster (N=16 x 64 channels:
        // 25.6 against 31.7 us; N=14 x 256: 31.4 against 35.2 us; from N=18 x 64 = 1024 workgroups on, mid wins)
        const int cs_mid = chans_per_slab(p, CFG_MID);
        const uint64_t mid_slabs = (uint64_t)((n_ch + cs_mid - 1) / cs_mid);
        const uint64_t mid_grid = p->N > 10 ? mid_slabs << (p->N - 10) : mid_slabs >> (10 - p->N);     // single pass: whole workgroups
        if (p->tall_npass) cfg = CFG_TALL;
        else if (mid_grid >= 512 || p->force_cfg == CFG_MID) cfg = CFG_MID;
        // single-pass orders: the vector kernel wins at every size that fills one workgroup's set of slabs (N=10 x 256
        // channels x 4 periods: 12 against 29 us, x 4096: 32 against 57 us; N=8 x 1024: 12 against 28 us)
        else if (single && mid_grid >= 1) cfg = CFG_MID;
    }
    if (p->force_cfg > CFG_SMALL) {
        if (!(vec_in && vec_out)) { set_error("the wide tiles need vector-aligned contiguous channels"); return ZMLS_E_ARG; }
        cfg = p->force_cfg;
    } else if (!own_work && !single) {
        if (cfg == CFG_TALL && work_bytes < slab_bytes(p, CFG_TALL)) cfg = CFG_MID;
        if (cfg == CFG_MID && work_bytes < slab_bytes(p, CFG_MID)) cfg = CFG_SMALL;
    }
    // Few channels (the reference's own call is ONE): the pseudo-channel layout, one channel per 128-byte-row slab of
    // 2^(N-PCB) rows, any strides and sample types (fwht_kernels.inc::row_butterflies). A caller's workspace too small for
    // one such slab leaves the ordinary tiles to try.
    const int PCB = pc_bits_of(p->dtype);
    const int N_eff_pc = p->N - PCB;
    const size_t pc_slab = pc_slab_bytes(p);
    const bool pc = pc_applies(p, n_ch) && (p->pc_npass == 1 || own_work || work_bytes >= pc_slab);
    if (pc) cfg = CFG_MID;
    const TileCfg tc = kCfg[cfg];
    const int N_eff = pc ? N_eff_pc : p->N;                         // transform bits the passes see
    const int npass = pc ? p->pc_npass : (cfg == CFG_TALL) ? p->tall_npass : p->npass;
    const int *pass_bits = pc ? p->pc_bits : (cfg == CFG_TALL) ? p->tall_bits : p->bits;
    const int CS = pc ? 1 : chans_per_slab(p, cfg);
    const int64_t n_slabs = (n_ch + CS - 1) / CS;
    const size_t sbytes = pc ? pc_slab : slab_bytes(p, cfg);
    const int in_first = pc ? IN_GATHER_PC : vec_in ? IN_GATHER_VEC : IN_GATHER_ELEM;
    const int out_last = pc ? OUT_SCATTER_PC : vec_out ? OUT_SCATTER_VEC : OUT_SCATTER_ELEM;

    PassArgs a;
    memset(&a, 0, sizeof a);
    a.N = p->N;
    a.x = d_x; a.in_dtype = in_dtype; a.periods = (int)periods; a.skip_periods = skip_periods; a.divisor = (int)divisor; a.L = p->L;
    a.row_stride = row_stride; a.ch_stride = ch_stride; a.n_ch = n_ch;
    a.gather = p->d_gather;
    a.h = d_h; a.out_row_stride = out_row_stride; a.out_ch_stride = out_ch_stride;
    a.out_scale = ldexp(1.0 / (double)p->L, -p->in_scale_log2);

    if (pc && npass == 1) {
        // the whole transform of 2^LT channels per workgroup, in one launch
        const int K = pass_bits[0];
        a.single = 1; a.b = 0; a.n_macro = 1; a.ch_first = 0;
        a.scatter = p->d_scatter_nat;
        a.n_slab_groups = (uint32_t)((n_ch + (1 << log2_tiles(K, tc)) - 1) >> log2_tiles(K, tc));
        if (ev) ZHIP(hipEventRecord(ev[0], st));
        rc = launch(cfg, K, IN_GATHER_PC, OUT_SCATTER_PC, a, st);
        if (rc) return rc;
        if (ev) {
            float ms = 0.f;
            ZHIP(hipEventRecord(ev[1], st));
            ZHIP(hipEventSynchronize(ev[1]));
            ZHIP(hipEventElapsedTime(&ms, ev[0], ev[1]));
            pass_ms[0] += ms;
        }
        return ZMLS_OK;
    }
    if (single && !pc) {
        const int K = p->bits[0];
        const int LT = log2_tiles(K, tc);
        a.single = 1; a.b = 0; a.n_macro = 1;
        a.scatter = p->d_scatter_nat;
        if (ev) ZHIP(hipEventRecord(ev[0], st));
        int64_t done = 0;
        if (cfg != CFG_SMALL) {
            // the mid tile holds the whole transforms of 2^LT slabs in registers and moves 16-byte vectors; it is launched
            // on whole macro tiles only, the channels left over go through the small tile's element kernel below
            const int64_t per_wg = (int64_t)CS << LT;
            a.ch_first = 0;
            a.n_slab_groups = (uint32_t)(n_ch / per_wg);
            if (a.n_slab_groups) {
                rc = launch(cfg, K, IN_GATHER_VEC, OUT_SCATTER_VEC, a, st);
                if (rc) return rc;
            }
            done = (int64_t)a.n_slab_groups * per_wg;
        }
        if (done < n_ch) {
            const int cs_small = chans_per_slab(p, CFG_SMALL), lt_small = log2_tiles(K, kCfg[CFG_SMALL]);
            const int64_t slabs = (n_ch - done + cs_small - 1) / cs_small;
            a.ch_first = done;
            a.n_slab_groups = (uint32_t)((slabs + (1 << lt_small) - 1) >> lt_small);
            rc = launch(CFG_SMALL, K, IN_GATHER_ELEM, OUT_SCATTER_ELEM, a, st);
            if (rc) return rc;
        }
        if (ev) {
            float ms = 0.f;
            ZHIP(hipEventRecord(ev[1], st));
            ZHIP(hipEventSynchronize(ev[1]));
            ZHIP(hipEventElapsedTime(&ms, ev[0], ev[1]));
            pass_ms[0] += ms;
        }
        return ZMLS_OK;
    }

    int64_t cap = (int64_t)(work_bytes / sbytes);
    if (cap < 1) {
        if (own_work) set_error("plan workspace (%zu bytes) holds no channel slab (%zu bytes each): call zmls_plan_reserve first", work_bytes, sbytes);
        else set_error("order %d needs %zu bytes of workspace for one channel slab; the caller's workspace holds %zu", p->N, sbytes, work_bytes);
        return ZMLS_E_WORKSPACE;
    }
    a.work = work;
    a.scatter = pc ? p->d_scatter_nat : p->d_scatter_perm[cfg];
    if (!a.scatter) { set_error("tile configuration %d has not been prepared for this plan", cfg); return ZMLS_E_UNSUPPORTED; }

    // ---- the cache-resident schedule (round 3). With one or two averaged periods every pass moves the whole batch through
    // HBM, so the bytes moved are two or three times the algorithmic ones. The 256 MiB Infinity Cache takes one of those
    // round trips away when the batch runs in chunks whose intermediate fits it and the intermediate's stores and loads
    // use the default cache policy while the capture and the responses stream past non-temporally (tools/ic_probe.hip:
    // the two-pass shape moves 3.3-3.4 TB/s of capture + response bytes in chunks of 96-256 MiB against 2.2-2.3 TB/s
    // beyond 512 MiB). Two-pass orders: chunks of channel slabs, both passes per chunk. Three-pass orders: passes 1 and 2
    // are closed under the low K0+K1 bits, so they run chunk by chunk over ranges of the top bits (contiguous macro tiles
    // in both), the second in place; the third pass then runs over everything from HBM. With many periods the capture
    // stream is what counts and the non-temporal pair of round 2 stays (fwht_kernels.inc).
    const size_t budget = p->resident_bytes ? p->resident_bytes : kResidentBytes;
    const bool want_resident = p->work_layout == 0 && (p->resident_mode > 0 || (p->resident_mode < 0 && periods <= kResidentMaxPeriods));
    const bool resident2 = want_resident && npass == 2 && sbytes <= budget;       // (the placement probe's lone gather pass runs in the same chunks)
    const bool resident3 = want_resident && !gather_pass_only && npass == 3 && pass_bits[1] >= log2_tiles(pass_bits[0], tc) && pass_bits[0] >= log2_tiles(pass_bits[1], tc);
    // One launch for both passes (tuning flag 8; fwht_kernels.inc::fwht_pipeline_kernel): wide vector-aligned batches of a two-pass
    // order on the mid tile whose intermediate fits the workspace slab by slab (nothing is written twice within the launch).
    if (want_resident && (p->resident_flags & 8) && p->d_pipe && npass == 2 && !pc && !gather_pass_only && cfg == CFG_MID && vec_in && vec_out &&
        in_dtype == p->dtype && n_slabs <= (int64_t)kPipeMaxSlabs && (int64_t)(work_bytes / sbytes) >= n_slabs && sbytes < ((size_t)1 << 31) &&
        pass_bits[1] >= 7 && n_ch % CS == 0) {
        const int K0 = pass_bits[0], K1 = pass_bits[1];
        PipeArgs pa;
        memset(&pa, 0, sizeof pa);
        a.ch_first = 0; a.n_slab_groups = (uint32_t)n_slabs; a.macro_first = 0; a.slab_fastest = 0; a.swz_group = 0;
        a.work_slab_stride = work_rows_total(N_eff) * U0(tc); a.work_row_stride = U0(tc);
        a.work_slab_bytes = (uint32_t)sbytes;
        a.n_macro = (uint32_t)(1u << (N_eff - K0 - log2_tiles(K0, tc)));
        pa.p1 = a; pa.p1.b = 0; pa.p1.work_keep_st = 2; pa.p1.work_keep_ld = 1;
        pa.p2 = a; pa.p2.b = K0; pa.p2.work_keep_st = 1; pa.p2.work_keep_ld = 1;
        pa.p2.n_macro = (uint32_t)(1u << (N_eff - K1 - log2_tiles(K1, tc)));
        pa.state = p->d_pipe;
        pa.tiles = pa.p1.n_macro;
        pa.n_slabs = (uint32_t)n_slabs;
        // stream B starts one slab and a launch's worth of resident workgroups behind stream A: when a scatter tile takes its ticket,
        // the last gather tile of its slab took its own at least that many tickets ago
        const uint64_t nA = (uint64_t)pa.tiles * pa.n_slabs;
        static const uint32_t env_delay = [] { const char *v = getenv("ZMLS_PIPE_DELAY"); return (v && *v) ? (uint32_t)strtoul(v, nullptr, 10) : 0u; }();   // tuning
        const uint64_t want_delay = (uint64_t)pa.tiles + (p->pipe_delay ? p->pipe_delay : env_delay ? env_delay : 768u);
        pa.delay = (uint32_t)std::min<uint64_t>(want_delay, nA);
        if (pa.p1.n_macro == pa.p2.n_macro && nA * 2 <= 0x7FFFFFFFull) {
            ZHIP(hipMemsetAsync(p->d_pipe, 0, (((size_t)4 + pa.n_slabs) * sizeof(uint32_t) + 15) & ~(size_t)15, st));
            if (ev) ZHIP(hipEventRecord(ev[0], st));
            ++p->last_launches[0];
            rc = (p->dtype == ZMLS_F64 ? launch_pipeline_f64 : launch_pipeline_f32)(K0, K1, pa, st);
            if (rc != ZMLS_E_UNSUPPORTED) {
                if (rc) return rc;
                if (ev) {
                    float ms = 0.f;
                    ZHIP(hipEventRecord(ev[1], st));
                    ZHIP(hipEventSynchronize(ev[1]));
                    ZHIP(hipEventElapsedTime(&ms, ev[0], ev[1]));
                    pass_ms[0] += ms;
                }
                return ZMLS_OK;
            }
            --p->last_launches[0];
            rc = ZMLS_OK;       // no kernel for this split: the launches below
        }
    }
    // Two lanes (tuning flag 4): consecutive chunks alternate between the caller's stream and the plan's second one, each
    // with half the chunk size, so that one chunk's launch tails run beside the other's bodies; forked and joined with
    // events, nothing of it is visible to the caller's stream order.
    const bool two = (resident2 || resident3) && (p->resident_flags & 4) && p->aux_stream && !ev;
    const size_t lane_budget = two ? budget / 2 : budget;
    if (resident2) {
        int64_t want = std::max<int64_t>(1, (int64_t)(lane_budget / sbytes));
        if (two && cap < 2 * want) want = cap / 2;                  // both lanes' chunks live in the workspace at once
        if (want >= 1 && want < cap) cap = want;
    }
    const bool lanes2 = two && (!resident2 || (int64_t)(work_bytes / sbytes) >= 2 * cap);
    hipStream_t lane[2] = {st, lanes2 ? p->aux_stream : st};
    if (lanes2) {
        ZHIP(hipEventRecord(p->ev_fork, st));
        ZHIP(hipStreamWaitEvent(p->aux_stream, p->ev_fork, 0));
    }
    const uint64_t U = 1ull << tc.lu;
    auto swizzle_of = [&](int i, uint32_t n_macro, uint32_t ns) -> uint32_t {
        if (cfg == CFG_BIG || p->swizzle == 0) return 0;
        if (i == 0 || p->work_layout) return (n_macro % 8 == 0) ? ns : 0;
        return (((uint64_t)n_macro * ns) % 16 == 0) ? 2 : 0;
    };
    int chunk_no = 0;
    for (int64_t s0 = 0; s0 < n_slabs; s0 += cap) {
        const uint32_t ns = (uint32_t)((n_slabs - s0 < cap) ? n_slabs - s0 : cap);
        a.ch_first = s0 * CS;
        a.n_slab_groups = ns;
        a.macro_first = 0;
        a.slab_fastest = p->work_layout;
        a.work_slab_stride = p->work_layout ? U : work_rows_total(N_eff) * U;
        a.work_row_stride = p->work_layout ? (uint64_t)ns * U : U;
        // two-pass orders on two lanes: chunk k runs on lane k & 1 in that lane's half of the workspace
        hipStream_t gs = st;
        void *gwork = work;
        if (resident2 && lanes2) {
            gs = lane[chunk_no & 1];
            gwork = (char *)work + (size_t)(chunk_no & 1) * (size_t)cap * sbytes;
            ++chunk_no;
        }
        a.work = gwork;
        int b = 0, first_pass = 0;
        if (resident3) {
            const int K0 = pass_bits[0], K1 = pass_bits[1], K2 = pass_bits[2];
            const int LT0 = log2_tiles(K0, tc), LT1 = log2_tiles(K1, tc);
            const size_t per_top = ((size_t)1 << (K0 + K1)) * (size_t)(U * kVecBytes);       // bytes of one slab's rows under one value of the top bits
            const uint32_t sub = (uint32_t)std::min<uint64_t>(ns, std::max<uint64_t>(1, lane_budget / per_top));
            const uint32_t tops = (uint32_t)std::min<uint64_t>(1ull << K2, std::max<uint64_t>(1, lane_budget / (per_top * sub)));
            if (ev) ZHIP(hipEventRecord(ev[0], st));
            if (s0 > 0 && lanes2) {
                // the previous group's third pass reads the workspace this group's chunks are about to overwrite
                ZHIP(hipEventRecord(p->ev_fork, st));
                ZHIP(hipStreamWaitEvent(p->aux_stream, p->ev_fork, 0));
            }
            for (uint32_t ss = 0; ss < ns; ss += sub) {
                const uint32_t nsub = std::min(sub, ns - ss);
                a.ch_first = (s0 + ss) * CS;
                a.n_slab_groups = nsub;
                a.work = (char *)work + (size_t)ss * a.work_slab_stride * kVecBytes;
                for (uint32_t t0 = 0; t0 < (1u << K2); t0 += tops) {
                    const uint32_t nt = std::min(tops, (1u << K2) - t0);
                    hipStream_t cs = lane[chunk_no++ & 1];
                    a.b = 0; a.macro_first = t0 << (K1 - LT0); a.n_macro = nt << (K1 - LT0);
                    a.swz_group = swizzle_of(0, a.n_macro, nsub);
                    a.work_keep_ld = 0; a.work_keep_st = 1;
                    rc = launch(cfg, K0, in_first, OUT_WORK, a, cs);
                    if (rc) return rc;
                    a.b = K0; a.macro_first = t0 << (K0 - LT1); a.n_macro = nt << (K0 - LT1);
                    a.swz_group = swizzle_of(1, a.n_macro, nsub);
                    a.work_keep_ld = 1; a.work_keep_st = p->resident_flags & 1;      // on its way to HBM: streams unless told otherwise (tuning)
                    rc = launch(cfg, K1, IN_WORK, OUT_WORK, a, cs);
                    if (rc) return rc;
                }
            }
            if (lanes2) {
                ZHIP(hipEventRecord(p->ev_join, p->aux_stream));
                ZHIP(hipStreamWaitEvent(st, p->ev_join, 0));
            }
            if (ev) ZHIP(hipEventRecord(ev[1], st));
            a.ch_first = s0 * CS; a.n_slab_groups = ns; a.work = work; a.macro_first = 0;
            b = K0 + K1;
            first_pass = 2;
        }
        for (int i = first_pass; i < (gather_pass_only ? 1 : npass); ++i) {
            const int K = pass_bits[i];
            const int LT = log2_tiles(K, tc);
            a.b = b;
            a.n_macro = (uint32_t)(1u << (N_eff - K - LT));
            const int inm = (i == 0) ? in_first : IN_WORK;
            const int outm = (i == npass - 1) ? out_last : OUT_WORK;
            a.swz_group = swizzle_of(i, a.n_macro, ns);
            a.work_keep_st = resident2 ? 1 : 0;
            a.work_keep_ld = (resident2 && !(p->resident_flags & 2)) ? 1 : 0;
            if (ev) ZHIP(hipEventRecord(ev[2 * i], st));
            rc = launch(cfg, K, inm, outm, a, gs);
            if (rc) return rc;
            if (ev) ZHIP(hipEventRecord(ev[2 * i + 1], st));
            b += K;
        }
        if (ev)
            for (int i = 0; i < (gather_pass_only ? 1 : npass); ++i) {
                if (resident3 && i == 1) continue;          // passes 1 and 2 ran chunk by chunk: their time is under pass 1
                float ms = 0.f;
                ZHIP(hipEventSynchronize(ev[2 * i + 1]));
                ZHIP(hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1]));
                pass_ms[i] += ms;
            }
    }
    if (resident2 && lanes2) {
        ZHIP(hipEventRecord(p->ev_join, p->aux_stream));
        ZHIP(hipStreamWaitEvent(st, p->ev_join, 0));
    }
    return ZMLS_OK;
}

// the passes on one channel chunk, and the way out: straight into the caller's responses, or - responses contiguous in
// time - into the second staging array and transposed from there. Pad columns ride along when both sides are staged.
static int passes_and_layout_out(zmls_plan *p, const Scratch &sc, const void *src, int src_dtype, int64_t src_periods, int src_skip,
                                 int64_t divisor, int64_t n, int64_t Cs, bool staged_in, int64_t src_rs, int64_t src_cs, void *d_h,
                                 int64_t out_row_stride, int64_t out_ch_stride, bool out_cm, hipStream_t st, hipEvent_t *ev, float *pass_ms)
{
    const int64_t n_pass = (staged_in && out_cm) ? Cs : n;
    void *dst = d_h;
    int64_t dst_rs = out_row_stride, dst_cs = out_ch_stride;
    if (out_cm) { dst = sc.stage[1]; dst_rs = Cs; dst_cs = 1; }
    int rc = run_passes(p, src, src_dtype, src_periods, src_skip, divisor, n_pass, src_rs, src_cs, dst, dst_rs, dst_cs, st, ev, pass_ms, 0,
                        sc.work, sc.work_bytes);
    if (rc) return rc;
    if (out_cm) rc = launch_transpose_out(sc.stage[1], p->dtype, p->L, Cs, n, d_h, out_ch_stride, st);
    return rc;
}

// Layout adapters (layout.hip, period_sum.hip) around run_passes for multi-pass plans:
//   time-contiguous captures (a stack of single-channel arrays; also the reference's own (n,) and (n,1) inputs)
//       -> periods summed and transposed into a frame-major staging array of sums, passes on that
//   frame-major captures that cannot be read as 16-byte vectors (int16/int32/float32-into-f64, odd channel counts)
//       -> periods summed (coalesced) into the same staging array
//   time-contiguous responses -> passes write a frame-major staging array, transposed out.
// Without staging (zmls_plan_reserve_staging) everything still works through the element path, slowly.
int impulse_device_core(zmls_plan *p, const void *d_x, int in_dtype, int64_t periods, int skip_periods, int64_t divisor,
                        int64_t n_ch, int64_t row_stride, int64_t ch_stride, void *d_h, int64_t out_row_stride,
                        int64_t out_ch_stride, hipStream_t st, hipEvent_t *ev, float *pass_ms, int gather_pass_only,
                        const Scratch *shared)
{
    if (!p || !d_x || (!d_h && !gather_pass_only)) { set_error("null argument"); return ZMLS_E_ARG; }
    if (int rc = sequence_only(p)) return rc;
    void *const work = shared ? shared->work : nullptr;
    const size_t work_bytes = shared ? shared->work_bytes : 0;
    void *const *stage = shared ? shared->stage : p->d_stage;
    const size_t stage_bytes = shared ? shared->stage_bytes : p->stage_bytes;
    const size_t is = dtype_size(in_dtype), ts = dtype_size(p->dtype);
    if (!is) { set_error("unknown input dtype %d", in_dtype); return ZMLS_E_ARG; }
    if (n_ch < 1) { set_error("need at least one channel"); return ZMLS_E_ARG; }
    if (row_stride < 1 || ch_stride < 1 || out_row_stride < 1 || out_ch_stride < 1) { set_error("strides must be positive"); return ZMLS_E_ARG; }
    const int E = (int)(kVecBytes / ts);
    const bool multi = p->npass > 1 && !gather_pass_only;
    const bool vec_in = in_dtype == p->dtype && ch_stride == 1 && (row_stride * (int64_t)ts) % kVecBytes == 0 &&
                        ((uintptr_t)d_x % kVecBytes) == 0 && n_ch % E == 0;
    const bool vec_out = out_ch_stride == 1 && (out_row_stride * (int64_t)ts) % kVecBytes == 0 && ((uintptr_t)d_h % kVecBytes) == 0 && n_ch % E == 0;
    if (pc_applies(p, n_ch) && !gather_pass_only) {
        // Few channels - the reference's own call is one - take the pseudo-channel layout (run_passes), which reads and writes
        // any strides and sample types element by element: no layout adapters. Its 8-byte gathers want ONE period to gather
        // from, so with staging the periods are added up first, in time order and period order (coalesced; the same sums the
        // gather would form), and the gather divides by their count.
        const void *src = d_x;
        int src_dtype = in_dtype, src_skip = skip_periods;
        int64_t src_periods = periods, src_rs = row_stride, src_cs = ch_stride;
        if (periods > 1 && stage[0] && stage_bytes >= (size_t)p->L * (size_t)n_ch * ts) {
            ZDEVICE(p->device);
            int rc = launch_period_sum(d_x, in_dtype, p->dtype, p->L, periods, skip_periods, row_stride, ch_stride, 1, 0, 0, n_ch, n_ch, stage[0], st);
            if (rc) return rc;
            src = stage[0]; src_dtype = p->dtype; src_skip = 0; src_periods = 1; src_rs = n_ch; src_cs = 1;
        }
        return run_passes(p, src, src_dtype, src_periods, src_skip, divisor, n_ch, src_rs, src_cs, d_h, out_row_stride, out_ch_stride, st, ev,
                          pass_ms, 0, work, work_bytes);
    }
    const int64_t per_ch = p->L * (int64_t)ts;                                    // staging bytes per channel
    int64_t cc = (int64_t)(stage_bytes / (size_t)per_ch) / E * E;                 // channels a staging array holds
    const bool in_cm = multi && !vec_in && row_stride == 1 && cc >= E;
    const bool in_fm = multi && !vec_in && !in_cm && ch_stride == 1 && cc >= E;
    const bool out_cm = multi && !vec_out && out_row_stride == 1 && cc >= E;
    if (!in_cm && !in_fm && !out_cm)
        return run_passes(p, d_x, in_dtype, periods, skip_periods, divisor, n_ch, row_stride, ch_stride, d_h, out_row_stride, out_ch_stride,
                          st, ev, pass_ms, gather_pass_only, work, work_bytes);
    ZDEVICE(p->device);
    const Scratch own = {work, work_bytes, {stage[0], stage[1]}, stage_bytes};
    int rc = ZMLS_OK;
    for (int64_t c0 = 0; c0 < n_ch; c0 += cc) {
        const int64_t n = (n_ch - c0 < cc) ? n_ch - c0 : cc;
        const int64_t Cs = (n + E - 1) / E * E;
        const void *src = (const char *)d_x + (size_t)(c0 * ch_stride) * is;
        int src_dtype = in_dtype, src_skip = skip_periods;
        int64_t src_periods = periods, src_rs = row_stride, src_cs = ch_stride;
        if (in_cm || in_fm) {
            if (in_cm) rc = launch_sum_transpose(d_x, in_dtype, p->dtype, p->L, periods, skip_periods, ch_stride, c0, n_ch, stage[0], Cs, st);
            else rc = launch_period_sum(d_x, in_dtype, p->dtype, p->L, periods, skip_periods, row_stride, 1, 1, 0, c0, Cs, n_ch, stage[0], st);
            if (rc) return rc;
            src = stage[0]; src_dtype = p->dtype; src_skip = 0; src_periods = 1; src_rs = Cs; src_cs = 1;
        }
        rc = passes_and_layout_out(p, own, src, src_dtype, src_periods, src_skip, divisor, n, Cs, in_cm || in_fm, src_rs, src_cs,
                                   (char *)d_h + (size_t)(c0 * out_ch_stride) * ts, out_row_stride, out_ch_stride, out_cm, st, ev, pass_ms);
        if (rc) return rc;
    }
    return ZMLS_OK;
}

// One period of sums already in a staging-shaped array (L frames x Cs channels of the plan's dtype, Cs a whole number of
// vectors, columns n..Cs-1 zero): what impulse_device_core does after its own staging step. For callers that made the
// sums themselves (sources.hip: one set of sums, several plans).
int staged_sums_to_responses(zmls_plan *p, const Scratch &sc, const void *d_sums, int64_t Cs, int64_t divisor, int64_t n, void *d_h,
                             int64_t out_row_stride, int64_t out_ch_stride, hipStream_t st)
{
    const size_t ts = dtype_size(p->dtype);
    const int E = (int)(kVecBytes / ts);
    const bool vec_out = out_ch_stride == 1 && (out_row_stride * (int64_t)ts) % kVecBytes == 0 && ((uintptr_t)d_h % kVecBytes) == 0 && n % E == 0;
    const bool out_cm = p->npass > 1 && !vec_out && out_row_stride == 1 && sc.stage[1] && sc.stage_bytes >= (size_t)p->L * (size_t)Cs * ts;
    return passes_and_layout_out(p, sc, d_sums, p->dtype, 1, 0, divisor, n, Cs, true, Cs, 1, d_h, out_row_stride, out_ch_stride, out_cm, st,
                                 nullptr, nullptr);
}
} // namespace zmls

namespace zmls {
// a capture of R >= 2 periods: the first is thrown away, the rest are averaged (mls.py:284-306, 322-323)
int capture_periods(const zmls_plan *p, int64_t n_rows, int64_t *periods)
{
    if (!p) { set_error("null argument"); return ZMLS_E_ARG; }
    if (n_rows <= p->L) { set_error("The first sequence will be thrown away: need more than L=%lld frames", (long long)p->L); return ZMLS_E_ARG; }
    if (n_rows % p->L) { set_error("frame count %lld is not a multiple of L=%lld", (long long)n_rows, (long long)p->L); return ZMLS_E_ARG; }
    *periods = n_rows / p->L - 1;
    if (*periods > 0x7FFFFFFF) { set_error("too many periods"); return ZMLS_E_ARG; }
    return ZMLS_OK;
}
} // namespace zmls

extern "C" {

int zmls_impulse_device(zmls_plan *p, const void *d_x, int in_dtype, int64_t n_rows, int64_t n_ch, int64_t row_stride,
                        int64_t ch_stride, void *d_h, int64_t out_row_stride, int64_t out_ch_stride, void *stream)
{
    int64_t periods = 0;
    int rc = capture_periods(p, n_rows, &periods);
    if (rc) return rc;
    return impulse_device_core(p, d_x, in_dtype, periods, 1, periods, n_ch, row_stride, ch_stride, d_h, out_row_stride,
                               out_ch_stride, (hipStream_t)stream, nullptr, nullptr, 0);
}

int zmls_impulse_device_timed(zmls_plan *p, const void *d_x, int in_dtype, int64_t n_rows, int64_t n_ch, int64_t row_stride,
                              int64_t ch_stride, void *d_h, int64_t out_row_stride, int64_t out_ch_stride, void *stream,
                              float pass_ms[4])
{
    if (!p || !pass_ms) { set_error("null argument"); return ZMLS_E_ARG; }
    int64_t periods = 0;
    int rc = capture_periods(p, n_rows, &periods);
    if (rc) return rc;
    ZDEVICE(p->device);
    hipEvent_t ev[8];
    int made = 0;
    for (; made < 8; ++made)
        if (hipEventCreate(&ev[made]) != hipSuccess) break;
    for (int i = 0; i < 4; ++i) pass_ms[i] = 0.f;
    if (made == 8)
        rc = impulse_device_core(p, d_x, in_dtype, periods, 1, periods, n_ch, row_stride, ch_stride, d_h, out_row_stride,
                                 out_ch_stride, (hipStream_t)stream, ev, pass_ms, 0);
    else { set_error("could not create HIP events"); rc = ZMLS_E_HIP; }
    for (int i = 0; i < made; ++i) (void)hipEventDestroy(ev[i]);
    return rc;
}

// Device buffers of the host entry point: kept with the plan (grow-only) while they are small, so that the
// reference's own call pattern - one short capture per call - does not pay two hipMalloc/hipFree pairs (each a
// device synchronisation) per call; large ones are allocated and freed per call.
static int host_call_buffer(zmls_plan *p, int which, size_t need, void **out, bool *per_call)
{
    constexpr size_t kKeep = (size_t)256 << 20;
    *per_call = need > kKeep;
    if (*per_call) { ZHIP(hipMalloc(out, need)); return ZMLS_OK; }
    if (p->host_call_bytes[which] < need) {
        if (p->d_host_call[which]) { ZHIP(hipFree(p->d_host_call[which])); p->d_host_call[which] = nullptr; p->host_call_bytes[which] = 0; }
        size_t cap = (size_t)1 << 20;
        while (cap < need) cap <<= 1;
        ZHIP(hipMalloc(&p->d_host_call[which], cap));
        p->host_call_bytes[which] = cap;
    }
    *out = p->d_host_call[which];
    return ZMLS_OK;
}

int zmls_impulse_host(zmls_plan *p, const void *x, int in_dtype, int64_t n_rows, int64_t n_ch, int64_t row_stride,
                      int64_t ch_stride, void *h, int64_t out_row_stride, int64_t out_ch_stride)
{
    if (!p || !x || !h) { set_error("null argument"); return ZMLS_E_ARG; }
    const size_t is = dtype_size(in_dtype), ts = dtype_size(p->dtype);
    if (!is) { set_error("unknown input dtype %d", in_dtype); return ZMLS_E_ARG; }
    if (n_ch < 1 || n_rows < 1 || row_stride < 1 || ch_stride < 1 || out_row_stride < 1 || out_ch_stride < 1) { set_error("bad shape/strides"); return ZMLS_E_ARG; }
    int64_t periods = 0;
    int rc = capture_periods(p, n_rows, &periods);
    if (rc) return rc;
    ZDEVICE(p->device);
    rc = zmls_plan_reserve(p, n_ch);
    if (rc) return rc;
    {
        // anything but 16-byte-aligned frame-major data of the plan's own type goes through the staging arrays
        const int64_t E = (int64_t)(kVecBytes / ts);
        const bool plain = in_dtype == p->dtype && ch_stride == 1 && out_ch_stride == 1 && n_ch % E == 0 && row_stride % E == 0 && out_row_stride % E == 0;
        if (!plain) {
            rc = zmls_plan_reserve_staging(p, n_ch);
            if (rc) return rc;
        }
    }
    // The first period is thrown away (mls.py:284): it never crosses PCIe. The device copy starts at frame L.
    const int64_t rows = periods * p->L;
    const bool stacked = row_stride == 1 && n_ch > 1 && ch_stride >= n_rows;       // (channels, frames): one run of frames per channel
    const int64_t dev_cs = stacked ? rows : ch_stride;
    const size_t in_span = stacked ? (size_t)rows * (size_t)n_ch * is
                                   : (size_t)((rows - 1) * row_stride + (n_ch - 1) * ch_stride + 1) * is;
    const size_t out_span = (size_t)((p->L - 1) * out_row_stride + (n_ch - 1) * out_ch_stride + 1) * ts;
    const bool out_dense = out_span == (size_t)p->L * (size_t)n_ch * ts;          // every byte of the span is ours
    const char *x_from = (const char *)x + (size_t)p->L * (size_t)row_stride * is;
    // ZMLS_TRACE_HOST=1 (diagnostics): where the call's time goes, to stderr
    static const bool trace = [] { const char *v = getenv("ZMLS_TRACE_HOST"); return v && *v == '1'; }();
    auto now = [] { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e6 + t.tv_nsec * 1e-3; };
    const double t_begin = trace ? now() : 0.0;
    double t_up = 0.0, t_launch = 0.0;
    void *d_x = nullptr, *d_h = nullptr;
    bool free_x = false, free_h = false;
    rc = host_call_buffer(p, 0, in_span, &d_x, &free_x);
    if (!rc) rc = host_call_buffer(p, 1, out_span, &d_h, &free_h);
    if (!rc) {
        // large captures go up in chunks through pinned bounce buffers, several at a time (hostcopy.hip)
        if (stacked && (size_t)rows * is < ((size_t)8 << 20)) {
            hipError_t e = hipMemcpy2D(d_x, (size_t)rows * is, x_from, (size_t)ch_stride * is, (size_t)rows * is, (size_t)n_ch, hipMemcpyHostToDevice);
            if (e != hipSuccess) rc = hip_fail(e, "impulse_host upload");
        } else if (stacked) {
            for (int64_t c = 0; c < n_ch && !rc; ++c)
                rc = host_copy(p->device, (char *)d_x + (size_t)c * (size_t)rows * is, (void *)(x_from + (size_t)c * (size_t)ch_stride * is), (size_t)rows * is, true);
        } else rc = host_copy(p->device, d_x, (void *)x_from, in_span, true);
        // a strided view of a larger array: keep the bytes between the elements we own
        if (!rc && !out_dense) rc = host_copy(p->device, d_h, h, out_span, true);
    }
    if (trace) t_up = now();
    if (!rc) rc = impulse_device_core(p, d_x, in_dtype, periods, 0, periods, n_ch, row_stride, dev_cs, d_h, out_row_stride, out_ch_stride,
                                      nullptr, nullptr, nullptr, 0);
    if (trace) t_launch = now();
    if (!rc) rc = host_copy(p->device, d_h, h, out_span, false);                 // ordered after the kernels on the null stream
    if (trace) fprintf(stderr, "zmls_impulse_host N=%d: reserve + upload of %zu bytes %.0f us, launches %.0f us, kernels + download of %zu bytes %.0f us\n",
                       p->N, in_span, t_up - t_begin, t_launch - t_up, out_span, now() - t_launch);
    if (free_x && d_x) (void)hipFree(d_x);
    if (free_h && d_h) (void)hipFree(d_h);
    return rc;
}

// Where a capture lies in physical memory decides how fast the gather pass reads it: on MI355X / ROCm 7.2 the same
// 73 GB capture (N=20, 512 channels, 17 periods) at the same virtual address is gathered in 12.6-12.8 ms on some
// allocations and 13.3-15.6 ms on others, stable for the lifetime of the allocation (tools/alloc_probe.py; neither
// hipDeviceMallocContiguous nor hipMemCreate/hipMemMap placement changed that). So a capture buffer is chosen: the
// search itself is placement_search (zmls_host.cpp); here are its three callbacks on the real device.
int zmls_capture_alloc_stats(zmls_plan *p, int in_dtype, int64_t n_rows, int64_t n_ch, int tries, void **d_x, float stats[8])
{
    if (!p || !d_x) { set_error("null argument"); return ZMLS_E_ARG; }
    *d_x = nullptr;
    if (stats) memset(stats, 0, sizeof(float) * 8);
    if (int rc0 = sequence_only(p)) return rc0;
    const size_t is = dtype_size(in_dtype);
    if (!is) { set_error("unknown input dtype %d", in_dtype); return ZMLS_E_ARG; }
    int64_t periods = 0;
    int rc = capture_periods(p, n_rows, &periods);
    if (rc) return rc;
    if (n_ch < 1) { set_error("need at least one channel"); return ZMLS_E_ARG; }
    if (tries < 1) tries = 4;
    ZDEVICE(p->device);
    const size_t bytes = (size_t)n_rows * (size_t)n_ch * is;
    // small captures, and plans that do everything in one LDS-resident pass, have nothing to choose
    const bool probe = tries > 1 && p->npass > 1 && bytes >= ((size_t)1 << 30);
    if (!probe) {
        ZHIP(hipMalloc(d_x, bytes));
        return ZMLS_OK;
    }
    rc = zmls_plan_reserve(p, n_ch);
    if (rc) return rc;
    struct Ctx { zmls_plan *p; int in_dtype; int64_t periods, n_ch; size_t bytes; hipEvent_t ev[2]; } ctx = {p, in_dtype, periods, n_ch, bytes, {nullptr, nullptr}};
    ZHIP(hipEventCreate(&ctx.ev[0]));
    ZHIP(hipEventCreate(&ctx.ev[1]));
    PlacementOps ops;
    ops.ctx = &ctx;
    ops.alloc = [](void *c) -> void * {
        Ctx *x = (Ctx *)c;
        void *cand = nullptr;
        if (hipMalloc(&cand, x->bytes) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
        if (hipMemsetAsync(cand, 0, x->bytes, 0) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(cand); return nullptr; }
        return cand;
    };
    ops.release = [](void *, void *q) { (void)hipFree(q); };
    // gather-pass time on a candidate: the first run warms up, the faster of the next two counts
    ops.rate = [](void *c, void *cand, float *ms_out) -> int {
        Ctx *x = (Ctx *)c;
        float ms = 0.f, pass_ms[4];
        for (int rep = 0; rep < 3; ++rep) {
            pass_ms[0] = 0.f;
            int r = impulse_device_core(x->p, cand, x->in_dtype, x->periods, 1, x->periods, x->n_ch, x->n_ch, 1, nullptr, x->n_ch, 1, 0, x->ev, pass_ms, 1);
            if (r) return r;
            if (rep == 1 || (rep == 2 && pass_ms[0] < ms)) ms = pass_ms[0];
        }
        *ms_out = ms;
        return ZMLS_OK;
    };
    // What the gather pass of this shape takes on a good placement: the capture read at 6.36 TB/s plus the intermediate
    // written at 5.05 TB/s, the rates the pass (non-temporal streams, round 2) reaches on MI355X: fits 11.65 ms at
    // N=20 x 512 x 16 periods, 3.05 ms at N=20 x 1024 x 1, 12.2 (measured 11.8) ms at N=24 x 256 x 1; DESIGN.md §4.
    const double cells = (double)p->L * (double)n_ch;
    const double model_ms = cells * (double)periods * (double)is / 6.36e9 + cells * (double)dtype_size(p->dtype) / 5.05e9;
    PlacementStats st;
    rc = placement_search(ops, tries, model_ms, d_x, &st);
    (void)hipEventDestroy(ctx.ev[0]);
    (void)hipEventDestroy(ctx.ev[1]);
    if (rc) return rc;
    if (stats) {
        stats[0] = (float)st.candidates; stats[1] = st.first_ms; stats[2] = st.final_ms; stats[3] = st.model_ms;
        stats[4] = (float)st.rounds; stats[5] = (float)st.alloc_failed;
    }
    return ZMLS_OK;
}

int zmls_capture_alloc(zmls_plan *p, int in_dtype, int64_t n_rows, int64_t n_ch, int tries, void **d_x, float *probe_ms)
{
    float stats[8];
    const int rc = zmls_capture_alloc_stats(p, in_dtype, n_rows, n_ch, tries, d_x, stats);
    if (probe_ms) *probe_ms = rc ? 0.f : stats[2];
    return rc;
}

int zmls_capture_free(zmls_plan *p, void *d_x)
{
    if (!p) { set_error("null argument"); return ZMLS_E_ARG; }
    if (!d_x) return ZMLS_OK;
    ZDEVICE(p->device);
    int rc = ZMLS_OK;
    ZHIP(hipFree(d_x));
    return ZMLS_OK;
}

int zmls_average_host(zmls_plan *p, const void *x, int in_dtype, int64_t n_rows, int64_t n_ch, int64_t row_stride,
                      int64_t ch_stride, void *out)
{
    if (!p || !x || !out) { set_error("null argument"); return ZMLS_E_ARG; }
    const size_t is = dtype_size(in_dtype);
    if (!is) { set_error("unknown input dtype %d", in_dtype); return ZMLS_E_ARG; }
    if (n_ch < 1 || row_stride < 1 || ch_stride < 1) { set_error("bad shape/strides"); return ZMLS_E_ARG; }
    if (n_rows <= p->L || n_rows % p->L) { set_error("frame count %lld must be a multiple of L=%lld and larger than L", (long long)n_rows, (long long)p->L); return ZMLS_E_ARG; }
    ZDEVICE(p->device);
    int rc = ZMLS_OK;
    const size_t os = (in_dtype == ZMLS_F32) ? 4 : 8;
    const size_t in_span = (size_t)((n_rows - 1) * row_stride + (n_ch - 1) * ch_stride + 1) * is;
    const size_t out_bytes = (size_t)p->L * n_ch * os;
    void *d_x = nullptr, *d_o = nullptr;
    hipError_t e = hipMalloc(&d_x, in_span);
    if (e == hipSuccess) e = hipMalloc(&d_o, out_bytes);
    if (e != hipSuccess) rc = hip_fail(e, "average_host staging");
    if (!rc) rc = host_copy(p->device, d_x, (void *)x, in_span, true);
    if (!rc) rc = launch_average(d_x, in_dtype, p->L, n_rows / p->L - 1, n_ch, row_stride, ch_stride, d_o, 0);
    if (!rc) {
        rc = host_copy(p->device, d_o, out, out_bytes, false);
    }
    if (d_x) (void)hipFree(d_x);
    if (d_o) (void)hipFree(d_o);
    return rc;
}

int zmls_synth_device(zmls_plan *p, void *d_x, int dtype, int64_t repeats, int64_t n_ch, int64_t ch_first, int64_t row_stride,
                      int64_t ch_stride, uint64_t seed, double sigma, void *stream)
{
    if (!p || !d_x || repeats < 1 || n_ch < 1 || row_stride < 1 || ch_stride < 1) { set_error("bad synth arguments"); return ZMLS_E_ARG; }
    ZDEVICE(p->device);
    int rc = ZMLS_OK;
    return launch_synth(p->d_bits, p->L, d_x, dtype, repeats, n_ch, ch_first, row_stride, ch_stride, seed, sigma, (hipStream_t)stream);
}

int zmls_synth_system(const zmls_plan *p, int64_t channel, uint64_t seed, double *taps16, int64_t *delay)
{
    if (!p || !taps16 || !delay) { set_error("null argument"); return ZMLS_E_ARG; }
    synth_system(p->L, channel, seed, taps16, delay);
    return ZMLS_OK;
}

int zmls_plan_export_table(zmls_plan *p, int which, uint32_t *out)
{
    if (!p || !out || which < 0 || which > 1) { set_error("bad export arguments"); return ZMLS_E_ARG; }
    if (int rc0 = sequence_only(p)) return rc0;
    ZDEVICE(p->device);
    int rc = ZMLS_OK;
    return host_copy(p->device, which == 0 ? p->d_gather : p->d_scatter_nat, out, ((size_t)1 << p->N) * 4, false);
}

// tuning only: replace the gather table (which = 0) by the caller's, e.g. a sorted one, to time the gather pass on
// rows that are not scattered; the responses are then meaningless
int zmls_plan_import_table(zmls_plan *p, int which, const uint32_t *in)
{
    if (!p || !in || which != 0) { set_error("bad import arguments"); return ZMLS_E_ARG; }
    if (int rc0 = sequence_only(p)) return rc0;
    ZDEVICE(p->device);
    return host_copy(p->device, p->d_gather, (void *)in, ((size_t)1 << p->N) * 4, true);
}

// ---- thin HIP wrappers for ctypes callers ---------------------------------------------------
int zmls_dev_alloc(int device, void **d_ptr, size_t bytes)
{
    ZDEVICE(device);
    int rc = ZMLS_OK;
    ZHIP(settled_malloc(d_ptr, bytes));
    return ZMLS_OK;
}
int zmls_dev_free(int device, void *d_ptr)
{
    ZDEVICE(device);
    int rc = ZMLS_OK;
    ZHIP(hipFree(d_ptr));
    return ZMLS_OK;
}
int zmls_dev_upload(int device, void *d_dst, const void *src, size_t bytes)
{
    ZDEVICE(device);
    int rc = ZMLS_OK;
    return host_copy(device, d_dst, (void *)src, bytes, true);
}
int zmls_dev_download(int device, void *dst, const void *d_src, size_t bytes)
{
    ZDEVICE(device);
    int rc = ZMLS_OK;
    return host_copy(device, (void *)d_src, dst, bytes, false);
}
int zmls_dev_memset(int device, void *d_ptr, int value, size_t bytes)
{
    ZDEVICE(device);
    int rc = ZMLS_OK;
    ZHIP(hipMemset(d_ptr, value, bytes));
    return ZMLS_OK;
}
int zmls_dev_copy(int device, void *d_dst, const void *d_src, size_t bytes, void *stream)
{
    if (!d_dst || !d_src) { set_error("null argument"); return ZMLS_E_ARG; }
    ZDEVICE(device);
    int rc = ZMLS_OK;
    ZHIP(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return ZMLS_OK;
}
int zmls_dev_sync(int device)
{
    ZDEVICE(device);
    int rc = ZMLS_OK;
    ZHIP(hipDeviceSynchronize());
    return ZMLS_OK;
}
int zmls_dev_mem_info(int device, size_t *free_bytes, size_t *total_bytes)
{
    ZDEVICE(device);
    int rc = ZMLS_OK;
    ZHIP(hipMemGetInfo(free_bytes, total_bytes));
    return ZMLS_OK;
}
int zmls_stream_create(int device, void **stream)
{
    ZDEVICE(device);
    int rc = ZMLS_OK;
    hipStream_t s;
    ZHIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    *stream = (void *)s;
    return ZMLS_OK;
}
int zmls_stream_destroy(int device, void *stream)
{
    ZDEVICE(device);
    int rc = ZMLS_OK;
    ZHIP(hipStreamDestroy((hipStream_t)stream));
    return ZMLS_OK;
}
int zmls_stream_sync(int device, void *stream)
{
    ZDEVICE(device);
    int rc = ZMLS_OK;
    ZHIP(hipStreamSynchronize((hipStream_t)stream));
    return ZMLS_OK;
}
int zmls_event_create(int device, void **event)
{
    ZDEVICE(device);
    int rc = ZMLS_OK;
    hipEvent_t ev;
    ZHIP(hipEventCreate(&ev));
    *event = (void *)ev;
    return ZMLS_OK;
}
int zmls_event_destroy(int device, void *event)
{
    ZDEVICE(device);
    int rc = ZMLS_OK;
    ZHIP(hipEventDestroy((hipEvent_t)event));
    return ZMLS_OK;
}
int zmls_event_record(int device, void *event, void *stream)
{
    ZDEVICE(device);
    int rc = ZMLS_OK;
    ZHIP(hipEventRecord((hipEvent_t)event, (hipStream_t)stream));
    return ZMLS_OK;
}
int zmls_event_elapsed_ms(int device, void *start, void *stop, float *ms)
{
    ZDEVICE(device);
    int rc = ZMLS_OK;
    ZHIP(hipEventSynchronize((hipEvent_t)stop));
    ZHIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return ZMLS_OK;
}
int zmls_dev_download_channel(int device, void *dst, const void *d_src, int dtype, int64_t n_rows, int64_t row_stride,
                              int64_t ch_stride, int64_t channel)
{
    ZDEVICE(device);
    int rc = ZMLS_OK;
    const size_t es = dtype_size(dtype);
    if (!es) { set_error("bad dtype"); return ZMLS_E_ARG; }
    void *d_tmp = nullptr;
    ZHIP(hipMalloc(&d_tmp, (size_t)n_rows * es));
    rc = launch_extract_channel(d_tmp, d_src, dtype, n_rows, row_stride, ch_stride, channel, 0);
    if (!rc) rc = host_copy(device, d_tmp, dst, (size_t)n_rows * es, false);
    (void)hipFree(d_tmp);
    return rc;
}

} // extern "C"
