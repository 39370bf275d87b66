// Multi-GPU: one process per GPU, RCCL over xGMI for the one exchange the path has.
//
// Channels are independent, so the normal way to use several GPUs needs nothing from this file: every
// rank runs zmls_impulse_device on its own channels. This file is for captures whose PERIODS are spread
// over the GPUs (second mode): trim_and_avgerage (zignal/measure/mls.py:284-306) is a
// sum over periods and xcorr_circular (mls.py:169-256) is linear, so every rank adds up the periods it
// holds (period_sum.hip), the partial sums are reduce-scattered over the channel axis (rank q receives
// the complete sums of channel block q), and rank q divides, transforms and scatters its block locally.
// Reducing before the transform moves the same bytes as reducing after it and needs 1/P of the transforms.
//
// The exchange is chunked over channels and double-buffered: while RCCL moves chunk k on the communicator's
// own stream, the caller's stream adds up chunk k+1 and transforms chunk k-1.
//
// librccl is loaded on first use (dlopen), so single-GPU users never pay for it.
#include <dlfcn.h>
#include <stdlib.h>
#include <string.h>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <rccl/rccl.h>
#include "zmls_internal.h"

namespace zmls {
namespace {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*ReduceScatter)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;
    char path[512] = "";          // the file the loader resolved (dladdr), for the benchmark's line
    char why[256] = "";
    bool ok = false;
    Rccl()
    {
        // The ROCm installation's own RCCL first, by absolute path: a host program that has loaded PyTorch has that
        // wheel's bundled librccl (another ROCm release) first in line for the bare soname, and this library is built
        // against the installation's HIP. $ZMLS_RCCL_LIBRARY overrides; the bare sonames remain as the fallback.
        const char *names[] = {getenv("ZMLS_RCCL_LIBRARY"), "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so", "librccl.so.1", "librccl.so"};
        for (const char *n : names) {
            if (!n || !*n) continue;
            handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if (handle) break;
        }
        if (!handle) { snprintf(why, sizeof why, "cannot load librccl: %s", dlerror()); return; }
#define ZSYM(field, name)                                                                      \
    field = (decltype(field))dlsym(handle, name);                                              \
    if (!field) { snprintf(why, sizeof why, "librccl has no symbol %s", name); return; }
        ZSYM(GetUniqueId, "ncclGetUniqueId")
        ZSYM(CommInitRank, "ncclCommInitRank")
        ZSYM(CommDestroy, "ncclCommDestroy")
        ZSYM(ReduceScatter, "ncclReduceScatter")
        ZSYM(AllGather, "ncclAllGather")
        ZSYM(GetErrorString, "ncclGetErrorString")
        ZSYM(GetVersion, "ncclGetVersion")
#undef ZSYM
        Dl_info di;
        if (dladdr((void *)GetUniqueId, &di) && di.dli_fname) snprintf(path, sizeof path, "%s", di.dli_fname);
        ok = true;
    }
};

Rccl *rccl()
{
    static Rccl r;      // loaded once, thread-safe
    if (!r.ok) { set_error("%s", r.why); return nullptr; }
    return &r;
}

#define ZNCCL(R, call)                                                                          \
    do {                                                                                        \
        ncclResult_t r__ = (call);                                                              \
        if (r__ != ncclSuccess) { set_error("RCCL error %d (%s) in %s", (int)r__, (R)->GetErrorString(r__), #call); return ZMLS_E_COMM; } \
    } while (0)

ncclDataType_t nccl_type(int dtype) { return dtype == ZMLS_F64 ? ncclFloat64 : ncclFloat32; }

} // namespace

// ---- loopback transport (tests only, zmls_testing.h): P "ranks" that are threads of ONE process on ONE GPU, each with
// its own plan, communicator and streams. A test box has a single GPU and RCCL refuses two ranks on one device, so this
// is how the P > 1 side of zmls_impulse_period_sharded_device - block layout of the exchange, chunking, buffer reuse,
// event order - runs at all there. The collective keeps RCCL's stream semantics: it is enqueued on the rank's stream,
// reads every rank's send buffer once all of them have produced it, and is complete on a rank only when all ranks
// have finished reading that rank's send buffer.
constexpr int kMaxLoopRanks = 16;
struct LoopHub {
    int n = 0;
    std::mutex mu;
    std::condition_variable cv;
    int arrived = 0;
    uint64_t generation = 0;
    bool broken = false;
    const void *send[kMaxLoopRanks] = {};
    hipEvent_t ready[kMaxLoopRanks] = {}, done[kMaxLoopRanks] = {};
};

namespace {

// all ranks meet here on the host; false if one of them does not show up (the test then fails instead of hanging)
bool hub_barrier(LoopHub *h)
{
    std::unique_lock<std::mutex> lk(h->mu);
    if (h->broken) return false;
    const uint64_t gen = h->generation;
    if (++h->arrived == h->n) {
        h->arrived = 0;
        ++h->generation;
        h->cv.notify_all();
        return true;
    }
    if (!h->cv.wait_for(lk, std::chrono::seconds(30), [&] { return h->generation != gen || h->broken; })) h->broken = true;
    if (h->broken) { h->cv.notify_all(); return false; }
    return true;
}

struct LoopPeers { const void *send[kMaxLoopRanks]; };      // by value in the kernel arguments: fixed at launch

template <typename T>
__global__ void __launch_bounds__(256) loop_reduce_kernel(const LoopPeers peers, int n, uint64_t first, uint64_t count, T *__restrict__ recv)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (uint64_t)gridDim.x * blockDim.x) {
        T s = ((const T *)peers.send[0])[first + i];
        for (int q = 1; q < n; ++q) s += ((const T *)peers.send[q])[first + i];      // rank order, as a ring sum would not be: equal to rounding
        recv[i] = s;
    }
}

} // namespace
} // namespace zmls

struct zmls_comm {
    int rank, n_ranks, device;
    ncclComm_t nccl;
    hipStream_t cs;             // the collectives run here, beside the caller's stream
    hipEvent_t e_sum[2], e_rs[2];
    void *d_send[2], *d_recv[2];
    size_t send_bytes, recv_bytes;   // per buffer
    int64_t chunk_ch;           // channels of one rank's share that go through the exchange at a time
    int staging_dtype;
    int64_t staging_L;
    // loopback transport (tests): null for RCCL
    zmls::LoopHub *hub;
    hipEvent_t loop_ready, loop_done;
};

using namespace zmls;

static void free_staging(zmls_comm *c)
{
    for (int i = 0; i < 2; ++i) {
        if (c->d_send[i]) (void)hipFree(c->d_send[i]);
        if (c->d_recv[i]) (void)hipFree(c->d_recv[i]);
        c->d_send[i] = c->d_recv[i] = nullptr;
    }
    c->send_bytes = c->recv_bytes = 0;
    c->chunk_ch = 0;
}

// the one exchange of the path: recv[i] = sum over ranks of their send[rank * recv_count + i], on stream cs
static int reduce_scatter(zmls_comm *c, const void *send, void *recv, size_t recv_count, int dtype, hipStream_t cs)
{
    if (!c->hub) {
        Rccl *R = rccl();
        if (!R) return ZMLS_E_COMM;
        ZNCCL(R, R->ReduceScatter(send, recv, recv_count, nccl_type(dtype), ncclSum, c->nccl, cs));
        return ZMLS_OK;
    }
    LoopHub *h = c->hub;
    ZHIP(hipEventRecord(c->loop_ready, cs));                   // my send buffer is complete at this point of my stream
    h->send[c->rank] = send;
    h->ready[c->rank] = c->loop_ready;
    if (!hub_barrier(h)) { set_error("loopback exchange: a rank did not arrive"); return ZMLS_E_COMM; }
    LoopPeers peers;
    memset(&peers, 0, sizeof peers);
    for (int q = 0; q < h->n; ++q) {
        peers.send[q] = h->send[q];
        ZHIP(hipStreamWaitEvent(cs, h->ready[q], 0));
    }
    const uint32_t grid = (uint32_t)((recv_count + 255) / 256 > 4096 ? 4096 : (recv_count + 255) / 256);
    if (dtype == ZMLS_F64)
        hipLaunchKernelGGL(loop_reduce_kernel<double>, dim3(grid), dim3(256), 0, cs, peers, h->n, (uint64_t)c->rank * recv_count, (uint64_t)recv_count, (double *)recv);
    else
        hipLaunchKernelGGL(loop_reduce_kernel<float>, dim3(grid), dim3(256), 0, cs, peers, h->n, (uint64_t)c->rank * recv_count, (uint64_t)recv_count, (float *)recv);
    ZHIP(hipGetLastError());
    ZHIP(hipEventRecord(c->loop_done, cs));
    h->done[c->rank] = c->loop_done;
    if (!hub_barrier(h)) { set_error("loopback exchange: a rank did not arrive"); return ZMLS_E_COMM; }
    for (int q = 0; q < h->n; ++q) ZHIP(hipStreamWaitEvent(cs, h->done[q], 0));      // complete here only when nobody reads my send buffer any more
    if (!hub_barrier(h)) { set_error("loopback exchange: a rank did not arrive"); return ZMLS_E_COMM; }     // nobody re-publishes before all have read the hub
    return ZMLS_OK;
}

extern "C" {

int zmls_comm_loopback_hub_create(void **hub, int n_ranks)
{
    if (!hub || n_ranks < 1 || n_ranks > kMaxLoopRanks) { set_error("loopback hubs hold 1..%d ranks", kMaxLoopRanks); return ZMLS_E_ARG; }
    LoopHub *h = new LoopHub;
    h->n = n_ranks;
    *hub = h;
    return ZMLS_OK;
}

int zmls_comm_loopback_hub_destroy(void *hub)
{
    delete (LoopHub *)hub;
    return ZMLS_OK;
}

int zmls_comm_create_loopback(zmls_comm **out, void *hub, int rank, int device)
{
    if (!out || !hub) { set_error("null argument"); return ZMLS_E_ARG; }
    *out = nullptr;
    LoopHub *h = (LoopHub *)hub;
    if (rank < 0 || rank >= h->n) { set_error("rank %d outside 0..%d", rank, h->n - 1); return ZMLS_E_ARG; }
    ZDEVICE(device);
    zmls_comm *c = (zmls_comm *)calloc(1, sizeof(zmls_comm));
    if (!c) { set_error("out of host memory"); return ZMLS_E_ARG; }
    c->rank = rank; c->n_ranks = h->n; c->device = device; c->hub = h;
    hipError_t e = hipStreamCreateWithFlags(&c->cs, hipStreamNonBlocking);
    for (int i = 0; i < 2 && e == hipSuccess; ++i) {
        e = hipEventCreateWithFlags(&c->e_sum[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->e_rs[i], hipEventDisableTiming);
    }
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->loop_ready, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->loop_done, hipEventDisableTiming);
    if (e != hipSuccess) {
        const int rc = hip_fail(e, "loopback communicator");
        (void)zmls_comm_destroy(c);
        return rc;
    }
    *out = c;
    return ZMLS_OK;
}

int zmls_comm_library_info(char *path, size_t path_bytes, int *version)
{
    Rccl *r = rccl();
    if (!r) return ZMLS_E_COMM;
    if (path && path_bytes) snprintf(path, path_bytes, "%s", r->path);
    if (version) ZNCCL(r, r->GetVersion(version));
    return ZMLS_OK;
}

int zmls_comm_unique_id(uint8_t *id)
{
    if (!id) { set_error("null id"); return ZMLS_E_ARG; }
    Rccl *R = rccl();
    if (!R) return ZMLS_E_COMM;
    static_assert(sizeof(ncclUniqueId) == ZMLS_COMM_ID_BYTES, "id size");
    ncclUniqueId u;
    ZNCCL(R, R->GetUniqueId(&u));
    memcpy(id, &u, sizeof u);
    return ZMLS_OK;
}

int zmls_comm_create(zmls_comm **out, const uint8_t *id, int rank, int n_ranks, int device)
{
    if (!out || !id) { set_error("null argument"); return ZMLS_E_ARG; }
    *out = nullptr;
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks) { set_error("rank %d outside 0..%d", rank, n_ranks - 1); return ZMLS_E_ARG; }
    ZDEVICE(device);
    int rc = ZMLS_OK;
    Rccl *R = rccl();
    if (!R) return ZMLS_E_COMM;
    zmls_comm *c = (zmls_comm *)calloc(1, sizeof(zmls_comm));
    if (!c) { set_error("out of host memory"); return ZMLS_E_ARG; }
    c->rank = rank; c->n_ranks = n_ranks; c->device = device;
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    ncclResult_t r = R->CommInitRank(&c->nccl, n_ranks, u, rank);
    if (r != ncclSuccess) {
        set_error("RCCL error %d (%s) in ncclCommInitRank(rank %d of %d)", (int)r, R->GetErrorString(r), rank, n_ranks);
        free(c);
        return ZMLS_E_COMM;
    }
    hipError_t e = hipStreamCreateWithFlags(&c->cs, hipStreamNonBlocking);
    for (int i = 0; i < 2 && e == hipSuccess; ++i) {
        e = hipEventCreateWithFlags(&c->e_sum[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->e_rs[i], hipEventDisableTiming);
    }
    if (e != hipSuccess) {
        rc = hip_fail(e, "communicator stream/events");
        (void)zmls_comm_destroy(c);
        return rc;
    }
    *out = c;
    return ZMLS_OK;
}

int zmls_comm_destroy(zmls_comm *c)
{
    if (!c) return ZMLS_OK;
    DeviceScope on_device(c->device);
    if (!on_device.rc) {
        (void)hipDeviceSynchronize();
        free_staging(c);
        for (int i = 0; i < 2; ++i) {
            if (c->e_sum[i]) (void)hipEventDestroy(c->e_sum[i]);
            if (c->e_rs[i]) (void)hipEventDestroy(c->e_rs[i]);
        }
        if (c->cs) (void)hipStreamDestroy(c->cs);
        if (c->loop_ready) (void)hipEventDestroy(c->loop_ready);
        if (c->loop_done) (void)hipEventDestroy(c->loop_done);
    }
    if (!c->hub) {
        Rccl *R = rccl();
        if (R && c->nccl) (void)R->CommDestroy(c->nccl);
    }
    free(c);
    return ZMLS_OK;
}

int zmls_comm_info(const zmls_comm *c, int64_t info[8])
{
    if (!c || !info) { set_error("null argument"); return ZMLS_E_ARG; }
    memset(info, 0, sizeof(int64_t) * 8);
    info[0] = c->rank; info[1] = c->n_ranks; info[2] = c->device; info[3] = c->chunk_ch;
    info[4] = (int64_t)(2 * (c->send_bytes + c->recv_bytes));
    return ZMLS_OK;
}

int zmls_comm_reserve(zmls_comm *c, zmls_plan *p, int64_t chunk_ch)
{
    if (!c || !p || chunk_ch < 1) { set_error("bad reserve arguments"); return ZMLS_E_ARG; }
    if (p->device != c->device) { set_error("plan is on device %d, communicator on device %d", p->device, c->device); return ZMLS_E_ARG; }
    ZDEVICE(c->device);
    int rc = ZMLS_OK;
    rc = zmls_plan_reserve(p, chunk_ch);
    if (rc) return rc;
    const size_t recv = (size_t)p->L * (size_t)chunk_ch * dtype_size(p->dtype);
    const size_t send = recv * (size_t)c->n_ranks;
    if (send <= c->send_bytes && recv <= c->recv_bytes && c->staging_dtype == p->dtype && c->staging_L == p->L) {
        c->chunk_ch = chunk_ch;
        return ZMLS_OK;
    }
    ZHIP(hipDeviceSynchronize());
    free_staging(c);
    for (int i = 0; i < 2; ++i) {
        ZHIP(settled_malloc(&c->d_send[i], send));
        ZHIP(settled_malloc(&c->d_recv[i], recv));
    }
    c->send_bytes = send; c->recv_bytes = recv; c->chunk_ch = chunk_ch;
    c->staging_dtype = p->dtype; c->staging_L = p->L;
    return ZMLS_OK;
}

int zmls_comm_reduce_scatter_device(zmls_comm *c, const void *d_send, void *d_recv, int64_t recv_count, int dtype, void *stream)
{
    if (!c || !d_send || !d_recv || recv_count < 1) { set_error("bad reduce_scatter arguments"); return ZMLS_E_ARG; }
    if (dtype != ZMLS_F64 && dtype != ZMLS_F32) { set_error("reduce_scatter moves f64 or f32"); return ZMLS_E_ARG; }
    ZDEVICE(c->device);
    return reduce_scatter(c, d_send, d_recv, (size_t)recv_count, dtype, (hipStream_t)stream);
}

int zmls_comm_all_gather_device(zmls_comm *c, const void *d_send, void *d_recv, int64_t send_count, int dtype, void *stream)
{
    if (!c || !d_send || !d_recv || send_count < 1) { set_error("bad all_gather arguments"); return ZMLS_E_ARG; }
    if (dtype != ZMLS_F64 && dtype != ZMLS_F32) { set_error("all_gather moves f64 or f32"); return ZMLS_E_ARG; }
    if (c->hub) { set_error("the loopback transport has the reduce-scatter only"); return ZMLS_E_UNSUPPORTED; }
    ZDEVICE(c->device);
    int rc = ZMLS_OK;
    Rccl *R = rccl();
    if (!R) return ZMLS_E_COMM;
    ZNCCL(R, R->AllGather(d_send, d_recv, (size_t)send_count, nccl_type(dtype), c->nccl, (hipStream_t)stream));
    return ZMLS_OK;
}

int zmls_period_sum_device(zmls_plan *p, const void *d_x, int in_dtype, int64_t n_rows, int skip_periods, int64_t row_stride,
                           int64_t ch_stride, int64_t n_blocks, int64_t block_ch_stride, int64_t ch_first, int64_t block_ch,
                           void *d_out, void *stream)
{
    if (!p || !d_x || !d_out) { set_error("null argument"); return ZMLS_E_ARG; }
    if (!dtype_size(in_dtype)) { set_error("unknown input dtype %d", in_dtype); return ZMLS_E_ARG; }
    if (skip_periods < 0 || n_rows < 1 || n_rows % p->L || n_rows / p->L <= skip_periods) {
        set_error("frame count %lld must be a multiple of L=%lld holding more than the %d skipped period(s)", (long long)n_rows, (long long)p->L, skip_periods);
        return ZMLS_E_ARG;
    }
    if (row_stride < 1 || ch_stride < 1 || n_blocks < 1 || block_ch_stride < 0 || ch_first < 0 || block_ch < 1) { set_error("bad shape/strides"); return ZMLS_E_ARG; }
    ZDEVICE(p->device);
    int rc = ZMLS_OK;
    return launch_period_sum(d_x, in_dtype, p->dtype, p->L, n_rows / p->L - skip_periods, skip_periods, row_stride, ch_stride, n_blocks,
                             block_ch_stride, ch_first, block_ch, -1, d_out, (hipStream_t)stream);
}

int zmls_impulse_sums_device(zmls_plan *p, const void *d_s, int64_t total_periods, int64_t n_ch, int64_t row_stride, int64_t ch_stride,
                             void *d_h, int64_t out_row_stride, int64_t out_ch_stride, void *stream)
{
    if (!p) { set_error("null argument"); return ZMLS_E_ARG; }
    return impulse_device_core(p, d_s, p->dtype, 1, 0, total_periods, n_ch, row_stride, ch_stride, d_h, out_row_stride, out_ch_stride,
                               (hipStream_t)stream, nullptr, nullptr, 0);
}

int zmls_impulse_period_sharded_device(zmls_plan *p, zmls_comm *c, const void *d_x, int in_dtype, int64_t n_rows, int skip_periods,
                                       int64_t total_periods, int64_t n_ch, int64_t row_stride, int64_t ch_stride, void *d_h,
                                       int64_t out_row_stride, int64_t out_ch_stride, void *stream)
{
    if (!p || !c || !d_h) { set_error("null argument"); return ZMLS_E_ARG; }
    if (p->device != c->device) { set_error("plan is on device %d, communicator on device %d", p->device, c->device); return ZMLS_E_ARG; }
    if (!dtype_size(in_dtype)) { set_error("unknown input dtype %d", in_dtype); return ZMLS_E_ARG; }
    if (n_rows < 0 || n_rows % p->L || skip_periods < 0 || n_rows / p->L < skip_periods) {
        set_error("frame count %lld must be a multiple of L=%lld holding the %d skipped period(s)", (long long)n_rows, (long long)p->L, skip_periods);
        return ZMLS_E_ARG;
    }
    const int64_t local_periods = n_rows / p->L - skip_periods;
    if (local_periods > 0 && !d_x) { set_error("null capture"); return ZMLS_E_ARG; }
    if (total_periods < 1 || total_periods > 0x7FFFFFFF || local_periods > total_periods) { set_error("total period count %lld out of range", (long long)total_periods); return ZMLS_E_ARG; }
    if (n_ch < 1 || n_ch % c->n_ranks) { set_error("%lld channels do not split evenly over %d ranks", (long long)n_ch, c->n_ranks); return ZMLS_E_ARG; }
    if (row_stride < 1 || ch_stride < 1 || out_row_stride < 1 || out_ch_stride < 1) { set_error("strides must be positive"); return ZMLS_E_ARG; }
    if (c->chunk_ch < 1 || c->staging_dtype != p->dtype || c->staging_L != p->L) { set_error("call zmls_comm_reserve(comm, plan, chunk_channels) first"); return ZMLS_E_WORKSPACE; }
    ZDEVICE(c->device);
    int rc = ZMLS_OK;
    hipStream_t st = (hipStream_t)stream;
    const size_t ts = dtype_size(p->dtype);
    const int64_t share = n_ch / c->n_ranks;                 // channels this rank ends up with
    const int64_t n_chunks = (share + c->chunk_ch - 1) / c->chunk_ch;

    auto add_up = [&](int64_t k) -> int {
        const int buf = (int)(k & 1);
        const int64_t cc = (share - k * c->chunk_ch < c->chunk_ch) ? share - k * c->chunk_ch : c->chunk_ch;
        if (local_periods == 0) ZHIP(hipMemsetAsync(c->d_send[buf], 0, (size_t)c->n_ranks * p->L * cc * ts, st));
        else {
            int r = launch_period_sum(d_x, in_dtype, p->dtype, p->L, local_periods, skip_periods, row_stride, ch_stride, c->n_ranks, share,
                                      k * c->chunk_ch, cc, -1, c->d_send[buf], st);
            if (r) return r;
        }
        ZHIP(hipEventRecord(c->e_sum[buf], st));
        ZHIP(hipStreamWaitEvent(c->cs, c->e_sum[buf], 0));
        const int rs = reduce_scatter(c, c->d_send[buf], c->d_recv[buf], (size_t)(p->L * cc), p->dtype, c->cs);
        if (rs) return rs;
        ZHIP(hipEventRecord(c->e_rs[buf], c->cs));
        return ZMLS_OK;
    };

    // chunk k+1 is added up and handed to RCCL before chunk k is transformed; send/recv buffer k&1 is
    // free again by stream order (chunk k+2 is issued on `st` after the transform of chunk k)
    rc = add_up(0);
    if (rc) return rc;
    for (int64_t k = 0; k < n_chunks; ++k) {
        if (k + 1 < n_chunks) {
            rc = add_up(k + 1);
            if (rc) return rc;
        }
        const int buf = (int)(k & 1);
        const int64_t cc = (share - k * c->chunk_ch < c->chunk_ch) ? share - k * c->chunk_ch : c->chunk_ch;
        ZHIP(hipStreamWaitEvent(st, c->e_rs[buf], 0));
        rc = impulse_device_core(p, c->d_recv[buf], p->dtype, 1, 0, total_periods, cc, cc, 1,
                                 (char *)d_h + (size_t)(k * c->chunk_ch * out_ch_stride) * ts, out_row_stride, out_ch_stride, st, nullptr, nullptr, 0);
        if (rc) return rc;
    }
    return ZMLS_OK;
}

} // extern "C"
