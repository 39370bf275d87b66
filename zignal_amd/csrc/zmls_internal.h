// Internal declarations shared by the translation units of libzmls.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/zmls.h"
#include "../../include/zmls_testing.h"
#include "geometry.h"

namespace zmls {

constexpr uint32_t kSentinel = 0xFFFFFFFFu;
// zignal's tap table ends at N = 32 (mlstaps.py). The bit sequence is generated for all of it; impulse responses
// (permutation tables + transform) up to N = 30: one 64-byte-row slab of the intermediate of a 2^31-point transform
// is 137 GB, of a 2^32-point one 275 GB, next to 16 / 32 GiB of tables and the capture itself in 288 GB of HBM.
constexpr int kMaxTransformOrder = 30;
constexpr int kMaxSequenceOrder = 32;
constexpr int kSynthTaps = 16;
// Which batches take the pseudo-channel layout on their own (one channel per slab, fwht_kernels.inc). Measured, device-resident
// f64 captures, ordinary slabs against the layout (tools/single_channel_timing.py, profiles/r03_single_channel.txt): ONE channel - the
// reference's call - breaks even at N = 18 (0.049 ms both) and wins from there (N = 20 x 17 periods 0.371 -> 0.082 ms, N = 24 x 2
// 2.95 -> 0.60 ms); below, both are two 10 us launches and the ordinary grid has more workgroups (N = 16: 0.024 against 0.036 ms).
// Two to four channels win at N = 20..22 only (4 channels, N = 20 x 17: 0.41 -> 0.18 ms): above, one period of the capture no longer
// fits the caches and the 8-byte gathers of the layout fetch a whole line per sample, where the ordinary slab reads all channels of
// a row in one piece.
constexpr int kPcMaxChannels = 4;
constexpr bool pc_pays(int N, int64_t n_ch) { return (n_ch == 1 && N >= 18) || (n_ch <= kPcMaxChannels && N >= 20 && N <= 22); }
constexpr int pc_bits_of(int dtype) { return dtype == ZMLS_F64 ? 4 : 5; }       // log2 of the elements of a 128-byte row
// cache-resident schedule (api.hip::run_passes): chunk size that keeps the intermediate in the 256 MiB Infinity Cache
// (measured best at 256 MiB for every shape tried, profiles/r03_resident_sweep.txt: 128 MiB chunks launch too few
// workgroups at a time, 384-512 MiB fall out of the cache). Used at every period count: the headline batch (16 periods)
// gains as much from it as the single-period ones (13.05-13.26 against 14.2-15.3 ms on one box).
constexpr size_t kResidentBytes = (size_t)256 << 20;
constexpr int kResidentMaxPeriods = 0x7FFFFFFF;

// tile configurations (geometry.h). CFG_SMALL is the general one (any channel count, any strides,
// single-pass orders); CFG_BIG is the streaming one for wide, vector-aligned channel blocks.
// CFG_TALL holds 2^11 rows of 128 bytes: orders 21 and 22 then take two passes (11+10, 11+11) instead of three.
enum { CFG_SMALL = 0, CFG_BIG = 1, CFG_MID = 2, CFG_TALL = 3, CFG_COUNT = 4 };
constexpr TileCfg kCfg[CFG_COUNT] = {{12, 4, 2, 12}, {14, 5, 4, 13}, {13, 5, 3, 12}, {14, 5, 3, 13}};

void set_error(const char *fmt, ...);
int hip_fail(hipError_t e, const char *what);
#define ZHIP(call)                                                         \
    do {                                                                   \
        hipError_t e__ = (call);                                           \
        if (e__ != hipSuccess) return ::zmls::hip_fail(e__, #call);        \
    } while (0)

// ---- GF(2) linear maps on N-bit states (N <= 32), column form: col[i] = A * e_i
struct Gf2Map { uint32_t col[32]; };
ZHD uint32_t gf2_apply(const Gf2Map &A, uint32_t s, int N)
{
    uint32_t r = 0;
    for (int i = 0; i < N; ++i) r ^= A.col[i] & (0u - ((s >> i) & 1u));
    return r;
}
// Fibonacci register: bit i of the state is sequence bit (n+i). One step drops bit 0 and
// appends parity(state & G) on top. G has bit (N-t) set for every tap t.
ZHD uint32_t fib_step(uint32_t s, uint32_t G, int N)
{
    return (s >> 1) | ((uint32_t)(__builtin_popcount(s & G) & 1) << (N - 1));
}
// Galois (dual) register: C[k+1] = x*C[k] mod p(x); C[k] are the coefficients of sequence bit
// (n+k) in terms of bits n..n+N-1.
ZHD uint32_t gal_step(uint32_t c, uint32_t G, int N)
{
    const uint32_t top = (c >> (N - 1)) & 1u;
    const uint32_t mask = (N == 32) ? 0xFFFFFFFFu : ((1u << N) - 1u);
    return ((c << 1) & mask) ^ (G & (0u - top));
}
struct JumpTables {
    Gf2Map fib[32];   // fib[k] = F^(2^k)
    Gf2Map gal[32];   // gal[k] = Gm^(2^k)
};
void build_jump_tables(int N, uint32_t G, JumpTables *jt);            // host
uint32_t host_fib_state(int N, uint32_t G, const JumpTables &jt, uint64_t steps);
uint32_t host_fib_state_at(int N, uint32_t G, const JumpTables &jt, uint64_t steps);     // any 64-bit position, no wrap at L
bool period_is_maximal(int N, const JumpTables &jt);                  // host; used where no tables are built

// ---- pass launch description
// IN_GATHER_PC / OUT_SCATTER_PC: the pseudo-channel layout of a single channel (fwht_kernels.inc): the low 4 (f64) or 5 (f32)
// bits of the transform index ride in the positions that adjacent channels have in the other modes
enum InMode { IN_WORK = 0, IN_GATHER_VEC = 1, IN_GATHER_ELEM = 2, IN_GATHER_PC = 3 };
enum OutMode { OUT_WORK = 0, OUT_SCATTER_VEC = 1, OUT_SCATTER_ELEM = 2, OUT_SCATTER_PC = 3 };

struct PassArgs {
    int N;                    // transform order
    int b;                    // bit offset of this pass
    int single;               // 1: the whole transform is this pass; tile bits select extra slabs
    uint32_t n_macro;         // macro tiles per slab (single: 1)
    uint32_t n_slab_groups;   // slabs (single: groups of 2^LT slabs) in this launch
    uint32_t swz_group;       // XCD swizzle group size (0 = identity)
    int slab_fastest;         // later passes: consecutive blocks are the slabs of one tile (else the tiles of one slab)
    void *work;               // row d of the transform of slab s at vector  s*work_slab_stride + work_row(d)*work_row_stride
    uint64_t work_slab_stride, work_row_stride;   // in 16-byte vectors: slab-major [slab][row][2^LU] or row-major [row][slab][2^LU]
    int work_keep_ld, work_keep_st;   // 1: default cache policy for the intermediate's loads / stores (cache-resident schedule), 0: non-temporal; stores 2: write-through (one-launch schedule)
    uint32_t work_slab_bytes; // one-launch schedule: bytes of one slab of the workspace (< 2^31: the stores go through a buffer descriptor)
    uint32_t macro_first;     // first macro tile of this launch (row-range chunks of the cache-resident schedule)
    // gather-in
    const void *x;
    int in_dtype;
    int periods;              // periods added up, starting at period skip_periods of the buffer
    int skip_periods;         // 1: the buffer starts with the period the reference throws away (mls.py:284)
    int divisor;              // the sum is divided by this (= periods, unless it is a partial sum)
    int64_t L;
    int64_t row_stride, ch_stride;   // elements
    int64_t n_ch;             // channels of the capture
    int64_t ch_first;         // first channel handled by slab 0 of this launch
    const uint32_t *gather;   // Q, natural order
    // scatter-out
    void *h;
    int64_t out_row_stride, out_ch_stride;
    const uint32_t *scatter;  // last pass's thread order (single: natural order)
    double out_scale;         // 1/L, times 2^-k for raw PCM captures (zmls_plan_set_input_scale_log2)
};

// Device scratch of a launch sequence: the intermediate between the passes and the two layout staging arrays. A plan
// brings its own; plans that run one after the other on one capture (several sources, sources.hip) share one set.
struct Scratch { void *work; size_t work_bytes; void *stage[2]; size_t stage_bytes; };
// the launch sequence of one impulse-response batch (api.hip); ev/pass_ms null unless timed; shared null: the plan's own scratch
int impulse_device_core(zmls_plan *p, const void *d_x, int in_dtype, int64_t periods, int skip_periods, int64_t divisor,
                        int64_t n_ch, int64_t row_stride, int64_t ch_stride, void *d_h, int64_t out_row_stride,
                        int64_t out_ch_stride, hipStream_t st, hipEvent_t *ev, float *pass_ms, int gather_pass_only,
                        const Scratch *shared = nullptr);
// the same from one period of sums the caller has staged (L x Cs frames x channels of the plan's dtype, n of them valid)
int staged_sums_to_responses(zmls_plan *p, const Scratch &sc, const void *d_sums, int64_t Cs, int64_t divisor, int64_t n, void *d_h,
                             int64_t out_row_stride, int64_t out_ch_stride, hipStream_t st);
// one period of de-emphasised period sums of a capture (iir.hip): *sums is float64, element (j, c) at j*srs + c*scs
int filtered_period_sums(zmls_plan *p, zmls_iir *f, const void *d_x, int in_dtype, int64_t n_rows, int64_t n_ch, int64_t row_stride,
                         int64_t ch_stride, const double **sums, int64_t *srs, int64_t *scs, hipStream_t st);
// the passes on buffers as they are (no layout adapters); work = null: the plan's own workspace
int run_passes(zmls_plan *p, const void *d_x, int in_dtype, int64_t periods, int skip_periods, int64_t divisor,
               int64_t n_ch, int64_t row_stride, int64_t ch_stride, void *d_h, int64_t out_row_stride,
               int64_t out_ch_stride, hipStream_t st, hipEvent_t *ev, float *pass_ms, int gather_pass_only,
               void *work, size_t work_bytes);
int capture_periods(const zmls_plan *p, int64_t n_rows, int64_t *periods);
int sequence_only(const zmls_plan *p);      // ZMLS_E_UNSUPPORTED for the plans of orders 31, 32 (no tables, no transform)
// the library's static device workspace (static_ws.hip): resolved once per device, at plan creation
int static_workspace(int device, void **ptr, size_t *bytes);
// capture placement search (zmls_host.cpp): host logic over callbacks
struct PlacementOps {
    void *ctx;
    void *(*alloc)(void *ctx);                      // a new candidate buffer, or null when memory is full
    void (*release)(void *ctx, void *buf);
    int (*rate)(void *ctx, void *buf, float *ms);   // gather-pass milliseconds on this buffer
};
struct PlacementStats { int candidates; float first_ms, final_ms, model_ms; int rounds, alloc_failed; };
int placement_search(const PlacementOps &ops, int tries, double model_ms, void **out, PlacementStats *stats);
size_t dtype_size(int dt);
int use_device(int device);
struct DeviceScope {
    int prev = -1, rc = 0;
    explicit DeviceScope(int device);
    ~DeviceScope();
    DeviceScope(const DeviceScope &) = delete;
    DeviceScope &operator=(const DeviceScope &) = delete;
};
#define ZCAT2(a, b) a##b
#define ZCAT(a, b) ZCAT2(a, b)
#define ZDEVICE(dev)                                          \
    ::zmls::DeviceScope ZCAT(zscope_, __LINE__)(dev);         \
    if (ZCAT(zscope_, __LINE__).rc) return ZCAT(zscope_, __LINE__).rc
hipError_t settled_malloc(void **ptr, size_t bytes);
bool pc_applies(const zmls_plan *p, int64_t n_ch);       // api.hip: the pseudo-channel layout serves a batch of n_ch channels
size_t pc_slab_bytes(const zmls_plan *p);                // ... in slabs of this size, one per channel
// pageable host memory <-> device memory, large copies in chunks through pinned bounce buffers on several streams (hostcopy.hip)
int host_copy(int device, void *d_ptr, void *h_ptr, size_t bytes, bool to_device);

// both passes of a two-pass order in one launch (fwht_kernels.inc::fwht_pipeline_kernel)
struct PipeArgs {
    PassArgs p1, p2;
    uint32_t *state;            // [0] ticket counter, [1] time-out flag, [4 + s] gather tiles of slab s done; zeroed before the launch
    uint32_t tiles;             // macro tiles per slab and pass
    uint32_t n_slabs;
    uint32_t delay;             // tiles by which stream B runs behind stream A (<= tiles * n_slabs)
};
constexpr uint32_t kPipeMaxSlabs = 65536;
int launch_pipeline_f64(int K0, int K1, const PipeArgs &pa, hipStream_t st);
int launch_pipeline_f32(int K0, int K1, const PipeArgs &pa, hipStream_t st);
int launch_pass_f64(int cfg, int K, int inm, int outm, const PassArgs &a, hipStream_t st);
int launch_pass_f32(int cfg, int K, int inm, int outm, const PassArgs &a, hipStream_t st);

// ---- plan-time kernels (lfsr.hip)
int launch_lfsr_tables(int N, uint32_t G, uint32_t s0, const JumpTables *d_jt, uint32_t *d_bits,
                       uint64_t n_words, uint32_t *d_gather, uint32_t *d_scatter_nat, int wrap, hipStream_t st);
int launch_scatter_perm(int N, int K, TileCfg cfg, int b, const uint32_t *d_scatter_nat, uint32_t *d_scatter_perm,
                        hipStream_t st);
int launch_sequence_expand(const uint32_t *d_bits, int64_t L, void *d_out, int dtype, int64_t repeats,
                           int64_t stride, hipStream_t st);
int launch_synth(const uint32_t *d_bits, int64_t L, void *d_x, int dtype, int64_t repeats, int64_t n_ch,
                 int64_t ch_first, int64_t row_stride, int64_t ch_stride, uint64_t seed, double sigma,
                 hipStream_t st);
int launch_average(const void *d_x, int in_dtype, int64_t L, int64_t periods, int64_t n_ch, int64_t row_stride,
                   int64_t ch_stride, void *d_out, hipStream_t st);
void synth_system(int64_t L, int64_t channel, uint64_t seed, double *taps, int64_t *delay);
int launch_period_sum(const void *d_x, int in_dtype, int out_dtype, int64_t L, int64_t periods, int skip_periods,
                      int64_t row_stride, int64_t ch_stride, int64_t n_blocks, int64_t block_ch_stride, int64_t ch_first,
                      int64_t block_ch, int64_t ch_limit /* channels at or beyond it read as zero; -1: none */, void *d_out, hipStream_t st);
int launch_sum_transpose(const void *d_x, int in_dtype, int out_dtype, int64_t L, int64_t periods, int skip_periods, int64_t ch_stride,
                         int64_t ch_first, int64_t n_ch_valid, void *d_S, int64_t Cs, hipStream_t st);
int launch_transpose_out(const void *d_H, int dtype, int64_t L, int64_t Cs, int64_t n_ch, void *d_out, int64_t out_ch_stride, hipStream_t st);
int launch_extract_channel(void *d_dst, const void *d_src, int dtype, int64_t n_rows, int64_t row_stride,
                           int64_t ch_stride, int64_t channel, hipStream_t st);

} // namespace zmls

struct zmls_plan {
    int N;
    int64_t L;
    int ntaps;
    int32_t taps[32];
    uint32_t G;
    int dtype;
    int device;
    int npass;
    int bits[4];
    int tall_npass;               // pass split of the tall tile (11 bits per pass), 0 when it saves no pass
    int tall_bits[4];
    int pc_npass;                 // pass split of the N - 4 (f64) / N - 5 (f32) row bits of the pseudo-channel layout, 0 when the order is too small for it
    int pc_bits[4];
    int last_launches[3];         // pass-kernel launches of the most recent run_passes call: first (gather) pass, middle passes, last (scatter) pass
    int force_pc;                 // -1: few-channel batches take the pseudo-channel layout, 0: never, 1: whenever it applies (tests, tuning)
    uint32_t s0;                  // Fibonacci state at n = 0 (first N sequence bits)
    int maximal;                  // the register's period is 2^N - 1 (else: the sequence is generated, the transform refused)
    uint32_t *d_bits;             // packed sequence bits, n_words uint32
    uint64_t n_words;
    uint32_t *d_gather;           // Q[d], 2^N
    uint32_t *d_scatter_nat;      // M[d], 2^N
    uint32_t *d_scatter_perm[zmls::CFG_COUNT];   // M in the last pass's thread order, per tile configuration (multi-pass plans)
    zmls::JumpTables *d_jt;
    void *d_work;
    size_t work_bytes;
    void *d_stage[2];             // frame-major staging (sums in, responses out) for captures in other layouts
    size_t stage_bytes;           // each
    void *d_host_call[2];         // capture / response buffers of zmls_impulse_host, kept while small
    size_t host_call_bytes[2];
    int force_cfg;                // -1: choose per call; else a tile configuration (tests, tuning)
    int swizzle;                  // XCD-aware block mapping on/off (tuning)
    int in_scale_log2;            // captures are raw integers with full scale 2^k: responses come out divided by 2^k
    int work_layout;              // 0: workspace slab-major, 1: row-major with slab-fastest block order in every pass
    int resident_mode;            // cache-resident schedule: -1 auto (few periods), 0 off, 1 on
    size_t resident_bytes;        // its chunk size (0: kResidentBytes)
    hipStream_t aux_stream;       // second lane of the cache-resident schedule and its fork / join events (created on demand)
    hipEvent_t ev_fork, ev_join;
    int resident_flags;           // tuning: 4 = chunks alternate between two streams, 1 = pass 2 of three keeps its stores in the cache too, 2 = non-temporal loads of the resident intermediate, 8 = both passes of a two-pass order in one launch
    uint32_t *d_pipe;             // one-launch schedule: ticket counter, time-out flag and one counter per slab (4 + kPipeMaxSlabs words, made on demand)
    uint32_t pipe_delay;          // ... tiles by which the scatter stream trails the gather stream beyond one slab (0: default)
};
