// Host-only parts of libzmls.so: error text, GF(2) jump-ahead tables, and the helpers the CPU
// test tier calls (no GPU needed for anything in this file).
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include "zmls_internal.h"

namespace zmls {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

int hip_fail(hipError_t e, const char *what)
{
    set_error("HIP error %d (%s) in %s", (int)e, hipGetErrorString(e), what);
    return (e == hipErrorNoDevice || e == hipErrorInvalidDevice) ? ZMLS_E_NODEVICE : ZMLS_E_HIP;
}

// A^(2^(k+1)) = A^(2^k) applied to its own columns
static void square(const Gf2Map &A, Gf2Map *out, int N)
{
    for (int i = 0; i < 32; ++i) out->col[i] = (i < N) ? gf2_apply(A, A.col[i], N) : 0u;
}

void build_jump_tables(int N, uint32_t G, JumpTables *jt)
{
    memset(jt, 0, sizeof *jt);
    for (int i = 0; i < N; ++i) {
        jt->fib[0].col[i] = fib_step(1u << i, G, N);
        jt->gal[0].col[i] = gal_step(1u << i, G, N);
    }
    for (int k = 1; k < 32; ++k) {
        square(jt->fib[k - 1], &jt->fib[k], N);
        square(jt->gal[k - 1], &jt->gal[k], N);
    }
}

static uint32_t taps_to_mask(int N, const int32_t *taps, int ntaps)
{
    uint32_t G = 0;
    // a tap listed twice cancels, as XOR does in mls.py:106-107; a tap N < t <= 2N reads shiftregister[N - t], a negative
    // index, which Python counts from the newest cell: the same cell as tap t - N
    for (int i = 0; i < ntaps; ++i) G ^= 1u << (N - (taps[i] > N ? taps[i] - N : taps[i]));
    return G;
}

static uint32_t initial_window(int N, uint32_t G)
{
    // mls.py:101: cells 0,...,0,1 = sequence bits a[-N..-1]; window at n=-N has only its top bit set
    uint32_t s = 1u << (N - 1);
    for (int i = 0; i < N; ++i) s = fib_step(s, G, N);
    return s;
}

uint32_t host_fib_state(int N, uint32_t G, const JumpTables &jt, uint64_t steps)
{
    uint32_t s = initial_window(N, G);
    for (int k = 0; k < 32; ++k)
        if ((steps >> k) & 1ull) s = gf2_apply(jt.fib[k], s, N);
    return s;
}

// the window after `steps` steps for any 64-bit step count (registers that are not maximum-length do not wrap at L)
uint32_t host_fib_state_at(int N, uint32_t G, const JumpTables &jt, uint64_t steps)
{
    uint32_t s = initial_window(N, G);
    Gf2Map P = jt.fib[31], Q;
    for (int k = 0; k < 64; ++k) {
        if (k >= 32) { square(P, &Q, N); P = Q; }          // F^(2^k) beyond the table
        if ((steps >> k) & 1ull) s = gf2_apply(k < 32 ? jt.fib[k] : P, s, N);
    }
    return s;
}

// Is the register's period 2^N - 1? F^L must be the identity and F^(L/q) must not be, for every prime q dividing L.
// Only needed where the permutation tables (whose completeness proves the same thing, api.hip::build_plan) are not
// built: N = 31 (L is prime) and N = 32 (L = 3 * 5 * 17 * 257 * 65537).
static Gf2Map map_mul(const Gf2Map &A, const Gf2Map &B, int N)      // A * B
{
    Gf2Map C;
    for (int i = 0; i < 32; ++i) C.col[i] = (i < N) ? gf2_apply(A, B.col[i], N) : 0u;
    return C;
}
static bool map_is_identity(const Gf2Map &A, int N)
{
    for (int i = 0; i < N; ++i)
        if (A.col[i] != (1u << i)) return false;
    return true;
}
static Gf2Map map_power(const JumpTables &jt, uint64_t e, int N)
{
    Gf2Map R;
    for (int i = 0; i < 32; ++i) R.col[i] = (i < N) ? (1u << i) : 0u;
    for (int k = 0; k < 32; ++k)
        if ((e >> k) & 1ull) R = map_mul(jt.fib[k], R, N);
    return R;
}
bool period_is_maximal(int N, const JumpTables &jt)
{
    // (a register whose taps leave out N has a singular step matrix: no power of it is the identity)
    const uint64_t L = (1ull << N) - 1;
    if (!map_is_identity(map_power(jt, L, N), N)) return false;
    uint64_t rest = L;
    for (uint64_t q = 2; q * q <= rest; ++q) {
        if (rest % q) continue;
        while (rest % q == 0) rest /= q;
        if (map_is_identity(map_power(jt, L / q, N), N)) return false;
    }
    if (rest > 1 && rest < L && map_is_identity(map_power(jt, L / rest, N), N)) return false;
    return true;
}

int validate_taps(int N, const int32_t *taps, int ntaps, uint32_t *G, uint32_t *s0)
{
    if (N < 2 || N > kMaxSequenceOrder) { set_error("MLS order N=%d outside the supported range 2..%d", N, kMaxSequenceOrder); return ZMLS_E_ARG; }
    if (!taps || ntaps < 1 || ntaps > 32) { set_error("need 1..32 feedback taps"); return ZMLS_E_ARG; }
    // the reference checks nothing (mls.py:64-68): every tap its register can index generates a sequence, maximum-length
    // or not (taps 1..N, and N+1..2N through Python's negative indices); anything else is its IndexError at the first bit
    for (int i = 0; i < ntaps; ++i)
        if (taps[i] < 1 || taps[i] > 2 * N) { set_error("deque index out of range: tap %d of a register of N=%d cells", (int)taps[i], N); return ZMLS_E_ARG; }
    *G = taps_to_mask(N, taps, ntaps);
    *s0 = initial_window(N, *G);
    return ZMLS_OK;
}

// The capture placement search (zmls_capture_alloc, api.hip), as host logic over three callbacks so that it can be
// unit-tested without a GPU (zmls_host_placement_sim).
//
// Candidates are allocated one after the other and rated with the plan's own gather pass; the fastest is kept.
// Rejected candidates stay allocated while the search runs (at most two at a time), so the driver cannot hand the
// same memory back. Placements come in plateaus, so two candidates that agree can both sit on an upper one:
// model_ms, what the pass takes on a good placement, tells a plateau from the floor. A placement can also lose its
// speed when the candidates it was compared with are given back (seen about once in twelve searches: 12.6 ms before,
// 13.6 ms after, for the life of the buffer), so the winner is rated again after the others are freed; if it no
// longer holds its time, or sits on a plateau above a floor that is in reach, the search is repeated with it kept
// allocated (so that its memory is not handed out again).
//
// When memory is short the search degrades, it does not fail: an allocation that fails ends the current round and
// the best buffer so far is kept. A capture of more than a third of the free memory gets two candidates at most,
// one of more than half gets exactly one - the first allocation, rated once (stats->alloc_failed says so).
int placement_search(const PlacementOps &ops, int tries, double model_ms, void **out, PlacementStats *stats)
{
    PlacementStats st = {};
    st.model_ms = (float)model_ms;
    auto at_floor = [&](float ms) { return ms <= (float)(model_ms * 1.025); };
    auto floor_in_reach = [&](float ms) { return ms <= (float)(model_ms * 1.15); };     // shapes the model fits at all
    void *rejected[2] = {nullptr, nullptr};
    auto reject = [&](void *q) {
        if (rejected[0]) ops.release(ops.ctx, rejected[0]);      // at most two rejected candidates are held at a time
        rejected[0] = rejected[1];
        rejected[1] = q;
    };
    int rc = ZMLS_OK;
    void *keep = nullptr;
    float keep_ms = 0.f;
    for (int round = 0; round < 3 && !rc; ++round) {
        float second_ms = 0.f, best_ms = 0.f;
        int rated = 0;
        void *best = nullptr;
        st.rounds = round + 1;
        for (int t = 0; t < tries; ++t) {
            void *cand = ops.alloc(ops.ctx);
            if (!cand) {
                st.alloc_failed = 1;
                if (!best && !keep) { set_error("no device memory for the capture"); rc = ZMLS_E_HIP; }
                break;                                    // memory is full of candidates: settle for the best so far
            }
            float ms = 0.f;
            rc = ops.rate(ops.ctx, cand, &ms);
            if (rc) { ops.release(ops.ctx, cand); break; }
            ++rated;
            if (++st.candidates == 1) st.first_ms = ms;
            if (!best || ms < best_ms) {
                second_ms = best ? best_ms : 0.f;
                if (best) reject(best);
                best = cand; best_ms = ms;
            } else {
                if (second_ms == 0.f || ms < second_ms) second_ms = ms;
                reject(cand);
            }
            if (at_floor(best_ms)) break;
            // two candidates within 2 % of each other at the top: a plateau. Where the model says a lower one is in
            // reach, a third candidate has to agree before it is taken for the floor
            if (second_ms > 0.f && second_ms <= best_ms * 1.02f && (!floor_in_reach(best_ms) || rated >= 3)) break;
        }
        for (int i = 0; i < 2; ++i) {
            if (rejected[i]) ops.release(ops.ctx, rejected[i]);
            rejected[i] = nullptr;
        }
        if (rc) { if (best) ops.release(ops.ctx, best); break; }
        if (!best) break;                                  // nothing new could be allocated: keep what there is
        if (keep && keep_ms <= best_ms) ops.release(ops.ctx, best);
        else {
            if (keep) ops.release(ops.ctx, keep);
            keep = best; keep_ms = best_ms;
        }
        if (st.candidates == 1 && st.alloc_failed) break;  // no second buffer fits: nothing was freed, nothing to rate again
        float again = 0.f;
        rc = ops.rate(ops.ctx, keep, &again);
        if (rc) break;
        const bool holds = again <= keep_ms * 1.03f;
        keep_ms = again;
        if (holds && (at_floor(again) || !floor_in_reach(again))) break;
    }
    if (rc) { if (keep) ops.release(ops.ctx, keep); keep = nullptr; }
    st.final_ms = keep_ms;
    if (stats) *stats = st;
    *out = keep;
    return rc;
}

} // namespace zmls

using namespace zmls;

extern "C" {

// The search above on a made-up machine: the i-th buffer ever allocated is gathered in ms_of_alloc[i % n_ms]
// milliseconds (every time it is rated), and at most max_live buffers fit in memory at once. out: [0] index of the
// allocation that was kept (-1: none), [1] candidates rated, [2] first candidate's ms x 1000, [3] final ms x 1000,
// [4] rounds, [5] 1 if an allocation failed, [6] buffers still allocated at the end (must be 1), [7] most buffers live at once.
int zmls_host_placement_sim(const float *ms_of_alloc, int n_ms, int max_live, int tries, float model_ms, int64_t out[8])
{
    if (!ms_of_alloc || n_ms < 1 || max_live < 0 || !out) { set_error("bad placement simulation arguments"); return ZMLS_E_ARG; }
    struct Sim { const float *ms; int n_ms, max_live, live, peak, next; std::vector<int> alive; } sim = {ms_of_alloc, n_ms, max_live, 0, 0, 0, {}};
    PlacementOps ops;
    ops.ctx = &sim;
    ops.alloc = [](void *c) -> void * {
        Sim *s = (Sim *)c;
        if (s->live >= s->max_live) return nullptr;
        s->alive.push_back(1);
        ++s->live;
        if (s->live > s->peak) s->peak = s->live;
        return (void *)(intptr_t)(++s->next);               // handle = 1 + allocation index
    };
    ops.release = [](void *c, void *q) {
        Sim *s = (Sim *)c;
        s->alive[(size_t)((intptr_t)q - 1)] = 0;
        --s->live;
    };
    ops.rate = [](void *c, void *q, float *ms) -> int {
        Sim *s = (Sim *)c;
        *ms = s->ms[((intptr_t)q - 1) % s->n_ms];
        return ZMLS_OK;
    };
    void *kept = nullptr;
    PlacementStats st;
    const int rc = placement_search(ops, tries < 1 ? 4 : tries, (double)model_ms, &kept, &st);
    out[0] = kept ? (int64_t)((intptr_t)kept - 1) : -1;
    out[1] = st.candidates; out[2] = (int64_t)(st.first_ms * 1000.f + 0.5f); out[3] = (int64_t)(st.final_ms * 1000.f + 0.5f);
    out[4] = st.rounds; out[5] = st.alloc_failed; out[6] = sim.live; out[7] = sim.peak;
    return rc;
}

const char *zmls_last_error(void) { return g_err; }
int zmls_version(void) { return 100; }

int zmls_host_pass_split(int N, int bits[4])
{
    if (N < 2 || N > kMaxTransformOrder) { set_error("N outside 2..%d", kMaxTransformOrder); return ZMLS_E_ARG; }
    return pass_split(N, bits);
}

int zmls_host_lfsr_state(int N, const int32_t *taps, int ntaps, uint64_t steps, uint32_t *state)
{
    uint32_t G, s0;
    int rc = validate_taps(N, taps, ntaps, &G, &s0);
    if (rc) return rc;
    if (steps >> 32) { set_error("steps must fit 32 bits"); return ZMLS_E_ARG; }
    JumpTables jt;
    build_jump_tables(N, G, &jt);
    *state = host_fib_state(N, G, jt, steps);
    return ZMLS_OK;
}

// Walk the geometry exactly as the kernels do and check it is a consistent tiling.
int zmls_host_geometry_selfcheck(int K, int cfg_id, int first_pass)
{
    if (cfg_id < 0 || cfg_id >= CFG_COUNT) { set_error("bad tile configuration"); return ZMLS_E_ARG; }
    const TileCfg C = kCfg[cfg_id];
    if (K < 1 || K > (cfg_id == CFG_TALL ? kTallPassBits : kMaxPassBits) || log2_tiles(K, C) < 0) { set_error("bad geometry arguments"); return ZMLS_E_ARG; }
    const int NB = cfg_logical_bits(C);
    const uint32_t n_log = 1u << NB;
    const int LUP = C.lu - cfg_log2_phases(C);
    const int U = 1 << LUP;
    const uint32_t slots = lds_slots_total(C);
    if ((size_t)slots * kVecBytes > 160 * 1024) { set_error("LDS image larger than a CU's 160 KiB"); return -99; }
    std::vector<int> slot_owner(slots, -1);
    for (uint32_t l = 0; l < n_log; ++l)
        for (int u = 0; u < U; ++u) {
            const uint32_t s = lds_slot(l, (uint32_t)u, LUP);
            if (s >= slots || slot_owner[s] != -1) { set_error("LDS slot clash at l=%u u=%d", l, u); return -100; }
            slot_owner[s] = (int)l;
        }
    int butterflied[16] = {0};
    for (int r = 0; r < num_rounds(K, C); ++r) {
        const RoundGeo g = make_round(K, C, first_pass != 0, r);
        for (int j = 0; j < g.nbf; ++j) butterflied[g.regbit[j]]++;
        if (r == num_rounds(K, C) - 1 && fused_bit(K, C) >= 0) {
            // the bit folded into the exchange before the last round must be held by the threads there, not in registers
            bool on_thread = false;
            for (int q = 0; q < g.nthr; ++q) on_thread |= g.thrbit[q] == fused_bit(K, C);
            if (!on_thread) { set_error("fused bit %d is not a thread bit of the last round", fused_bit(K, C)); return -106; }
            butterflied[fused_bit(K, C)]++;
        }
        if (g.nthr != NB - C.rb) { set_error("round %d: %d thread bits, expected %d", r, g.nthr, NB - C.rb); return -101; }
        std::vector<int> seen(n_log, 0);
        for (uint32_t tlog = 0; tlog < (uint32_t)(cfg_threads(C) >> C.lu); ++tlog)
            for (uint32_t reg = 0; reg < (1u << C.rb); ++reg) {
                const uint32_t l = logical_of_thread(g, tlog) | logical_of_reg(g, reg, C.rb);
                if (l >= n_log) { set_error("round %d: logical index out of range", r); return -102; }
                seen[l]++;
            }
        for (uint32_t l = 0; l < n_log; ++l)
            if (seen[l] != 1) { set_error("round %d: element %u held %d times", r, l, seen[l]); return -103; }
        // butterflied bits must be exactly rows [4r, min(4r+4,K)) and sit in the low register bits
        for (int j = 0; j < g.nbf; ++j)
            if (g.regbit[j] != C.rb * r + j) { set_error("round %d: register bit %d carries logical bit %d", r, j, g.regbit[j]); return -104; }
    }
    for (int q = 0; q < NB; ++q)
        if (butterflied[q] != (q < K ? 1 : 0)) { set_error("logical bit %d is butterflied %d times", q, butterflied[q]); return -107; }
    // rows of a pass at bit offset b: every row of the 2^(K+LT+b')-row neighbourhood exactly once
    const int LT = log2_tiles(K, C);
    for (int b = 0; b <= (first_pass ? 0 : 6); ++b) {
        const uint64_t n_macro = 4;   // a few macro tiles are enough to see overlaps
        const uint64_t span = (n_macro << (LT + K)) + ((uint64_t)1 << (b + K + 8));
        std::vector<uint8_t> hit((size_t)span, 0);
        for (uint64_t macro = 0; macro < n_macro; ++macro)
            for (uint32_t l = 0; l < n_log; ++l) {
                const uint64_t d = row_of_logical(macro, l, b, K, LT);
                if (d >= span || hit[(size_t)d]) { set_error("row clash at b=%d macro=%llu l=%u", b, (unsigned long long)macro, l); return -105; }
                hit[(size_t)d] = 1;
            }
    }
    return ZMLS_OK;
}

} // extern "C"
