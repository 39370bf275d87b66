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
    for (int i = 0; i < ntaps; ++i) G ^= 1u << (N - taps[i]);   // a tap listed twice cancels, as XOR does in mls.py:106-107
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

int validate_taps(int N, const int32_t *taps, int ntaps, uint32_t *G, uint32_t *s0)
{
    if (N < 2 || N > 30) { set_error("MLS order N=%d outside the supported range 2..30", N); return ZMLS_E_ARG; }
    if (!taps || ntaps < 1 || ntaps > 32) { set_error("need 1..32 feedback taps"); return ZMLS_E_ARG; }
    for (int i = 0; i < ntaps; ++i)
        if (taps[i] < 1 || taps[i] > N) { set_error("tap %d outside 1..N (N=%d)", (int)taps[i], N); return ZMLS_E_ARG; }
    *G = taps_to_mask(N, taps, ntaps);
    if (!(*G & 1u)) { set_error("feedback taps must include N=%d (register length)", N); return ZMLS_E_ARG; }
    *s0 = initial_window(N, *G);
    return ZMLS_OK;
}

} // namespace zmls

using namespace zmls;

extern "C" {

const char *zmls_last_error(void) { return g_err; }
int zmls_version(void) { return 100; }

int zmls_host_pass_split(int N, int bits[4])
{
    if (N < 2 || N > 30) { set_error("N outside 2..30"); return ZMLS_E_ARG; }
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
