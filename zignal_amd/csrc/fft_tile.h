// Tile geometry and arithmetic of the hand-written DFT behind the magnitude response (spectrum.hip, fft.hip) - shared by the
// device kernels and the host-side emulation (zmls_host_fft_selfcheck), so there is one definition of "which thread holds
// which element in which round", checked on the CPU against a direct DFT before it ever runs on a GPU.
//
// A transform of M = 2^m complex points (double) runs as a few PASSES over HBM. Every pass works on TILES of 4096 points
// (64 KiB, one workgroup of 256 threads, 16 points per thread) that hold S = 4096 / R independent sequences of R = 2^K points:
//   strided pass  (K <= 9):  the R points of a sequence lie C points apart (C = the size of the sub-transforms still to
//                            come); the tile takes S >= 8 adjacent columns, so every row of the tile is S x 16 contiguous bytes;
//   contiguous pass (K <= 12): the last level; the tile is 4096 contiguous points = S adjacent sequences.
// Forward (decimation in frequency, "four-step"): for every level, DFT_R over the rows, multiply by w_B^(k c) (B = R C the
// block the level splits, c the column, k the output row), recurse into the C-point sub-transforms. X[k0 + R0 (k1 + R1 (...))]
// ends up at position k0 C0 + k1 C1 + ...; nothing is transposed, because the only consumer of the frequency domain is a
// point-wise product (Bluestein) or a gather (power-of-two sizes). The inverse runs the levels backwards - contiguous pass,
// then the strided ones with the twiddle BEFORE the row transform - from that order back to natural order. One tile routine
// serves both: an inverse row transform is conj -> forward DFT_R -> conj.
//
// Inside a tile a sequence is transformed in one to three ROUNDS of radix <= 16 on registers with an exchange through LDS in
// between (the same shape as the FWHT passes, fwht_kernels.inc, plus twiddles): r = d0 W0 + d1 W1 + d2, round i replaces digit
// d_i by output digit k_i in place and multiplies by w^(k_i * lower digits); the last round writes natural order k0 + R0 k1 + R0 R1 k2.
#pragma once
#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#define ZFD __host__ __device__ __forceinline__
#else
#define ZFD inline
#endif

namespace zmls {
namespace fft {

constexpr int kTileLog = 12;
constexpr int kTile = 1 << kTileLog;            // points per tile
constexpr int kThreads = 256;
constexpr int kPerThread = kTile / kThreads;    // 16
constexpr int kMaxStridedBits = 9;              // S >= 8 columns: 128-byte rows
constexpr int kLdsSlots = kTile + (kTile >> 4); // one slot of padding per 16 (keeps strided exchanges off one bank group)

struct alignas(16) cd { double x, y; };
ZFD cd operator+(cd a, cd b) { return {a.x + b.x, a.y + b.y}; }
ZFD cd operator-(cd a, cd b) { return {a.x - b.x, a.y - b.y}; }
ZFD cd cmul(cd a, cd b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
ZFD cd conj(cd a) { return {a.x, -a.y}; }
// exp(-i pi t)
ZFD cd unit_neg(double t)
{
    double s, c;
#if defined(__HIP_DEVICE_COMPILE__)
    sincospi(t, &s, &c);
#else
    // host emulation: reduce to [-1, 1) first so that large arguments keep their accuracy
    t = t - 2.0 * floor(t * 0.5 + 0.5);
    s = sin(M_PI * t); c = cos(M_PI * t);
#endif
    return {c, -s};
}

ZFD constexpr int num_rounds(int K) { return K <= 4 ? 1 : K <= 8 ? 2 : 3; }
// radix 16 first: every round but the last multiplies by twiddles - one sincospi per butterfly - and a radix-16 round has one
// butterfly per thread where two radix-8 rounds have two each (9 bits: 4 + 4 + 1, not 3 + 3 + 3)
ZFD constexpr int round_bits(int K, int rho)
{
    const int left = K - 4 * rho;
    return left > 4 ? 4 : left;
}
// weight (log2) of round rho's digit inside the in-place row position: the bits of the rounds after it
ZFD constexpr int round_weight_log(int K, int rho)
{
    int w = 0;
    for (int i = rho + 1; i < num_rounds(K); ++i) w += round_bits(K, i);
    return w;
}
ZFD constexpr uint32_t lds_slot(uint32_t idx) { return idx + (idx >> 4); }

ZFD constexpr uint32_t bit_reverse(uint32_t v, int bits)
{
    uint32_t r = 0;
    for (int i = 0; i < bits; ++i) r |= ((v >> i) & 1u) << (bits - 1 - i);
    return r;
}

// w_16^j = exp(-2 pi i j / 16), j = 0..7
ZFD constexpr cd w16(int j)
{
    constexpr double c1 = 0.92387953251128675613, s1 = 0.38268343236508977173, h = 0.70710678118654752440;
    switch (j & 7) {
    case 0: return {1.0, 0.0};
    case 1: return {c1, -s1};
    case 2: return {h, -h};
    case 3: return {s1, -c1};
    case 4: return {0.0, -1.0};
    case 5: return {-s1, -c1};
    case 6: return {-h, -h};
    default: return {-c1, -s1};
    }
}

// forward DFT of Q = 2^q points held in v[g0 .. g0 + Q): natural order in, BIT-REVERSED order out (output k sits in
// v[g0 + bit_reverse(k, q)]); radix-2 decimation in frequency, everything resolved at compile time under full unrolling
template <int q>
ZFD void dft_regs(cd (&v)[kPerThread], int g0)
{
    constexpr int Q = 1 << q;
#pragma unroll
    for (int hb = q - 1; hb >= 0; --hb) {
        const int h = 1 << hb;
#pragma unroll
        for (int a = 0; a < Q; ++a) {
            if (a & h) continue;
            const int j = a & (h - 1);                  // position inside the half block
            const cd x = v[g0 + a], y = v[g0 + a + h];
            v[g0 + a] = x + y;
            const cd t = x - y;
            const int e = (j * (8 >> hb)) & 7;          // w_(2h)^j = w_16^(j * 16 / (2h))
            if (e == 0) v[g0 + a + h] = t;
            else if (e == 4) v[g0 + a + h] = cd{t.y, -t.x};     // times -i
            else v[g0 + a + h] = cmul(t, w16(e));
        }
    }
}

// The geometry of one round of one tile shape. bf = g * 256 + tid numbers the radix-Q butterflies of the tile.
template <int K, bool STRIDED, int RHO>
struct Round {
    static constexpr int q = round_bits(K, RHO);
    static constexpr int Q = 1 << q;
    static constexpr int G = kPerThread / Q;                    // butterflies per thread
    static constexpr int wl = round_weight_log(K, RHO);         // bits below the digit
    static constexpr int ul = K - q - wl;                       // bits above it
    static constexpr int LOGS = kTileLog - K;
    static constexpr bool last = RHO == num_rounds(K) - 1;

    struct Where { uint32_t s, lo, up; };
    static ZFD Where decode(uint32_t bf)
    {
        Where w;
        if (STRIDED) {
            w.s = bf & ((1u << LOGS) - 1u);
            const uint32_t t = bf >> LOGS;
            w.lo = t & ((1u << wl) - 1u);
            w.up = t >> wl;
        } else {
            w.lo = bf & ((1u << wl) - 1u);
            const uint32_t t = bf >> wl;
            w.up = t & ((1u << ul) - 1u);
            w.s = t >> ul;
        }
        return w;
    }
    // tile index (= offset in LDS before padding) of digit value d of this butterfly, in-place layout
    static ZFD uint32_t index(const Where &w, uint32_t d)
    {
        const uint32_t r = (((w.up << q) | d) << wl) | w.lo;
        return STRIDED ? (r << LOGS) | w.s : (w.s << K) | r;
    }
    // ... and of output digit d of the LAST round in natural order: k = k0 + R0 k1 + R0 R1 k2
    static ZFD uint32_t index_natural(const Where &w, uint32_t d)
    {
        constexpr int q0 = round_bits(K, 0), q1 = num_rounds(K) > 2 ? round_bits(K, 1) : 0;
        uint32_t k;
        if (num_rounds(K) == 1) k = d;
        else if (num_rounds(K) == 2) k = w.up | (d << q0);
        else k = (w.up >> q1) | ((w.up & ((1u << q1) - 1u)) << q0) | (d << (q0 + q1));
        return STRIDED ? (k << LOGS) | w.s : (w.s << K) | k;
    }
};

// one thread's share of round RHO: read 16 points from the tile image, butterfly, twiddle. The caller writes them back with
// round_write (in place at once for all but the last round; behind a barrier for the last one, whose positions differ).
template <int K, bool STRIDED, int RHO>
ZFD void round_read_compute(const cd *lds, uint32_t tid, cd (&v)[kPerThread])
{
    using Rd = Round<K, STRIDED, RHO>;
#pragma unroll
    for (int g = 0; g < Rd::G; ++g) {
        const typename Rd::Where w = Rd::decode((uint32_t)g * kThreads + tid);
#pragma unroll
        for (int d = 0; d < Rd::Q; ++d) v[g * Rd::Q + d] = lds[lds_slot(Rd::index(w, (uint32_t)d))];
        dft_regs<Rd::q>(v, g * Rd::Q);
        if (Rd::wl > 0) {
            // times w_(Q Wt)^(k lo): powers of one unit root per butterfly, each from two of the earlier ones
            cd pw[Rd::Q];
            pw[0] = cd{1.0, 0.0};
            if (Rd::Q > 1) pw[1] = unit_neg(ldexp((double)w.lo, 1 - Rd::q - Rd::wl));
#pragma unroll
            for (int k = 2; k < Rd::Q; ++k) pw[k] = cmul(pw[k / 2], pw[k - k / 2]);
#pragma unroll
            for (int k = 1; k < Rd::Q; ++k) {
                const int reg = g * Rd::Q + (int)bit_reverse((uint32_t)k, Rd::q);
                v[reg] = cmul(v[reg], pw[k]);
            }
        }
    }
}

template <int K, bool STRIDED, int RHO>
ZFD void round_write(cd *lds, uint32_t tid, const cd (&v)[kPerThread])
{
    using Rd = Round<K, STRIDED, RHO>;
#pragma unroll
    for (int g = 0; g < Rd::G; ++g) {
        const typename Rd::Where w = Rd::decode((uint32_t)g * kThreads + tid);
#pragma unroll
        for (int k = 0; k < Rd::Q; ++k) {
            const uint32_t idx = Rd::last ? Rd::index_natural(w, (uint32_t)k) : Rd::index(w, (uint32_t)k);
            lds[lds_slot(idx)] = v[g * Rd::Q + (int)bit_reverse((uint32_t)k, Rd::q)];
        }
    }
}

// ---- the passes of one transform size
struct Level {
    int K;          // bits of the row transform
    int strided;    // 1: rows C points apart; 0: the contiguous last level
    int logC;       // log2 of the column count of the block (0 for the contiguous level)
};
struct Plan {
    int m;                  // M = 2^m
    int n_levels;
    Level level[5];
};
// strided levels of 4 ... 9 bits, as even as possible, in front of one contiguous level of at most 12 bits (2^13 ... 2^15 points:
// a 4-bit strided level and a shorter contiguous one - with 16 or more rows per tile column a thread's 16 points of a strided
// tile share their column, which is what the twiddle and chirp progressions of phase_load / phase_store rest on)
constexpr int kMinStridedBits = 4;
ZFD Plan make_plan(int m)
{
    Plan p{};
    p.m = m;
    int last = m < kTileLog ? m : kTileLog;
    int rest = m - last;
    if (rest > 0 && rest < kMinStridedBits) { rest = kMinStridedBits; last = m - rest; }
    const int ns = (rest + kMaxStridedBits - 1) / kMaxStridedBits;
    int logB = m;
    for (int i = 0; i < ns; ++i) {
        const int K = rest / ns + (i < rest % ns ? 1 : 0);
        p.level[i].K = K; p.level[i].strided = 1; p.level[i].logC = logB - K;
        logB -= K;
    }
    p.level[ns].K = last; p.level[ns].strided = 0; p.level[ns].logC = 0;
    p.n_levels = ns + 1;
    return p;
}
// position of X[k] after the forward passes
ZFD uint64_t position_of(const Plan &p, uint64_t k)
{
    uint64_t pos = 0;
    for (int i = 0; i < p.n_levels; ++i) {
        if (!p.level[i].strided) { pos += k; break; }
        pos += (k & ((1ull << p.level[i].K) - 1ull)) << p.level[i].logC;
        k >>= p.level[i].K;
    }
    return pos;
}

// Where the tiles of a pass lie. Tile numbers run pair-fastest (transforms of one batch step through the same table entries
// and twiddles together), then over the M / 4096 tiles of a transform.
struct TileAt {
    uint64_t base;      // offset of the tile's first point in the work array (transform p at p * M)
    uint64_t c0;        // strided: its first column inside the block
    uint64_t pair;
};
ZFD TileAt tile_at(uint64_t tile, uint64_t n_transforms, int m, const Level &lv)
{
    TileAt t;
    if (m < kTileLog) {         // several whole transforms per tile (contiguous level only)
        t.base = tile << kTileLog; t.c0 = 0; t.pair = 0;
        return t;
    }
    t.pair = tile % n_transforms;
    const uint64_t in = tile / n_transforms;                    // tile inside the transform
    if (lv.strided) {
        const int logS = kTileLog - lv.K;
        const uint64_t cg = in & ((1ull << (lv.logC - logS)) - 1ull), block = in >> (lv.logC - logS);
        t.c0 = cg << logS;
        t.base = (t.pair << m) + (block << (lv.logC + lv.K)) + t.c0;
    } else {
        t.c0 = 0;
        t.base = (t.pair << m) + (in << kTileLog);
    }
    return t;
}
// offset (from the tile's base) of tile point e = i * 256 + tid, and its row / column
ZFD uint64_t point_offset(uint32_t e, const Level &lv)
{
    if (!lv.strided) return e;
    const int logS = kTileLog - lv.K;
    return ((uint64_t)(e >> logS) << lv.logC) + (e & ((1u << logS) - 1u));
}
// the level's twiddle w_B^(row * column) = exp(-2 pi i row col / B), B = 2^(K + logC)
ZFD cd level_twiddle(uint32_t row, uint64_t col, const Level &lv)
{
    const int logB = lv.K + lv.logC;
    const uint64_t j = ((uint64_t)row * col) & ((1ull << logB) - 1ull);
    return unit_neg(ldexp((double)j, 1 - logB));
}

// x mod 2F for x < 2^63 (one multiplication by the reciprocal; the quotient is off by two at most)
ZFD uint64_t mod_2F(uint64_t x, uint64_t F)
{
    const uint64_t twoF = 2 * F;
    const uint64_t q = (uint64_t)((double)x * (0.5 / (double)F));
    int64_t r = (int64_t)(x - q * twoF);
    while (r < 0) r += (int64_t)twoF;
    while (r >= (int64_t)twoF) r -= (int64_t)twoF;
    return (uint64_t)r;
}
// exp(-i pi j / F) for an integer j < 2F
ZFD cd chirp_root(uint64_t j, uint64_t F) { return unit_neg((double)j / (double)F); }
// Bluestein's chirp exp(-i pi t^2 / F): t^2 mod 2F exactly in integers (t < 2^31), then one unit root
ZFD cd chirp(uint64_t t, uint64_t F) { return chirp_root(mod_2F(t * t, F), F); }
// ... along t = t0 + i d, i = 0..15: c(t0 + i d) = c(t0) A^i B^(i^2) with A = exp(-2 i pi t0 d / F) (one unit root per thread, its
// powers by for_twiddle_sequence) and B^(i^2) = exp(-i pi i^2 d^2 / F), sixteen constants of the PASS (d is the same for every
// thread of a pass) that the host computes in long double (chirp_pass_constants) and hands over with the kernel arguments.
ZFD cd chirp_ratio(uint64_t t0, uint64_t d, uint64_t F) { return chirp_root(mod_2F(2 * mod_2F(t0 * d, F), F), F); }

// ---- one pass = every tile through: load -> rounds -> store. The phases are separate functions of (tile, thread) so that the
// device kernel (barriers in between) and the host emulation (a loop over the threads per phase) run the very same code.
enum { LOAD_WORK = 0, LOAD_STAGE = 1, LOAD_STAGE_CHIRP = 2, LOAD_BSEQ = 3 };
enum { STORE_WORK = 0, STORE_WORK_TIMES_TABLE = 1, STORE_FINAL_CHIRP = 2 };
struct PassArgs {
    cd *work;                   // n_transforms x M, transformed in place
    uint64_t n_transforms;
    uint64_t total;             // n_transforms << m
    int m;
    Level lv;
    int inv;                    // 1: inverse row transforms (conj - forward - conj), twiddle before them
    int load_mode, store_mode;
    const cd *stage;            // LOAD_STAGE*: n_stage points per transform, zero beyond
    uint64_t n_stage;
    uint64_t F;                 // Bluestein: the DFT size the chirps belong to
    const cd *table;            // STORE_WORK_TIMES_TABLE: M points (the chirp filter's spectrum, in the passes' own order)
    cd *Y;                      // STORE_FINAL_CHIRP: F points per transform
    double scale;
    cd chirp_i2[kPerThread];    // chirp passes: exp(-i pi i^2 d^2 / F) for the pass's point distance d (chirp_pass_constants)
};

ZFD cd pass_load_point(const PassArgs &a, uint64_t g)
{
    const uint64_t M = 1ull << a.m, pos = g & (M - 1ull), tr = g >> a.m;
    switch (a.load_mode) {
    case LOAD_WORK: return a.work[g];
    case LOAD_STAGE: return pos < a.n_stage ? a.stage[tr * a.n_stage + pos] : cd{0.0, 0.0};
    case LOAD_STAGE_CHIRP: return pos < a.n_stage ? cmul(a.stage[tr * a.n_stage + pos], chirp(pos, a.F)) : cd{0.0, 0.0};
    default: {
        // the chirp filter b[u] = exp(+i pi u^2 / F), |u| < F, wrapped around M
        const uint64_t u = pos < a.F ? pos : (M - pos < a.F ? M - pos : M);
        return u < M ? conj(chirp(u, a.F)) : cd{0.0, 0.0};
    }
    }
}
ZFD void pass_store_point(const PassArgs &a, uint64_t g, cd v)
{
    const uint64_t M = 1ull << a.m, pos = g & (M - 1ull), tr = g >> a.m;
    switch (a.store_mode) {
    case STORE_WORK: a.work[g] = v; break;
    case STORE_WORK_TIMES_TABLE: a.work[g] = cmul(v, a.table[pos]); break;
    default:
        if (pos < a.F) {
            const cd o = cmul(v, chirp(pos, a.F));
            a.Y[tr * a.F + pos] = cd{o.x * a.scale, o.y * a.scale};
        }
        break;
    }
}

// The 16 points of a thread, e = i * 256 + tid. From 2^12 points up a tile lies inside one transform and those points are an
// arithmetic progression there - contiguous tile: 256 apart; strided tile (K >= 4, so S <= 256): one column, rows 2^(K-4) apart -
// so the level's twiddles are base * step^i and the chirp is c(t0) A^i times sixteen constants of the pass (chirp_ratio) instead of
// one sincospi and one modular square per point.
template <int K, bool STRIDED>
struct ThreadPoints {
    static constexpr int LOGS = kTileLog - K;
    uint64_t g0, step;      // work offsets g0 + i * step
    uint64_t pos0, tr;      // position inside the transform, transform number
    uint32_t row0, drow;    // strided: rows row0 + i * drow of column col
    uint64_t col;
    ZFD ThreadPoints(const PassArgs &a, const TileAt &at, uint32_t tid)
    {
        if (STRIDED) {
            row0 = tid >> LOGS; drow = kThreads >> LOGS;
            const uint32_t s = tid & ((1u << LOGS) - 1u);
            col = at.c0 + s;
            g0 = at.base + ((uint64_t)row0 << a.lv.logC) + s;
            step = (uint64_t)drow << a.lv.logC;
        } else {
            row0 = 0; drow = 0; col = 0;
            g0 = at.base + tid; step = kThreads;
        }
        pos0 = g0 & ((1ull << a.m) - 1ull);
        tr = g0 >> a.m;
    }
};
// w^(row0 col) * (w^(drow col))^i, i = 0..15, each from the one with its lowest set bit cleared: at most four products deep
template <class F>
ZFD void for_twiddle_sequence(cd base, cd step, F &&f)
{
    cd s1 = step, s2 = cmul(s1, s1), s4 = cmul(s2, s2), s8 = cmul(s4, s4);
    cd t0 = base, t8 = cmul(base, s8);
#pragma unroll
    for (int hi = 0; hi < 2; ++hi) {
        const cd b = hi ? t8 : t0;
        const cd b4 = cmul(b, s4);
#pragma unroll
        for (int mid = 0; mid < 2; ++mid) {
            const cd c = mid ? b4 : b;
            const cd c2 = cmul(c, s2);
            f(hi * 8 + mid * 4 + 0, c);
            f(hi * 8 + mid * 4 + 1, cmul(c, s1));
            f(hi * 8 + mid * 4 + 2, c2);
            f(hi * 8 + mid * 4 + 3, cmul(c2, s1));
        }
    }
}

template <int K, bool STRIDED>
ZFD void phase_load(const PassArgs &a, const TileAt &at, uint32_t tid, cd *lds)
{
    constexpr int LOGS = kTileLog - K;
    if (a.m < kTileLog || a.load_mode == LOAD_BSEQ || (STRIDED && K < kMinStridedBits)) {
        // several transforms per tile / the chirp filter (once per size): point by point
#pragma unroll 1
        for (int i = 0; i < kPerThread; ++i) {
            const uint32_t e = (uint32_t)i * kThreads + tid;
            const uint64_t g = at.base + point_offset(e, a.lv);
            cd v{0.0, 0.0};
            if (g < a.total) {
                v = pass_load_point(a, g);
                if (a.inv) {
                    v = conj(v);
                    if (STRIDED) v = cmul(v, level_twiddle(e >> LOGS, at.c0 + (e & ((1u << LOGS) - 1u)), a.lv));
                }
            }
            lds[lds_slot(e)] = v;
        }
        return;
    }
    const ThreadPoints<K, STRIDED> tp(a, at, tid);
    cd v[kPerThread];
    if (a.load_mode == LOAD_WORK) {
#pragma unroll
        for (int i = 0; i < kPerThread; ++i) v[i] = a.work[tp.g0 + (uint64_t)i * tp.step];
    } else {
        const cd *src = a.stage + tp.tr * a.n_stage;
#pragma unroll
        for (int i = 0; i < kPerThread; ++i) {
            const uint64_t t = tp.pos0 + (uint64_t)i * tp.step;
            v[i] = t < a.n_stage ? src[t] : cd{0.0, 0.0};
        }
        if (a.load_mode == LOAD_STAGE_CHIRP && tp.pos0 < a.n_stage)
            for_twiddle_sequence(chirp(tp.pos0, a.F), chirp_ratio(tp.pos0, tp.step, a.F),
                                 [&](int i, cd w) { v[i] = cmul(v[i], cmul(w, a.chirp_i2[i])); });
    }
    if (a.inv) {
#pragma unroll
        for (int i = 0; i < kPerThread; ++i) v[i] = conj(v[i]);
        if (STRIDED)
            for_twiddle_sequence(level_twiddle(tp.row0, tp.col, a.lv), level_twiddle(tp.drow, tp.col, a.lv),
                                 [&](int i, cd w) { v[i] = cmul(v[i], w); });
    }
#pragma unroll
    for (int i = 0; i < kPerThread; ++i) lds[lds_slot((uint32_t)i * kThreads + tid)] = v[i];
}

template <int K, bool STRIDED>
ZFD void phase_store(const PassArgs &a, const TileAt &at, uint32_t tid, const cd *lds)
{
    constexpr int LOGS = kTileLog - K;
    if (a.m < kTileLog || (STRIDED && K < kMinStridedBits)) {
#pragma unroll 1
        for (int i = 0; i < kPerThread; ++i) {
            const uint32_t e = (uint32_t)i * kThreads + tid;
            const uint64_t g = at.base + point_offset(e, a.lv);
            if (g >= a.total) continue;
            cd v = lds[lds_slot(e)];
            if (a.inv) v = conj(v);
            else if (STRIDED) v = cmul(v, level_twiddle(e >> LOGS, at.c0 + (e & ((1u << LOGS) - 1u)), a.lv));
            pass_store_point(a, g, v);
        }
        return;
    }
    const ThreadPoints<K, STRIDED> tp(a, at, tid);
    cd v[kPerThread];
#pragma unroll
    for (int i = 0; i < kPerThread; ++i) v[i] = lds[lds_slot((uint32_t)i * kThreads + tid)];
    if (a.inv) {
#pragma unroll
        for (int i = 0; i < kPerThread; ++i) v[i] = conj(v[i]);
    } else if (STRIDED) {
        for_twiddle_sequence(level_twiddle(tp.row0, tp.col, a.lv), level_twiddle(tp.drow, tp.col, a.lv),
                             [&](int i, cd w) { v[i] = cmul(v[i], w); });
    }
    if (a.store_mode == STORE_WORK) {
#pragma unroll
        for (int i = 0; i < kPerThread; ++i) a.work[tp.g0 + (uint64_t)i * tp.step] = v[i];
    } else if (a.store_mode == STORE_WORK_TIMES_TABLE) {
        cd w[kPerThread];
#pragma unroll
        for (int i = 0; i < kPerThread; ++i) w[i] = a.table[tp.pos0 + (uint64_t)i * tp.step];
#pragma unroll
        for (int i = 0; i < kPerThread; ++i) a.work[tp.g0 + (uint64_t)i * tp.step] = cmul(v[i], w[i]);
    } else if (tp.pos0 < a.F) {
        cd *dst = a.Y + tp.tr * a.F;
        for_twiddle_sequence(chirp(tp.pos0, a.F), chirp_ratio(tp.pos0, tp.step, a.F),
                             [&](int i, cd w) { v[i] = cmul(v[i], cmul(w, a.chirp_i2[i])); });
#pragma unroll
        for (int i = 0; i < kPerThread; ++i) {
            const uint64_t t = tp.pos0 + (uint64_t)i * tp.step;
            if (t < a.F) dst[t] = cd{v[i].x * a.scale, v[i].y * a.scale};
        }
    }
}

} // namespace fft
} // namespace zmls
