// Tile geometry of one FWHT pass — shared by the device kernels, the table builders and the
// host-side self check (zmls_host_geometry_selfcheck), so there is exactly one definition of
// "which thread holds which element in which round".
//
// A workgroup owns a MACRO TILE of 2^LOGV sixteen-byte vectors:
//     2^LT tiles  x  2^K rows  x  2^LU vectors per row        (LT = LOGV - LU - K)
// A "vector" is 16 bytes of adjacent channels (2 doubles or 4 floats); a row of a tile is
// 2^LU vectors = one contiguous row segment of a channel slab. The pass butterflies over the K
// row bits; tiles of a macro tile are independent (they only share the workgroup).
//
// Every thread keeps 2^RB vectors in registers (2^(LOGV-RB) threads). A pass is ceil(K/RB)
// ROUNDS (two for K = 2*RB+1, see fused_bit); round r butterflies row bits [RB*r, min(RB*r+RB, K)) entirely in registers, and between
// rounds the macro tile is exchanged through LDS. For each round the LOGV-LU "logical bits" (row
// bits 0..K-1, then tile bits K..K+LT-1) are split into RB register-index bits and the thread bits.
//
// Configurations built (zmls_internal.h):
//   small  LOGV=12 RB=4 LU=2 :  64 KiB tile, 256 threads, 2 workgroups/CU,  64-byte row segments
//   mid    LOGV=13 RB=5 LU=3 : 128 KiB tile, 256 threads, 2 workgroups/CU, 128-byte row segments
//   big    LOGV=14 RB=5 LU=4 : 256 KiB tile, 512 threads, 1 workgroup/CU,  256-byte row segments
//   tall   LOGV=14 RB=5 LU=3 : 256 KiB tile, 512 threads, 1 workgroup/CU,  128-byte row segments, 2^11 rows
// The wider ones exist because random row gathers from HBM run at 3.2 / 5.2 / 5.7 TB/s for 64 / 128 /
// 256-byte segments on MI355X (tools/gather_probe.hip); their tiles live in the register file
// (512 KiB per CU) and go through LDS in two halves.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define ZHD __host__ __device__ inline
#else
#define ZHD inline
#endif

namespace zmls {

constexpr int kMaxPassBits = 10;
constexpr int kTallPassBits = 11;   // the tall tile's passes (zmls_internal.h)
constexpr int kVecBytes = 16;

struct TileCfg {
    int logv;   // log2(vectors per macro tile)
    int rb;     // log2(vectors per thread)
    int lu;     // log2(vectors per row segment)
    int llds;   // log2(vectors the workgroup's LDS image holds at a time)
};

ZHD constexpr int cfg_threads(TileCfg c) { return 1 << (c.logv - c.rb); }
ZHD constexpr int cfg_logical_bits(TileCfg c) { return c.logv - c.lu; }
ZHD constexpr int cfg_thread_logical_bits(TileCfg c) { return c.logv - c.lu - c.rb; }
// tiles bigger than the LDS image live in registers and are exchanged in phases, one group of
// channel vectors per phase (channels never mix, so each group is self-contained)
ZHD constexpr int cfg_log2_phases(TileCfg c) { return c.logv > c.llds ? c.logv - c.llds : 0; }

struct RoundGeo {
    int regbit[5];  // logical bit carried by bit j of the register index
    int nbf;        // the first nbf of them are butterflied in this round
    int thrbit[12]; // logical bit carried by bit q of the thread's logical id (tid >> LU)
    int nthr;
};

// A pass of 2*RB+1 bits gets no third round for its last bit: that butterfly is folded into the read side of the LDS
// exchange before round 1 (every thread reads both partners along the bit; the stages of a WHT commute). It saves a
// whole exchange for one more LDS read per element. -1: the pass has no such bit.
ZHD constexpr int fused_bit(int K, TileCfg c) { return K == 2 * c.rb + 1 ? K - 1 : -1; }
ZHD constexpr int num_rounds(int K, TileCfg c) { return fused_bit(K, c) >= 0 ? 2 : (K + c.rb - 1) / c.rb; }
ZHD constexpr int log2_tiles(int K, TileCfg c) { return c.logv - c.lu - K; }

// first_pass: the pass works on bit offset 0, its tiles are contiguous runs of rows, and the
// lanes should walk rows first. Otherwise adjacent TILES are adjacent in memory (they differ
// in a row bit below the pass's offset), so lanes walk tiles first.
ZHD constexpr RoundGeo make_round(int K, TileCfg c, bool first_pass, int r)
{
    const int NB = cfg_logical_bits(c);
    RoundGeo g{};
    bool used[16] = {false, false, false, false, false, false, false, false,
                     false, false, false, false, false, false, false, false};
    int n = 0;
    const int lo = c.rb * r;
    const int hi = (c.rb * r + c.rb < K) ? c.rb * r + c.rb : K;
    for (int q = lo; q < hi; ++q) { g.regbit[n++] = q; used[q] = true; }
    g.nbf = n;
    // fill the register index with bits that are not butterflied now: highest rows first
    // (low rows stay on the lanes, where they keep global accesses contiguous), then tiles
    for (int q = K - 1; q >= 0 && n < c.rb; --q)
        if (!used[q]) { g.regbit[n++] = q; used[q] = true; }
    for (int q = NB - 1; q >= K && n < c.rb; --q)
        if (!used[q]) { g.regbit[n++] = q; used[q] = true; }
    int m = 0;
    if (first_pass) {
        for (int q = 0; q < NB; ++q) if (!used[q]) g.thrbit[m++] = q;
    } else {
        for (int q = K; q < NB; ++q) if (!used[q]) g.thrbit[m++] = q;
        for (int q = 0; q < K; ++q) if (!used[q]) g.thrbit[m++] = q;
    }
    g.nthr = m;
    return g;
}

// logical index (row bits low, tile bits above) of the element a thread holds in register `reg`
ZHD constexpr uint32_t logical_of_thread(const RoundGeo &g, uint32_t tlog)
{
    uint32_t l = 0;
    for (int q = 0; q < g.nthr; ++q) l |= ((tlog >> q) & 1u) << g.thrbit[q];
    return l;
}
ZHD constexpr uint32_t logical_of_reg(const RoundGeo &g, uint32_t reg, int rb)
{
    uint32_t l = 0;
    for (int j = 0; j < rb; ++j) l |= ((reg >> j) & 1u) << g.regbit[j];
    return l;
}

// LDS slot (in 16-byte units) of vector `up` (index within the phase's group of 2^lup vectors)
// of logical element l. Row segments of 64 bytes get two padding terms that keep the strided
// round-to-round exchanges off a single bank group; 128-byte segments need none
// (tools/lds_conflicts.cpp).
ZHD constexpr uint32_t lds_slot(uint32_t l, uint32_t up, int lup)
{
    return (l << lup) + up + (lup <= 2 ? (((l >> 4) + (l >> 8)) << 2) : 0u);
}
ZHD constexpr uint32_t lds_slots_total(TileCfg c)
{
    const int lup = c.lu - cfg_log2_phases(c);
    const uint32_t last = (1u << cfg_logical_bits(c)) - 1u;
    return lds_slot(last, (1u << lup) - 1u, lup) + 1u;
}

// Global row d (index into the 2^N-point transform) of row i of tile `tile` for a pass that
// works on bits [b, b+K): the tile number supplies all the other bits.
ZHD constexpr uint64_t row_of(uint64_t tile, uint32_t i, int b, int K)
{
    return ((tile >> b) << (b + K)) | ((uint64_t)i << b) | (tile & ((1ull << b) - 1ull));
}
// ... and of logical element l of macro tile `macro` (tile = macro*2^LT + (l >> K))
ZHD constexpr uint64_t row_of_logical(uint64_t macro, uint32_t l, int b, int K, int LT)
{
    return row_of((macro << LT) | (l >> K), l & ((1u << K) - 1u), b, K);
}

// Row d of the transform lives at row work_row(d) of a workspace slab: one row of padding after every
// 2^10 and every 2^20 rows, so that the rows of a strided pass (stride 2^b rows, b a multiple of ~10)
// do not all fall on the same HBM channel.
ZHD constexpr uint64_t work_row(uint64_t d) { return d + (d >> 10) + (d >> 20); }
ZHD constexpr uint64_t work_rows_total(int N) { return work_row((1ull << N) - 1ull) + 1ull; }

// How N transform bits are split into passes of at most max_bits bits (low bits first).
ZHD constexpr int pass_split(int N, int bits[4], int max_bits = kMaxPassBits)
{
    const int np = (N + max_bits - 1) / max_bits;
    const int base = N / np, rem = N % np;
    for (int i = 0; i < np; ++i) bits[i] = base + (i < rem ? 1 : 0);
    for (int i = np; i < 4; ++i) bits[i] = 0;
    return np;
}

} // namespace zmls
