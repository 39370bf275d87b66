// Tile geometry of one FWHT pass — shared by the device kernels, the table builders and the
// host-side self check (zmls_host_geometry_selfcheck), so there is exactly one definition of
// "which thread holds which element in which round".
//
// A workgroup of 256 threads owns a MACRO TILE of 4096 sixteen-byte vectors (64 KiB):
//     2^LT tiles  x  2^K rows  x  2^LU vectors per row        (LT = 12 - LU - K)
// A "vector" is 16 bytes of adjacent channels (2 doubles or 4 floats); a row of a tile is
// 2^LU vectors = one contiguous row segment of a channel slab. The pass butterflies over the K
// row bits; tiles of a macro tile are independent (they only share the workgroup).
//
// Every thread keeps 16 vectors in registers. A pass is ceil(K/4) ROUNDS; round r butterflies
// row bits [4r, min(4r+4,K)) entirely in registers, and between rounds the macro tile is
// exchanged through LDS. For each round the 12-LU "logical bits" (row bits 0..K-1, then tile
// bits K..K+LT-1) are split into 4 register-index bits and 8-LU thread bits.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define ZHD __host__ __device__ inline
#else
#define ZHD inline
#endif

namespace zmls {

constexpr int kLogVecs = 12;     // log2(vectors per macro tile)
constexpr int kThreads = 256;
constexpr int kMaxPassBits = 10; // K <= 10: at most three rounds, two LDS exchanges
constexpr int kVecBytes = 16;

struct RoundGeo {
    int regbit[4];  // logical bit carried by bit j of the register index
    int nbf;        // the first nbf of them are butterflied in this round
    int thrbit[10]; // logical bit carried by bit q of the thread's logical id (tid >> LU)
    int nthr;
};

ZHD constexpr int num_rounds(int K) { return (K + 3) / 4; }
ZHD constexpr int log2_tiles(int K, int LU) { return kLogVecs - LU - K; }

// first_pass: the pass works on bit offset 0, its tiles are contiguous runs of rows, and the
// lanes should walk rows first. Otherwise adjacent TILES are adjacent in memory (they differ
// in a row bit below the pass's offset), so lanes walk tiles first.
ZHD constexpr RoundGeo make_round(int K, int LU, bool first_pass, int r)
{
    const int NB = kLogVecs - LU;
    RoundGeo g{};
    bool used[16] = {false, false, false, false, false, false, false, false,
                     false, false, false, false, false, false, false, false};
    int n = 0;
    const int lo = 4 * r;
    const int hi = (4 * r + 4 < K) ? 4 * r + 4 : K;
    for (int q = lo; q < hi; ++q) { g.regbit[n++] = q; used[q] = true; }
    g.nbf = n;
    for (int q = K - 1; q >= 0 && n < 4; --q)
        if (!used[q]) { g.regbit[n++] = q; used[q] = true; }
    for (int q = NB - 1; q >= K && n < 4; --q)
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
ZHD constexpr uint32_t logical_of_reg(const RoundGeo &g, uint32_t reg)
{
    uint32_t l = 0;
    for (int j = 0; j < 4; ++j) l |= ((reg >> j) & 1u) << g.regbit[j];
    return l;
}

// LDS slot (in 16-byte units) of vector u of logical element l. The two padding terms keep
// the strided round-to-round exchanges off a single bank group (tools/lds_conflicts.cpp).
ZHD constexpr uint32_t lds_slot(uint32_t l, uint32_t u, int LU)
{
    return (l << LU) + u + (((l >> 4) + (l >> 8)) << 2);
}
ZHD constexpr uint32_t lds_slots_total(int LU)
{
    const uint32_t last = (1u << (kLogVecs - LU)) - 1u;
    return lds_slot(last, (1u << LU) - 1u, LU) + 1u;
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

// How N transform bits are split into passes of at most kMaxPassBits bits (low bits first).
ZHD constexpr int pass_split(int N, int bits[4])
{
    const int np = (N + kMaxPassBits - 1) / kMaxPassBits;
    const int base = N / np, rem = N % np;
    for (int i = 0; i < np; ++i) bits[i] = base + (i < rem ? 1 : 0);
    for (int i = np; i < 4; ++i) bits[i] = 0;
    return np;
}

} // namespace zmls
