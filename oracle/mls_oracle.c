/* ORACLE — test infrastructure, not product code.
 *
 * Plain-C restatement of the integer part of the reference's MLS path
 * (zignal v0.6.0, zignal/measure/mls.py) plus the definition of the circular
 * cross-correlation it computes. Used by tests/ and bench.py's cpu_baseline leg
 * only; nothing under zignal_amd/ links or loads it.
 *
 * Why C as well as oracle/mls_oracle.py: the reference's per-bit Python loop
 * needs ~30 s for one N=24 period. This file runs the SAME recurrence
 * (same register, same tap indexing, same initial state) a few thousand times
 * faster so the full-size GPU parity tests can check every bit. It is pinned by
 * tests/test_oracle_golden.py against the SHA-256 digests captured from the real
 * reference (tests/golden/seq_kat.json) and against mls_oracle.py.
 *
 * Build: make -C oracle   (gcc -O2 -shared -fPIC) -> oracle/_build/libmls_oracle.so
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* Shift register of mls.py:101-108: cells[0] is the oldest bit, cells[N-1] the
 * newest; initial contents 0,...,0,1. One step: bit = XOR cells[N - tap] over
 * the taps; drop cells[0]; append bit. */
typedef struct {
    int N, ntaps;
    int taps[64];
    uint8_t cells[64];
} lfsr_t;

static int lfsr_init(lfsr_t *r, int N, const int32_t *taps, int ntaps)
{
    if (N < 1 || N > 64 || ntaps < 1 || ntaps > 64) return -1;
    r->N = N;
    r->ntaps = ntaps;
    for (int i = 0; i < ntaps; i++) {
        /* cells[N - tap]: a tap beyond N makes the index negative, which the reference's deque counts from the newest
         * cell (mls.py:107): tap N+k reads the cell tap k reads. Tap 0 and taps beyond 2N are its IndexError. */
        if (taps[i] < 1 || taps[i] > 2 * N) return -1;
        r->taps[i] = taps[i] > N ? taps[i] - N : taps[i];
    }
    memset(r->cells, 0, sizeof r->cells);
    r->cells[N - 1] = 1;
    return 0;
}

static inline int lfsr_step(lfsr_t *r)
{
    int bit = 0;
    for (int i = 0; i < r->ntaps; i++) bit ^= r->cells[r->N - r->taps[i]];
    memmove(r->cells, r->cells + 1, (size_t)(r->N - 1));
    r->cells[r->N - 1] = (uint8_t)bit;
    return bit;
}

/* nbits consecutive bits from the start of the generator (mls.py:91-128),
 * one byte per bit. Continues across the period boundary like the reference. */
int mls_oracle_bits(int N, const int32_t *taps, int ntaps, uint8_t *out, int64_t nbits)
{
    lfsr_t r;
    if (lfsr_init(&r, N, taps, ntaps)) return -1;
    for (int64_t i = 0; i < nbits; i++) out[i] = (uint8_t)lfsr_step(&r);
    return 0;
}

/* Same bits packed LSB-first (numpy.packbits(bitorder="little")); out has
 * (nbits+7)/8 bytes, unused high bits of the last byte are 0. */
int mls_oracle_bits_packed(int N, const int32_t *taps, int ntaps, uint8_t *out, int64_t nbits)
{
    lfsr_t r;
    if (lfsr_init(&r, N, taps, ntaps)) return -1;
    memset(out, 0, (size_t)((nbits + 7) / 8));
    for (int64_t i = 0; i < nbits; i++)
        if (lfsr_step(&r)) out[i >> 3] |= (uint8_t)(1u << (i & 7));
    return 0;
}

/* The same register held in one 64-bit word (bit k = cells[k]): the XOR over cells[N - tap] is the parity of
 * (register & mask), dropping cells[0] is a shift, the new bit goes in at position N-1 (mls.py:101-108 again, a
 * few nanoseconds per bit instead of a memmove). For the two orders whose period is too long for the byte-cell
 * loop above in a test (N = 31, 32: 2 and 4 Gbit); tests/test_oracle_golden.py holds it to the same reference
 * digests and to mls_oracle_bits_packed. */
int mls_oracle_bits_packed_word(int N, const int32_t *taps, int ntaps, uint8_t *out, int64_t nbits)
{
    if (N < 1 || N > 64 || ntaps < 1 || ntaps > 64) return -1;
    uint64_t mask = 0, reg = 1ull << (N - 1);
    for (int i = 0; i < ntaps; i++) {
        if (taps[i] < 1 || taps[i] > 2 * N) return -1;
        mask ^= 1ull << (N - (taps[i] > N ? taps[i] - N : taps[i]));   /* a tap listed twice cancels, as it does in the XOR loop */
    }
    memset(out, 0, (size_t)((nbits + 7) / 8));
    for (int64_t i = 0; i < nbits; i++) {
        const uint64_t bit = (uint64_t)__builtin_parityll(reg & mask);
        reg = (reg >> 1) | (bit << (N - 1));
        if (bit) out[i >> 3] |= (uint8_t)(1u << (i & 7));
    }
    return 0;
}

/* (repeats*L) samples: bit 0 -> +1.0, bit 1 -> -1.0 (mls.py:150), one fresh
 * period tiled (mls.py:162-164). */
int mls_oracle_sequence(int N, const int32_t *taps, int ntaps, double *out, int64_t repeats)
{
    lfsr_t r;
    if (lfsr_init(&r, N, taps, ntaps)) return -1;
    int64_t L = ((int64_t)1 << N) - 1;
    for (int64_t i = 0; i < L; i++) out[i] = lfsr_step(&r) ? -1.0 : 1.0;
    for (int64_t k = 1; k < repeats; k++) memcpy(out + k * L, out, (size_t)L * sizeof(double));
    return 0;
}

/* trim_and_avgerage (mls.py:284-306) for one channel stored with a stride:
 * drop period 0, add periods 1..R-1 in order, divide by their count. */
int mls_oracle_average(const double *x, int64_t n_rows, int64_t stride, int64_t L, double *avg)
{
    if (n_rows <= L || n_rows % L) return -1;
    int64_t periods = n_rows / L - 1;
    for (int64_t j = 0; j < L; j++) {
        double s = x[(L + j) * stride];
        for (int64_t p = 1; p < periods; p++) s += x[((p + 1) * L + j) * stride];
        avg[j] = s / (double)periods;
    }
    return 0;
}

/* What mls.py:219-251 computes, written as its definition:
 * h[m] = (1/L) * sum_i ref[i] * avg[(i+m) mod L].  O(L^2): small orders only. */
int mls_oracle_xcorr_direct(const double *avg, const double *ref, int64_t L, double *h)
{
    for (int64_t m = 0; m < L; m++) {
        double s = 0.0;
        int64_t k = m;
        for (int64_t i = 0; i < L; i++) {
            s += ref[i] * avg[k];
            if (++k == L) k = 0;
        }
        h[m] = s / (double)L;
    }
    return 0;
}
