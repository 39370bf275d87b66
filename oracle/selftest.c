/* ORACLE — test infrastructure. Sanitizer run of the C restatement (tests/test_sanitizers_cpu.py builds this file with
 * mls_oracle.c under -fsanitize=address,undefined): both LFSR forms agree, the packed form matches the byte form, the
 * known-answer example of the reference (mls.py:179-205) comes out, and the buffers are exactly as long as declared. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

int mls_oracle_bits(int N, const int32_t *taps, int ntaps, uint8_t *out, int64_t nbits);
int mls_oracle_bits_packed(int N, const int32_t *taps, int ntaps, uint8_t *out, int64_t nbits);
int mls_oracle_bits_packed_word(int N, const int32_t *taps, int ntaps, uint8_t *out, int64_t nbits);
int mls_oracle_sequence(int N, const int32_t *taps, int ntaps, double *out, int64_t repeats);
int mls_oracle_average(const double *x, int64_t n_rows, int64_t stride, int64_t L, double *avg);
int mls_oracle_xcorr_direct(const double *avg, const double *ref, int64_t L, double *h);

#define CHECK(c) do { if (!(c)) { printf("FAILED line %d: %s\n", __LINE__, #c); return 1; } } while (0)

int main(void)
{
    const struct { int N; int32_t taps[4]; int nt; } sets[] = {{3, {3, 2}, 2}, {5, {5, 4, 3, 2}, 4}, {10, {10, 7}, 2}, {16, {16, 15, 13, 4}, 4},
                                                                {31, {31, 28}, 2}, {32, {32, 22, 2, 1}, 4}};
    for (size_t s = 0; s < sizeof sets / sizeof sets[0]; s++) {
        const int N = sets[s].N;
        const int64_t nbits = N <= 16 ? (((int64_t)1 << N) - 1) + 77 : 100003;
        uint8_t *bits = malloc((size_t)nbits), *pk = malloc((size_t)((nbits + 7) / 8)), *pw = malloc((size_t)((nbits + 7) / 8));
        CHECK(bits && pk && pw);
        CHECK(mls_oracle_bits(N, sets[s].taps, sets[s].nt, bits, nbits) == 0);
        CHECK(mls_oracle_bits_packed(N, sets[s].taps, sets[s].nt, pk, nbits) == 0);
        CHECK(mls_oracle_bits_packed_word(N, sets[s].taps, sets[s].nt, pw, nbits) == 0);
        CHECK(memcmp(pk, pw, (size_t)((nbits + 7) / 8)) == 0);
        for (int64_t i = 0; i < nbits; i++) CHECK(((pk[i >> 3] >> (i & 7)) & 1) == bits[i]);
        free(bits); free(pk); free(pw);
    }
    /* mls.py:179-205 */
    const int32_t t3[2] = {3, 2};
    double seq[14], avg[7], h[7];
    CHECK(mls_oracle_sequence(3, t3, 2, seq, 2) == 0);
    const double want[7] = {1, -1, -1, -1, 1, 1, -1};
    for (int i = 0; i < 14; i++) CHECK(seq[i] == want[i % 7]);
    CHECK(mls_oracle_average(seq, 14, 1, 7, avg) == 0);
    CHECK(mls_oracle_xcorr_direct(avg, seq, 7, h) == 0);
    CHECK(fabs(h[0] - 1.0) < 1e-15);
    for (int i = 1; i < 7; i++) CHECK(fabs(h[i] + 1.0 / 7.0) < 1e-15);
    CHECK(mls_oracle_average(seq, 7, 1, 7, avg) != 0 && mls_oracle_average(seq, 13, 1, 7, avg) != 0);
    const int32_t bad[1] = {9};
    CHECK(mls_oracle_bits(4, bad, 1, (uint8_t *)h, 4) != 0);
    printf("oracle selftest ok\n");
    return 0;
}
