// Host-only self test of libzmls's host logic, built with AddressSanitizer + UndefinedBehaviorSanitizer on the host
// side (tests/test_sanitizers_cpu.py). No GPU call is made: everything here is the C ABI's host-only part
// (include/zmls_testing.h: zmls_host_*).
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <vector>
#include "../../include/zmls.h"
#include "../../include/zmls_testing.h"

#define CHECK(c) do { if (!(c)) { printf("FAILED %s:%d: %s (%s)\n", __FILE__, __LINE__, #c, zmls_last_error()); return 1; } } while (0)

// bit-serial Fibonacci register, straight from the definition
static uint32_t window_after(int N, const int32_t *taps, int ntaps, uint64_t steps)
{
    std::vector<uint8_t> a(steps + (uint64_t)N + 64, 0);
    std::vector<uint8_t> hist(N, 0);           // a[-N..-1]
    hist[N - 1] = 1;
    auto at = [&](int64_t i) -> int { return i < 0 ? hist[(size_t)(N + i)] : a[(size_t)i]; };
    for (uint64_t n = 0; n < steps + (uint64_t)N; ++n) {
        int b = 0;
        for (int t = 0; t < ntaps; ++t) b ^= at((int64_t)n - taps[t]);
        a[n] = (uint8_t)b;
    }
    uint32_t w = 0;
    for (int i = 0; i < N; ++i) w |= (uint32_t)a[steps + i] << i;
    return w;
}

int main()
{
    // pass planner
    for (int N = 2; N <= 30; ++N) {
        int bits[4] = {-1, -1, -1, -1};
        const int n = zmls_host_pass_split(N, bits);
        CHECK(n >= 1 && n <= 3);
        int sum = 0;
        for (int i = 0; i < n; ++i) { CHECK(bits[i] >= 1 && bits[i] <= 10); sum += bits[i]; }
        CHECK(sum == N);
    }
    int dummy[4];
    CHECK(zmls_host_pass_split(31, dummy) == ZMLS_E_ARG && zmls_host_pass_split(1, dummy) == ZMLS_E_ARG);
    // tile geometry of every configuration and pass width
    for (int cfg = 0; cfg <= 2; ++cfg)
        for (int K = 1; K <= 10; ++K)
            for (int first = 0; first <= 1; ++first) CHECK(zmls_host_geometry_selfcheck(K, cfg, first) == 0);
    for (int K = 10; K <= 11; ++K)
        for (int first = 0; first <= 1; ++first) CHECK(zmls_host_geometry_selfcheck(K, 3, first) == 0);
    CHECK(zmls_host_geometry_selfcheck(12, 3, 0) != 0 && zmls_host_geometry_selfcheck(3, 9, 0) == ZMLS_E_ARG);
    // the transform passes of the magnitude response, thread by thread on the CPU, against a direct DFT
    for (int m = 1; m <= 13; ++m) CHECK(zmls_host_fft_selfcheck(0, (int64_t)1 << m) == 0);
    CHECK(zmls_host_fft_selfcheck(1, 1023) == 0 && zmls_host_fft_selfcheck(1, 5000) == 0 && zmls_host_fft_selfcheck(0, 1000) != 0);
    // GF(2) jump-ahead against the bit-serial register, orders 2..32 (32 exercises every shift at its limit)
    const struct { int N; int32_t taps[4]; int nt; } sets[] = {{2, {2, 1}, 2}, {3, {3, 2}, 2}, {16, {16, 15, 13, 4}, 4}, {20, {20, 17}, 2},
                                                                {24, {24, 23, 22, 17}, 4}, {31, {31, 28}, 2}, {32, {32, 22, 2, 1}, 4}};
    for (const auto &s : sets)
        for (uint64_t steps : {0ull, 1ull, 31ull, 32ull, 33ull, 1023ull, 4097ull, 70001ull}) {
            uint32_t got = 0;
            CHECK(zmls_host_lfsr_state(s.N, s.taps, s.nt, steps, &got) == 0);
            CHECK(got == window_after(s.N, s.taps, s.nt, steps));
        }
    uint32_t st = 0;
    const int32_t bad[2] = {9, 3}, zero[1] = {0}, alias[2] = {5, 3};
    CHECK(zmls_host_lfsr_state(4, bad, 2, 0, &st) == ZMLS_E_ARG);          // tap beyond 2N: the reference's IndexError
    CHECK(zmls_host_lfsr_state(4, zero, 1, 0, &st) == ZMLS_E_ARG);
    CHECK(zmls_host_lfsr_state(33, bad, 2, 0, &st) == ZMLS_E_ARG);
    // taps N+1..2N read the cell of tap t-N (Python's negative deque index, mls.py:107); taps without N, non-primitive sets
    // and single taps run too (the reference checks nothing)
    const struct { int N; int32_t taps[4]; int nt; int32_t same[4]; } odd[] = {{4, {5, 3}, 2, {1, 3}}, {6, {12, 7}, 2, {6, 1}}, {8, {8, 4}, 2, {8, 4}},
                                                                              {9, {3, 3}, 1, {3, 3}}, {7, {14, 14}, 1, {7, 7}}};
    for (const auto &s : odd)
        for (uint64_t steps : {0ull, 1ull, 5ull, 64ull, 1000ull, 70001ull}) {
            uint32_t got = 0;
            CHECK(zmls_host_lfsr_state(s.N, s.taps, s.nt, steps, &got) == 0);
            CHECK(got == window_after(s.N, s.same, s.nt, steps));
        }
    (void)alias;
    // capture placement search on simulated machines: every buffer it takes is given back but the one it returns
    const float ms[6] = {14.5f, 13.5f, 12.6f, 13.4f, 12.61f, 15.0f};
    for (int max_live = 0; max_live <= 5; ++max_live)
        for (int tries = 1; tries <= 6; ++tries)
            for (int n = 1; n <= 6; ++n) {
                int64_t out[8];
                const int rc = zmls_host_placement_sim(ms, n, max_live, tries, 12.5f, out);
                CHECK((rc == 0) == (max_live > 0));
                CHECK(out[6] == (max_live > 0 ? 1 : 0) && out[7] <= max_live && out[7] <= 5);   // the kept one, the best, two rejected, one being rated
                if (rc == 0) CHECK(out[0] >= 0 && out[3] > 0 && out[3] <= out[2] + 1);
            }
    // null handles and bad arguments are refused before anything touches a device
    CHECK(zmls_plan_info(nullptr, nullptr) == ZMLS_E_ARG && zmls_device_count(nullptr) == ZMLS_E_ARG);
    CHECK(zmls_plan_set_input_scale_log2(nullptr, 3) == ZMLS_E_ARG && zmls_plan_force_tile(nullptr, 0) == ZMLS_E_ARG);
    CHECK(strlen(zmls_last_error()) > 0);
    printf("host selftest ok\n");
    return 0;
}
