/* The C ABI from plain C (C99, nothing but include/zmls.h and libzmls.so): the loop-back measurement of zignal's
 * `python -m zignal.measure.mls` (zignal/measure/mls.py:469-491) - generate an MLS, feed it back as the capture, recover the
 * impulse response - through the host entry points, so the program needs no HIP header.
 *
 *   gcc -std=c99 -Wall -Wextra -pedantic -Iinclude examples/c_api.c -Lzignal_amd/lib -lzmls -Wl,-rpath,$PWD/zignal_amd/lib -o c_api
 *
 * Exit status: 0 = measured and checked, 3 = no usable GPU (the library has no CPU path), 1 = anything else. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include "zmls.h"

static int fail(const char *what, int rc)
{
    fprintf(stderr, "%s failed (%d): %s\n", what, rc, zmls_last_error());
    return rc == ZMLS_E_NODEVICE ? 3 : 1;
}

int main(void)
{
    const int N = 14, repeats = 3, n_ch = 2;
    const int32_t taps[] = {14, 13, 12, 2};                       /* TAPS[14][0], zignal/measure/mlstaps.py */
    const int64_t L = ((int64_t)1 << N) - 1;
    zmls_plan *plan = NULL;
    int n_dev = 0, rc;

    printf("libzmls version %d\n", zmls_version());
    rc = zmls_device_count(&n_dev);
    if (rc || n_dev < 1) {
        printf("no usable GPU (%d devices): libzmls has no CPU path\n", n_dev);
        return 3;
    }
    rc = zmls_plan_create(&plan, N, taps, 4, ZMLS_F64, 0);
    if (rc) return fail("zmls_plan_create", rc);

    /* stimulus: `repeats` periods of +1 / -1 (get_full_sequence, mls.py:154-167) */
    double *seq = (double *)malloc(sizeof(double) * (size_t)(repeats * L));
    double *x = (double *)malloc(sizeof(double) * (size_t)(repeats * L * n_ch));
    double *h = (double *)malloc(sizeof(double) * (size_t)(L * n_ch));
    if (!seq || !x || !h) return 1;
    rc = zmls_sequence_f64(plan, seq, repeats);
    if (rc) return fail("zmls_sequence_f64", rc);

    /* capture: channel 0 hears the stimulus as it is, channel 1 half as loud and 100 samples late; frames x channels */
    for (int64_t j = 0; j < repeats * L; ++j) {
        x[j * n_ch + 0] = seq[j];
        x[j * n_ch + 1] = 0.5 * seq[(j + repeats * L - 100) % (repeats * L)];
    }
    rc = zmls_impulse_host(plan, x, ZMLS_F64, repeats * L, n_ch, n_ch, 1, h, n_ch, 1);
    if (rc) return fail("zmls_impulse_host", rc);

    /* a loop-back response is a unit impulse on a floor of -1/L (mls.py:205) */
    double worst = 0.0;
    for (int64_t m = 0; m < L; ++m) {
        const double want0 = (m == 0) ? 1.0 : -1.0 / (double)L;
        const double want1 = 0.5 * ((m == 100) ? 1.0 : -1.0 / (double)L);
        const double e0 = fabs(h[m * n_ch + 0] - want0), e1 = fabs(h[m * n_ch + 1] - want1);
        if (e0 > worst) worst = e0;
        if (e1 > worst) worst = e1;
    }
    printf("loop-back: h0[0] = %.12f, h1[100] = %.12f, worst deviation %.3e\n", h[0], h[100 * n_ch + 1], worst);

    /* errors come back as codes with a message, as the reference's assertions do (mls.py:322-323) */
    rc = zmls_impulse_host(plan, x, ZMLS_F64, L, n_ch, n_ch, 1, h, n_ch, 1);
    printf("one period only -> %d (%s)\n", rc, zmls_last_error());
    const int refused = rc == ZMLS_E_ARG;

    zmls_plan_destroy(plan);
    free(seq); free(x); free(h);
    if (worst > 1e-12 || !refused) return 1;
    printf("c_api ok\n");
    return 0;
}
