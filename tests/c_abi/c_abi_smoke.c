/* Plain-C client of libgpry_hip.so: no Python, no C++, no torch -- only include/gpry_hip.h.
 * Reads a tiny problem from stdin (text), drives the hot path through the C ABI and prints the
 * results; tests/test_c_abi_gpu.py compares them with the oracle.
 *
 * stdin:  N d M kernel_id
 *         theta[0..d]           (log C, log l_1..l_d)
 *         N rows: x_1..x_d y alpha
 *         M rows: x_1..x_d
 * stdout: "lml <v>", "grad <d+1 values>", M lines "pred <mean> <std> <acq>", "top <idx> <acq>" x K
 */
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include "gpry_hip.h"

#define CHECK(call)                                                                  \
    do {                                                                             \
        int rc_ = (call);                                                            \
        if (rc_ != 0) {                                                              \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, gpry_last_error(ctx)); \
            return 2;                                                                \
        }                                                                            \
    } while (0)

int main(void) {
    long N, M; int d, kid;
    gpry_ctx* ctx = NULL;
    if (scanf("%ld %d %ld %d", &N, &d, &M, &kid) != 4) return 1;
    double* theta = malloc(sizeof(double) * (d + 1));
    double* X = malloc(sizeof(double) * N * d);
    double* y = malloc(sizeof(double) * N);
    double* alpha = malloc(sizeof(double) * N);
    double* Xc = malloc(sizeof(double) * M * d);
    for (int k = 0; k <= d; k++) if (scanf("%lf", &theta[k]) != 1) return 1;
    for (long i = 0; i < N; i++) {
        for (int k = 0; k < d; k++) if (scanf("%lf", &X[i * d + k]) != 1) return 1;
        if (scanf("%lf %lf", &y[i], &alpha[i]) != 2) return 1;
    }
    for (long i = 0; i < M * d; i++) if (scanf("%lf", &Xc[i]) != 1) return 1;

    CHECK(gpry_ctx_create(0, &ctx));
    CHECK(gpry_set_train(ctx, X, y, alpha, N, d));
    CHECK(gpry_set_theta(ctx, kid, theta));
    int info = -1;
    CHECK(gpry_factorize(ctx, &info));
    if (info != 0) { fprintf(stderr, "not positive definite: info=%d\n", info); return 3; }

    double lml = 0.0;
    double* grad = malloc(sizeof(double) * (d + 1));
    CHECK(gpry_lml(ctx, theta, 1, &lml, grad, &info));
    printf("lml %.17g\ngrad", lml);
    for (int k = 0; k <= d; k++) printf(" %.17g", grad[k]);
    printf("\n");

    double* mean = malloc(sizeof(double) * M);
    double* sd = malloc(sizeof(double) * M);
    double* acq = malloc(sizeof(double) * M);
    int64_t n_nan = 0;
    /* zeta = 0.25, baseline = 0, sigma_n = 1e-3: LogExp.f on the device */
    CHECK(gpry_sweep_logexp(ctx, Xc, M, NULL, 0.25, 0.0, 1e-3, mean, sd, acq, &n_nan));
    for (long m = 0; m < M; m++) printf("pred %.17g %.17g %.17g\n", mean[m], sd[m], acq[m]);
    gpry_cand top[5];
    int64_t n_out = 0; double bound = 0.0;
    CHECK(gpry_sweep_topk(ctx, 5, NULL, 0, top, &n_out, &bound));
    for (int64_t t = 0; t < n_out; t++) printf("top %lld %.17g\n", (long long)top[t].idx, top[t].acq);
    printf("bound %.17g nan %lld\n", bound, (long long)n_nan);
    CHECK(gpry_ctx_destroy(ctx));
    return 0;
}
