/*
 * fit_bayes.c -- the reference's only native program, C/fit-bayes.c (RWMH for Bayesian logistic
 * regression, GSL), as a plain-C client of the C ABI in include/logreg_hip.h: same input file
 * (../pima.data or argv[1]), same stdout format (header "beta0 ... beta7", one line per kept
 * sample), same tuning (start beta = (-10,0,...,0), ll = -1e80, proposal sd 0.2 for the intercept
 * and 0.02 for the rest, C/fit-bayes.c:98-118,153-166; ITERS = 10000, THIN = 1000, :21-22).
 * The MCMC loop runs on the MI355X through lr_run_rwmh; no GSL.  Also shows the boundary is
 * usable from C, not only through ctypes.
 *
 *   gcc -O2 -I include examples/fit_bayes.c -L logreg_amd/lib -llogreg_hip -Wl,-rpath,$PWD/logreg_amd/lib -o fit_bayes
 *   ./fit_bayes [pima.data] [iters] [thin] [chains] > fit-bayes.tsv
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "logreg_hip.h"

#define N 200
#define P 8

int main(int argc, char** argv) {
    const char* path = argc > 1 ? argv[1] : "../pima.data";
    const long iters = argc > 2 ? atol(argv[2]) : 10000;
    const long thin = argc > 3 ? atol(argv[3]) : 1000;
    const long chains = argc > 4 ? atol(argv[4]) : 1;
    static double x[N * P], y[N];
    char tmps[32];
    fprintf(stderr, "RW MH for Bayesian logistic regression in C (MI355X through liblogreg_hip)\n");
    FILE* s = fopen(path, "r");
    if (s == NULL) {
        perror("error opening data file, pima.data");
        return 1;
    }
    char line[512];
    for (int i = 0; i < N;) { /* C/fit-bayes.c:54-67; '#' lines (provenance header of the shipped copy) skipped */
        if (!fgets(line, sizeof line, s)) { fprintf(stderr, "data file ends at row %d\n", i); return 1; }
        if (line[0] == '#' || line[0] == '\n') continue;
        double* r = &x[i * P];
        r[0] = 1.0;
        if (sscanf(line, "%lf %lf %lf %lf %lf %lf %lf %31s", r + 1, r + 2, r + 3, r + 4, r + 5, r + 6, r + 7, tmps) != 8) {
            fprintf(stderr, "bad data row %d\n", i);
            return 1;
        }
        y[i] = strcmp(tmps, "Yes") == 0 ? 1.0 : 0.0;
        ++i;
    }
    fclose(s);
    fprintf(stderr, "Data read and file closed\n");

    const double prior_sd[P] = {10, 1, 1, 1, 1, 1, 1, 1};                     /* :137-145 */
    const double prop_sd[P] = {0.2, 0.02, 0.02, 0.02, 0.02, 0.02, 0.02, 0.02}; /* :156-158 */
    lr_model* m = NULL;
    if (lr_model_create(x, y, N, P, prior_sd, LR_F32, 0, &m) != LR_OK) {
        fprintf(stderr, "lr_model_create: %s\n", lr_last_error());
        return 2;
    }
    float* state = (float*)calloc((size_t)chains * P, sizeof(float));
    double* ll = (double*)malloc((size_t)chains * sizeof(double));
    float* out = (float*)malloc((size_t)iters * chains * P * sizeof(float));
    unsigned* acc = (unsigned*)calloc((size_t)chains, sizeof(unsigned));
    for (long c = 0; c < chains; c++) {
        state[c * P] = -10.0f; /* :102 */
        ll[c] = -1e80;         /* :103 */
    }
    lr_run_opts o;
    memset(&o, 0, sizeof o);
    o.n_chains = chains;
    o.thin = thin;
    o.iters = iters;
    o.seed = 20240101;
    o.mode = LR_MODE_AUTO;
    if (lr_run_rwmh(m, state, ll, prop_sd, &o, out, acc) != LR_OK) {
        fprintf(stderr, "lr_run_rwmh: %s\n", lr_last_error());
        return 3;
    }
    for (int i = 0; i < P; i++) printf("beta%d ", i); /* :104-107 */
    printf("\n");
    for (long i = 0; i < iters; i++) /* chain 0, like the reference; further chains follow chain-major */
        for (long c = 0; c < chains; c++) {
            for (int j = 0; j < P; j++) printf("%f ", out[(i * chains + c) * P + j]);
            printf("\n");
        }
    fprintf(stderr, "accepted %u of %ld proposals (chain 0)\n", acc[0], iters * thin);
    lr_model_destroy(m);
    free(state); free(ll); free(out); free(acc);
    return 0;
}
