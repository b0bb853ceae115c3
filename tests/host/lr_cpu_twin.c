/*
 * lr_cpu_twin.c -- a CPU TEST DOUBLE of the C ABI in include/logreg_hip.h, built on the parity oracle (oracle/lr_oracle.c).
 *
 * TEST INFRASTRUCTURE ONLY.  The product (logreg_amd/) never builds, loads or falls back to this file: logreg_amd/_lib.py loads
 * liblogreg_hip.so and nothing else, and raises without it.  tests/twin.py builds this twin into a temporary directory and
 * INJECTS it into logreg_amd._lib for the duration of a test module, so that the host side of the product -- the Python face
 * (model closures, kernel objects, ChainSet chunking / checkpoint / streaming statistics, mcmc(), mcmc_sharded()) and the
 * plain-C client examples/fit_bayes.c -- runs in the GPU-less build container against the same ABI, same conventions, same
 * Philox stream: host-side defects show there, not in metered GPU minutes ("the same header is
 * implemented twice, which is what lets the API be tested in this GPU-less container").
 *
 * What it is NOT: a second implementation of the kernels.  All arithmetic is the oracle's float64 restatement of the reference
 * (a float32 model stores float32 states and rounds the state to float32 after every kept sample, so that chunked runs equal
 * monolithic ones as they do on the device); "device" memory is host memory; streams and events are tokens; the planner
 * answers one fixed variant.  The precision policy is accepted and ignored (every evaluation exact).
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include "../../oracle/lr_oracle.c"

#include "../../include/logreg_hip.h"

struct lr_model {
    orc_model om;
    double *X, *y, *sd;
    int64_t n;
    int32_t p, dtype, device;
};

static __thread char g_err[512];
static int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

LR_API const char *lr_last_error(void) { return g_err; }
LR_API const char *lr_build_id(void) { return "cpu-twin (tests/host/lr_cpu_twin.c)"; }
LR_API int lr_sizeof_run_opts(void) { return (int)sizeof(lr_run_opts); }
LR_API int lr_device_count(void) { return 1; }
LR_API int lr_device_cus(int device) { return device == 0 ? 256 : fail(LR_ERR_INVALID, "no device %d", device); }
LR_API int lr_device_info(int device, char *buf, int len) {
    if (!buf || len <= 0) return fail(LR_ERR_INVALID, "NULL / empty buffer");
    if (device != 0) return fail(LR_ERR_INVALID, "no device %d", device);
    snprintf(buf, (size_t)len, "pci=twin:%ld uuid=%032lx name=cpu-twin cus=256", (long)getpid(), (long)getpid());
    return LR_OK;
}

LR_API int lr_model_create(const double *X, const double *y, int64_t n, int32_t p, const double *prior_sd, int32_t dtype,
                           int32_t device, lr_model **out) {
    if (!X || !y || !prior_sd || !out) return fail(LR_ERR_INVALID, "NULL argument");
    if (n <= 0 || p <= 0 || p > 128) return fail(LR_ERR_INVALID, "n must be positive and 1 <= p <= 128 (got n=%lld p=%d)", (long long)n, p);
    if (dtype != LR_F32 && dtype != LR_F64) return fail(LR_ERR_INVALID, "dtype must be LR_F32 or LR_F64");
    if (device != 0) return fail(LR_ERR_INVALID, "no device %d", device);
    for (int j = 0; j < p; ++j)
        if (!(prior_sd[j] > 0) || !isfinite(prior_sd[j])) return fail(LR_ERR_INVALID, "prior_sd[%d] must be positive and finite", j);
    for (int64_t i = 0; i < n; ++i)
        if (y[i] != 0.0 && y[i] != 1.0) return fail(LR_ERR_INVALID, "y[%lld] must be 0 or 1", (long long)i);
    lr_model *m = (lr_model *)calloc(1, sizeof(lr_model));
    if (!m) return fail(LR_ERR_NOMEM, "model");
    m->X = (double *)malloc(sizeof(double) * (size_t)n * p);
    m->y = (double *)malloc(sizeof(double) * (size_t)n);
    m->sd = (double *)malloc(sizeof(double) * (size_t)p);
    if (!m->X || !m->y || !m->sd) return fail(LR_ERR_NOMEM, "model data");
    memcpy(m->X, X, sizeof(double) * (size_t)n * p);
    memcpy(m->y, y, sizeof(double) * (size_t)n);
    memcpy(m->sd, prior_sd, sizeof(double) * (size_t)p);
    if (dtype == LR_F32)  /* the device copy of a float32 model holds float32 rows */
        for (int64_t i = 0; i < n * p; ++i) m->X[i] = (double)(float)m->X[i];
    m->om.n = n;
    m->om.p = p;
    m->om.X = m->X;
    m->om.y = m->y;
    m->om.pscale = m->sd;
    m->n = n;
    m->p = p;
    m->dtype = dtype;
    m->device = device;
    *out = m;
    return LR_OK;
}

LR_API void lr_model_destroy(lr_model *m) {
    if (!m) return;
    free(m->X);
    free(m->y);
    free(m->sd);
    free(m);
}

LR_API int lr_plan_run_info(const lr_model *m, int32_t kind, const lr_run_opts *opts, lr_plan_info *out) {
    if (!out) return fail(LR_ERR_INVALID, "out is NULL");
    memset(out, 0, sizeof *out);
    return lr_plan_run(m, kind, opts, &out->mode, &out->group, &out->rows); /* the double has one variant, never a second part */
}
LR_API int lr_model_debug_opts(const lr_model *m, char *buf, int len) {
    if (!m || !buf || len <= 0) return fail(LR_ERR_INVALID, "NULL argument / empty buffer");
    buf[0] = 0; /* the double has no switches */
    return LR_OK;
}
LR_API int lr_model_interior_format(const lr_model *m, int32_t *format) {
    if (!m || !format) return fail(LR_ERR_INVALID, "NULL argument");
    *format = LR_INTERIOR_NONE; /* the double computes everything in float64 */
    return LR_OK;
}
LR_API int lr_model_info(const lr_model *m, int64_t *n, int32_t *p, int32_t *dtype, int32_t *device, int32_t *padded_p) {
    if (!m) return fail(LR_ERR_INVALID, "model is NULL");
    if (n) *n = m->n;
    if (p) *p = m->p;
    if (dtype) *dtype = m->dtype;
    if (device) *device = m->device;
    if (padded_p) *padded_p = m->p <= 4 ? 4 : m->p <= 8 ? 8 : m->p <= 16 ? 16 : m->p <= 32 ? 32 : m->p <= 64 ? 64 : 128;
    return LR_OK;
}

static double get(const void *a, int32_t dtype, int64_t i) { return dtype == LR_F32 ? (double)((const float *)a)[i] : ((const double *)a)[i]; }
static void put(void *a, int32_t dtype, int64_t i, double v) {
    if (dtype == LR_F32) ((float *)a)[i] = (float)v;
    else ((double *)a)[i] = v;
}

static int check_opts(const lr_model *m, const lr_run_opts *o, int run) {
    if (!m) return fail(LR_ERR_INVALID, "model is NULL");
    if (!o) return fail(LR_ERR_INVALID, "opts is NULL");
    if (o->n_chains <= 0) return fail(LR_ERR_INVALID, "n_chains must be positive (got %lld)", (long long)o->n_chains);
    if (o->plan_chains < 0) return fail(LR_ERR_INVALID, "plan_chains must be 0 (= n_chains) or positive (got %d)", o->plan_chains);
    if (o->group < 0) return fail(LR_ERR_INVALID, "group must be >= 0");
    if (o->mode < LR_MODE_AUTO || o->mode > LR_MODE_MIXED) return fail(LR_ERR_INVALID, "unknown mode %d", o->mode);
    if (run) {
        if (o->thin <= 0 || o->iters < 0) return fail(LR_ERR_INVALID, "thin must be > 0 and iters >= 0");
        if (o->chain_offset < 0 || o->iter_offset < 0) return fail(LR_ERR_INVALID, "offsets must be >= 0");
        if (o->stats) {
            if (o->stats_batch < 1) return fail(LR_ERR_INVALID, "stats_batch must be >= 1");
            if (o->stats_first < 0 || o->stats_first + o->iters > o->stats_slots * o->stats_batch)
                return fail(LR_ERR_INVALID, "statistics window too small: stats_first + iters = %lld > slots * batch = %lld",
                            (long long)(o->stats_first + o->iters), (long long)(o->stats_slots * o->stats_batch));
        }
    }
    return LR_OK;
}

LR_API int lr_eval(lr_model *m, const void *beta, void *ll, void *lprior, void *lpost, void *grad, const lr_run_opts *o) {
    int rc = check_opts(m, o, 0);
    if (rc) return rc;
    if (!beta) return fail(LR_ERR_INVALID, "beta is NULL");
    const int p = m->p;
    for (int64_t c = 0; c < o->n_chains; ++c) {
        double b[ORC_MAXP], g[ORC_MAXP];
        for (int j = 0; j < p; ++j) b[j] = get(beta, m->dtype, c * p + j);
        const double l = orc_ll(&m->om, b), pr = orc_lprior(&m->om, b);
        if (ll) put(ll, m->dtype, c, l);
        if (lprior) put(lprior, m->dtype, c, pr);
        if (lpost) put(lpost, m->dtype, c, l + pr);
        if (grad) {
            orc_glp(&m->om, b, g);
            for (int j = 0; j < p; ++j) put(grad, m->dtype, c * p + j, g[j]);
        }
    }
    return LR_OK;
}

/* one lr_run_* call: kept sample by kept sample (the state takes the model's storage type in between, as on the device) */
static int run_common(lr_model *m, const orc_kernel *k, const lr_run_opts *o, void *state, double *lp_state, void *out, uint32_t *accepts) {
    int rc = check_opts(m, o, 1);
    if (rc) return rc;
    if (!state) return fail(LR_ERR_INVALID, "state is NULL");
    const int p = m->p;
    const int64_t C = o->n_chains;
    double *st = (double *)malloc(sizeof(double) * (size_t)C * p);
    uint64_t *acc = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)C);
    if (!st || !acc) return fail(LR_ERR_NOMEM, "run scratch");
    for (int64_t e = 0; e < C * p; ++e) st[e] = get(state, m->dtype, e);
    for (int64_t i = 0; i < o->iters; ++i) {
        rc = orc_run(&m->om, k, st, lp_state, C, o->chain_offset, o->thin, 1, o->iter_offset + i * o->thin, o->seed, NULL, NULL, NULL, acc,
                     NULL, 0);
        if (rc) {
            free(st);
            free(acc);
            return fail(LR_ERR_UNSUPPORTED, "oracle run failed (%d)", rc);
        }
        for (int64_t e = 0; e < C * p; ++e) {
            if (m->dtype == LR_F32) st[e] = (double)(float)st[e];
            if (out) put(out, m->dtype, i * C * p + e, st[e]);
        }
        if (accepts)
            for (int64_t c = 0; c < C; ++c) accepts[c] += (uint32_t)acc[c];
        if (o->stats) { /* Welford per batch slot; the first sample of a slot initialises it */
            const int64_t idx = o->stats_first + i, b = idx / o->stats_batch, kk = idx - b * o->stats_batch;
            for (int64_t c = 0; c < C; ++c) {
                double *s = o->stats + ((b * C + c) * 2) * p;
                for (int j = 0; j < p; ++j) {
                    const double x = st[c * p + j];
                    if (kk == 0) {
                        s[j] = x;
                        s[p + j] = 0.0;
                    } else {
                        const double d = x - s[j];
                        s[j] += d / (double)(kk + 1);
                        s[p + j] += d * (x - s[j]);
                    }
                }
            }
        }
    }
    for (int64_t e = 0; e < C * p; ++e) put(state, m->dtype, e, st[e]);
    free(st);
    free(acc);
    return LR_OK;
}

static int positive_vec(const char *name, const double *v, int p) {
    if (!v) return fail(LR_ERR_INVALID, "%s is NULL", name);
    for (int j = 0; j < p; ++j)
        if (!(v[j] > 0) || !isfinite(v[j])) return fail(LR_ERR_INVALID, "%s[%d] must be positive and finite", name, j);
    return LR_OK;
}

LR_API int lr_run_rwmh(lr_model *m, void *state, double *lp_state, const double *prop_sd, const lr_run_opts *o, void *out, uint32_t *accepts) {
    if (!m) return fail(LR_ERR_INVALID, "model is NULL");
    int rc = positive_vec("prop_sd", prop_sd, m->p);
    if (rc) return rc;
    if (!lp_state) return fail(LR_ERR_INVALID, "lp_state is NULL");
    const orc_kernel k = {ORC_RWMH, 0.0, 0, prop_sd};
    return run_common(m, &k, o, state, lp_state, out, accepts);
}
LR_API int lr_run_mala(lr_model *m, void *state, double *lp_state, double dt, const double *pre, const lr_run_opts *o, void *out,
                       uint32_t *accepts) {
    if (!m) return fail(LR_ERR_INVALID, "model is NULL");
    int rc = positive_vec("pre", pre, m->p);
    if (rc) return rc;
    if (!(dt > 0) || !isfinite(dt)) return fail(LR_ERR_INVALID, "dt must be positive and finite");
    if (!lp_state) return fail(LR_ERR_INVALID, "lp_state is NULL");
    const orc_kernel k = {ORC_MALA, dt, 0, pre};
    return run_common(m, &k, o, state, lp_state, out, accepts);
}
LR_API int lr_run_ul(lr_model *m, void *state, double dt, const double *pre, const lr_run_opts *o, void *out, uint32_t *accepts) {
    if (!m) return fail(LR_ERR_INVALID, "model is NULL");
    int rc = positive_vec("pre", pre, m->p);
    if (rc) return rc;
    if (!(dt > 0) || !isfinite(dt)) return fail(LR_ERR_INVALID, "dt must be positive and finite");
    const orc_kernel k = {ORC_UL, dt, 0, pre};
    return run_common(m, &k, o, state, NULL, out, accepts);
}
LR_API int lr_run_hmc(lr_model *m, void *state, double eps, int32_t l, const double *dmm, const lr_run_opts *o, void *out, uint32_t *accepts) {
    if (!m) return fail(LR_ERR_INVALID, "model is NULL");
    int rc = positive_vec("dmm", dmm, m->p);
    if (rc) return rc;
    if (!(eps > 0) || !isfinite(eps) || l < 1) return fail(LR_ERR_INVALID, "eps must be positive and finite, l >= 1");
    const orc_kernel k = {ORC_HMC, eps, l, dmm};
    return run_common(m, &k, o, state, NULL, out, accepts);
}

LR_API int lr_hessian(lr_model *m, const double *beta, double *lpost, double *grad, double *hess, void *stream) {
    (void)stream;
    if (!m) return fail(LR_ERR_INVALID, "model is NULL");
    if (!beta) return fail(LR_ERR_INVALID, "beta is NULL");
    const int p = m->p;
    if (lpost) *lpost = orc_lpost(&m->om, beta);
    if (grad) orc_glp(&m->om, beta, grad);
    if (hess) { /* X^T W X + diag(1 / sd^2), W = sigma (1 - sigma): fit-jax-hmc.py:61-79 */
        for (int e = 0; e < p * p; ++e) hess[e] = 0.0;
        for (int64_t i = 0; i < m->n; ++i) {
            double eta = 0.0;
            for (int j = 0; j < p; ++j) eta += m->X[i * p + j] * beta[j];
            const double s = 1.0 / (1.0 + exp(-eta)), w = s * (1.0 - s);
            for (int a = 0; a < p; ++a)
                for (int b = 0; b < p; ++b) hess[a * p + b] += w * m->X[i * p + a] * m->X[i * p + b];
        }
        for (int j = 0; j < p; ++j) hess[j * p + j] += 1.0 / (m->sd[j] * m->sd[j]);
    }
    return LR_OK;
}

typedef struct { double n, mean, m2; } moments;
static void merge(moments *a, double nb, double mb, double m2b) { /* Chan et al. */
    if (nb <= 0.0) return;
    if (a->n <= 0.0) {
        a->n = nb;
        a->mean = mb;
        a->m2 = m2b;
        return;
    }
    const double tot = a->n + nb, d = mb - a->mean;
    a->mean += d * (nb / tot);
    a->m2 += m2b + d * d * (a->n * nb / tot);
    a->n = tot;
}

LR_API int lr_stats_reduce(int device, const double *stats, int64_t C, int32_t p, int64_t B, int64_t kept, const double *pivot, double *sums,
                           void *stream) {
    (void)device;
    (void)stream;
    if (!stats || !pivot || !sums) return fail(LR_ERR_INVALID, "NULL argument");
    if (C <= 0 || p <= 0 || B < 1 || kept < 0) return fail(LR_ERR_INVALID, "bad sizes");
    for (int e = 0; e < LR_STATS_ROWS * p; ++e) sums[e] = 0.0;
    const int64_t nb = kept / B, rem = kept - nb * B, stride = C * 2 * p;
    const int halves = nb >= 2 && (nb & 1) == 0;
    for (int64_t c = 0; c < C; ++c)
        for (int j = 0; j < p; ++j) {
            const double *s = stats + (c * 2) * p + j, piv = pivot[j];
            moments h[2] = {{0, 0, 0}, {0, 0, 0}};
            for (int64_t b = 0; b < nb; ++b) merge(&h[halves && b >= nb / 2 ? 1 : 0], (double)B, s[b * stride], s[b * stride + p]);
            moments full = h[0];
            merge(&full, h[1].n, h[1].mean, h[1].m2);
            double bm = 0.0;
            for (int64_t b = 0; b < nb; ++b) bm += (s[b * stride] - full.mean) * (s[b * stride] - full.mean);
            moments all = full;
            if (rem > 0) merge(&all, (double)rem, s[nb * stride], s[nb * stride + p]);
            const double dm = all.mean - piv;
            sums[0 * p + j] += all.n * dm;
            sums[1 * p + j] += all.n * dm * dm;
            sums[2 * p + j] += all.m2;
            if (halves)
                for (int w = 0; w < 2; ++w) {
                    const double dh = h[w].mean - piv;
                    sums[3 * p + j] += dh;
                    sums[4 * p + j] += dh * dh;
                    sums[5 * p + j] += h[w].n > 1.0 ? h[w].m2 / (h[w].n - 1.0) : 0.0;
                }
            sums[6 * p + j] += nb >= 2 ? bm : 0.0;
        }
    return LR_OK;
}

/* the planner: one variant (rows in "device" memory, one lane per chain) */
LR_API int lr_plan(const lr_model *m, int64_t n_chains, int32_t group, int32_t mode, int32_t *mode_out, int32_t *group_out, int32_t *rows_out) {
    if (!m) return fail(LR_ERR_INVALID, "model is NULL");
    if (n_chains <= 0) return fail(LR_ERR_INVALID, "n_chains must be positive");
    (void)group;
    (void)mode;
    if (mode_out) *mode_out = LR_MODE_GLOBAL;
    if (group_out) *group_out = 1;
    if (rows_out) *rows_out = 0;
    return LR_OK;
}
LR_API int lr_plan_run(const lr_model *m, int32_t kind, const lr_run_opts *o, int32_t *mode_out, int32_t *group_out, int32_t *rows_out) {
    int rc = check_opts(m, o, 0);
    if (rc) return rc;
    if (kind < LR_KIND_RWMH || kind > LR_KIND_UL) return fail(LR_ERR_INVALID, "unknown kernel family %d", kind);
    return lr_plan(m, o->plan_chains > 0 ? o->plan_chains : o->n_chains, o->group, o->mode, mode_out, group_out, rows_out);
}

/* "device" memory, streams, events */
LR_API int lr_malloc(int device, uint64_t bytes, void **dptr) {
    (void)device;
    if (!dptr) return fail(LR_ERR_INVALID, "dptr is NULL");
    *dptr = malloc(bytes ? bytes : 1);
    return *dptr ? LR_OK : fail(LR_ERR_NOMEM, "%llu bytes", (unsigned long long)bytes);
}
LR_API int lr_free(int device, void *dptr) {
    (void)device;
    free(dptr);
    return LR_OK;
}
LR_API int lr_memcpy_h2d(int device, void *dst, const void *src, uint64_t bytes, void *stream) {
    (void)device;
    (void)stream;
    memmove(dst, src, bytes);
    return LR_OK;
}
LR_API int lr_memcpy_d2h(int device, void *dst, const void *src, uint64_t bytes, void *stream) {
    (void)device;
    (void)stream;
    memmove(dst, src, bytes);
    return LR_OK;
}
LR_API int lr_memset(int device, void *dst, int value, uint64_t bytes, void *stream) {
    (void)device;
    (void)stream;
    memset(dst, value, bytes);
    return LR_OK;
}
LR_API int lr_stream_create(int device, void **stream) { return lr_malloc(device, 1, stream); }
LR_API int lr_stream_destroy(int device, void *stream) { return lr_free(device, stream); }
LR_API int lr_stream_sync(int device, void *stream) {
    (void)device;
    (void)stream;
    return LR_OK;
}
/* events hold the wall clock of their record call (the work is synchronous, so that is when it finished) */
LR_API int lr_event_create(int device, void **event) { return lr_malloc(device, sizeof(double), event); }
LR_API int lr_event_destroy(int device, void *event) { return lr_free(device, event); }
LR_API int lr_event_record(int device, void *event, void *stream) {
    (void)device;
    (void)stream;
    if (!event) return fail(LR_ERR_INVALID, "NULL event");
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    *(double *)event = ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
    return LR_OK;
}
LR_API int lr_event_elapsed_ms(int device, void *start, void *stop, float *ms) {
    (void)device;
    if (!start || !stop) return fail(LR_ERR_INVALID, "NULL event");
    if (ms) *ms = (float)(*(double *)stop - *(double *)start);
    return LR_OK;
}

/* The C-level exchange on the test double: ranks are PROCESSES of one host that meet in a POSIX shared-memory segment named after the
 * communicator id (lr_comm_unique_id: random; the caller hands it to the other ranks, as it hands RCCL's id on the GPU -- a file, a
 * launcher variable).  The semantics are the header's and RCCL's: lr_gather delivers every rank's `bytes` block to `recv` of the root,
 * in rank order (recv is not touched elsewhere); lr_allreduce_sum_f64 leaves in every rank's buffer the sum over the ranks, added in
 * rank order (so all ranks hold the same bits); both return when the data is there.  A sense-reversing barrier in the segment orders
 * the phases.  tests/test_distributed_gloo.py runs two and three processes against torch.distributed (gloo) on the same blocks. */
#include <fcntl.h>
#include <sched.h>
#include <stdatomic.h>
#include <sys/mman.h>
#include <sys/stat.h>
#define TWIN_MAX_WORLD 8
#define TWIN_SLOT_BYTES ((size_t)8 << 20) /* staging per rank (sparse: /dev/shm pages exist once touched) */
typedef struct twin_seg {
    atomic_int arrived, generation, attached;
    unsigned char pad[64 - 3 * sizeof(atomic_int)];
    unsigned char slot[TWIN_MAX_WORLD][TWIN_SLOT_BYTES];
} twin_seg;
struct lr_comm { int rank, world; twin_seg *seg; char name[80]; };
static void twin_barrier(lr_comm *c) {
    if (c->world == 1) return;
    const int gen = atomic_load(&c->seg->generation);
    if (atomic_fetch_add(&c->seg->arrived, 1) + 1 == c->world) {
        atomic_store(&c->seg->arrived, 0);
        atomic_fetch_add(&c->seg->generation, 1);
    } else {
        while (atomic_load(&c->seg->generation) == gen) sched_yield();
    }
}
LR_API int lr_comm_unique_id(void *id) {
    if (!id) return fail(LR_ERR_INVALID, "NULL id");
    memset(id, 0, LR_COMM_ID_BYTES);
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    snprintf((char *)id, LR_COMM_ID_BYTES, "lrtwin-%d-%lld-%ld", (int)getpid(), (long long)ts.tv_sec, ts.tv_nsec);
    return LR_OK;
}
LR_API int lr_comm_create(const void *id, int32_t rank, int32_t world, int device, lr_comm **out) {
    (void)device;
    if (!id || !out) return fail(LR_ERR_INVALID, "NULL id / out");
    if (world < 1 || world > TWIN_MAX_WORLD || rank < 0 || rank >= world) return fail(LR_ERR_INVALID, "rank %d of world %d (the test double takes up to %d ranks)", rank, world, TWIN_MAX_WORLD);
    lr_comm *c = (lr_comm *)calloc(1, sizeof(lr_comm));
    c->rank = rank;
    c->world = world;
    if (world > 1) {
        char tag[LR_COMM_ID_BYTES + 1];
        memcpy(tag, id, LR_COMM_ID_BYTES);
        tag[LR_COMM_ID_BYTES] = 0;
        for (char *q = tag; *q; ++q)
            if (*q == '/') *q = '_';
        snprintf(c->name, sizeof c->name, "/%.70s", tag[0] ? tag : "lrtwin-default");
        const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, (off_t)sizeof(twin_seg)) != 0) {
            if (fd >= 0) close(fd);
            free(c);
            return fail(LR_ERR_HIP, "shared-memory segment %s of the test double's exchange could not be made", c->name);
        }
        c->seg = (twin_seg *)mmap(NULL, sizeof(twin_seg), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        close(fd);
        if (c->seg == MAP_FAILED) {
            free(c);
            return fail(LR_ERR_NOMEM, "mapping the exchange segment failed");
        }
        atomic_fetch_add(&c->seg->attached, 1);
        while (atomic_load(&c->seg->attached) < world) sched_yield(); /* (a fresh segment is zero-filled: the counters start at 0) */
        twin_barrier(c);
    }
    *out = c;
    return LR_OK;
}
LR_API int lr_comm_destroy(lr_comm *comm) {
    if (!comm) return LR_OK;
    if (comm->world > 1) {
        twin_barrier(comm);
        munmap(comm->seg, sizeof(twin_seg));
        if (comm->rank == 0) shm_unlink(comm->name);
    }
    free(comm);
    return LR_OK;
}
LR_API int lr_gather(lr_comm *comm, const void *send, void *recv, uint64_t bytes, int32_t root, void *stream) {
    (void)stream;
    if (!comm || !send) return fail(LR_ERR_INVALID, "comm / send is NULL");
    if (root < 0 || root >= comm->world) return fail(LR_ERR_INVALID, "root %d of world %d", root, comm->world);
    if (comm->rank == root && !recv) return fail(LR_ERR_INVALID, "recv is NULL on the root rank");
    if (comm->world == 1) {
        memmove(recv, send, bytes);
        return LR_OK;
    }
    if (bytes > TWIN_SLOT_BYTES) return fail(LR_ERR_UNSUPPORTED, "the test double stages at most %zu bytes per rank", TWIN_SLOT_BYTES);
    memcpy(comm->seg->slot[comm->rank], send, bytes);
    twin_barrier(comm);
    if (comm->rank == root)
        for (int r = 0; r < comm->world; ++r) memcpy((unsigned char *)recv + (uint64_t)r * bytes, comm->seg->slot[r], bytes);
    twin_barrier(comm); /* the slots are free again */
    return LR_OK;
}
LR_API int lr_allreduce_sum_f64(lr_comm *comm, double *buf, uint64_t count, void *stream) {
    (void)stream;
    if (!comm || !buf) return fail(LR_ERR_INVALID, "comm / buf is NULL");
    if (comm->world == 1) return LR_OK;
    if (count * 8 > TWIN_SLOT_BYTES) return fail(LR_ERR_UNSUPPORTED, "the test double stages at most %zu bytes per rank", TWIN_SLOT_BYTES);
    memcpy(comm->seg->slot[comm->rank], buf, count * 8);
    twin_barrier(comm);
    for (uint64_t i = 0; i < count; ++i) {
        double s = ((const double *)comm->seg->slot[0])[i];
        for (int r = 1; r < comm->world; ++r) s += ((const double *)comm->seg->slot[r])[i];
        buf[i] = s;
    }
    twin_barrier(comm);
    return LR_OK;
}
