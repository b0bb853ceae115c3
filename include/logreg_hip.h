/*
 * logreg_hip.h -- C ABI of liblogreg_hip.so: many-chain MCMC for Bayesian logistic regression
 * on AMD MI355X (gfx950).  Plain C, no exceptions, no torch types: pointers and sizes only.
 *
 * The reference (darrenjw/logreg) has no FFI for this path: its whole "interface" is the set of
 * Python closures in Python/fit-numpy.py, Python/fit-np-mala.py, Python/fit-np-hmc.py and
 * Python/fit-np-ul.py.  Each entry point below names the reference closure(s) it replaces
 * (file:line under the reference root); logreg_amd/ binds them with ctypes and re-exposes the
 * reference's Python names and signatures (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - every function returns 0 on success, a negative lr_status on failure;
 *     lr_last_error() returns a thread-local message for the last failure on this thread.
 *   - a model handle is bound to one device; HOST calls on one handle must be serialised by the
 *     caller; different handles may be used from different threads/processes.  The DEVICE work of
 *     on_device calls issued on different streams may overlap: the scratch state of the stepwise engine
 *     is kept per (handle, stream), so two chain sets of one model on two streams do not disturb each other.
 *   - "dtype" is the arithmetic type the device path computes in and the element type of every
 *     `void*` array below: LR_F32 (float) or LR_F64 (double).  Model inputs and kernel tuning
 *     vectors are always host doubles (they are tiny and converted once).
 *   - arrays are row-major; C = number of chains, p = number of parameters.
 *   - opts->on_device = 0: array pointers are HOST memory; the call copies in, runs, copies out
 *     and returns when the results are in host memory.
 *     opts->on_device = 1: array pointers are DEVICE memory on the model's device; work is
 *     enqueued on opts->stream (a hipStream_t, NULL = default stream) and the call returns
 *     without synchronising.
 *   - randomness: Philox4x32-10, key = seed, counter = (chain_offset + c, iter_offset + t,
 *     block).  Results depend only on (seed, global chain id, global iteration index): any
 *     split of a run into launches (iter_offset) or shards (chain_offset) reproduces the
 *     monolithic run bit for bit.
 */
#ifndef LOGREG_HIP_H
#define LOGREG_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LR_API __attribute__((visibility("default")))

typedef enum lr_status {
    LR_OK = 0,
    LR_ERR_INVALID = -1,     /* bad argument */
    LR_ERR_HIP = -2,         /* HIP runtime error (no device, launch failure, ...) */
    LR_ERR_UNSUPPORTED = -3, /* no compiled kernel variant for this (dtype, p, n, group, mode) */
    LR_ERR_NOMEM = -4
} lr_status;

enum { LR_F32 = 0, LR_F64 = 1 };
/* where the data rows live / which pipe does the matvecs: REG/LDS/GLOBAL use the vector ALU with rows in
 * VGPRs / LDS / memory; MFMA is the fused matrix-core chain kernel (5 <= p <= 32; `group` selects the row-split
 * ways S in {1,4} instead of lanes per chain; its operands live in VGPRs, in LDS or in device memory as n grows --
 * lr_plan reports tiles of 16 rows per wave, 0 or -1 as rows_out -- up to 8192 rows); STEPWISE is the tall-data engine: two
 * small kernels per log-posterior evaluation, the rows split into slices across the whole chip (`group` = 0
 * lets the library choose the slice count from the chain count, `group` > 0 requests that many slices; lr_plan
 * reports the slice count as group_out and the slice length as rows_out).  Slice partials are summed in slice
 * order, so launches of different chain counts are bit-identical only under the same slicing: a sharded run
 * that wants bit-exact agreement with the one-GPU run sets lr_run_opts.plan_chains to the whole run's chain count
 * (which pins the interior kernels' own slicing as well; an explicit group > 0 pins the end-point slicing only).
 * (float64 models: MFMA exists for HMC at 5 <= p <= 8, n <= 208, group = 1 -- lr_mfma_f64.h.)
 * MIXED (float64 models, HMC only, p <= 16, rows within the float32 register variants: n <= 1024 at p <= 8, 512 at p <= 16): float64
 * rows in LDS for the end points of a trajectory, the same rows rounded to float32 in VGPRs for its interior gradients (see
 * LR_PREC_*); `group` = lanes per chain (16 / 32 / 64). */
enum { LR_MODE_AUTO = -1, LR_MODE_REG = 0, LR_MODE_LDS = 1, LR_MODE_GLOBAL = 2, LR_MODE_MFMA = 3, LR_MODE_STEPWISE = 4, LR_MODE_MIXED = 5 };

typedef struct lr_model lr_model;

typedef struct lr_run_opts {
    int64_t n_chains;     /* C: chains in this call */
    int64_t chain_offset; /* global id of chain 0 of this call (sharding) */
    int64_t thin;         /* iterations per kept sample      (mcmc(thin=...)) */
    int64_t iters;        /* kept samples in this call       (mcmc(iters=...)) */
    int64_t iter_offset;  /* global index of this call's first iteration (chunked runs) */
    uint64_t seed;
    int32_t group;        /* 0 = choose automatically; else per mode: lanes per chain 1,2,4,...,64 (REG / LDS / GLOBAL), row-split
                           * ways 1, 4 or 8 (MFMA), slice count >= 1 (STEPWISE and every model with p > 32) */
    int32_t mode;         /* LR_MODE_*: where the data rows live during the launch */
    int32_t on_device;    /* see conventions */
    void* stream;         /* hipStream_t when on_device = 1 */
    /* streaming posterior statistics (optional; see "Streaming statistics" below) */
    double* stats;        /* NULL, or [stats_slots][C][2][p] running (mean, M2) per batch of kept samples */
    int64_t stats_batch;  /* B >= 1: kept samples per batch slot */
    int64_t stats_first;  /* index, within the statistics window, of this call's first kept sample */
    int64_t stats_slots;  /* slots in the buffer: stats_first + iters <= stats_slots * stats_batch */
    int32_t precision;    /* LR_PREC_*: arithmetic of HMC's INTERIOR leapfrog gradients (see below) */
    int32_t plan_chains;  /* 0, or the chain count to PLAN for instead of n_chains: a shard of a larger run passes the whole
                           * run's chain count, so that every choice that depends on the chain count (kernel variant, row
                           * slicing of the stepwise engine and of its interior kernels, trajectory kernels) is the one-GPU
                           * run's and the shard's output is bit-identical to the same chains of that run */
    /* (layout history: `plan_first` was appended in round 4 -- lr_sizeof_run_opts() lets a binding check its struct against the
     *  library's; a caller that zero-initialises an OLDER, shorter struct must not be linked against this library.  With
     *  plan_chains > 0 the launched chains [chain_offset, chain_offset + n_chains) must lie inside the planned run
     *  [plan_first, plan_first + plan_chains): LR_ERR_INVALID otherwise -- a chain_offset used only to decorrelate runs goes with
     *  plan_chains = 0.) */
    int64_t plan_first;   /* read only when plan_chains > 0: global id (chain_offset units) of the FIRST chain of the run being
                           * planned for.  A run planned in two parts (lr_plan_info.split: an exactly-filled head on narrow lane
                           * groups, the remainder on wide ones) assigns a chain to its part by its position in the WHOLE run,
                           * chain_offset + c - plan_first, so a shard needs to know where the run starts.  0 for a run whose
                           * chains are numbered from 0 (logreg_amd.distributed.mcmc_sharded). */
} lr_run_opts;

/*
 * Interior-gradient precision of lr_run_hmc.  The L - 1 gradient evaluations strictly inside a trajectory only
 * steer it: the leapfrog map stays volume-preserving and reversible for any deterministic force, and the
 * Metropolis test uses the log-posterior at the two END points, which is always evaluated in the model's full
 * precision (as are the end-point half-kicks).  So the interior evaluations may be cheaper without biasing the
 * sampler; the only possible cost is acceptance rate.
 *   LR_PREC_AUTO   (= 0: what a zeroed lr_run_opts asks for, and the default of the Python face's mcmc()) the library's
 *                  choice: reduced-precision interior gradients wherever such a kernel exists for a float32 model --
 *                    narrow models, 8 <= padded p <= 32, n <= 8192: the fused matrix-core chain kernel (LR_MODE_MFMA; rows
 *                      and beta in two bf16 pieces each, w = sigma(-eta) in one), planned from about 4 chains per CU
 *                      (p > 8) / 16 per CU (p <= 8) upward, i.e. 1024 - 4096 chains on MI355X;
 *                    tall models on the stepwise engine (the same scheme with the rows streamed: lr_tall_mx.h);
 *                    wide models, 32 < p <= 128: rows, beta and w in ONE half-precision (f16) piece each where every |x| <= 2^15 and every
 *                      column reaches 2^-10 (config 5: the exact interior's acceptance rate), else rows in one bf16 piece, beta in two (lr_wide_bf16.h);
 *                  ... and for a float64 model with p <= 16 and rows within the register variants (LR_MODE_MIXED): float32 interior gradients -- the
 *                    trajectory's position and momentum, both end-point evaluations, the half kicks, the kinetic energies and
 *                    the Metropolis test stay float64; only the force applied inside the trajectory is computed from the
 *                    position and the rows rounded to float32 (4 - 5 x the all-float64 rate); from 33 chains per CU (n <= 208) the
 *                    fused matrix-core kernel takes over (LR_MODE_MFMA, bf16 interior force, the same float64 everything else);
 *                    tall and wide float64 models run the stepwise engine's bf16 interior kernels on a float64 state;
 *                  elsewhere (other float64 models -- 17 <= p <= 32 with rows in LDS run all-float64 on the distributed-state
 *                    kernel --, p < 5, few chains on register/LDS-resident data) it is LR_PREC_FULL.
 *                  A default HMC run is therefore NOT step-for-step comparable with a float64 reference run (the
 *                  posterior is the same; acceptance rates measured within 0.001 - 0.01 of the exact-gradient run);
 *   LR_PREC_FULL   every evaluation in the model's dtype (comparable with the float64 oracle step by step)
 *   LR_PREC_BF16   request the reduced-precision interior kernels (ignored where none exists); on the wide models' two-tile trajectory
 *                  kernel rows and beta in ONE bf16 piece each (config 5 whole: 20.0 us per evaluation against the default's 20.9,
 *                  acceptance 0.738 against 0.758)
 * RWMH, MALA and UL ignore the field (every evaluation of theirs enters an accept ratio or is the sample itself).
 */
enum { LR_PREC_AUTO = 0, LR_PREC_FULL = 1, LR_PREC_BF16 = 2 };

LR_API const char* lr_last_error(void);
/* content hash of the sources this library was compiled from (logreg_amd/build.py: source_hash): the binding
 * refuses a library whose id differs from the sources beside it */
LR_API const char* lr_build_id(void);
/* sizeof(lr_run_opts) as this library was compiled: a binding in another language checks its struct layout */
LR_API int lr_sizeof_run_opts(void);
LR_API int lr_device_count(void);
/* number of compute units of `device` (<0 on error) */
LR_API int lr_device_cus(int device);
/* identity of `device` as a NUL-terminated string "pci=<domain:bus:dev.fn> uuid=<32 hex digits> name=<marketing name> cus=<n>"
 * written to buf (at most len bytes, always terminated): what a multi-process run compares across its ranks to show that
 * every rank drives a different GPU (bench.py: `devices`).  The reference has no counterpart (single process, CPU). */
LR_API int lr_device_info(int device, char* buf, int len);

/*
 * Model = data block + model closures.
 * Replaces the data block Python/fit-np-hmc.py:12-19 (X with intercept column, y in {0,1}) and
 * the closure state of ll/lprior/lpost/glp (fit-np-hmc.py:23-47; `pscale` fit-np-hmc.py:31).
 * X [n,p] and y [n] and prior_sd [p] are host doubles, copied; caller keeps ownership.
 */
LR_API int lr_model_create(const double* X, const double* y, int64_t n, int32_t p, const double* prior_sd,
                           int32_t dtype, int32_t device, lr_model** out);
LR_API void lr_model_destroy(lr_model* m);
LR_API int lr_model_info(const lr_model* m, int64_t* n, int32_t* p, int32_t* dtype, int32_t* device,
                         int32_t* padded_p);
/* A/B switches this model runs with.  The library reads ONE environment variable, LOGREG_DEBUG_OPTS="key=value,...", once, in
 * lr_model_create (an unknown key or value fails the creation); nothing else on the run path consults the environment:
 *   residency_cap=0   fused chain kernels launched without the LDS request that spreads a grid evenly over the CUs
 *   tall_mx16=0       tall models: 4-wave interior kernel with separate update launches instead of the fused 16-wave form
 *   wide_traj=0|1|2   wide models: forbid / force the one-launch trajectory kernel with 1 / 2 chain tiles per workgroup (default: by chain count and design size)
 *   wide_waves=4|8    wide models: waves (x 16 chains) per workgroup of the exact-split / chain-split kernels (default: by chain count)
 *   wide_f16=0|1|2    wide models, reduced-precision interior steps: 0 = bf16 rows x two bf16 pieces of beta even where the rows fit the
 *                     one-piece half-precision (f16) format (the default, 1), 2 = f16 also where LR_PREC_BF16 would take bf16 x one piece
 * Writes "" to buf for a model on the defaults (what a benchmark must run with), else all the settings.  No reference counterpart. */
LR_API int lr_model_debug_opts(const lr_model* m, char* buf, int len);
/* Operand format of the reduced-precision interior leapfrog steps (LR_PREC_AUTO / LR_PREC_BF16) this model's HMC runs take where such a
 * matrix-pipe kernel is planned: LR_INTERIOR_NONE (none exists for the shape: p < 5), LR_INTERIOR_BF16 (bf16 operand pieces),
 * LR_INTERIOR_F16 (wide models whose design fits IEEE half precision: every |x| <= 2^15, no column below 2^-10 throughout).
 * No reference counterpart (the reference has one arithmetic). */
enum { LR_INTERIOR_NONE = 0, LR_INTERIOR_BF16 = 1, LR_INTERIOR_F16 = 2 };
LR_API int lr_model_interior_format(const lr_model* m, int32_t* format);

/*
 * ll(beta), lprior(beta), lpost(beta), glp(beta) for C parameter vectors at once.
 * Replaces fit-np-hmc.py:23-24 (ll), :33-34 (lprior), :36-37 (lpost), :44-47 (glp).
 * beta [C,p]; ll/lprior/lpost [C]; grad [C,p]; any output may be NULL.
 * Only n_chains, group, mode, on_device, stream of `opts` are read.
 */
LR_API int lr_eval(lr_model* m, const void* beta, void* ll, void* lprior, void* lpost, void* grad,
                   const lr_run_opts* opts);

/*
 * mcmc(init, kernel, thin, iters) with the kernel fused in: every call advances all chains by
 * iters*thin iterations on the device and writes the thinned states.
 * Replaces mcmc() fit-numpy.py:64-79 / fit-np-mala.py:80-95 / fit-np-hmc.py:89-103 /
 * fit-np-ul.py:70-84 together with the kernel named per function.
 *   state    [C,p]  in/out  current states
 *   lp_state [C]    in/out  (double, always) the log-density threaded through mhKernel for
 *                           RWMH/MALA; -inf is allowed and means "accept the first proposal"
 *                           (fit-np-mala.py:82)
 *   out      [iters,C,p] or NULL   row i = states after (i+1)*thin iterations
 *   accepts  [C] or NULL           incremented by the number of accepted proposals
 */
/* mhKernel(lpost, rprop) with rprop(beta) = beta + prop_sd*N(0,I): fit-numpy.py:53-62, :81-84 */
LR_API int lr_run_rwmh(lr_model* m, void* state, double* lp_state, const double* prop_sd,
                       const lr_run_opts* opts, void* out, uint32_t* accepts);
/* malaKernel(lpi, glpi, dt, pre) inside mhKernel: fit-np-mala.py:61-78 */
LR_API int lr_run_mala(lr_model* m, void* state, double* lp_state, double dt, const double* pre,
                       const lr_run_opts* opts, void* out, uint32_t* accepts);
/* ulKernel(glpi, dt, pre): fit-np-ul.py:61-68 (no accept step; accepts counts iterations) */
LR_API int lr_run_ul(lr_model* m, void* state, double dt, const double* pre, const lr_run_opts* opts,
                     void* out, uint32_t* accepts);
/* hmcKernel(lpi, glpi, eps, l, dmm) with its own mhKernel: fit-np-hmc.py:56-87
 * (opts->precision selects the arithmetic of the l - 1 interior leapfrog gradients: LR_PREC_*) */
LR_API int lr_run_hmc(lr_model* m, void* state, double eps, int32_t l, const double* dmm,
                      const lr_run_opts* opts, void* out, uint32_t* accepts);

/*
 * lpost, glp and the NEGATIVE Hessian of lpost at ONE beta, in float64 arithmetic whatever the model's dtype:
 *     hess = X^T W X + diag(1 / prior_sd^2),  W = diag(sigma(eta) (1 - sigma(eta)))       [p,p], symmetric
 * One pass over the rows on the device.  Serves the warm start that replaces the reference's SciPy call
 * `minimize(-lpost, init, jac=-glp, method='BFGS')` (fit-np-hmc.py:49) with Newton's method, the closed-form
 * Hessian being the one of the reference's JAX variant (fit-jax-hmc.py:61-79).  beta [p], grad [p], hess [p,p]
 * and lpost are HOST doubles; any output may be NULL.  Synchronous.
 */
LR_API int lr_hessian(lr_model* m, const double* beta, double* lpost, double* grad, double* hess, void* stream);

/*
 * Streaming statistics: the on-device replacement for keeping every thinned sample when only a posterior summary
 * is wanted.  The reference post-processes the full [iters, p] matrix: scipy.stats.describe in
 * fit-np-hmc.py:113-117 and smfsb::mcmcSummary in Python/analyse.R:17-19; with 65 536 chains that matrix is tens
 * of GB.  When opts->stats is set every lr_run_* call folds each kept sample into the running (mean, M2) of its
 * batch slot (Welford; the first sample of a slot initialises it, so the buffer needs no clearing):
 *     stats[b][c][0][j] = mean, stats[b][c][1][j] = sum of squared deviations, of parameter j of chain c over
 *     the kept samples [b*B, (b+1)*B) of the statistics window
 * lr_stats_reduce turns the buffer (DEVICE memory) into chain-pooled sums, rows of `sums` [LR_STATS_ROWS][p]
 * (HOST doubles), with n = kept samples per chain, nb = n / B full batches, piv = pivot[j]:
 *     0  sum_c n (mean_c - piv)              1  sum_c n (mean_c - piv)^2       2  sum_c M2_c
 *     3  sum_h (mean_h - piv)   over the 2C half-chains (first / second nb/2 batches; nb even, else 0)
 *     4  sum_h (mean_h - piv)^2              5  sum_h M2_h / (n_h - 1)
 *     6  sum_c sum_b (batchmean_cb - mean_c)^2   over the full batches
 * Sums are additive over chain shards (one all-reduce across GPUs); logreg_amd/diagnostics.py turns them into
 * mean, sd (ddof = 1), split-R-hat and batch-means ESS.  `pivot` only has to be near the posterior (it removes
 * cancellation from the squared sums); every shard must pass the same one.
 */
#define LR_STATS_ROWS 7
LR_API int lr_stats_reduce(int device, const double* stats, int64_t n_chains, int32_t p, int64_t batch, int64_t kept,
                           const double* pivot, double* sums, void* stream);

/*
 * Which kernel variant the library would launch for (model, n_chains, group, mode):
 * mode_out in LR_MODE_*, group_out lanes per chain, rows_out rows per lane (REG mode, else 0; MFMA: 16-row tiles per
 * wave, 0 = operands in LDS, -1 = operands in device memory; STEPWISE: rows per slice).
 */
LR_API int lr_plan(const lr_model* m, int64_t n_chains, int32_t group, int32_t mode, int32_t* mode_out,
                   int32_t* group_out, int32_t* rows_out);

/*
 * The variant lr_run_<kind> would launch for these options (n_chains, group, mode, precision are read): unlike
 * lr_plan it knows the kernel family -- HMC runs whose interior gradients may use the bf16 matrix pipe
 * (LR_PREC_AUTO / LR_PREC_BF16) are planned onto the fused matrix-core kernels when there are enough chains.
 */
enum { LR_KIND_RWMH = 0, LR_KIND_MALA = 1, LR_KIND_HMC = 2, LR_KIND_UL = 3 };
LR_API int lr_plan_run(const lr_model* m, int32_t kind, const lr_run_opts* opts, int32_t* mode_out, int32_t* group_out,
                       int32_t* rows_out);
/*
 * The same plan in full.  Chain counts between the ones that fill the chip exactly lose to wave quantisation (5120 chains on 16
 * lanes per chain: two waves on a quarter of the SIMDs, one on the rest, and the launch takes the time of 8192): the planner then
 * splits a run of register-resident kernels in TWO launches -- chains [0, split) of the run on (group, rows), the exactly-filled
 * head; chains [split, n) on (tail_group, tail_rows), wider lane groups that finish the remainder in one short launch.
 * The remainder's launch runs beside the head's on a stream owned by the model handle, forked from opts->stream (after everything
 * enqueued there so far) and joined back into it before the call returns: to the caller the call is still "enqueued on opts->stream".
 * The same holds for HMC on the fused matrix-core kernel under LR_PREC_AUTO when the remainder is at most a quarter of an exactly-
 * filling chain count (5120 chains: 4096 on the matrix cores + 1024 on 64 lanes per chain, 0.538 -> 0.440 ms per 20 iterations).
 * split = 0: one part (every field of the tail is 0).  A chain's variant is a function of its position in the planned run
 * (lr_run_opts.plan_chains / plan_first), so chunked and sharded runs reproduce the one-launch-sequence run bit for bit.
 */
typedef struct lr_plan_info {
    int32_t mode, group, rows;     /* as lr_plan_run */
    int32_t tail_group, tail_rows; /* the remainder's variant ... */
    int64_t split;                 /* chains of the planned run in the first part; 0 = no second part */
    int32_t tail_mode;             /* ... and its mode: LR_MODE_REG (also behind a matrix-core head: HMC under LR_PREC_AUTO, where the
                                    * remainder's chains then run with exact interior gradients -- the policy permits reduced precision,
                                    * it does not require it) */
    int32_t reserved;
} lr_plan_info;
LR_API int lr_plan_run_info(const lr_model* m, int32_t kind, const lr_run_opts* opts, lr_plan_info* out);

/* device memory + stream + event helpers so a host language needs no other GPU runtime */
LR_API int lr_malloc(int device, uint64_t bytes, void** dptr);
LR_API int lr_free(int device, void* dptr);
LR_API int lr_memcpy_h2d(int device, void* dst, const void* src, uint64_t bytes, void* stream);
LR_API int lr_memcpy_d2h(int device, void* dst, const void* src, uint64_t bytes, void* stream);
LR_API int lr_memset(int device, void* dst, int value, uint64_t bytes, void* stream);
LR_API int lr_stream_create(int device, void** stream);
LR_API int lr_stream_destroy(int device, void* stream);
LR_API int lr_stream_sync(int device, void* stream);
LR_API int lr_event_create(int device, void** event);
LR_API int lr_event_destroy(int device, void* event);
LR_API int lr_event_record(int device, void* event, void* stream);
/* synchronises on `stop`, then returns the time between the two events in milliseconds */
LR_API int lr_event_elapsed_ms(int device, void* start, void* stop, float* ms);

/*
 * The multi-GPU exchange of the path at the C level (SURVEY.md section 8(b) `lr_gather`): one process per GPU, chains sharded in
 * contiguous blocks (lr_run_opts.chain_offset / plan_chains / plan_first), ONE exchange -- the gather of the thinned samples to a root
 * rank, or the sum of the statistics rows (lr_stats_reduce) over the ranks.  A thin wrapper over RCCL (point-to-point send / receive
 * group and all-reduce over xGMI), loaded at first use (dlopen of librccl.so.1: the library has no link-time dependency on it), so
 * that a host language needs no other collective runtime; the Python face uses torch.distributed (the same RCCL) and does not call
 * these.  The reference has no counterpart (single process).
 *   lr_comm_unique_id   one rank makes the LR_COMM_ID_BYTES identifier and hands it to the others by any means (a file, a socket)
 *   lr_comm_create      every rank, with the same identifier: collective; binds the communicator to `device`
 *   lr_gather           every rank sends `bytes` bytes at `send` (device memory); rank `root` receives world blocks, in rank order, at
 *                       `recv` (device memory, world * bytes; ignored elsewhere).  Enqueued on `stream`.
 *   lr_allreduce_sum_f64  in place over `count` doubles of device memory (the [LR_STATS_ROWS][p] sums + a chain count)
 */
typedef struct lr_comm lr_comm;
#define LR_COMM_ID_BYTES 128
LR_API int lr_comm_unique_id(void* id);
LR_API int lr_comm_create(const void* id, int32_t rank, int32_t world, int device, lr_comm** out);
LR_API int lr_comm_destroy(lr_comm* comm);
LR_API int lr_gather(lr_comm* comm, const void* send, void* recv, uint64_t bytes, int32_t root, void* stream);
LR_API int lr_allreduce_sum_f64(lr_comm* comm, double* buf, uint64_t count, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LOGREG_HIP_H */
