// lr_kernels.h -- the fused many-chain kernels: one launch runs `iters x thin` MCMC iterations
// for every chain and writes only the thinned states (the whole of the reference's mcmc() double
// loop, Python/fit-np-hmc.py:89-103, lives inside the kernel).
#pragma once
#include <type_traits>

#include "lr_device.h"
#include "lr_stamps.h"

namespace lr {

enum Kind { KIND_RWMH = 0, KIND_MALA = 1, KIND_HMC = 2, KIND_UL = 3 };
enum Mode { MODE_REG = 0, MODE_LDS = 1, MODE_GLOBAL = 2, MODE_MFMA = 3, MODE_STEPWISE = 4, MODE_MIXED = 5 };

template <typename T, int P> struct ModelArgs {
    const T* rows;  // [n][P] signed rows (2y-1)*x, zero-padded to P columns (device)
    const float* rows_tw;  // float32 models, P <= 32: the same rows as twisted row pairs (ScalarRowPairs), else null
    const float* rows_mf;  // float32 models whose matrix-core operands live in LDS: fp32 MFMA operand images per 16-row tile
                           // (lr_mfma.h mf_image_floats), else null
    const unsigned char* ops_mf;  // float32 models beyond that: the bf16 operand images in device memory (MfmaRowsLds<P, 4, true>), else null
    int64_t n;
    Prior<T, P> prior;
};

template <typename T> struct EvalArgs {
    const T* beta;  // [C][p]
    int64_t C;
    int p;  // real parameter count (<= P)
    T* ll;  // [C] each, any may be null
    T* lprior;
    T* lpost;
    T* grad;  // [C][p]
};

// kind-specific vectors, precomputed on the host in double (zero in padded coordinates):
//   RWMH : a = proposal sd                                   (0.02*pre, fit-numpy.py:83-84)
//   MALA : a = 0.5*pre*dt, b = sqrt(pre*dt), c = 1/(pre*dt)  (fit-np-mala.py:72-78)
//   UL   : a = 0.5*pre*dt, b = sqrt(pre*dt)                  (fit-np-ul.py:61-68)
//   HMC  : a = sqrt(dmm),  b = eps/dmm,      c = 1/dmm       (fit-np-hmc.py:65-87)
//          d = b*k, e = prior inv_var/k with k = ExpScale<T>::k: the leapfrog position is carried as k*q, which
//          is what the row dot products want (exp2), so interior evaluations do no rescaling at all
template <typename T, int P> struct ChainArgs {
    T* state;           // [C][p] in/out
    double* lp_state;   // [C] threaded log-density of RWMH/MALA (in/out, -inf allowed); null for HMC/UL
    T* out;             // [iters][C][p] or null
    uint32_t* accepts;  // [C], incremented; or null
    int64_t C;             // chains of the CALL: the stride of out / stats, the bound of state / lp_state / accepts
    int64_t first, count;  // the chains THIS LAUNCH advances: [first, first + count) of the call's arrays (a call planned in two
                           // parts -- lr_plan.h: exactly-filled head on narrow groups, remainder on wide ones -- is two launches)
    int64_t chain_offset;  // global id of local chain 0 (Philox counter)
    int64_t iters, thin;
    int64_t iter_offset;  // global index of this launch's first iteration (Philox counter)
    uint64_t seed;
    int p;
    int l;   // HMC leapfrog steps
    T step;  // HMC eps
    T a[P], b[P], c[P];
    T d[P], e[P];  // HMC: d = b * ExpScale<T>::k, e = prior inv_var / ExpScale<T>::k (trajectory in scaled units)
    StatsArgs stats;  // streaming (mean, M2) per batch of kept samples; buf = null: off
    int interior_bf16;  // HMC, matrix-core variants: interior leapfrog gradients from bf16 operands (LR_PREC_*)
};

// --------------------------------------------------------------------------------------------
template <typename T, int P, int G, int MODE, int R> struct RowsOf {
    using type = StridedRows<T, P, G>;
};
template <typename T, int P, int G, int R> struct RowsOf<T, P, G, MODE_REG, R> {
    using type = RegRows<T, P, R, G>;
};
template <int P, int G, int R> struct RowsOf<float, P, G, MODE_REG, R> {
    using type = RegRowPairs<P, R, G>;
};
template <typename T, int P, int R> struct RowsOf<T, P, 1, MODE_GLOBAL, R> {
    using type = ScalarRows<T, P>;  // lane-per-chain: rows broadcast through the scalar unit
};
template <int R> struct RowsOf<float, 4, 1, MODE_GLOBAL, R> { using type = ScalarRowPairs<4>; };
template <int R> struct RowsOf<float, 8, 1, MODE_GLOBAL, R> { using type = ScalarRowPairs<8>; };
template <int R> struct RowsOf<float, 16, 1, MODE_GLOBAL, R> { using type = ScalarRowPairs<16>; };  // P = 32: 64 SGPRs per pair

template <typename T, int P, int G, int MODE, int R>
__device__ __forceinline__ typename RowsOf<T, P, G, MODE, R>::type make_rows(const ModelArgs<T, P>& m, int gl,
                                                                             T* smem) {
    typename RowsOf<T, P, G, MODE, R>::type rows;
    if constexpr (MODE == MODE_REG) {
        rows.load(m.rows, m.n, gl);
    } else if constexpr (MODE == MODE_LDS) {
        // stage all rows once; coalesced copy by the whole workgroup
        const int64_t tot = m.n * P;
        for (int64_t i = threadIdx.x; i < tot; i += blockDim.x) smem[i] = m.rows[i];
        __syncthreads();
        rows.base = smem;
        rows.n = m.n;
        rows.gl = gl;
    } else if constexpr (is_scalar_pairs<typename RowsOf<T, P, G, MODE, R>::type>::value) {
        rows.base = m.rows_tw;
        rows.k0 = 0;
        rows.k1 = (m.n + 1) / 2;
        rows.zero_rows = (int)(m.n & 1);
    } else if constexpr (G == 1) {
        rows.base = m.rows;
        rows.i0 = 0;
        rows.i1 = m.n;
    } else {
        rows.base = m.rows;
        rows.n = m.n;
        rows.gl = gl;
    }
    return rows;
}

// --------------------------------------------------------------------------------------------
// batched ll / lprior / lpost / glp   (reference fit-np-hmc.py:23-47)
template <typename T, int P, int G, int MODE, int R>
__global__ void __launch_bounds__(256) k_eval(ModelArgs<T, P> m, EvalArgs<T> a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int gl = threadIdx.x % G;
    int64_t chain = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const bool live = chain < a.C;
    if (!live) chain = a.C - 1;
    const auto rows = make_rows<T, P, G, MODE, R>(m, gl, reinterpret_cast<T*>(smem_raw));
    T beta[P], grad[P];
#pragma unroll
    for (int j = 0; j < P; ++j) beta[j] = j < a.p ? a.beta[chain * a.p + j] : T(0);
    double ll, lpr;
    eval_lpost<T, P, G, true, true>(rows, m.prior, beta, grad, ll, lpr);
    if (live && gl == 0) {
        if (a.ll) a.ll[chain] = (T)ll;
        if (a.lprior) a.lprior[chain] = (T)lpr;
        if (a.lpost) a.lpost[chain] = (T)(ll + lpr);
        if (a.grad) {
#pragma unroll
            for (int j = 0; j < P; ++j)
                if (j < a.p) a.grad[chain * a.p + j] = grad[j];
        }
    }
}

// --------------------------------------------------------------------------------------------
// HMC interior leapfrog loop for the 16-lanes-per-chain register kernel (float, P = 8): l - 1 x (drift, gradient,
// kick) with the position and momentum of a chain DISTRIBUTED over its 16 lanes -- quad q owns coordinates 2q, 2q + 1
// -- instead of replicated in all of them.  Per step: all-gather the position (8 v_mov_b32_dpp row_share), the row
// pass (unchanged), a reduce-scatter of the 8 gradient sums (16 v_add_f32_dpp instead of the all-reduce's 32) and
// kick + drift on ONE coordinate pair (3 v_pk_fma_f32 instead of 12): 27 instructions where there were 44, of ~190.
// On entry xk = k * position (all lanes), pm = momentum after the first half kick; on exit both are replicated again.
template <int R>
__device__ __forceinline__ void hmc_interior_rs16(const RegRowPairs<8, R, 16>& rows, const float (&d)[8], const float (&e)[8],
                                                  float step, int nsteps, float (&xk)[8], float (&pm)[8]) {
    const int q = (threadIdx.x >> 2) & 3;
    auto pick = [&](const float (&v)[8]) {
        const float x01 = q & 1 ? v[2] : v[0], x23 = q & 1 ? v[6] : v[4];
        const float y01 = q & 1 ? v[3] : v[1], y23 = q & 1 ? v[7] : v[5];
        return f32x2{q & 2 ? x23 : x01, q & 2 ? y23 : y01};
    };
    const f32x2 dq = pick(d), eq = pick(e), st = {step, step};
    f32x2 xq = pick(xk), pq = pick(pm);
    f32x2 bb[4];
    for (int i = 0; i < nsteps; ++i) {
        xq = __builtin_elementwise_fma(dq, pq, xq);  // drift
        group16_allgather_pairs(xq, bb);
        f32x2 gpp[4];
        float unused = 0.0f;
        row_pairs_eval<8, R, 16, false, true>(rows, bb, gpp, unused);
        const float gv[8] = {gpp[0].x, gpp[0].y, gpp[1].x, gpp[1].y, gpp[2].x, gpp[2].y, gpp[3].x, gpp[3].y};
        float u0, u1;
        group16_reduce_scatter8(gv, u0, u1);
        const f32x2 gq = __builtin_elementwise_fma(-xq, eq, f32x2{u0, u1});  // + prior:  g - (k q)(ivar / k)
        pq = __builtin_elementwise_fma(st, gq, pq);                          // kick
    }
    xq = __builtin_elementwise_fma(dq, pq, xq);  // the last drift (its gradient is the end-point evaluation)
    group16_allgather_pairs(xq, bb);
    f32x2 pp[4];
    group16_allgather_pairs(pq, pp);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        xk[2 * j] = bb[j].x;
        xk[2 * j + 1] = bb[j].y;
        pm[2 * j] = pp[j].x;
        pm[2 * j + 1] = pp[j].y;
    }
}

// --------------------------------------------------------------------------------------------
// MALA / RWMH for the 16-lanes-per-chain register kernel (float, P = 8) with the chain state DISTRIBUTED over the
// group for the whole launch (quad q owns coordinates 2q, 2q + 1), as hmc_interior_rs16 does inside a trajectory:
// proposal, drift terms, proposal-density difference and the accept select act on ONE coordinate pair per lane
// (12 packed ops instead of 45), the gradient is reduce-scattered (16 DPP adds instead of 32), the proposal is
// all-gathered for the row pass (8 row_share moves), scalars over the coordinates are summed across the quads with
// two symmetric DPP adds.  With two waves per SIMD (8192 chains) the saved instructions are saved time.
// Same Philox stream, same accept rule (fit-np-mala.py:61-78, fit-numpy.py:53-62) as k_chain.
template <int R, int KIND>
__global__ void __launch_bounds__(256) k_chain_rs16(ModelArgs<float, 8> m, ChainArgs<float, 8> a) {
    static_assert(KIND == KIND_MALA || KIND == KIND_RWMH, "threaded-ll kernels");
    constexpr int P = 8, G = 16;
    const int gl = threadIdx.x % G;
    int64_t chain = a.first + ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const bool live = chain < a.first + a.count;
    if (!live) chain = a.first + a.count - 1;
    const bool writer = live && gl == 0;
    RegRowPairs<P, R, G> rows;
    rows.load(m.rows, m.n, gl);
    const uint64_t gchain = (uint64_t)(a.chain_offset + chain);
    const int q = (threadIdx.x >> 2) & 3;
    auto pick = [&](const float (&v)[8]) {
        const float x01 = q & 1 ? v[2] : v[0], x23 = q & 1 ? v[6] : v[4];
        const float y01 = q & 1 ? v[3] : v[1], y23 = q & 1 ? v[7] : v[5];
        return f32x2{q & 2 ? x23 : x01, q & 2 ? y23 : y01};
    };
    const f32x2 aq = pick(a.a), bq = pick(a.b), cq = pick(a.c), ivq = pick(m.prior.inv_var);
    float x8[P];
#pragma unroll
    for (int j = 0; j < P; ++j) x8[j] = j < a.p ? a.state[chain * a.p + j] : 0.0f;
    f32x2 xq = pick(x8), gq = {0.0f, 0.0f};

    // ll, lprior (both replicated in the group) and -- GRAD -- the lpost gradient of the lane's pair, at the pair pq
    auto evaluate = [&](auto want_value, auto want_grad, const f32x2& pq, f32x2& grad_q, double& ll, double& lpr) {
        constexpr bool VALUE = decltype(want_value)::value, GRAD = decltype(want_grad)::value;
        f32x2 bb[4];
        group16_allgather_pairs(pq * f32x2{ExpScale<float>::k, ExpScale<float>::k}, bb);
        f32x2 gpp[4];
        float v = 0.0f;
        row_pairs_eval<P, R, G, VALUE, GRAD>(rows, bb, gpp, v);
        if constexpr (GRAD) {
            const float gv[8] = {gpp[0].x, gpp[0].y, gpp[1].x, gpp[1].y, gpp[2].x, gpp[2].y, gpp[3].x, gpp[3].y};
            float u0, u1;
            group16_reduce_scatter8(gv, u0, u1);
            grad_q = __builtin_elementwise_fma(-pq, ivq, f32x2{u0, u1});
        }
        if constexpr (VALUE) {
            ll = group_sum<G>((double)(v + rows.value_fixup()));
            const f32x2 sq = pq * pq * ivq;
            lpr = m.prior.lprior_const - 0.5 * (double)group16_quad_sum(sq.x + sq.y);
        }
    };
    using True = std::integral_constant<bool, true>;
    using False = std::integral_constant<bool, false>;

    double lp = a.lp_state[chain];
    uint32_t nacc = 0;
    if constexpr (KIND == KIND_MALA) {
        double d0, d1;
        evaluate(False{}, True{}, xq, gq, d0, d1);
    }
    DrawBatch<float, P, G> draws;
    draws.reset();
    LR_RS16_PHASES_BEGIN  // development builds: shader cycles per phase of an iteration (lr_stamps.h; profiles/r3_mala_phases.txt)
    for (int64_t it = 0; it < a.iters; ++it) {
        for (int64_t jt = 0; jt < a.thin; ++jt) {
            const uint64_t iter = (uint64_t)(a.iter_offset + it * a.thin + jt);
            float zx, zy, logu_f;
            LR_RS16_PHASE(3);
            draws.next_pair(a.seed, gchain, iter, gl, q, zx, zy, logu_f);
            LR_RS16_PHASE(0);
            const f32x2 zq = {zx, zy};
            const double logu = (double)logu_f;
            f32x2 xp, gp = {0.0f, 0.0f};
            double llp = 0, lprp = 0, logr;
            if constexpr (KIND == KIND_RWMH) {
                xp = __builtin_elementwise_fma(aq, zq, xq);  // prop = x + sd z                      fit-numpy.py:83-84
                evaluate(True{}, False{}, xp, gp, llp, lprp);
                logr = (llp + lprp) - lp;
            } else {
                const f32x2 advx = __builtin_elementwise_fma(aq, gq, xq);  // advance(x)             fit-np-mala.py:76
                xp = __builtin_elementwise_fma(bq, zq, advx);
                evaluate(True{}, True{}, xp, gp, llp, lprp);
                LR_RS16_PHASE(1);
                const f32x2 advp = __builtin_elementwise_fma(aq, gp, xp);
                const f32x2 d1 = xq - advp, d2 = xp - advx;
                const f32x2 t = cq * __builtin_elementwise_fma(-d2, d2, d1 * d1);
                logr = (llp + lprp) - lp - 0.5 * (double)group16_quad_sum(t.x + t.y);
            }
            const bool acc = logu < logr;  // NaN -> reject
            if (acc) {
                ++nacc;
                lp = llp + lprp;
            }
            xq = acc ? xp : xq;
            if constexpr (KIND == KIND_MALA) gq = acc ? gp : gq;
            LR_RS16_PHASE(2);
        }
        if ((a.out || a.stats.buf) && live) {  // group-uniform: the kept sample, gathered back into the writer lane
            f32x2 all[4];
            group16_allgather_pairs(xq, all);
            const float xs[P] = {all[0].x, all[0].y, all[1].x, all[1].y, all[2].x, all[2].y, all[3].x, all[3].y};
            if (writer) {
                if (a.out) {
                    float* o = a.out + (it * a.C + chain) * a.p;
#pragma unroll
                    for (int j = 0; j < P; ++j)
                        if (j < a.p) o[j] = xs[j];
                }
                if (a.stats.buf) stats_update<float, P>(a.stats, it, a.C, chain, a.p, xs);
            }
        }
    }
    {
        f32x2 all[4];
        group16_allgather_pairs(xq, all);
        const float xs[P] = {all[0].x, all[0].y, all[1].x, all[1].y, all[2].x, all[2].y, all[3].x, all[3].y};
        if (writer) {
#pragma unroll
            for (int j = 0; j < P; ++j)
                if (j < a.p) a.state[chain * a.p + j] = xs[j];
            if (a.accepts) a.accepts[chain] += nacc;
            a.lp_state[chain] = lp;
        }
    }
    LR_RS16_PHASES_REPORT(KIND, a.iters * a.thin)
}

// --------------------------------------------------------------------------------------------
// HMC on a FLOAT64 model (padded p = 8, n <= 16 R) with the l - 1 INTERIOR leapfrog gradients in float32 (LR_MODE_MIXED; the
// float64 model's form of the interior-precision policy, include/logreg_hip.h LR_PREC_*).  The reference computes in float64
// (fit-np-hmc.py:65-87): here the Metropolis test, both end-point evaluations of log-posterior and gradient, the half kicks,
// the kinetic energies AND the trajectory's position and momentum are float64 -- rows in LDS, as the float64 LDS variant has
// them -- while the force inside the trajectory is the float32 register kernel's row pass (hmc_interior_rs16's: twisted row
// pairs of the rows rounded to float32, v_pk_fma_f32, reduce-scatter) applied to the float64 position rounded to float32.  A
// force that is a deterministic function of the position keeps drift and kick shears of (q, p) in float64 arithmetic, exactly as
// reversible and volume-preserving as the reference's own leapfrog; only the trajectory's energy error sees the rounding.
// One float64 evaluation (~5.5 float32 ones) + l - 1 float32 ones per iteration instead of l float64 ones.
constexpr int kMixedStashDoubles = 16;
template <int R>
__global__ void __launch_bounds__(256) k_chain_mixed(ModelArgs<double, 8> m, ChainArgs<double, 8> a) {
    constexpr int P = 8, G = 16;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int gl = threadIdx.x % G;
    int64_t chain = a.first + ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const bool live = chain < a.first + a.count;
    if (!live) chain = a.first + a.count - 1;
    const bool writer = live && gl == 0;
    const auto rows = make_rows<double, P, G, MODE_LDS, 0>(m, gl, reinterpret_cast<double*>(smem_raw));  // float64 rows: end points
    RegRowPairs<P, R, G> rows32;                                                                          // float32 rows: interior
    rows32.load(m.rows, m.n, gl);
    // (dynamic LDS: the float64 rows, then kMixedStashDoubles doubles per lane -- lr_plan.h plan_mixed_hmc sizes it)
    double* const stash = reinterpret_cast<double*>(smem_raw) + m.n * P + threadIdx.x;
    const uint64_t gchain = (uint64_t)(a.chain_offset + chain);

    // The chain's state is DISTRIBUTED over its 16 lanes for the whole launch, as in k_chain_rs16: quad qd owns coordinates 2 qd,
    // 2 qd + 1 of position, gradient and momentum (float64 pairs); only the end-point evaluation sees all 8 coordinates.
    const int qd = (threadIdx.x >> 2) & 3;
    auto pick = [&](const double (&v)[8], double& lo, double& hi) {
        const double x01 = qd & 1 ? v[2] : v[0], x23 = qd & 1 ? v[6] : v[4];
        const double y01 = qd & 1 ? v[3] : v[1], y23 = qd & 1 ? v[7] : v[5];
        lo = qd & 2 ? x23 : x01;
        hi = qd & 2 ? y23 : y01;
    };
    auto gather = [&](double lo, double hi, double (&v)[8]) {  // lane 4 j of the 16-lane row holds coordinates 2 j, 2 j + 1
        v[0] = dpp_mov<0x150>(lo); v[1] = dpp_mov<0x150>(hi); v[2] = dpp_mov<0x154>(lo); v[3] = dpp_mov<0x154>(hi);
        v[4] = dpp_mov<0x158>(lo); v[5] = dpp_mov<0x158>(hi); v[6] = dpp_mov<0x15C>(lo); v[7] = dpp_mov<0x15C>(hi);
    };
    auto quad_sum = [&](double v) {  // over the four quads of a value identical inside each quad; bit-identical in all 16 lanes
        v += dpp_mov<0x141>(v);
        v += dpp_mov<0x140>(v);
        return v;
    };
    double a0, a1, b0, b1, c0, c1, e0, e1, x0, x1, g0, g1;
    pick(a.a, a0, a1);  // sqrt(dmm)
    pick(a.b, b0, b1);  // eps / dmm
    pick(a.c, c0, c1);  // 1 / dmm
    pick(m.prior.inv_var, e0, e1);
    double lp;
    uint32_t nacc = 0;
    {
        double x8[P], g8[P], ll0 = 0, lpr0 = 0;
#pragma unroll
        for (int j = 0; j < P; ++j) x8[j] = j < a.p ? a.state[chain * a.p + j] : 0.0;
        eval_lpost<double, P, G, true, true>(rows, m.prior, x8, g8, ll0, lpr0);
        lp = ll0 + lpr0;
        pick(x8, x0, x1);
        pick(g8, g0, g1);
    }
    const double heps = 0.5 * a.step, step = a.step;
    constexpr float kf = ExpScale<float>::k;
    DrawBatch<double, P, G> draws;
    draws.reset();
    for (int64_t it = 0; it < a.iters; ++it) {
        for (int64_t jt = 0; jt < a.thin; ++jt) {
            const uint64_t iter = (uint64_t)(a.iter_offset + it * a.thin + jt);
            double zx, zy, logu;
            draws.next_pair(a.seed, gchain, iter, gl, qd, zx, zy, logu);
            // p ~ N(0, dmm); leapfrog l steps; a = alpi(prop) - alpi(x)     fit-np-hmc.py:65-87
            double p0 = zx * a0, p1 = zy * a1;
            const double k0_ = quad_sum(__builtin_fma(c0 * p0, p0, c1 * p1 * p1));
            p0 = __builtin_fma(heps, g0, p0);
            p1 = __builtin_fma(heps, g1, p1);
            double xp0 = x0, xp1 = x1;
            // Everything the trajectory does not touch waits in LDS (16 doubles per lane): with it in registers the interior loop
            // enters with 62 VGPRs it never uses (tools/isa_liveness.py) and its breadth-first row pass wants ~180 of the 256.
            stash[0 * 256] = x0, stash[1 * 256] = x1, stash[2 * 256] = g0, stash[3 * 256] = g1, stash[4 * 256] = lp, stash[5 * 256] = k0_;
            stash[6 * 256] = logu, stash[7 * 256] = a0, stash[8 * 256] = a1, stash[9 * 256] = c0, stash[10 * 256] = c1;
            stash[11 * 256] = draws.mine[0], stash[12 * 256] = draws.mine[1], stash[13 * 256] = draws.mine[2], stash[14 * 256] = draws.mine[3];
            stash[15 * 256] = draws.lu;
            asm volatile("" ::: "memory");  // (no store-to-load forwarding across the loop)
            for (int i = 0; i < a.l - 1; ++i) {
                xp0 = __builtin_fma(b0, p0, xp0);  // drift (float64)
                xp1 = __builtin_fma(b1, p1, xp1);
                f32x2 bb[4];
                group16_allgather_pairs(f32x2{(float)xp0, (float)xp1} * f32x2{kf, kf}, bb);
                f32x2 gpp[4];
                row_pairs_grad_bf<P, R, G>(rows32, bb, gpp);
                const float gv[8] = {gpp[0].x, gpp[0].y, gpp[1].x, gpp[1].y, gpp[2].x, gpp[2].y, gpp[3].x, gpp[3].y};
                float u0, u1;
                group16_reduce_scatter8(gv, u0, u1);
                p0 = __builtin_fma(step, __builtin_fma(-xp0, e0, (double)u0), p0);  // kick (float64): float32 likelihood force + prior
                p1 = __builtin_fma(step, __builtin_fma(-xp1, e1, (double)u1), p1);
            }
            asm volatile("" ::: "memory");
            x0 = stash[0 * 256], x1 = stash[1 * 256], g0 = stash[2 * 256], g1 = stash[3 * 256], lp = stash[4 * 256];
            const double k0 = stash[5 * 256];
            logu = stash[6 * 256], a0 = stash[7 * 256], a1 = stash[8 * 256], c0 = stash[9 * 256], c1 = stash[10 * 256];
            draws.mine[0] = stash[11 * 256], draws.mine[1] = stash[12 * 256], draws.mine[2] = stash[13 * 256], draws.mine[3] = stash[14 * 256];
            draws.lu = stash[15 * 256];
            xp0 = __builtin_fma(b0, p0, xp0);  // the last drift: its gradient is the end-point evaluation, in float64
            xp1 = __builtin_fma(b1, p1, xp1);
            double gp0, gp1, llp = 0, lprp = 0;
            {
                double xp8[P], gp8[P];
                gather(xp0, xp1, xp8);
                eval_lpost<double, P, G, true, true>(rows, m.prior, xp8, gp8, llp, lprp);
                pick(gp8, gp0, gp1);
            }
            p0 = __builtin_fma(heps, gp0, p0);
            p1 = __builtin_fma(heps, gp1, p1);
            const double k1 = quad_sum(__builtin_fma(c0 * p0, p0, c1 * p1 * p1));
            const double logr = ((llp + lprp) - lp) - 0.5 * (k1 - k0);
            const bool acc = logu < logr;  // NaN -> reject, as `np.log(np.random.rand()) < a`
            if (acc) {
                ++nacc;
                lp = llp + lprp;
            }
            x0 = acc ? xp0 : x0;
            x1 = acc ? xp1 : x1;
            g0 = acc ? gp0 : g0;
            g1 = acc ? gp1 : g1;
        }
        if ((a.out || a.stats.buf) && live) {  // group-uniform: the kept sample, gathered back into the writer lane
            double xs[P];
            gather(x0, x1, xs);
            if (writer) {
                if (a.out) {
                    double* o = a.out + (it * a.C + chain) * a.p;
#pragma unroll
                    for (int j = 0; j < P; ++j)
                        if (j < a.p) o[j] = xs[j];
                }
                if (a.stats.buf) stats_update<double, P>(a.stats, it, a.C, chain, a.p, xs);
            }
        }
    }
    {
        double xs[P];
        gather(x0, x1, xs);
        if (writer) {
#pragma unroll
            for (int j = 0; j < P; ++j)
                if (j < a.p) a.state[chain * a.p + j] = xs[j];
            if (a.accepts) a.accepts[chain] += nacc;
        }
    }
}

// The same policy on 32 / 64 lanes per chain (few chains: a wave alone on its SIMD runs 7 / 4 rows per lane instead of 13), with
// the float64 state REPLICATED in the group's lanes as k_chain has it: all-reduce of the float32 gradient, drift and kick on all 8
// coordinates in every lane.
// Also the form for the other padded widths (p <= 4, 9 <= p <= 32) and for p <= 8 beyond 256 rows, at any lanes per chain.
template <int P, int G, int R>
__global__ void __launch_bounds__(256) k_chain_mixed_rep(ModelArgs<double, P> m, ChainArgs<double, P> a) {
    static_assert(G == 16 || G == 32 || G == 64, "lanes per chain");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int gl = threadIdx.x % G;
    int64_t chain = a.first + ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const bool live = chain < a.first + a.count;
    if (!live) chain = a.first + a.count - 1;
    const bool writer = live && gl == 0;
    const auto rows = make_rows<double, P, G, MODE_LDS, 0>(m, gl, reinterpret_cast<double*>(smem_raw));
    RegRowPairs<P, R, G> rows32;
    rows32.load(m.rows, m.n, gl);
    const uint64_t gchain = (uint64_t)(a.chain_offset + chain);
    double x[P], g[P];
#pragma unroll
    for (int j = 0; j < P; ++j) x[j] = j < a.p ? a.state[chain * a.p + j] : 0.0;
    double lp;
    uint32_t nacc = 0;
    {
        double ll0 = 0, lpr0 = 0;
        eval_lpost<double, P, G, true, true>(rows, m.prior, x, g, ll0, lpr0);
        lp = ll0 + lpr0;
    }
    const double heps = 0.5 * a.step;
    constexpr float kf = ExpScale<float>::k;
    DrawBatch<double, P, G> draws;
    draws.reset();
    for (int64_t it = 0; it < a.iters; ++it) {
        for (int64_t jt = 0; jt < a.thin; ++jt) {
            const uint64_t iter = (uint64_t)(a.iter_offset + it * a.thin + jt);
            double z[P], logu;
            if constexpr (DrawBatch<double, P, G>::kEnabled) draws.next(a.seed, gchain, iter, gl, z, logu);
            else draw_group<double, P, G>(a.seed, gchain, iter, gl, z, logu);
            double pm[P], xp[P], gp[P];
#pragma unroll
            for (int j = 0; j < P; ++j) {
                pm[j] = z[j] * a.a[j];
                xp[j] = x[j];
            }
            const double k0 = vquad<double, P>(a.c, pm);
            vfma_s<double, P>(heps, g, pm);
            for (int i = 0; i < a.l - 1; ++i) {
                vfma_v<double, P>(a.b, pm, xp);  // drift (float64)
                f32x2 bb[P / 2], gpp[P / 2];
#pragma unroll
                for (int j = 0; j < P / 2; ++j) bb[j] = f32x2{(float)xp[2 * j], (float)xp[2 * j + 1]} * f32x2{kf, kf};
                row_pairs_grad_bf<P, R, G>(rows32, bb, gpp);
                float gv[P];
#pragma unroll
                for (int j = 0; j < P / 2; ++j) gv[2 * j] = gpp[j].x, gv[2 * j + 1] = gpp[j].y;
                group_sum_levels<G, P>(gv);
#pragma unroll
                for (int j = 0; j < P; ++j) pm[j] = __builtin_fma(a.step, __builtin_fma(-xp[j], m.prior.inv_var[j], (double)gv[j]), pm[j]);  // kick
            }
            vfma_v<double, P>(a.b, pm, xp);  // the last drift: its gradient is the end-point evaluation, in float64
            double llp = 0, lprp = 0;
            eval_lpost<double, P, G, true, true>(rows, m.prior, xp, gp, llp, lprp);
            vfma_s<double, P>(heps, gp, pm);
            const double k1 = vquad<double, P>(a.c, pm);
            const double logr = ((llp + lprp) - lp) - 0.5 * (k1 - k0);
            const bool acc = logu < logr;  // NaN -> reject
            if (acc) {
                ++nacc;
                lp = llp + lprp;
            }
#pragma unroll
            for (int j = 0; j < P; ++j) {
                x[j] = acc ? xp[j] : x[j];
                g[j] = acc ? gp[j] : g[j];
            }
        }
        if (a.out && writer) {
            double* o = a.out + (it * a.C + chain) * a.p;
#pragma unroll
            for (int j = 0; j < P; ++j)
                if (j < a.p) o[j] = x[j];
        }
        if (a.stats.buf && writer) stats_update<double, P>(a.stats, it, a.C, chain, a.p, x);
    }
    if (writer) {
#pragma unroll
        for (int j = 0; j < P; ++j)
            if (j < a.p) a.state[chain * a.p + j] = x[j];
        if (a.accepts) a.accepts[chain] += nacc;
    }
}

// --------------------------------------------------------------------------------------------
// the chain kernel
template <typename T, int P, int G, int MODE, int R, int KIND>
__global__ void __launch_bounds__(256) k_chain(ModelArgs<T, P> m, ChainArgs<T, P> a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int gl = threadIdx.x % G;
    int64_t chain = a.first + ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const bool live = chain < a.first + a.count;
    if (!live) chain = a.first + a.count - 1;  // whole waves stay converged for the DPP reductions; stores are masked
    const bool writer = live && gl == 0;
    const auto rows = make_rows<T, P, G, MODE, R>(m, gl, reinterpret_cast<T*>(smem_raw));
    const uint64_t gchain = (uint64_t)(a.chain_offset + chain);

    T x[P];
#pragma unroll
    for (int j = 0; j < P; ++j) x[j] = j < a.p ? a.state[chain * a.p + j] : T(0);

    Prior<T, P> prior_k;  // HMC: inv_var / ExpScale<T>::k, for positions carried as k * q
#pragma unroll
    for (int j = 0; j < P; ++j) prior_k.inv_var[j] = a.e[j];
    prior_k.lprior_const = 0.0;

    T g[P];        // gradient at x (MALA / HMC / UL)
    double lp;     // log-density attached to x: threaded value (RWMH/MALA) or lpost(x) (HMC)
    uint32_t nacc = 0;
    {
        double ll0 = 0, lpr0 = 0;
        if constexpr (KIND == KIND_HMC) {
            eval_lpost<T, P, G, true, true>(rows, m.prior, x, g, ll0, lpr0);
            lp = ll0 + lpr0;
        } else if constexpr (KIND == KIND_MALA) {
            eval_lpost<T, P, G, false, true>(rows, m.prior, x, g, ll0, lpr0);
            lp = a.lp_state[chain];
        } else if constexpr (KIND == KIND_UL) {
            eval_lpost<T, P, G, false, true>(rows, m.prior, x, g, ll0, lpr0);
            lp = 0;
        } else {
            lp = a.lp_state[chain];
        }
    }

    DrawBatch<T, P, G> draws;  // randomness of several consecutive iterations per generator pass (lr_device.h)
    draws.reset();
    for (int64_t it = 0; it < a.iters; ++it) {
        for (int64_t jt = 0; jt < a.thin; ++jt) {
            const uint64_t iter = (uint64_t)(a.iter_offset + it * a.thin + jt);
            T z[P];
            T logu_t;
            if constexpr (DrawBatch<T, P, G>::kEnabled) draws.next(a.seed, gchain, iter, gl, z, logu_t);
            else draw_group<T, P, G>(a.seed, gchain, iter, gl, z, logu_t);

            if constexpr (KIND == KIND_UL) {
                // x <- x + 0.5*pre*dt*glp(x) + sqrt(pre*dt)*z            fit-np-ul.py:65-67
#pragma unroll
                for (int j = 0; j < P; ++j) x[j] = fma_t(a.b[j], z[j], fma_t(a.a[j], g[j], x[j]));
                double d0, d1;
                eval_lpost<T, P, G, false, true>(rows, m.prior, x, g, d0, d1);
                ++nacc;
            } else {
                const double logu = (double)logu_t;
                T xp[P], gp[P];
                double llp = 0, lprp = 0, logr;
                if constexpr (KIND == KIND_RWMH) {
                    // prop = x + sd*z; a = lpost(prop) - ll                  fit-numpy.py:53-62,83-84
#pragma unroll
                    for (int j = 0; j < P; ++j) xp[j] = fma_t(a.a[j], z[j], x[j]);
                    eval_lpost<T, P, G, true, false>(rows, m.prior, xp, gp, llp, lprp);
                    logr = (llp + lprp) - lp;
                } else if constexpr (KIND == KIND_MALA) {
                    // prop = advance(x) + sqrt(pre*dt) z ; advance(x) = x + 0.5*pre*dt*glp(x)
                    // a = lp' - ll + dprop(x,prop) - dprop(prop,x)           fit-np-mala.py:61-78
                    T advx[P], advp[P];
                    vfma_o<T, P>(a.a, g, x, advx);
                    vfma_o<T, P>(a.b, z, advx, xp);
                    eval_lpost<T, P, G, true, true>(rows, m.prior, xp, gp, llp, lprp);
                    vfma_o<T, P>(a.a, gp, xp, advp);
                    // dprop(x, prop) - dprop(prop, x): (x - advance(prop))^2 - (prop - advance(x))^2, weighted 1/(pre dt)
                    const T dq = vdiffsq<T, P>(a.c, x, advp, xp, advx);
                    logr = (llp + lprp) - lp - 0.5 * (double)dq;
                } else {  // HMC
                    // p ~ N(0, dmm); leapfrog l steps; a = alpi(prop) - alpi(x)     fit-np-hmc.py:65-87
                    T pm[P];
#pragma unroll
                    for (int j = 0; j < P; ++j) {
                        pm[j] = z[j] * a.a[j];
                        xp[j] = x[j];
                        gp[j] = g[j];
                    }
                    const T k0 = vquad<T, P>(a.c, pm);
                    const T heps = T(0.5) * a.step;
                    vfma_s<T, P>(heps, gp, pm);
                    T xk[P];  // k * position
                    vscale<T, P>(ExpScale<T>::k, xp, xk);
                    if constexpr (G == 16 && P == 8 && sizeof(T) == 4 && MODE == MODE_REG) {
                        hmc_interior_rs16<R>(rows, a.d, a.e, a.step, a.l - 1, xk, pm);  // includes the last drift
                    } else {
                        for (int i = 0; i < a.l - 1; ++i) {
                            vfma_v<T, P>(a.d, pm, xk);  // drift
                            double d0, d1;
                            eval_lpost<T, P, G, false, true, true>(rows, prior_k, xk, gp, d0, d1);
                            vfma_s<T, P>(a.step, gp, pm);  // kick
                        }
                        vfma_v<T, P>(a.d, pm, xk);
                    }
                    vscale<T, P>(ExpScale<T>::inv, xk, xp);
                    eval_lpost<T, P, G, true, true>(rows, m.prior, xp, gp, llp, lprp);
                    vfma_s<T, P>(heps, gp, pm);
                    const T k1 = vquad<T, P>(a.c, pm);
                    logr = ((llp + lprp) - lp) - 0.5 * ((double)k1 - (double)k0);
                }
                const bool acc = logu < logr;  // NaN -> reject, as `np.log(np.random.rand()) < a`
                if (acc) {
                    ++nacc;
                    lp = llp + lprp;
                }
#pragma unroll
                for (int j = 0; j < P; ++j) {
                    x[j] = acc ? xp[j] : x[j];
                    if constexpr (KIND != KIND_RWMH) g[j] = acc ? gp[j] : g[j];
                }
            }
        }
        if (a.out && writer) {
            T* o = a.out + (it * a.C + chain) * a.p;
#pragma unroll
            for (int j = 0; j < P; ++j)
                if (j < a.p) o[j] = x[j];
        }
        if (a.stats.buf && writer) stats_update<T, P>(a.stats, it, a.C, chain, a.p, x);
    }
    if (writer) {
#pragma unroll
        for (int j = 0; j < P; ++j)
            if (j < a.p) a.state[chain * a.p + j] = x[j];
        if (a.accepts) a.accepts[chain] += nacc;
        if constexpr (KIND == KIND_RWMH || KIND == KIND_MALA) a.lp_state[chain] = lp;
    }
}

}  // namespace lr
