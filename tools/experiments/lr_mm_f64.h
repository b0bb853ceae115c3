// lr_mm_f64.h -- HMC on a FLOAT64 model (padded p = 8) with EVERY evaluation in float64 -- the arithmetic the reference computes in
// (Python/fit-np-hmc.py:17-19, 44-47, 65-87) -- on the float64 MATRIX pipe, for chain counts around one chain tile per CU (the
// BASELINE configuration: 4096 chains on 256 CUs).  Round 6; replaces k_chain<double, 8, 16 lanes per chain, rows in LDS> there.
//
// Why: the lane-group kernel spends 47 vector instructions per (row, chain) -- 16 of them the two contractions, 4 LDS reads -- plus a
// 96-instruction butterfly per evaluation (no DPP form of v_add_f64), one wave per SIMD at 5.5 cycles per instruction: 0.21-0.22 of
// the float64 vector peak.  Here the contractions run as v_mfma_f64_16x16x4_f64 (eta of a 16-row x 16-chain tile: 2 MFMAs; its
// gradient contribution: 4), whose operand layout hands every lane exactly the coordinates it owns, so there is NO reduction and NO
// gather inside a wave, the rows live in registers as MFMA operands for the whole launch (no LDS reads in the row loop), and the vector
// ALU is left with the sigmoid alone: 27 instructions per (row, chain).
//
// Layout.  One workgroup = ONE chain tile of 16 chains = W waves (W = blockDim.x / 64: 4, 8 or 16); the ceil(n / 16) row tiles are
// dealt round-robin to the waves (wave w: tiles w, w + W, ... -- at most T per wave, T a template parameter), so 4096 chains x 13 row
// tiles of Pima-sized data put 3-4 waves on every SIMD.  Lane l = (c, k) = (l & 15, l >> 4) of every wave carries coordinates k and
// k + 4 of chain c: position, momentum, gradient, draws -- the whole chain state is REPLICATED over the W waves (each repeats the
// 2-coordinate updates and the generator; the row work is what is split).
//   eta tile   E[16 rows x 16 chains] = sum_h A_h (16 x 4) . B_h (4 x 16)
//                A_h: lane (c, k) = xs[row 16 t + c][4 h + k]  (register, loaded once)   B_h: lane (c, k) = q[chain c][4 h + k] = its OWN coordinate
//                D  : register r of lane (c, k) = E[row 16 t + 4 r + k][chain c]                       (the f64 16x16x4 layout: rows interleaved by 4)
//   weights    w_r = sigma(-E_r) in place: 4 per lane and tile (interior steps: rcp(1 + exp(t)), as row_term FASTW; end points: the
//              sign-symmetric form with the value term, as row_term)
//   gradient   G[16 coords x 16 chains] += sum_r A'_r (16 x 4) . B'_r (4 x 16)
//                A'_r: lane (c, k) = xs[row 16 t + 4 r + k][coord c] (0 for c >= 8)     B'_r: lane (c, k) = w_r                (no shuffle)
//                D   : register 0 / 1 of lane (c, k) = G[coord k / k + 4][chain c] -- the coordinates the lane owns; registers 2, 3: padding
//   exchange   the W partial gradients (2 doubles per lane) through LDS, double-buffered, ONE barrier per evaluation, summed in wave
//              order by every wave: bit-identical states in all W waves, so their Metropolis decisions agree.
// Summation order: per output element the matrix pipe's fma chain over K within a tile, tiles of a wave in order, waves in order -- a
// function of (n, W) only: chunked launches and shards of a planned run reproduce the whole run bit for bit (W is part of the plan).
#pragma once
#include <type_traits>

#include "lr_kernels.h"
#include "lr_mfma.h"

namespace lr {

typedef double f64x4 __attribute__((ext_vector_type(4)));

// sigma(-t) for HMC's interior force (row_term's FASTW form: the clamp keeps exp finite, 1 + e <= 1e304 needs no guard in the reciprocal)
__device__ __forceinline__ double mm_weight_fast(double ts) {
    const double s1 = 1.0 + exp_noguard(__builtin_fmin(__builtin_fmax(ts, -750.0), 700.0));
    double r = __builtin_amdgcn_rcp(s1);
    double err = __builtin_fma(-s1, r, 1.0);
    r = __builtin_fma(r, err, r);
    err = __builtin_fma(-s1, r, 1.0);
    return __builtin_fma(r, err, r);
}
// sigma(-t) and log sigma(t) from ONE exponential (row_term's float64 value + gradient form; the clamp is a select so that NaN stays NaN)
__device__ __forceinline__ double mm_weight_value(double ts, double& lv) {
    const double na = -__builtin_fabs(ts);
    const double e = exp_noguard(na < -750.0 ? -750.0 : na);
    const double s1 = 1.0 + e;
    double r = __builtin_amdgcn_rcp(s1);
    double err = __builtin_fma(-s1, r, 1.0);
    r = __builtin_fma(r, err, r);
    err = __builtin_fma(-s1, r, 1.0);
    r = __builtin_fma(r, err, r);
    lv = (ts >= 0.0 ? 0.0 : ts) - log1p_unit(e);  // (NaN: the comparison is false, ts goes through)
    return ts > 0.0 ? e * r : r;
}

// T: row tiles per wave; WMAX: the most waves per workgroup the instance is launched with (its register budget: 16 waves = 4 per SIMD
// leave 128 registers per lane -- one tile per wave; 8 waves leave 256)
template <int T, int WMAX>
__global__ void __launch_bounds__(64 * WMAX) k_chain_mm_f64(ModelArgs<double, 8> m, ChainArgs<double, 8> a) {
    constexpr int P = 8, NC = 2;
    __shared__ __attribute__((aligned(16))) double red[2][WMAX][64][NC];  // partial gradients of the waves, double-buffered
    __shared__ double redv[WMAX][64];                                     // partial log-likelihood values (end points only)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int W = (int)(blockDim.x >> 6);
    const int c = lane & 15, k = lane >> 4;
    int64_t chain = a.first + (int64_t)blockIdx.x * 16 + c;
    const bool live = chain < a.first + a.count;
    if (!live) chain = a.first + a.count - 1;
    const bool writer = live && wave == 0;
    const uint64_t gchain = (uint64_t)(a.chain_offset + chain);

    // the wave's row tiles as MFMA operands, in registers for the whole launch (rows beyond n, and tiles beyond the data, are zeros: a
    // zero row has eta = 0, weight 1/2 and no gradient contribution; its value term is masked below)
    const int64_t ntiles = (m.n + 15) / 16;
    double ae[T][NC], ag[T][4];
    int nt = 0;
#pragma unroll
    for (int i = 0; i < T; ++i) {
        const int64_t t = wave + (int64_t)i * W;
        if (t < ntiles) nt = i + 1;
#pragma unroll
        for (int h = 0; h < NC; ++h) {
            const int64_t row = 16 * t + c;
            ae[i][h] = t < ntiles && row < m.n ? m.rows[row * P + 4 * h + k] : 0.0;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t row = 16 * t + 4 * r + k;
            ag[i][r] = t < ntiles && row < m.n && c < P ? m.rows[row * P + c] : 0.0;
        }
    }
    nt = __builtin_amdgcn_readfirstlane(nt);

    auto pick = [&](const double (&v)[P], int h) {  // coordinate k + 4 h
        const double lo = k == 0 ? v[4 * h] : v[4 * h + 1], hi = k == 2 ? v[4 * h + 2] : v[4 * h + 3];
        return k < 2 ? lo : hi;
    };
    auto from = [&](int src_lane, double v) { return DrawBatch<double, P, 16>::fetch(v, src_lane * 4); };
    double inv_var[NC], ka[NC], kb[NC], kc[NC], x[NC], g[NC];
#pragma unroll
    for (int h = 0; h < NC; ++h) {
        inv_var[h] = pick(m.prior.inv_var, h);
        ka[h] = pick(a.a, h);
        kb[h] = pick(a.b, h);
        kc[h] = pick(a.c, h);
        const int j = k + 4 * h;
        x[h] = j < a.p ? a.state[chain * a.p + j] : 0.0;
    }

    int par = 0;
    // gradient of the log-posterior for the lane's own coordinates at q (distributed as the state is); VALUE: the log-likelihood too
    auto evaluate = [&](auto want_value, const double (&q)[NC], double (&grad)[NC], double& ll) {
        constexpr bool VALUE = decltype(want_value)::value;
        f64x4 ga = {0, 0, 0, 0}, gb = {0, 0, 0, 0};  // two accumulators: two independent MFMA chains
        double vsum = 0.0;
        f64x4 e[T];
#pragma unroll
        for (int i = 0; i < T; ++i) {  // every tile's eta first: the matrix pipe works ahead of the sigmoids
            e[i] = f64x4{0, 0, 0, 0};
            if (i < nt) {
#pragma unroll
                for (int h = 0; h < NC; ++h) e[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(ae[i][h], q[h], e[i], 0, 0, 0);
            }
        }
#pragma unroll
        for (int i = 0; i < T; ++i) {
            if (i < nt) {
                double w[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if constexpr (VALUE) {
                        double lv;
                        w[r] = mm_weight_value(e[i][r], lv);
                        const int64_t row = 16 * (wave + (int64_t)i * W) + 4 * r + k;
                        if (row < m.n) vsum += lv;
                    } else {
                        w[r] = mm_weight_fast(e[i][r]);
                    }
                }
                ga = __builtin_amdgcn_mfma_f64_16x16x4f64(ag[i][0], w[0], ga, 0, 0, 0);
                gb = __builtin_amdgcn_mfma_f64_16x16x4f64(ag[i][1], w[1], gb, 0, 0, 0);
                ga = __builtin_amdgcn_mfma_f64_16x16x4f64(ag[i][2], w[2], ga, 0, 0, 0);
                gb = __builtin_amdgcn_mfma_f64_16x16x4f64(ag[i][3], w[3], gb, 0, 0, 0);
            }
        }
        red[par][wave][lane][0] = ga[0] + gb[0];
        red[par][wave][lane][1] = ga[1] + gb[1];
        if constexpr (VALUE) redv[wave][lane] = vsum;
        __syncthreads();
        double g0 = red[par][0][lane][0], g1 = red[par][0][lane][1];
        for (int w = 1; w < W; ++w) {
            g0 += red[par][w][lane][0];
            g1 += red[par][w][lane][1];
        }
        par ^= 1;
        if constexpr (VALUE) {
            double v = redv[0][lane];
            for (int w = 1; w < W; ++w) v += redv[w][lane];
            ll = ksum(v);
            __syncthreads();  // redv is single-buffered; value passes are the two end points of a trajectory
        }
        grad[0] = __builtin_fma(-q[0], inv_var[0], g0);
        grad[1] = __builtin_fma(-q[1], inv_var[1], g1);
    };
    using True = std::integral_constant<bool, true>;
    using False = std::integral_constant<bool, false>;
    auto lprior_of = [&](const double (&q)[NC]) {
        return m.prior.lprior_const - 0.5 * ksum(__builtin_fma(q[0] * q[0], inv_var[0], q[1] * q[1] * inv_var[1]));
    };

    double lp;
    uint32_t nacc = 0;
    {
        double ll0 = 0;
        evaluate(True{}, x, g, ll0);
        lp = ll0 + lprior_of(x);
    }
    const double heps = 0.5 * a.step;
    for (int64_t it = 0; it < a.iters; ++it) {
        for (int64_t jt = 0; jt < a.thin; ++jt) {
            const uint64_t iter = (uint64_t)(a.iter_offset + it * a.thin + jt);
            // lane (c, k): Philox block k of the iteration (k = 0, 1: the normals of coordinates 4 k .. 4 k + 3; k = 2, 3: the accept
            // uniform), as k_chain_mfma_f64 deals them; coordinate k + 4 h = element k of block h.  Every wave draws the same numbers.
            double z[NC], logu;
            {
                const U4 w4 = philox4x32_10((uint32_t)gchain, (uint32_t)iter, (uint32_t)(iter >> 32), k < 2 ? (uint32_t)k : TAG_UNIFORM,
                                            (uint32_t)a.seed, (uint32_t)(a.seed >> 32));
                double nrm[4];
                box_muller(w4.x, w4.y, nrm[0], nrm[1]);
                box_muller(w4.z, w4.w, nrm[2], nrm[3]);
                logu = from(c + 32, log(u01<double>(w4.x)));
#pragma unroll
                for (int h = 0; h < NC; ++h) {
                    const double e0 = from(c + 16 * h, nrm[0]), e1 = from(c + 16 * h, nrm[1]);
                    const double e2 = from(c + 16 * h, nrm[2]), e3 = from(c + 16 * h, nrm[3]);
                    z[h] = k < 2 ? (k == 0 ? e0 : e1) : (k == 2 ? e2 : e3);
                }
            }
            // p ~ N(0, dmm); leapfrog l steps; a = alpi(prop) - alpi(x)     fit-np-hmc.py:65-87
            double pm[NC], xp[NC], gp[NC], k0 = 0.0;
#pragma unroll
            for (int h = 0; h < NC; ++h) {
                pm[h] = z[h] * ka[h];
                k0 = __builtin_fma(pm[h] * pm[h], kc[h], k0);
                xp[h] = x[h];
                pm[h] = __builtin_fma(heps, g[h], pm[h]);
            }
            for (int i = 0; i < a.l - 1; ++i) {
#pragma unroll
                for (int h = 0; h < NC; ++h) xp[h] = __builtin_fma(kb[h], pm[h], xp[h]);  // drift
                double d0;
                evaluate(False{}, xp, gp, d0);
#pragma unroll
                for (int h = 0; h < NC; ++h) pm[h] = __builtin_fma(a.step, gp[h], pm[h]);  // kick
            }
#pragma unroll
            for (int h = 0; h < NC; ++h) xp[h] = __builtin_fma(kb[h], pm[h], xp[h]);  // the last drift
            double llp = 0;
            evaluate(True{}, xp, gp, llp);
            const double lprp = lprior_of(xp);
            double k1 = 0.0;
#pragma unroll
            for (int h = 0; h < NC; ++h) {
                pm[h] = __builtin_fma(heps, gp[h], pm[h]);
                k1 = __builtin_fma(pm[h] * pm[h], kc[h], k1);
            }
            const double logr = ((llp + lprp) - lp) - 0.5 * ksum(k1 - k0);
            const bool acc = logu < logr;  // NaN -> reject
            if (acc) {
                ++nacc;
                lp = llp + lprp;
            }
#pragma unroll
            for (int h = 0; h < NC; ++h) {
                x[h] = acc ? xp[h] : x[h];
                g[h] = acc ? gp[h] : g[h];
            }
        }
        if (a.out && writer) {
            double* o = a.out + (it * a.C + chain) * a.p;
#pragma unroll
            for (int h = 0; h < NC; ++h)
                if (k + 4 * h < a.p) o[k + 4 * h] = x[h];
        }
        if (a.stats.buf && writer) {  // the lane owns coordinates k + 4 h
            const int64_t idx = a.stats.first + it, sb = idx / a.stats.batch, sk = idx - sb * a.stats.batch;
            const double inv = 1.0 / (double)(sk + 1);
            double* s = a.stats.buf + ((sb * a.C + chain) * 2) * a.p;
#pragma unroll
            for (int h = 0; h < NC; ++h)
                if (k + 4 * h < a.p) stats_fold(s + k + 4 * h, s + a.p + k + 4 * h, sk, inv, x[h]);
        }
    }
    if (writer) {
#pragma unroll
        for (int h = 0; h < NC; ++h)
            if (k + 4 * h < a.p) a.state[chain * a.p + k + 4 * h] = x[h];
        if (k == 0 && a.accepts) a.accepts[chain] += nacc;
    }
}

}  // namespace lr
