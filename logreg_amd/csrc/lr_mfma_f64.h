// lr_mfma_f64.h -- HMC on a FLOAT64 model (padded p = 8, n <= 16 NTW) for MANY chains: the fused matrix-core chain kernel's
// interior (lr_mfma.h: 16 chains per wave, rows as bf16 MFMA operands in registers, eta / sigmoid / gradient of a 16-row
// tile from two MFMAs) with everything else in float64 -- the float64 model's form of LR_PREC_AUTO beside k_chain_mixed
// (lr_kernels.h), which it overtakes from 40 chains per CU as k_chain_mfma overtakes the float32 register kernel.
//
//   float64: position and momentum (lane (c, k) of a wave owns coordinates k, k + 4 of chain c), drift and kick, the
//            end-point evaluations of log-posterior and gradient (rows in LDS, lane (c, k) takes rows k, k + 4, ... of chain c's
//            evaluation, the 8 gradient sums and the value reduced over the chain's 4 lanes), the half kicks, the kinetic
//            energies, the Metropolis test (fit-np-hmc.py:65-87);  draws: double Box-Muller on the shared Philox stream
//   bf16:    the force applied inside the trajectory -- MfmaRows::eval_bf16 on the position rounded to float32, from the
//            rows rounded to float32 and split into two bf16 pieces.  A deterministic function of the position: drift and kick
//            stay shears in float64 arithmetic.
// With LR_PREC_FULL (mode = LR_MODE_MFMA forced) the interior steps use the float64 evaluation as well.
#pragma once
#include "lr_kernels.h"
#include "lr_mfma.h"

namespace lr {

template <int NTW>
__global__ void __launch_bounds__(256) k_chain_mfma_f64(ModelArgs<double, 8> m, ChainArgs<double, 8> a) {
    constexpr int P = 8, NC = 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* const srows = reinterpret_cast<double*>(smem_raw);  // [n][8] float64 rows, shared by the four chain tiles
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = lane & 15, k = lane >> 4;
    const int64_t tile0 = ((int64_t)blockIdx.x * 4 + wave) * 16;
    int64_t chain = a.first + tile0 + c;
    const bool live = chain < a.first + a.count;
    if (!live) chain = a.first + a.count - 1;
    const uint64_t gchain = (uint64_t)(a.chain_offset + chain);
    {
        const int64_t tot = m.n * P;
        for (int64_t i = threadIdx.x; i < tot; i += blockDim.x) srows[i] = m.rows[i];
        __syncthreads();
    }
    MfmaRows<P, NTW, 1> rows;  // (only its bf16 operands are used: the fp32 end-point operands are dead code here)
    rows.load(m.rows, m.n, 0, lane);

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
    // float64 evaluation at q (distributed): likelihood value (VALUE) and the lpost gradient of the lane's coordinates
    auto evaluate = [&](auto want_value, const double (&q)[NC], double (&grad)[NC], double& ll) {
        constexpr bool VALUE = decltype(want_value)::value;
        double b8[P], g8[P], v = 0.0;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {  // all-gather over the chain's four lanes
            b8[kk] = from(c + 16 * kk, q[0]);
            b8[4 + kk] = from(c + 16 * kk, q[1]);
        }
#pragma unroll
        for (int j = 0; j < P; ++j) g8[j] = 0.0;
        StridedRows<double, P, 4> sr;
        sr.base = srows;
        sr.n = m.n;
        sr.gl = k;
        sr.for_each([&](const double(&xs)[P]) { row_term<double, P, VALUE, true>(xs, b8, g8, v); });
#pragma unroll
        for (int j = 0; j < P; ++j) g8[j] = ksum(g8[j]);
        if constexpr (VALUE) ll = ksum(v);
#pragma unroll
        for (int h = 0; h < NC; ++h) grad[h] = pick(g8, h) - q[h] * inv_var[h];
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
            // lane (c, k): Philox block k of the iteration (k = 0, 1: the normals of coordinates 4 k .. 4 k + 3; k = 2, 3: the
            // accept uniform), as k_chain_mfma deals them; coordinate k + 4 h = element k of block h
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
            if (a.interior_bf16) {
                for_pair_count<MfmaRows<P, NTW, 1>::NPAIR>((rows.ntile_live + 1) >> 1, [&](auto npl) {
                    for (int i = 0; i < a.l - 1; ++i) {
                        float qf[NC], gl[NC];
#pragma unroll
                        for (int h = 0; h < NC; ++h) {
                            xp[h] = __builtin_fma(kb[h], pm[h], xp[h]);  // drift (float64)
                            qf[h] = (float)xp[h];
                        }
                        rows.template eval_bf16<decltype(npl)::value>(qf, gl);
#pragma unroll
                        for (int h = 0; h < NC; ++h) pm[h] = __builtin_fma(a.step, __builtin_fma(-xp[h], inv_var[h], (double)gl[h]), pm[h]);  // kick
                    }
                });
            } else {
                for (int i = 0; i < a.l - 1; ++i) {
#pragma unroll
                    for (int h = 0; h < NC; ++h) xp[h] = __builtin_fma(kb[h], pm[h], xp[h]);
                    double d0;
                    evaluate(False{}, xp, gp, d0);
#pragma unroll
                    for (int h = 0; h < NC; ++h) pm[h] = __builtin_fma(a.step, gp[h], pm[h]);
                }
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
        if (a.out && live) {
            double* o = a.out + (it * a.C + chain) * a.p;
#pragma unroll
            for (int h = 0; h < NC; ++h)
                if (k + 4 * h < a.p) o[k + 4 * h] = x[h];
        }
        if (a.stats.buf && live) {  // the lane owns coordinates k + 4 h
            const int64_t idx = a.stats.first + it, sb = idx / a.stats.batch, sk = idx - sb * a.stats.batch;
            const double inv = 1.0 / (double)(sk + 1);
            double* s = a.stats.buf + ((sb * a.C + chain) * 2) * a.p;
#pragma unroll
            for (int h = 0; h < NC; ++h)
                if (k + 4 * h < a.p) stats_fold(s + k + 4 * h, s + a.p + k + 4 * h, sk, inv, x[h]);
        }
    }
    if (live) {
#pragma unroll
        for (int h = 0; h < NC; ++h)
            if (k + 4 * h < a.p) a.state[chain * a.p + k + 4 * h] = x[h];
        if (k == 0 && a.accepts) a.accepts[chain] += nacc;
    }
}

}  // namespace lr
