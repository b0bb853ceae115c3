// lr_f64x.h -- HMC on a FLOAT64 model, padded p = 8, EVERY evaluation float64 (Python/fit-np-hmc.py:17-19, 44-47, 65-87 in the
// reference's own arithmetic): the lane-group kernel k_chain<double, 8, G, ..> with the chain state DISTRIBUTED over the group and the
// group laid ACROSS the four DPP rows of the wave (round 6).  Measured (HMC L = 50, n = 200, chain-iterations/s,
// replicated | this, tools/f64_full_check.py): 1024 chains (64 lanes, rows in registers) 2.40 -> 3.6e7; 2048 (32 lanes, registers) 4.03 ->
// 5.0e7; 4096 (16 lanes, LDS) 4.67 -> 5.07e7 = 0.24 of the float64 vector peak; 8192 / 16 384 (8 lanes, LDS) 5.69 -> 5.9 / 6.2e7 = 0.28 / 0.29.
// (32 lanes x 7 register rows would run two waves per SIMD at 4096 chains if the kernel fitted 256 registers: it needs 280 -- the
// float64 Box-Muller beside 112 registers of rows -- and capped there it spills 17 to scratch, which the build refuses.)
//
// What was wrong with the replicated form (ISA of k_chain<double, 8, 16, 1, 0, HMC>, 952 instructions per evaluation and wave): v_add_f64
// has no DPP form, so the butterfly all-reduce of the 8 gradient sums over a 16-lane DPP row is 8 x 4 x (2 v_mov_b32_dpp + 1 add) = 96
// instructions plus pads; every lane then repeats drift, kick and prior on all 8 coordinates (24), and the five constant 8-vectors of
// the update live in SGPRs that spill (64 v_readlane per evaluation).  The float64 matrix pipe does not help: on gfx950 a
// v_mfma_f64_16x16x4_f64 occupies the SIMD's double-precision lanes for 72-80 cycles and float64 vector instructions do NOT run beside
// it (profiles/r6_mfma_f64_rate.txt) -- a matrix-pipe formulation of this kernel measured 4.4e7 it/s against this one's lane-group
// predecessor at 4.7e7.
//
// Layout (16 lanes per chain; the kernel is a template over G = 8 / 16 / 32 / 64 lanes per chain, LPR = G / 4 of them per DPP row):
// lane l of a wave = (row r = l >> 4, chain c = (l >> 2) & 3, k = l & 3): a wave carries 4 chains, a chain's 16 lanes are
// one QUAD in each of the four 16-lane DPP rows, lane (r, k) of a chain takes rows gl, gl + 16, ... of the design, gl = 4 r + k.
//   * reduce-scatter of the 8 gradient sums: v_permlane32_swap of (v[j], v[j + 4]) + add  ->  4 sums, coordinates 0-3 in rows 0, 1 and
//     4-7 in rows 2, 3;  v_permlane16_swap of (u[j], u[j + 2]) + add  ->  row r holds coordinates 2 r, 2 r + 1 summed over the four rows;
//     two quad_perm levels finish the sum over k.  8 + 4 swaps, 6 + 4 adds, 8 DPP moves: 30 instructions for 96.
//   * position, momentum, gradient, prior and step constants: ONE coordinate pair per lane (row r owns 2 r, 2 r + 1; the quad's four
//     lanes hold copies): drift + kick + prior = 6 instructions for 24, and no constant vector in SGPRs.
//   * all-gather of the position for the row pass: the two swap levels backwards (12 copies + 12 swaps).
// The row pass itself is k_chain's (row_term on StridedRows in LDS): same arithmetic per row.
// Randomness: every lane draws the Philox block of its own coordinate pair (block r >> 1, Box-Muller pair r & 1) and the accept
// uniform -- the library's stream (normal j = element j % 4 of block j / 4; the uniform = word 0 of block TAG_UNIFORM), no exchange.
#pragma once
#include "lr_kernels.h"

namespace lr {

__device__ __forceinline__ void f64x_split(double d, float& lo, float& hi) {
    const uint64_t b = __builtin_bit_cast(uint64_t, d);
    lo = __builtin_bit_cast(float, (uint32_t)b);
    hi = __builtin_bit_cast(float, (uint32_t)(b >> 32));
}
__device__ __forceinline__ double f64x_join(float lo, float hi) {
    return __builtin_bit_cast(double, ((uint64_t)__builtin_bit_cast(uint32_t, hi) << 32) | (uint64_t)__builtin_bit_cast(uint32_t, lo));
}

// 8 per-lane partial sums -> the totals over the chain's 4 LPR lanes of the lane's OWN coordinate pair (2 r, 2 r + 1), identical in the
// LPR lanes the chain has in the DPP row
template <int LPR> __device__ __forceinline__ void f64x_reduce_scatter(const double (&v)[8], double& t0, double& t1) {
    float a[8], b[8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f64x_split(v[j], a[2 * j], a[2 * j + 1]);
        f64x_split(v[j + 4], b[2 * j], b[2 * j + 1]);
    }
    swap_pairs<32, 8>(a, b);  // rows 0, 1: (own v[j], the other half's v[j]);  rows 2, 3: (the other half's v[j + 4], own v[j + 4])
    double u[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) u[j] = f64x_join(a[2 * j], a[2 * j + 1]) + f64x_join(b[2 * j], b[2 * j + 1]);
    float c[4], d[4];
    f64x_split(u[0], c[0], c[1]);
    f64x_split(u[1], c[2], c[3]);
    f64x_split(u[2], d[0], d[1]);
    f64x_split(u[3], d[2], d[3]);
    swap_pairs<16, 4>(c, d);  // even rows: (own u[0..1], the odd row's u[0..1]);  odd rows: (the even row's u[2..3], own u[2..3])
    t0 = f64x_join(c[0], c[1]) + f64x_join(d[0], d[1]);
    t1 = f64x_join(c[2], c[3]) + f64x_join(d[2], d[3]);
    t0 = group_sum<LPR>(t0);  // the in-row levels (symmetric DPP exchanges: bit-identical in the LPR lanes)
    t1 = group_sum<LPR>(t1);
}
// the lane's coordinate pair -> all 8 coordinates of the chain in every one of its lanes
__device__ __forceinline__ void f64x_all_gather(double x0, double x1, double (&all)[8]) {
    float a[4], b[4];
    f64x_split(x0, a[0], a[1]);
    f64x_split(x1, a[2], a[3]);
#pragma unroll
    for (int i = 0; i < 4; ++i) b[i] = a[i];
    swap_pairs<16, 4>(a, b);  // a: the even row's pair, b: the odd row's pair -- in both rows
    float e[8], f[8];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        e[i] = a[i];
        e[4 + i] = b[i];
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = e[i];
    swap_pairs<32, 8>(e, f);  // e: coordinates 0-3 (rows 0, 1), f: coordinates 4-7 (rows 2, 3) -- in both halves
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        all[j] = f64x_join(e[2 * j], e[2 * j + 1]);
        all[4 + j] = f64x_join(f[2 * j], f[2 * j + 1]);
    }
}
// sum over the chain's 4 LPR lanes of a per-lane value; bit-identical in all of them
template <int LPR> __device__ __forceinline__ double f64x_sum_group(double v) { return swap32_sum(swap16_sum(group_sum<LPR>(v))); }
// sum over the four DPP rows of a value that is identical inside the quad (a function of the row's coordinate pair)
__device__ __forceinline__ double f64x_sum_rows(double v) { return swap32_sum(swap16_sum(v)); }

// G lanes per chain (8, 16, 32 or 64: LPR = G / 4 of them in each DPP row); MODE / R: where the rows live, as k_chain's
template <int G, int MODE, int R>
__global__ void __launch_bounds__(256) k_chain_f64x(ModelArgs<double, 8> m, ChainArgs<double, 8> a) {
    constexpr int P = 8, LPR = G / 4, CPW = 64 / G;  // lanes of a chain per DPP row, chains per wave
    static_assert(G == 8 || G == 16 || G == 32 || G == 64, "a chain spans the four DPP rows of a wave");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane >> 4, c = (lane & 15) / LPR, k = lane % LPR, gl = LPR * r + k;
    int64_t chain = a.first + ((int64_t)blockIdx.x * 4 + wave) * CPW + c;
    const bool live = chain < a.first + a.count;
    if (!live) chain = a.first + a.count - 1;  // whole waves stay converged for the cross-lane exchanges; stores are masked
    const bool writer = live && k == 0;
    const auto rows = make_rows<double, P, G, MODE, R>(m, gl, reinterpret_cast<double*>(smem_raw));
    const uint64_t gchain = (uint64_t)(a.chain_offset + chain);

    auto own = [&](const double (&v)[P], int h) {  // coordinate 2 r + h
        const double lo = r == 0 ? v[h] : v[2 + h], hi = r == 2 ? v[4 + h] : v[6 + h];
        return r < 2 ? lo : hi;
    };
    double inv_var[2], ka[2], kb[2], kc[2], x[2], g[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        inv_var[h] = own(m.prior.inv_var, h);
        ka[h] = own(a.a, h);
        kb[h] = own(a.b, h);
        kc[h] = own(a.c, h);
        const int j = 2 * r + h;
        x[h] = j < a.p ? a.state[chain * a.p + j] : 0.0;
    }
    // gradient of the log-posterior for the lane's own pair at q (distributed); VALUE: the log-likelihood too.  INTERIOR: the weight as
    // rcp(1 + exp(t)) (row_term FASTW), as k_chain's interior steps
    auto evaluate = [&](auto want_value, auto interior, const double (&q)[2], double (&grad)[2], double& ll) {
        constexpr bool VALUE = decltype(want_value)::value, FASTW = decltype(interior)::value;
        double b8[P], g8[P], v = 0.0;
        f64x_all_gather(q[0], q[1], b8);
#pragma unroll
        for (int j = 0; j < P; ++j) g8[j] = 0.0;
        rows.for_each([&](const double(&xs)[P]) { row_term<double, P, VALUE, true, FASTW>(xs, b8, g8, v); });
        double t0, t1;
        f64x_reduce_scatter<LPR>(g8, t0, t1);
        if constexpr (VALUE) ll = f64x_sum_group<LPR>(v + rows.value_fixup());
        grad[0] = __builtin_fma(-q[0], inv_var[0], t0);
        grad[1] = __builtin_fma(-q[1], inv_var[1], t1);
    };
    using True = std::integral_constant<bool, true>;
    using False = std::integral_constant<bool, false>;
    auto lprior_of = [&](const double (&q)[2]) {
        return m.prior.lprior_const - 0.5 * f64x_sum_rows(__builtin_fma(q[0] * q[0], inv_var[0], q[1] * q[1] * inv_var[1]));
    };

    double lp;
    uint32_t nacc = 0;
    {
        double ll0 = 0;
        evaluate(True{}, False{}, x, g, ll0);
        lp = ll0 + lprior_of(x);
    }
    const double heps = 0.5 * a.step;
    for (int64_t it = 0; it < a.iters; ++it) {
        for (int64_t jt = 0; jt < a.thin; ++jt) {
            const uint64_t iter = (uint64_t)(a.iter_offset + it * a.thin + jt);
            double z[2], logu;
            {
                const U4 wn = philox4x32_10((uint32_t)gchain, (uint32_t)iter, (uint32_t)(iter >> 32), (uint32_t)(r >> 1), (uint32_t)a.seed,
                                            (uint32_t)(a.seed >> 32));
                if (r & 1) box_muller(wn.z, wn.w, z[0], z[1]);  // coordinates 2 r, 2 r + 1 = elements 2 (r & 1), 2 (r & 1) + 1 of block r >> 1
                else box_muller(wn.x, wn.y, z[0], z[1]);
                const U4 wu = philox4x32_10((uint32_t)gchain, (uint32_t)iter, (uint32_t)(iter >> 32), TAG_UNIFORM, (uint32_t)a.seed,
                                            (uint32_t)(a.seed >> 32));
                logu = log(u01<double>(wu.x));
            }
            // p ~ N(0, dmm); leapfrog l steps; a = alpi(prop) - alpi(x)     fit-np-hmc.py:65-87
            double pm[2], xp[2], gp[2], k0 = 0.0;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                pm[h] = z[h] * ka[h];
                k0 = __builtin_fma(pm[h] * pm[h], kc[h], k0);
                xp[h] = x[h];
                pm[h] = __builtin_fma(heps, g[h], pm[h]);
            }
            for (int i = 0; i < a.l - 1; ++i) {
#pragma unroll
                for (int h = 0; h < 2; ++h) xp[h] = __builtin_fma(kb[h], pm[h], xp[h]);  // drift
                double d0;
                evaluate(False{}, True{}, xp, gp, d0);
#pragma unroll
                for (int h = 0; h < 2; ++h) pm[h] = __builtin_fma(a.step, gp[h], pm[h]);  // kick
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) xp[h] = __builtin_fma(kb[h], pm[h], xp[h]);  // the last drift
            double llp = 0;
            evaluate(True{}, False{}, xp, gp, llp);
            const double lprp = lprior_of(xp);
            double k1 = 0.0;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                pm[h] = __builtin_fma(heps, gp[h], pm[h]);
                k1 = __builtin_fma(pm[h] * pm[h], kc[h], k1);
            }
            const double logr = ((llp + lprp) - lp) - 0.5 * f64x_sum_rows(k1 - k0);
            const bool acc = logu < logr;  // NaN -> reject, as `np.log(np.random.rand()) < a`
            if (acc) {
                ++nacc;
                lp = llp + lprp;
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                x[h] = acc ? xp[h] : x[h];
                g[h] = acc ? gp[h] : g[h];
            }
        }
        if (a.out && writer) {
            double* o = a.out + (it * a.C + chain) * a.p;
#pragma unroll
            for (int h = 0; h < 2; ++h)
                if (2 * r + h < a.p) o[2 * r + h] = x[h];
        }
        if (a.stats.buf && writer) {  // the lane owns coordinates 2 r + h
            const int64_t idx = a.stats.first + it, sb = idx / a.stats.batch, sk = idx - sb * a.stats.batch;
            const double inv = 1.0 / (double)(sk + 1);
            double* s = a.stats.buf + ((sb * a.C + chain) * 2) * a.p;
#pragma unroll
            for (int h = 0; h < 2; ++h)
                if (2 * r + h < a.p) stats_fold(s + 2 * r + h, s + a.p + 2 * r + h, sk, inv, x[h]);
        }
    }
    if (writer) {
#pragma unroll
        for (int h = 0; h < 2; ++h)
            if (2 * r + h < a.p) a.state[chain * a.p + 2 * r + h] = x[h];
        if (r == 0 && a.accepts) a.accepts[chain] += nacc;
    }
}

}  // namespace lr
