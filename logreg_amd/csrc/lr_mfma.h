// lr_mfma.h -- matrix-core formulation of the fused chain kernel (fp32 in, fp32 accumulate;
// padded p = 8, 16 or 32; gfx950 v_mfma_f32_16x16x4_f32, which is bit-for-bit a k-ordered fmaf chain).
//
// One wavefront owns 16 chains.  Lane l = (c, k): c = l & 15 is the chain, k = l >> 4 the
// parameter group; the lane OWNS the NC = p / 4 parameters k + 4h of chain c (position, momentum, gradient
// for those coordinates live only in this lane).  Written out below for p = 8 (h = 0, 1); wider models repeat
// the eta MFMAs per h (one accumulator) and the gradient MFMAs per set of four h.
// Data rows are processed in tiles of 16:
//
//   eta tile  E[16 rows x 16 chains] = Xs[16 x 8] . B^T[8 x 16]       2 MFMAs (K = 4 each)
//       A operand (lane l): Xs[16T + (l&15)][(l>>4) + 4h]       h = 0,1     (registers xa[t][h])
//       B operand (lane l): beta[chain l&15][(l>>4) + 4h]  = the lane's own two coordinates
//       D (lane l, reg r) : eta[row 16T + 4(l>>4) + r][chain l&15]
//   w = sigma(-eta) elementwise on the 4 D registers                    (VALU: exp2, add, rcp)
//   grad tile G^T[16 slots x 16 chains] += XsT[16 slots x 4 rows] . W[4 rows x 16 chains]   4 MFMAs
//       B operand for K-slice s (lane l): w[row 16T + 4(l>>4) + s][chain l&15] = D register s of
//                                         the eta MFMA -- NO data movement between the two GEMMs
//       A operand (lane l): Xs[16T + 4(l>>4) + s][mu(l&15)], slot m = 4k'+r' -> parameter
//                           mu(m) = k' + 4r' for r' < 2, empty (0) otherwise  (registers xg[t][s])
//       D (lane l, reg r) : gradient of parameter k + 4r of chain c for r = 0,1 = the lane's own
//                           two coordinates -- the leapfrog update is lane-local, no transposes.
//
// Only per-iteration scalars (energies, prior) cross lanes (2 permlane swaps over k).
// S = 4 ("row split"): the 4 waves of a workgroup share the same 16 chains and each takes every
// 4th tile; per leapfrog step the partial gradients are combined through LDS in a fixed order
// (bit-identical in all 4 waves).  S = 1: every wave has its own 16 chains.
#pragma once
#include <type_traits>

#include "lr_kernels.h"

namespace lr {

typedef float f32x4 __attribute__((ext_vector_type(4)));

typedef float mf_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 mf_bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 mf_bf16x8 __attribute__((ext_vector_type(8)));
typedef short mf_s16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t mf_u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t mf_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint32_t mf_pack_rne(float a, float b) {  // two fp32 -> packed bf16 (v_cvt_pk_bf16_f32, RNE)
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(mf_f32x2{a, b}, mf_bf16x2));
}
__device__ __forceinline__ float mf_hi_f32(uint32_t packed) { return __builtin_bit_cast(float, packed << 16); }

// rows[row][coord] for row < n, else 0 -- without a branch: a guarded load is a basic block of its own with its own
// vmcnt(0), and the operand builds below issue hundreds of them per launch (measured on the LDS variant at n = 2000:
// 190 serial L2 round trips = 25 us per launch of 10 iterations)
// (Src = double: a float64 model's rows rounded to float32 -- the bf16 interior operands of k_chain_mfma_f64)
template <int P, typename Src> __device__ __forceinline__ float mf_row_or_zero(const Src* __restrict__ rows, int64_t n, int64_t row, int coord) {
    const float v = (float)rows[(row < n ? row : n - 1) * P + coord];
    return row < n ? v : 0.0f;
}

template <int P> constexpr int mf_image_floats();
// End-point evaluation (fp32-input MFMAs) with the operands STREAMED from device memory -- the tile image `image` (mf_image_prepare
// layout; null: gathered from the row matrix) -- for this wave's tiles t S + wave, t < ntile_live.  Shared by the LDS / device-memory
// variants of the bf16 operands and by the register variant whose end-point operands do not fit beside them (MfmaRows END_MEM).
template <int P, int S, bool VALUE>
__device__ __forceinline__ void mf_eval_stream(const float* __restrict__ rows, const float* __restrict__ image, int64_t n, int wave, int lane,
                                               int ntile_live, const float (&q)[P / 4], float (&gl)[P / 4], float& vsum) {
    constexpr int NC = P / 4, NG = (NC + 3) / 4, HG = NC < 4 ? NC : 4;
    const int c = lane & 15, k = lane >> 4;
    const int rp = c & 3, kp = c >> 2;
    float bs[NC];
#pragma unroll
    for (int h = 0; h < NC; ++h) bs[h] = q[h] * ExpScale<float>::k;
    f32x4 ga[NG], gb[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) ga[g] = gb[g] = f32x4{0, 0, 0, 0};
    float v = 0.0f;
    int pad = 0;
    // the fp32 operands come from global memory (L2): blocks of TB tiles, the next block requested before the current
    // one is worked on (one tile ahead left 2200 cycles per tile, mostly L2 latency; the operands-in-registers
    // variant spends 730 on the same arithmetic)
    constexpr int TB = P <= 8 ? 4 : 2;
    float ca[TB][NC], cg[TB][NG][4];
    auto fetch = [&](int t0, float (&fa)[TB][NC], float (&fg)[TB][NG][4]) {
#pragma unroll
        for (int i = 0; i < TB; ++i) {
            const int64_t row0 = 16 * ((int64_t)(t0 + i) * S + wave);  // tiles past the wave's last one: rows >= n, masked to 0
            if (image) {
                const int64_t T = (t0 + i) * S + wave, tiles = (n + 15) / 16;
                const float* o = image + ((size_t)(T < tiles ? T : tiles - 1) * 64 + lane) * mf_image_floats<P>();
#pragma unroll
                for (int h = 0; h < NC; ++h) fa[i][h] = T < tiles ? o[h] : 0.0f;
#pragma unroll
                for (int g = 0; g < NG; ++g)
#pragma unroll
                    for (int s = 0; s < 4; ++s) fg[i][g][s] = T < tiles ? o[NC + 4 * g + s] : 0.0f;
                continue;
            }
#pragma unroll
            for (int h = 0; h < NC; ++h) fa[i][h] = mf_row_or_zero<P>(rows, n, row0 + c, k + 4 * h);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int g = 0; g < NG; ++g)
                    fg[i][g][s] = rp < HG ? mf_row_or_zero<P>(rows, n, row0 + 4 * k + s, kp + 4 * (rp < HG ? rp : 0) + 16 * g) : 0.0f;
        }
    };
    fetch(0, ca, cg);
    for (int t0 = 0; t0 < ntile_live; t0 += TB) {
        float na[TB][NC], ng[TB][NG][4];
        fetch(t0 + TB, na, ng);
#pragma unroll
        for (int i = 0; i < TB; ++i) {
            if (t0 + i < ntile_live) {
                const int64_t row0 = 16 * ((int64_t)(t0 + i) * S + wave);
                f32x4 e = {0, 0, 0, 0};
#pragma unroll
                for (int h = 0; h < NC; ++h) e = __builtin_amdgcn_mfma_f32_16x16x4f32(ca[i][h], bs[h], e, 0, 0, 0);
                float w[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float tr = VALUE ? min_keep_nan(e[r], 100.0f) : e[r];  // (see MfmaRows::eval)
                    const float d = 1.0f + __builtin_amdgcn_exp2f(tr);
                    w[r] = fast_rcp(d);
                    if constexpr (VALUE) v += tr - __builtin_amdgcn_logf(d);
                    if (row0 + 4 * k + r >= n) ++pad;
                }
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    ga[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(cg[i][g][0], w[0], ga[g], 0, 0, 0);
                    gb[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(cg[i][g][1], w[1], gb[g], 0, 0, 0);
                    ga[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(cg[i][g][2], w[2], ga[g], 0, 0, 0);
                    gb[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(cg[i][g][3], w[3], gb[g], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < TB; ++i) {
#pragma unroll
            for (int h = 0; h < NC; ++h) ca[i][h] = na[i][h];
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int s2 = 0; s2 < 4; ++s2) cg[i][g][s2] = ng[i][g][s2];
        }
    }
#pragma unroll
    for (int h = 0; h < NC; ++h) gl[h] = ga[h >> 2][h & 3] + gb[h >> 2][h & 3];
    if constexpr (VALUE) vsum = (v + (float)pad) * ExpScale<float>::inv;
}

// END_MEM (round 5): the fp32 end-point operands are NOT held -- an end-point evaluation streams them from device memory
// (mf_eval_stream).  For HMC at the largest tile counts (p = 16: 16 tiles per wave, p = 32: 8): end-point and interior operands
// together are 1.25 P registers per tile, 320 per lane there -- the kernels ran with 33 / 108 registers spilled to scratch.  One of an
// iteration's l evaluations is an end point, so the stream costs little; the l - 1 interior ones keep their operands in registers.
template <int P, int NTW, int S, bool END_MEM = false> struct MfmaRows {
    static constexpr int NC = P / 4;             // coordinates per lane: j_h = k + 4 h
    static constexpr int NU = NC / 2;            // coordinate pairs (h = 2u, 2u + 1): one bf16 MFMA each
    static constexpr int NG = (NC + 3) / 4;      // fp32 gradient MFMA sets of up to four h
    static constexpr int HG = NC < 4 ? NC : 4;   // h per set
    float xa[END_MEM ? 1 : NTW][NC];
    float xg[END_MEM ? 1 : NTW][NG][4];
    const float* rows_ = nullptr;  // END_MEM: the row matrix, the fp32 operand image (or null) and this lane's place
    const float* image = nullptr;
    int64_t n_ = 0;
    int wave_ = 0, lane_ = 0;
    int pad_rows;  // rows >= n among this lane's eta rows (each adds log sigma(0) = -log 2)
    // INTERIOR leapfrog steps on the bf16 matrix pipe (the scheme of lr_tall_mx.h with the rows in registers):
    //   xs = x log2 e = xh + xl, beta = bh + bl (two round-to-nearest bf16 pieces each); per pair u the lane's coordinates
    //   a = k + 8u, b = a + 4
    //   eta tile:  ONE v_mfma_f32_16x16x32_bf16 per pair (the K = 32 instruction costs what one K = 16 does: 8 slots per lane)
    //              A (lane (row, k)) = [xh_a xl_a xh_a xl_a xh_b xl_b xh_b xl_b],  B = [bh_a bh_a bl_a bl_a bh_b bh_b bl_b bl_b]:
    //              all four piece products of both coordinates.  A one-piece design (half the operand) was tried: on the
    //              unscaled Pima covariates its 2^-9 rounding raised the sd of the energy error from 0.152 to 0.191 and cost
    //              1.7 points of acceptance (0.957 -> 0.940), so the second piece stays.
    //   grad tile pair (K = 32 rows):  B = w = sigma(-eta) of the lane's own 2 x 4 accumulator values, one bf16 piece;
    //              A (lane (m', k')) = element m' & 3 of (xh_a xl_a xh_b xl_b) of group m' >> 2, rows of slot group k'
    //              D (lane (c, k), r) = sum_rows w (xh_a, xl_a, xh_b, xl_b)[r]:  g_a = D0 + D1, g_b = D2 + D3
    static constexpr int NPAIR = (NTW + 1) / 2;
    mf_u32x4 xe[NTW][NU];
    mf_u32x4 xq[NPAIR][NU];
    int ntile_live;  // this wave's tiles that contain at least one real row (all-padding tiles are skipped)

    template <typename Src> __device__ __forceinline__ void load(const Src* __restrict__ rows, int64_t n, int wave, int lane) {
        const int c = lane & 15, k = lane >> 4;
        const int rp = c & 3, kp = c >> 2;  // slot m = c = 4*kp + rp -> parameter kp + 4 rp (+ 16 per set) of the gradient A operand
        pad_rows = 0;
        if constexpr (END_MEM) {
            if constexpr (std::is_same<Src, float>::value) rows_ = rows;
            n_ = n;
            wave_ = wave;
            lane_ = lane;
        } else {
#pragma unroll
            for (int t = 0; t < NTW; ++t) {
                const int64_t base = 16 * ((int64_t)t * S + wave);
                const int64_t ra = base + c;
#pragma unroll
                for (int h = 0; h < NC; ++h) xa[t][h] = mf_row_or_zero<P>(rows, n, ra, k + 4 * h);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int64_t rg = base + 4 * k + s;
#pragma unroll
                    for (int g = 0; g < NG; ++g) xg[t][g][s] = rp < HG ? mf_row_or_zero<P>(rows, n, rg, kp + 4 * (rp < HG ? rp : 0) + 16 * g) : 0.0f;
                    if (rg >= n) ++pad_rows;
                }
            }
        }
        // bf16 operands of the interior steps
        auto piece = [&](int64_t row, int coord, int lo) {  // bf16 bit pattern of the hi / lo piece of X[row][coord]
            const float x = mf_row_or_zero<P>(rows, n, row, coord);
            const uint32_t h = mf_pack_rne(x, x) & 0xFFFFu;
            if (!lo) return h;
            return mf_pack_rne(x - mf_hi_f32(h), 0.0f) & 0xFFFFu;
        };
        ntile_live = 0;
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            const int64_t base = 16 * ((int64_t)t * S + wave);
            if (base < n) ntile_live = t + 1;
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                uint32_t hl[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float xs = mf_row_or_zero<P>(rows, n, base + c, k + 8 * u + 4 * h) * ExpScale<float>::k;
                    const uint32_t hi = mf_pack_rne(xs, xs) & 0xFFFFu;
                    hl[h] = hi | (mf_pack_rne(0.0f, xs - mf_hi_f32(hi)) & 0xFFFF0000u);
                }
                xe[t][u] = mf_u32x4{hl[0], hl[0], hl[1], hl[1]};
            }
        }
        const int grp = c >> 2, el = c & 3;  // gradient A operand: M-row c = 4 grp + el -> element el of group grp
        const int gcoord = grp + 4 * (el >> 1), glo = el & 1;
#pragma unroll
        for (int pi = 0; pi < NPAIR; ++pi) {
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                uint32_t v[8];
#pragma unroll
                for (int sl = 0; sl < 8; ++sl) {
                    const int t = 2 * pi + (sl >> 2);
                    const int64_t row = 16 * ((int64_t)t * S + wave) + 4 * k + (sl & 3);
                    v[sl] = t < NTW ? piece(row, gcoord + 8 * u, glo) : 0u;
                }
                xq[pi][u] = mf_u32x4{v[0] | (v[1] << 16), v[2] | (v[3] << 16), v[4] | (v[5] << 16), v[6] | (v[7] << 16)};
            }
        }
    }

    // interior-step gradient (likelihood part, this wave's tiles) for the lane's coordinates.  NPL = tile pairs
    // to run (compile time: a run-time skip of dead tiles inside the step put every tile behind its own branch, each with
    // the full MFMA -> exp latency exposed).  All-padding tiles inside the NPL pairs have zero operands and add exactly 0.
    template <int NPL> __device__ __forceinline__ void eval_bf16(const float (&q)[NC], float (&gl)[NC]) const {
        static_assert(NPL >= 1 && NPL <= NPAIR, "pairs");
        mf_u32x4 bb[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const uint32_t ha = mf_pack_rne(q[2 * u], q[2 * u]), hb = mf_pack_rne(q[2 * u + 1], q[2 * u + 1]);  // piece | piece << 16
            const float la = q[2 * u] - mf_hi_f32(ha), lb = q[2 * u + 1] - mf_hi_f32(hb);
            bb[u] = mf_u32x4{ha, mf_pack_rne(la, la), hb, mf_pack_rne(lb, lb)};
        }
        constexpr int NT = 2 * NPL < NTW ? 2 * NPL : NTW;
        // Issue order.  Measured at one wave per SIMD (HMC L=50, n=200, p=8; S=4 at 4096 chains | S=1 at 16 384 chains, ms per
        // 20 iterations): compiler's order (all eta MFMAs, then the exp / rcp work, gradient MFMAs) 0.390 | 0.786; next
        // pair's eta MFMAs fenced in front of each pair's exp / rcp work 0.401 | 0.724; MFMAs spread between the VALU
        // groups with sched_group_barrier 0.420 | 0.739; the same with v_add_f32 for v_pk_add_f32 0.432 | 0.752.  A wave
        // does not overlap its own MFMAs with its own VALU work to any useful degree here (SQ_ACTIVE_INST_VALU = 76 % of
        // the wave's cycles for 207 instructions per step, MFMA pipe busy 22 %): the step costs the sum of its issue
        // slots, so the order only decides how many s_nop states follow the MFMAs.  S = 1 (13 tiles) takes the fenced
        // order, the short row-split bodies the compiler's.
        auto eta = [&](int t) {
            f32x4 acc = {0, 0, 0, 0};
#pragma unroll
            for (int u = 0; u < NU; ++u)
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mf_bf16x8, xe[t][u]), __builtin_bit_cast(mf_bf16x8, bb[u]), acc, 0, 0, 0);
            return acc;
        };
        f32x4 e[2 * NPL];
        e[0] = eta(0);
        if constexpr (NT > 1) e[1] = eta(1);
        if constexpr (S == 1) __builtin_amdgcn_sched_barrier(0);
        f32x4 gacc[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) gacc[u] = f32x4{0, 0, 0, 0};
#pragma unroll
        for (int pi = 0; pi < NPL; ++pi) {
            uint32_t wq[4] = {0u, 0u, 0u, 0u};
            if (2 * pi + 2 < NT) e[2 * pi + 2] = eta(2 * pi + 2);
            if (2 * pi + 3 < NT) e[2 * pi + 3] = eta(2 * pi + 3);
            if constexpr (S == 1) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int T = 0; T < 2; ++T) {
                const int t = 2 * pi + T;
                if (t < NT) {
                    const mf_f32x2 d0 = mf_f32x2{__builtin_amdgcn_exp2f(e[t][0]), __builtin_amdgcn_exp2f(e[t][1])} + mf_f32x2{1.0f, 1.0f};
                    const mf_f32x2 d1 = mf_f32x2{__builtin_amdgcn_exp2f(e[t][2]), __builtin_amdgcn_exp2f(e[t][3])} + mf_f32x2{1.0f, 1.0f};
                    wq[2 * T] = mf_pack_rne(fast_rcp(d0.x), fast_rcp(d0.y));
                    wq[2 * T + 1] = mf_pack_rne(fast_rcp(d1.x), fast_rcp(d1.y));
                }
            }
            const mf_u32x4 wv = {wq[0], wq[1], wq[2], wq[3]};
#pragma unroll
            for (int u = 0; u < NU; ++u)
                gacc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mf_bf16x8, xq[pi][u]), __builtin_bit_cast(mf_bf16x8, wv), gacc[u], 0, 0, 0);
            if constexpr (S == 1) __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            gl[2 * u] = gacc[u][0] + gacc[u][1];
            gl[2 * u + 1] = gacc[u][2] + gacc[u][3];
        }
    }

    // likelihood part for the lane's coordinates over THIS wave's tiles:
    //   gl[h] = sum_rows sigma(-t) * xs[row][k+4h],  vsum = sum over the lane's eta rows of log sigma(t)
    template <bool VALUE>
    __device__ __forceinline__ void eval(const float (&q)[NC], float (&gl)[NC], float& vsum) const {
        if constexpr (END_MEM) {
            mf_eval_stream<P, S, VALUE>(rows_, image, n_, wave_, lane_, ntile_live, q, gl, vsum);
            return;
        }
        float bs[NC];
#pragma unroll
        for (int h = 0; h < NC; ++h) bs[h] = q[h] * ExpScale<float>::k;
        f32x4 ga[NG], gb[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) ga[g] = gb[g] = f32x4{0, 0, 0, 0};
        float v = 0.0f;
#pragma unroll
        for (int t = 0; t < NTW; ++t) {
            f32x4 e = {0, 0, 0, 0};
#pragma unroll
            for (int h = 0; h < NC; ++h) e = __builtin_amdgcn_mfma_f32_16x16x4f32(xa[t][h], bs[h], e, 0, 0, 0);
            float w[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                // value: log2 sigma = t - log2(1 + 2^t) from the same 2^t the gradient needs (pair_term's form, lr_device.h:
                // t clamped at 100 so that 2^t stays finite -- sigma(-t) < 2^-100 there; the cancellation at large t costs
                // an absolute ulp(t) ~ 1e-6 per row); one v_log per value instead of exp2 + log + 7 VALU
                const float tr = VALUE ? min_keep_nan(e[r], 100.0f) : e[r];
                const float d = 1.0f + __builtin_amdgcn_exp2f(tr);
                w[r] = fast_rcp(d);
                if constexpr (VALUE) v += tr - __builtin_amdgcn_logf(d);
            }
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                ga[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(xg[t][g][0], w[0], ga[g], 0, 0, 0);
                gb[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(xg[t][g][1], w[1], gb[g], 0, 0, 0);
                ga[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(xg[t][g][2], w[2], ga[g], 0, 0, 0);
                gb[g] = __builtin_amdgcn_mfma_f32_16x16x4f32(xg[t][g][3], w[3], gb[g], 0, 0, 0);
            }
        }
#pragma unroll
        for (int h = 0; h < NC; ++h) gl[h] = ga[h >> 2][h & 3] + gb[h >> 2][h & 3];
        if constexpr (VALUE) vsum = (v + (float)pad_rows) * ExpScale<float>::inv;  // v in log2 units; a zero row adds log2 sigma(0) = -1
    }
};

// The same operands kept in LDS instead of registers, for data beyond the register variants (16 S NTW < n): every wave of
// a row-split workgroup builds the bf16 images of ITS tiles in a private LDS region at kernel start (64 p/8 bytes per
// row: n <= 2400 at p = 8, 1200 at p = 16) and the interior steps stream them back (ds_read_b64 / b128 per
// lane and tile, 1.5 KB per tile and wave against ~100 cycles of MFMA + VALU work).  S = 1: the four chain tiles of a
// workgroup share ONE image (every wave writes the same bytes).  The end-point evaluations take their fp32 MFMA operands
// from the tile image below (device memory, built on the host at model creation), else gathered from the row matrix.
//
// fp32 operand image of the end-point evaluations, per GLOBAL 16-row tile T (row-split independent) and lane (c, k):
//   [T][lane][0 .. NC)            A operand of the eta MFMAs:      rows[16T + c][k + 4h]
//   [T][lane][NC + 4g + s]        A operand of the gradient MFMAs: rows[16T + 4k + s][c/4 + 4 (c%4) + 16g]  (0 if c%4 >= HG)
// so that a lane's operands of a tile are (NC + 4 NG) consecutive floats: two or three wide, fully coalesced loads
// per tile instead of NC + 4 NG scattered dword loads (which kept the TCP / L2 busy for 2000 cycles per tile).
template <int P> constexpr int mf_image_floats() { return P / 4 + 4 * ((P / 4 + 3) / 4); }
template <int P> inline void mf_image_prepare(const float* rows, int64_t n, float* out) {
    constexpr int NC = P / 4, NG = (NC + 3) / 4, HG = NC < 4 ? NC : 4, F = mf_image_floats<P>();
    const int64_t tiles = (n + 15) / 16;
    for (int64_t T = 0; T < tiles; ++T)
        for (int lane = 0; lane < 64; ++lane) {
            const int c = lane & 15, k = lane >> 4, rp = c & 3, kp = c >> 2;
            float* o = out + ((size_t)T * 64 + lane) * F;
            const int64_t ra = 16 * T + c;
            for (int h = 0; h < NC; ++h) o[h] = ra < n ? rows[ra * P + k + 4 * h] : 0.0f;
            for (int g = 0; g < NG; ++g)
                for (int s = 0; s < 4; ++s) {
                    const int64_t rg = 16 * T + 4 * k + s;
                    o[NC + 4 * g + s] = (rg < n && rp < HG) ? rows[rg * P + kp + 4 * rp + 16 * g] : 0.0f;
                }
        }
}

extern __shared__ __attribute__((aligned(16))) unsigned char lr_mfma_dyn_smem[];
// GLOBAL = true: the same images, built once per model by k_mfma_image_build into device memory (one region per wave of the
// row split, identical for every workgroup) for data beyond LDS; the interior steps then stream them from L2 with the
// operands of the next two tile pairs requested while the current two are worked on.
template <int P, int S, bool GLOBAL = false> struct MfmaRowsLds {
    static constexpr int NC = P / 4, NU = NC / 2, NG = (NC + 3) / 4, HG = NC < 4 ? NC : 4;
    static constexpr int NPAIR = 1;  // no per-pair-count code versions: the tile loops are run-time loops
    // device memory: PADP all-zero pairs behind the wave's last one, so that the two-trip look-ahead of the interior loop
    // needs no index clamps
    static constexpr int PADP = GLOBAL ? 4 : 0;
    static __host__ __device__ constexpr size_t bytes_per_wave(int64_t ntw) {  // eta images for an even number of tiles
        return (size_t)((ntw + 1) / 2 + PADP) * 2 * NU * 64 * 8 + (size_t)((ntw + 1) / 2 + PADP) * NU * 64 * 16;
    }
    const float* rows;
    const float* image;  // mf_image_prepare layout, or null (then the operands are gathered from `rows`)
    int64_t n;
    int wave, lane, ntw, ntile_live;
    mf_u32x2* xe;  // [ntw][NU][64]
    mf_u32x4* xq;  // [(ntw + 1) / 2][NU][64]

    // point at the wave's image region inside `store` (LDS or device memory); no data is touched
    __device__ __forceinline__ void attach(const float* __restrict__ rows_, int64_t n_, int wave_, int lane_, unsigned char* store) {
        rows = rows_;
        n = n_;
        wave = wave_;
        lane = lane_;
        const int64_t tiles = (n + 15) / 16;
        ntw = (int)((tiles + S - 1) / S);
        const int npair = (ntw + 1) / 2 + PADP;
        unsigned char* base = store + (size_t)wave * bytes_per_wave(ntw);
        xq = reinterpret_cast<mf_u32x4*>(base);
        xe = reinterpret_cast<mf_u32x2*>(base + (size_t)npair * NU * 64 * 16);
        const int64_t live = (tiles - wave + S - 1) / S;  // tiles t with 16 (t S + wave) < n
        ntile_live = (int)(live < 0 ? 0 : live);
    }
    __device__ __forceinline__ void load(const float* __restrict__ rows_, int64_t n_, int wave_, int lane_) {
        attach(rows_, n_, wave_, lane_, lr_mfma_dyn_smem);
        build();
    }
    // write the wave's images (called by every workgroup for LDS, once per model by k_mfma_image_build for device memory)
    __device__ __forceinline__ void build() {
        const int c = lane & 15, k = lane >> 4;
        const int npair = (ntw + 1) / 2 + PADP;
        auto piece = [&](int64_t row, int coord, int lo) {
            const float x = mf_row_or_zero<P>(rows, n, row, coord);
            const uint32_t h = mf_pack_rne(x, x) & 0xFFFFu;
            if (!lo) return h;
            return mf_pack_rne(x - mf_hi_f32(h), 0.0f) & 0xFFFFu;
        };
        for (int t = 0; t < 2 * npair; ++t) {  // (an odd tile count: one all-zero image at the end)
            const int64_t row0 = t < ntw ? 16 * ((int64_t)t * S + wave) : n;
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                uint32_t hl[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float xs = mf_row_or_zero<P>(rows, n, row0 + c, k + 8 * u + 4 * h) * ExpScale<float>::k;
                    const uint32_t hi = mf_pack_rne(xs, xs) & 0xFFFFu;
                    hl[h] = hi | (mf_pack_rne(0.0f, xs - mf_hi_f32(hi)) & 0xFFFF0000u);
                }
                xe[(t * NU + u) * 64 + lane] = mf_u32x2{hl[0], hl[1]};
            }
        }
        const int grp = c >> 2, el = c & 3;
        const int gcoord = grp + 4 * (el >> 1), glo = el & 1;
        for (int pi = 0; pi < npair; ++pi) {
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                uint32_t v[8];
#pragma unroll
                for (int sl = 0; sl < 8; ++sl) {
                    const int t = 2 * pi + (sl >> 2);
                    const int64_t row = 16 * ((int64_t)t * S + wave) + 4 * k + (sl & 3);
                    v[sl] = t < ntw ? piece(row, gcoord + 8 * u, glo) : 0u;
                }
                xq[(pi * NU + u) * 64 + lane] = mf_u32x4{v[0] | (v[1] << 16), v[2] | (v[3] << 16), v[4] | (v[5] << 16), v[6] | (v[7] << 16)};
            }
        }
    }

    template <int NPL> __device__ __forceinline__ void eval_bf16(const float (&q)[NC], float (&gl)[NC]) const {
        mf_u32x4 bb[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const uint32_t ha = mf_pack_rne(q[2 * u], q[2 * u]), hb = mf_pack_rne(q[2 * u + 1], q[2 * u + 1]);
            const float la = q[2 * u] - mf_hi_f32(ha), lb = q[2 * u + 1] - mf_hi_f32(hb);
            bb[u] = mf_u32x4{ha, mf_pack_rne(la, la), hb, mf_pack_rne(lb, lb)};
        }
        f32x4 gacc[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) gacc[u] = f32x4{0, 0, 0, 0};
        const int npl = (ntile_live + 1) >> 1;
        auto pair_work = [&](int pi) {
            mf_u32x2 he[2][NU];
            mf_u32x4 hq[NU];
#pragma unroll
            for (int T = 0; T < 2; ++T)
#pragma unroll
                for (int u = 0; u < NU; ++u) he[T][u] = xe[((2 * pi + T) * NU + u) * 64 + lane];
#pragma unroll
            for (int u = 0; u < NU; ++u) hq[u] = xq[(pi * NU + u) * 64 + lane];
            uint32_t wq[4];
#pragma unroll
            for (int T = 0; T < 2; ++T) {
                f32x4 e = {0, 0, 0, 0};
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    const mf_u32x4 av = {he[T][u][0], he[T][u][0], he[T][u][1], he[T][u][1]};
                    e = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mf_bf16x8, av), __builtin_bit_cast(mf_bf16x8, bb[u]), e, 0, 0, 0);
                }
                const mf_f32x2 d0 = mf_f32x2{__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])} + mf_f32x2{1.0f, 1.0f};
                const mf_f32x2 d1 = mf_f32x2{__builtin_amdgcn_exp2f(e[2]), __builtin_amdgcn_exp2f(e[3])} + mf_f32x2{1.0f, 1.0f};
                wq[2 * T] = mf_pack_rne(fast_rcp(d0.x), fast_rcp(d0.y));
                wq[2 * T + 1] = mf_pack_rne(fast_rcp(d1.x), fast_rcp(d1.y));
            }
            const mf_u32x4 wv = {wq[0], wq[1], wq[2], wq[3]};
#pragma unroll
            for (int u = 0; u < NU; ++u)
                gacc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mf_bf16x8, hq[u]), __builtin_bit_cast(mf_bf16x8, wv), gacc[u], 0, 0, 0);
        };
        if constexpr (GLOBAL) {
            // operands from device memory (L2): trips of two pairs, the next trip's operands requested before the current
            // trip is worked on; two register sets used alternately (no copies; a third set, i.e. two trips between request and
            // use, was measured: 81 -> 78 TF at n = 3000, so latency is not what limits this path).  Pairs past the wave's last
            // one are zero images and are not worked on.
            struct Trip { mf_u32x2 he[2][2][NU]; mf_u32x4 hq[2][NU]; };
            auto fetch = [&](int p0, Trip& tr) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int pj = p0 + j;  // up to npl + 3: the zero pairs behind the wave's last one
#pragma unroll
                    for (int T = 0; T < 2; ++T)
#pragma unroll
                        for (int u = 0; u < NU; ++u) tr.he[j][T][u] = xe[((2 * pj + T) * NU + u) * 64 + lane];
#pragma unroll
                    for (int u = 0; u < NU; ++u) tr.hq[j][u] = xq[(pj * NU + u) * 64 + lane];
                }
            };
            auto work = [&](int p0, const Trip& tr) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (p0 + j < npl) {
                        uint32_t wq[4];
#pragma unroll
                        for (int T = 0; T < 2; ++T) {
                            f32x4 e = {0, 0, 0, 0};
#pragma unroll
                            for (int u = 0; u < NU; ++u) {
                                const mf_u32x4 av = {tr.he[j][T][u][0], tr.he[j][T][u][0], tr.he[j][T][u][1], tr.he[j][T][u][1]};
                                e = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mf_bf16x8, av), __builtin_bit_cast(mf_bf16x8, bb[u]), e, 0, 0, 0);
                            }
                            const mf_f32x2 d0 = mf_f32x2{__builtin_amdgcn_exp2f(e[0]), __builtin_amdgcn_exp2f(e[1])} + mf_f32x2{1.0f, 1.0f};
                            const mf_f32x2 d1 = mf_f32x2{__builtin_amdgcn_exp2f(e[2]), __builtin_amdgcn_exp2f(e[3])} + mf_f32x2{1.0f, 1.0f};
                            wq[2 * T] = mf_pack_rne(fast_rcp(d0.x), fast_rcp(d0.y));
                            wq[2 * T + 1] = mf_pack_rne(fast_rcp(d1.x), fast_rcp(d1.y));
                        }
                        const mf_u32x4 wv = {wq[0], wq[1], wq[2], wq[3]};
#pragma unroll
                        for (int u = 0; u < NU; ++u)
                            gacc[u] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mf_bf16x8, tr.hq[j][u]), __builtin_bit_cast(mf_bf16x8, wv), gacc[u], 0, 0, 0);
                    }
                }
            };
            Trip ta, tb;
            fetch(0, ta);
            for (int p0 = 0; p0 < npl; p0 += 4) {
                fetch(p0 + 2, tb);
                work(p0, ta);
                fetch(p0 + 4, ta);
                work(p0 + 2, tb);
            }
        } else {
            // two pairs per trip: the scheduler lifts all eight LDS reads of the trip to its top, so the second pair's latency
            // hides under the first pair's work (one wave per SIMD: nothing else would hide it)
            int pi = 0;
            for (; pi + 1 < npl; pi += 2) {
                pair_work(pi);
                pair_work(pi + 1);
            }
            if (pi < npl) pair_work(pi);
        }
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            gl[2 * u] = gacc[u][0] + gacc[u][1];
            gl[2 * u + 1] = gacc[u][2] + gacc[u][3];
        }
    }

    template <bool VALUE>
    __device__ __forceinline__ void eval(const float (&q)[NC], float (&gl)[NC], float& vsum) const {
        mf_eval_stream<P, S, VALUE>(rows, image, n, wave, lane, ntile_live, q, gl, vsum);
    }
};

// f(integral_constant<int, max(n, 1)>) for a run-time 1 <= n <= N
template <int N, typename F> __device__ __forceinline__ void for_pair_count(int n, F&& f) {
    if constexpr (N <= 1) {
        f(std::integral_constant<int, 1>{});
    } else {
        if (n >= N) f(std::integral_constant<int, N>{});
        else for_pair_count<N - 1>(n, f);
    }
}

// sum over the 4 parameter groups k (lanes c, c+16, c+32, c+48); identical in all 4 lanes
template <typename T> __device__ __forceinline__ T ksum(T v) { return swap32_sum(swap16_sum(v)); }

// one workgroup of S waves: the bf16 operand images of MfmaRowsLds<P, S, true> into device memory, once per model
template <int P, int S>
__global__ void __launch_bounds__(64 * S) k_mfma_image_build(const float* rows, int64_t n, unsigned char* store) {
    MfmaRowsLds<P, S, true> r;
    r.attach(rows, n, threadIdx.x >> 6, threadIdx.x & 63, store);
    r.build();
}

template <int P, int NTW, int S, int KIND>
__global__ void __launch_bounds__((S == 1 ? 256 : 64 * S)) k_chain_mfma(ModelArgs<float, P> m, ChainArgs<float, P> a) {
    constexpr int NC = P / 4;                 // coordinates per lane
    constexpr int SW = S == 1 ? 4 : S;        // waves of the workgroup (S = 1: four independent chain tiles)
    __shared__ float red[2][SW][64][NC];      // row split: per-step partial gradients of the S waves, double-buffered
    __shared__ double redv[SW][64];           // row split: partial log-likelihood values
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // (scalar: loop bounds, addresses)
    const int c = lane & 15, k = lane >> 4;
    const int64_t tile0 = S == 1 ? ((int64_t)blockIdx.x * 4 + wave) * 16 : (int64_t)blockIdx.x * 16;
    int64_t chain = a.first + tile0 + c;  // (a launch advances chains [first, first + count) of the call's arrays: lr_kernels.h)
    const bool live = chain < a.first + a.count;
    if (!live) chain = a.first + a.count - 1;
    const bool writer = live && (S == 1 || wave == 0);
    const uint64_t gchain = (uint64_t)(a.chain_offset + chain);

    // NTW = 0: operands in LDS; NTW = -1: operands in device memory (built once per model)
    // (HMC at the largest tile counts: the end-point operands streamed -- MfmaRows END_MEM)
    constexpr bool kEndMem = KIND == KIND_HMC && NTW > 0 && NTW * P >= 256;
    using Rows = std::conditional_t<NTW == 0, MfmaRowsLds<P, S, false>,
                                    std::conditional_t<NTW < 0, MfmaRowsLds<P, S, true>, MfmaRows<P, (NTW <= 0 ? 1 : NTW), S, kEndMem>>>;
    Rows rows;
    if constexpr (NTW < 0) {  // device images: the S = 4 one first, the S = 8 one behind it
        const size_t skip = S == 8 ? 4 * MfmaRowsLds<P, 4, true>::bytes_per_wave((int64_t)(((m.n + 15) / 16 + 3) / 4)) : 0;
        rows.attach(m.rows, m.n, wave, lane, const_cast<unsigned char*>(m.ops_mf) + skip);
    }
    else rows.load(m.rows, m.n, S == 1 ? 0 : wave, lane);
    if constexpr (NTW <= 0 || kEndMem) rows.image = m.rows_mf;

    // the lane's coordinates: j_h = k + 4h
    auto pick = [&](const float (&v)[P], int h) {
        const float lo = k == 0 ? v[4 * h] : v[4 * h + 1], hi = k == 2 ? v[4 * h + 2] : v[4 * h + 3];
        return k < 2 ? lo : hi;
    };
    float inv_var[NC], ka[NC], kb[NC], kc[NC], x[NC], g[NC];
#pragma unroll
    for (int h = 0; h < NC; ++h) {
        inv_var[h] = pick(m.prior.inv_var, h);
        ka[h] = pick(a.a, h);
        kb[h] = pick(a.b, h);
        kc[h] = pick(a.c, h);
        const int j = k + 4 * h;
        x[h] = j < a.p ? a.state[chain * a.p + j] : 0.0f;
    }

    int step_parity = 0;
    // partial gradients of the 4 waves of a row-split workgroup, summed in a fixed order (identical in all 4 waves)
    auto combine = [&](float (&gl)[NC]) {
#pragma unroll
        for (int h = 0; h < NC; ++h) red[step_parity][wave][lane][h] = gl[h];
        __syncthreads();
#pragma unroll
        for (int h = 0; h < NC; ++h) {
            float q4[SW / 4];
#pragma unroll
            for (int w = 0; w < SW; w += 4)
                q4[w / 4] = (red[step_parity][w][lane][h] + red[step_parity][w + 1][lane][h]) +
                            (red[step_parity][w + 2][lane][h] + red[step_parity][w + 3][lane][h]);
            float tot = q4[0];
#pragma unroll
            for (int w = 1; w < SW / 4; ++w) tot += q4[w];
            gl[h] = tot;
        }
    };
    // full gradient (likelihood over all tiles + prior) for own coordinates; VALUE: ll (double, replicated)
    auto evaluate = [&](auto want_value, const float (&q)[NC], float (&grad)[NC], double& ll) {
        constexpr bool VALUE = decltype(want_value)::value;
        float gl[NC], vs = 0.0f;
        rows.template eval<VALUE>(q, gl, vs);
        if constexpr (S > 1) {
            if constexpr (VALUE) redv[wave][lane] = (double)vs;
            combine(gl);
            double dv = 0;
            if constexpr (VALUE) {
#pragma unroll
                for (int w = 0; w < SW; w += 4) dv += (redv[w][lane] + redv[w + 1][lane]) + (redv[w + 2][lane] + redv[w + 3][lane]);
                __syncthreads();  // redv is single-buffered; value passes are rare
            }
            step_parity ^= 1;
            if constexpr (VALUE) ll = ksum(dv);
        } else {
            if constexpr (VALUE) ll = ksum((double)vs);
        }
#pragma unroll
        for (int h = 0; h < NC; ++h) grad[h] = gl[h] - q[h] * inv_var[h];
    };
    // interior leapfrog step: gradient only, from the bf16 operands (LR_PREC_BF16 / AUTO), else the exact evaluation
    auto evaluate_interior = [&](auto npl, const float (&q)[NC], float (&grad)[NC]) {
        float gl[NC];
        rows.template eval_bf16<decltype(npl)::value>(q, gl);
        if constexpr (S > 1) {
            combine(gl);
            step_parity ^= 1;
        }
#pragma unroll
        for (int h = 0; h < NC; ++h) grad[h] = gl[h] - q[h] * inv_var[h];
    };
    using True = std::integral_constant<bool, true>;
    using False = std::integral_constant<bool, false>;
    auto lprior_of = [&](const float (&q)[NC]) {
        float acc = q[NC - 1] * q[NC - 1] * inv_var[NC - 1];
#pragma unroll
        for (int h = NC - 2; h >= 0; --h) acc = fma_t(q[h] * q[h], inv_var[h], acc);
        return m.prior.lprior_const - 0.5 * (double)ksum(acc);
    };

    double lp;
    uint32_t nacc = 0;
    {
        double ll0 = 0;
        if constexpr (KIND == KIND_HMC) {
            evaluate(True{}, x, g, ll0);
            lp = ll0 + lprior_of(x);
        } else if constexpr (KIND == KIND_MALA) {
            evaluate(False{}, x, g, ll0);
            lp = a.lp_state[chain];
        } else if constexpr (KIND == KIND_UL) {
            evaluate(False{}, x, g, ll0);
            lp = 0;
        } else {
            lp = a.lp_state[chain];
        }
    }

    for (int64_t it = 0; it < a.iters; ++it) {
        for (int64_t jt = 0; jt < a.thin; ++jt) {
            const uint64_t iter = (uint64_t)(a.iter_offset + it * a.thin + jt);
            // Philox blocks are dealt over the 4 lanes (c, 0..3) of a chain instead of every lane computing all of them:
            // in round r lane (c, k) generates normal block 4r + k (coordinates 4b .. 4b + 3 of block b, as everywhere);
            // p = 8 has two normal blocks, so its lanes k = 2, 3 generate the accept-uniform block in the same round, wider
            // models generate it in every lane.  Every lane then collects the normal of coordinate k + 4h = element k of
            // block h from lane (c, h & 3) and log(u).  Same values, bit for bit.
            float z[NC], logu_f;
            {
                constexpr int ROUNDS = (NC + 3) / 4;
                auto from = [&](int src_lane, float v) {
                    return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src_lane * 4, __builtin_bit_cast(int, v)));
                };
                auto block = [&](uint32_t blk) {
                    return philox4x32_10((uint32_t)gchain, (uint32_t)iter, (uint32_t)(iter >> 32), blk, (uint32_t)a.seed,
                                         (uint32_t)(a.seed >> 32));
                };
#pragma unroll
                for (int r = 0; r < ROUNDS; ++r) {
                    const uint32_t blk = 4 * r + k < NC ? (uint32_t)(4 * r + k) : TAG_UNIFORM;
                    const U4 w4 = block(blk);
                    float nrm[4];
                    box_muller(w4.x, w4.y, nrm[0], nrm[1]);
                    box_muller(w4.z, w4.w, nrm[2], nrm[3]);
                    if constexpr (NC < 4) logu_f = from(c + 32, log_u01(w4.x));
#pragma unroll
                    for (int hh = 0; hh < 4; ++hh) {
                        const int h = 4 * r + hh;
                        if (h < NC) {
                            const float e0 = from(c + 16 * hh, nrm[0]), e1 = from(c + 16 * hh, nrm[1]);
                            const float e2 = from(c + 16 * hh, nrm[2]), e3 = from(c + 16 * hh, nrm[3]);
                            z[h] = k < 2 ? (k == 0 ? e0 : e1) : (k == 2 ? e2 : e3);
                        }
                    }
                }
                if constexpr (NC >= 4) logu_f = log_u01(block(TAG_UNIFORM).x);
            }
            if constexpr (KIND == KIND_UL) {
#pragma unroll
                for (int h = 0; h < NC; ++h) x[h] = fma_t(kb[h], z[h], fma_t(ka[h], g[h], x[h]));
                double d0;
                evaluate(False{}, x, g, d0);
                ++nacc;
            } else {
                const double logu = (double)logu_f;
                float xp[NC], gp[NC];
                double llp = 0, lprp = 0, logr;
                if constexpr (KIND == KIND_RWMH) {
#pragma unroll
                    for (int h = 0; h < NC; ++h) xp[h] = fma_t(ka[h], z[h], x[h]);
                    evaluate(True{}, xp, gp, llp);
                    lprp = lprior_of(xp);
                    logr = (llp + lprp) - lp;
                } else if constexpr (KIND == KIND_MALA) {
                    float advx[NC];
#pragma unroll
                    for (int h = 0; h < NC; ++h) {
                        advx[h] = fma_t(ka[h], g[h], x[h]);
                        xp[h] = fma_t(kb[h], z[h], advx[h]);
                    }
                    evaluate(True{}, xp, gp, llp);
                    lprp = lprior_of(xp);
                    float dq = 0.0f;
#pragma unroll
                    for (int h = 0; h < NC; ++h) {
                        const float advp = fma_t(ka[h], gp[h], xp[h]);
                        const float d1 = x[h] - advp, d2 = xp[h] - advx[h];
                        dq = fma_t(kc[h], d1 * d1 - d2 * d2, dq);
                    }
                    logr = (llp + lprp) - lp - 0.5 * (double)ksum(dq);
                } else {  // HMC
                    float pm[NC];
                    float k0 = 0.0f;
#pragma unroll
                    for (int h = 0; h < NC; ++h) {
                        pm[h] = z[h] * ka[h];
                        k0 = fma_t(pm[h] * pm[h], kc[h], k0);
                        xp[h] = x[h];
                        gp[h] = g[h];
                    }
                    const float heps = 0.5f * a.step;
#pragma unroll
                    for (int h = 0; h < NC; ++h) pm[h] = fma_t(heps, gp[h], pm[h]);
                    if (a.interior_bf16) {
                        // the whole interior loop once per live pair count (wave-uniform; every wave of a row-split
                        // workgroup still meets the same l - 1 barriers)
                        for_pair_count<Rows::NPAIR>((rows.ntile_live + 1) >> 1, [&](auto npl) {
                            for (int i = 0; i < a.l - 1; ++i) {
#pragma unroll
                                for (int h = 0; h < NC; ++h) xp[h] = fma_t(kb[h], pm[h], xp[h]);
                                evaluate_interior(npl, xp, gp);
#pragma unroll
                                for (int h = 0; h < NC; ++h) pm[h] = fma_t(a.step, gp[h], pm[h]);
                            }
                        });
                    } else {
                        for (int i = 0; i < a.l - 1; ++i) {
#pragma unroll
                            for (int h = 0; h < NC; ++h) xp[h] = fma_t(kb[h], pm[h], xp[h]);
                            double d0;
                            evaluate(False{}, xp, gp, d0);
#pragma unroll
                            for (int h = 0; h < NC; ++h) pm[h] = fma_t(a.step, gp[h], pm[h]);
                        }
                    }
#pragma unroll
                    for (int h = 0; h < NC; ++h) xp[h] = fma_t(kb[h], pm[h], xp[h]);
                    evaluate(True{}, xp, gp, llp);
                    lprp = lprior_of(xp);
                    float k1 = 0.0f;
#pragma unroll
                    for (int h = 0; h < NC; ++h) {
                        pm[h] = fma_t(heps, gp[h], pm[h]);
                        k1 = fma_t(pm[h] * pm[h], kc[h], k1);
                    }
                    logr = ((llp + lprp) - lp) - 0.5 * (double)ksum(k1 - k0);
                }
                const bool acc = logu < logr;
                if (acc) {
                    ++nacc;
                    lp = llp + lprp;
                }
#pragma unroll
                for (int h = 0; h < NC; ++h) {
                    x[h] = acc ? xp[h] : x[h];
                    if constexpr (KIND != KIND_RWMH) g[h] = acc ? gp[h] : g[h];
                }
            }
        }
        if (a.out && writer) {
            float* o = a.out + (it * a.C + chain) * a.p;
#pragma unroll
            for (int h = 0; h < NC; ++h)
                if (k + 4 * h < a.p) o[k + 4 * h] = x[h];
        }
        if (a.stats.buf && writer) {  // the lane owns coordinates k + 4h
            const int64_t idx = a.stats.first + it, sb = idx / a.stats.batch, sk = idx - sb * a.stats.batch;
            const double inv = 1.0 / (double)(sk + 1);
            double* s = a.stats.buf + ((sb * a.C + chain) * 2) * a.p;
#pragma unroll
            for (int h = 0; h < NC; ++h)
                if (k + 4 * h < a.p) stats_fold(s + k + 4 * h, s + a.p + k + 4 * h, sk, inv, (double)x[h]);
        }
    }
    if (writer) {
#pragma unroll
        for (int h = 0; h < NC; ++h)
            if (k + 4 * h < a.p) a.state[chain * a.p + k + 4 * h] = x[h];
        if (k == 0) {
            if (a.accepts) a.accepts[chain] += nacc;
            if constexpr (KIND == KIND_RWMH || KIND == KIND_MALA) a.lp_state[chain] = lp;
        }
    }
}

}  // namespace lr
