// lr_chain8.h -- MALA / RWMH with EIGHT lanes per chain (float, padded p = 8): 8 chains per wave, the chain state distributed
// over the group for the whole launch (lane gl owns coordinate gl), the lane's rows partly in VGPRs and partly in LDS.
//
// Why.  k_chain_rs16 (16 lanes per chain, lr_kernels.h) is issue-bound: of its ~270 instructions per wave-iteration the row pass
// is 73 %, the draws 16 %, reductions + bookkeeping 11 % (profiles/r3_mala_phases.txt) -- and everything but the row pass is
// paid per WAVE, i.e. per 4 chains.  With 8 lanes per chain the same per-wave work serves 8 chains, the reductions are one
// level shorter, and 8 x 25 rows cover n = 200 exactly (16 x 13 pads 4 %).  The price is 25 rows per lane: 200 registers as
// twisted row pairs, which do not fit beside the state.  So the residency is split: the first RPV row pairs stay in VGPRs
// (RegRowPairs layout), the rest -- identical for every chain of the workgroup -- sit in LDS once per workgroup and are read
// back per evaluation, one ds_read_b128 per quarter pair, 8 distinct 16-byte chunks per instruction (conflict-free), each
// broadcast to the 8 chains of the wave.  8192 chains are then ONE wave per SIMD (512 registers available, 256 addressable).
//
// Same Philox stream, same accept rule, same arithmetic per row (pair_term) as k_chain / k_chain_rs16; the summation ORDER of a
// chain's rows differs (8 partial sums of 25 rows instead of 16 of 13), so results agree with the other variants statistically and
// to rounding, not bit for bit -- as between any two variants.  Reference: fit-np-mala.py:61-78, fit-numpy.py:53-62.
#pragma once
#include <type_traits>

#include "lr_kernels.h"
#include "lr_mfma.h"  // f32x4

namespace lr {

// sum over the 8 lanes of a group by symmetric exchanges (i <-> 7 - i, i <-> i ^ 2, i <-> i ^ 1): every lane adds the same two
// numbers at each level, so the result is bit-identical in all 8 lanes
__device__ __forceinline__ float group8_sum(float v) {
    v += dpp_mov<0x141>(v);  // row_half_mirror
    v += dpp_mov<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_mov<0xB1>(v);   // quad_perm [1,0,3,2]
    return v;
}

// 8 values over 8 lanes -> lane gl keeps the total of value gl (reduce-scatter: 8 + 6 + 3 instructions instead of the
// all-reduce's 24).  Level 1 halves the values with bank-masked DPP adds (banks = quads: lanes 0-3 of a group keep v0..v3,
// lanes 4-7 keep v4..v7); levels 2 and 3 exchange inside a quad, where no bank mask can select, so every lane SENDS the value
// its partner keeps and keeps the other.  hi2 = gl & 2, odd = gl & 1.
__device__ __forceinline__ float group8_reduce_scatter8(const float (&v)[8], bool hi2, bool odd) {
    float r0, r1, r2, r3;
    asm volatile(
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %4, %4 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %1, %5, %5 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %2, %6, %6 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %3, %7, %7 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %0, %8, %8 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %1, %9, %9 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %2, %10, %10 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %3, %11, %11 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "s_nop 1\n\t"
        : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3)
        : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]));
    const float k0 = hi2 ? r2 : r0, s0 = hi2 ? r0 : r2, k1 = hi2 ? r3 : r1, s1 = hi2 ? r1 : r3;
    const float u0 = k0 + dpp_mov<0x4E>(s0), u1 = k1 + dpp_mov<0x4E>(s1);
    const float k = odd ? u1 : u0, s = odd ? u0 : u1;
    return k + dpp_mov<0xB1>(s);
}

// every lane of an 8-lane group gets the values of all 8: value j comes from lane j of the group.  A 16-lane DPP row holds two
// groups, so each value takes two bank-masked row_share moves (lanes 0-7 <- lane j, lanes 8-15 <- lane 8 + j).
template <int J> __device__ __forceinline__ float group8_bcast(float v) {
    const int x = __builtin_bit_cast(int, v);
    int r = __builtin_amdgcn_update_dpp(0, x, 0x150 + J, 0xF, 0x3, false);
    r = __builtin_amdgcn_update_dpp(r, x, 0x158 + J, 0xF, 0xC, false);
    return __builtin_bit_cast(float, r);
}
__device__ __forceinline__ void group8_allgather(float mine, f32x2 (&bb)[4]) {
    bb[0] = f32x2{group8_bcast<0>(mine), group8_bcast<1>(mine)};
    bb[1] = f32x2{group8_bcast<2>(mine), group8_bcast<3>(mine)};
    bb[2] = f32x2{group8_bcast<4>(mine), group8_bcast<5>(mine)};
    bb[3] = f32x2{group8_bcast<6>(mine), group8_bcast<7>(mine)};
}

// The lane's R rows (gl, gl + 8, ...): pairs 0 .. RPV-1 in registers, pairs RPV .. R/2-1 and the odd last row in LDS.
// LDS image (shared by the workgroup; lanes with equal gl read equal addresses):
//     pair j >= RPV, quarter qd (coordinates 2qd, 2qd+1 -> the pair's q[2qd], q[2qd+1]), lane gl :  ((j - RPV) * 4 + qd) * 8 + gl   (x 16 B)
//     odd row, half h (coordinates 4h .. 4h+3 as s[2h], s[2h+1])                                 :  NLP * 32 + h * 8 + gl
template <int R, int RPV> struct Rows8 {
    static constexpr int P = 8, G = 8, RP = R / 2, NLP = RP - RPV;
    static constexpr bool ODD = (R & 1) != 0;
    static_assert(RPV >= 0 && RPV <= RP, "register-resident pairs");
    static constexpr int kLdsChunks = NLP * 32 + (ODD ? 16 : 0);  // 16-byte chunks
    static constexpr int kLdsBytes = kLdsChunks * 16;
    f32x2 q[RPV > 0 ? RPV : 1][P];
    const f32x4* img;  // LDS
    int gl;
    int pad_rows;

    // called by every thread of the workgroup (synchronises)
    __device__ __forceinline__ void load(const float* __restrict__ rows, int64_t n, int gl_, f32x4* smem) {
        gl = gl_;
        img = smem;
        auto at = [&](int lane_gl, int k, int j) {  // element j of lane_gl's k-th row (zero beyond n): clamped load, then mask
            const int64_t i = lane_gl + (int64_t)k * G;
            const float v = rows[(i < n ? i : n - 1) * P + j];
            return i < n ? v : 0.0f;
        };
        pad_rows = 0;
#pragma unroll
        for (int k = 0; k < R; ++k)
            if (gl + (int64_t)k * G >= n) ++pad_rows;
#pragma unroll
        for (int k = 0; k < RPV; ++k)
#pragma unroll
            for (int j = 0; j < P; j += 2) {
                q[k][j] = f32x2{at(gl, 2 * k, j), at(gl, 2 * k + 1, j + 1)};
                q[k][j + 1] = f32x2{at(gl, 2 * k, j + 1), at(gl, 2 * k + 1, j)};
            }
        for (int ch = threadIdx.x; ch < kLdsChunks; ch += blockDim.x) {
            f32x4 v;
            if (ch < NLP * 32) {
                const int lg = ch & 7, qd = (ch >> 3) & 3, kk = RPV + (ch >> 5), j = 2 * qd;
                v = f32x4{at(lg, 2 * kk, j), at(lg, 2 * kk + 1, j + 1), at(lg, 2 * kk, j + 1), at(lg, 2 * kk + 1, j)};
            } else {
                const int c2 = ch - NLP * 32, lg = c2 & 7, h = c2 >> 3;
                v = f32x4{at(lg, R - 1, 4 * h), at(lg, R - 1, 4 * h + 1), at(lg, R - 1, 4 * h + 2), at(lg, R - 1, 4 * h + 3)};
            }
            smem[ch] = v;
        }
        __syncthreads();
    }
    __device__ __forceinline__ void lds_pair(int jl, f32x2 (&out)[P]) const {  // pair RPV + jl
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const f32x4 v = img[(jl * 4 + qd) * 8 + gl];
            out[2 * qd] = f32x2{v[0], v[1]};
            out[2 * qd + 1] = f32x2{v[2], v[3]};
        }
    }
    __device__ __forceinline__ void lds_odd(f32x2 (&s)[P / 2]) const {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const f32x4 v = img[NLP * 32 + h * 8 + gl];
            s[2 * h] = f32x2{v[0], v[1]};
            s[2 * h + 1] = f32x2{v[2], v[3]};
        }
    }
    __device__ __forceinline__ float value_fixup() const { return float(pad_rows) * 0.693147180559945309f; }

    // all rows of the lane: gradient partial sums into gp[4] = (g_j, g_{j+1}) pairs, value (natural-log units) added to v.
    // The value as in row_pairs_eval: sum of ts minus log2 of the PRODUCT of the lane's 1 + 2^ts factors, per-lane fall-back to
    // per-row logs when the product overflowed.
    template <bool VALUE, bool GRAD>
    __device__ __forceinline__ void eval(const f32x2 (&bb)[P / 2], f32x2 (&gp)[P / 2], float& v) const {
        f32x2 hp[P / 2];
#pragma unroll
        for (int j = 0; j < P / 2; ++j) gp[j] = hp[j] = f32x2{0.0f, 0.0f};
        f32x2 vacc = {0.0f, 0.0f}, pacc = {1.0f, 1.0f};
        f32x2 nxt[P];
        if constexpr (NLP > 0) lds_pair(0, nxt);  // the first LDS pair travels under the register pairs
#pragma unroll
        for (int k = 0; k < RPV; ++k) pair_term<P, VALUE, GRAD, VALUE>(q[k], bb, gp, hp, vacc, &pacc);
#pragma unroll
        for (int jl = 0; jl < NLP; ++jl) {
            f32x2 cur[P];
#pragma unroll
            for (int j = 0; j < P; ++j) cur[j] = nxt[j];
            if (jl + 1 < NLP) lds_pair(jl + 1, nxt);
            pair_term<P, VALUE, GRAD, VALUE>(cur, bb, gp, hp, vacc, &pacc);
        }
        float vs = 0.0f, ts_odd = 0.0f;
        f32x2 so[P / 2];
        if constexpr (ODD) {
            lds_odd(so);
            f32x2 acc = so[0] * bb[0];
#pragma unroll
            for (int j = 1; j < P / 2; ++j) acc = __builtin_elementwise_fma(so[j], bb[j], acc);
            const float ts = acc.x + acc.y;
            const float d = 1.0f + ExpScale<float>::exp_scaled(ts);
            if constexpr (GRAD) {
                const float w = fast_rcp(d);
#pragma unroll
                for (int j = 0; j < P / 2; ++j) gp[j] = __builtin_elementwise_fma(f32x2{w, w}, so[j], gp[j]);
            }
            if constexpr (VALUE) {
                vs = ts;
                pacc.y *= d;
                ts_odd = ts;
            }
        }
        if constexpr (VALUE) {
            float val = ((vacc.x + vacc.y) + vs) - (__builtin_amdgcn_logf(pacc.x) + __builtin_amdgcn_logf(pacc.y));
            const bool bad = !(val > -3.0e38f);  // a product overflowed (-inf), or NaN input
            if (__builtin_amdgcn_ballot_w64(bad) != 0) {  // rare: per-row logs, selected per lane
                f32x2 va = {0.0f, 0.0f}, g0[P / 2], h0[P / 2];
#pragma unroll
                for (int k = 0; k < RPV; ++k) pair_term<P, true, false, false>(q[k], bb, g0, h0, va);
                for (int jl = 0; jl < NLP; ++jl) {
                    f32x2 cur[P];
                    lds_pair(jl, cur);
                    pair_term<P, true, false, false>(cur, bb, g0, h0, va);
                }
                float vo = 0.0f;
                if constexpr (ODD) {
                    const float tc = __builtin_fminf(ts_odd, 100.0f);
                    vo = tc - __builtin_amdgcn_logf(1.0f + ExpScale<float>::exp_scaled(tc));
                }
                const float safe = (va.x + va.y) + vo;
                val = bad ? safe : val;
            }
            v += val * ExpScale<float>::inv;
        }
        if constexpr (GRAD) {
#pragma unroll
            for (int j = 0; j < P / 2; ++j) gp[j] += __builtin_shufflevector(hp[j], hp[j], 1, 0);
        }
    }
};

// the draws of one iteration for a lane that owns ONE coordinate (gl) of a chain held by 8 lanes: z[gl] and log(u), from the
// batched generator (DrawBatch<float, 8, 8>: two iterations per generator pass)
__device__ __forceinline__ void draw8_next(DrawBatch<float, 8, 8>& d, uint64_t seed, uint64_t chain, uint64_t iter, int gl, float& z, float& logu) {
    typedef DrawBatch<float, 8, 8> D;
    if (d.pos >= D::NB) d.refill(seed, chain, iter, gl);  // wave-uniform
    const int base = ((int)(threadIdx.x & 63) - gl + d.pos * D::BPI) * 4;  // byte address of lane (group base + pos * BPI)
    const int src = base + 4 * (gl >> 2);                                   // block gl / 4 of this iteration
    const float e0 = D::fetch(d.mine[0], src), e1 = D::fetch(d.mine[1], src), e2 = D::fetch(d.mine[2], src), e3 = D::fetch(d.mine[3], src);
    const float lo = (gl & 1) ? e1 : e0, hi = (gl & 1) ? e3 : e2;
    z = (gl & 2) ? hi : lo;
    logu = D::fetch(d.lu, base + 4 * D::NBn);
    ++d.pos;
}

template <int R, int RPV, int KIND>
__global__ void __launch_bounds__(256) k_chain_rs8(ModelArgs<float, 8> m, ChainArgs<float, 8> a) {
    static_assert(KIND == KIND_MALA || KIND == KIND_RWMH, "threaded-ll kernels");
    constexpr int P = 8, G = 8;
    typedef Rows8<R, RPV> Rows;
    __shared__ __attribute__((aligned(16))) f32x4 smem[Rows::kLdsChunks > 0 ? Rows::kLdsChunks : 1];
    const int gl = threadIdx.x % G;
    int64_t chain = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) / G;
    const bool live = chain < a.C;
    if (!live) chain = a.C - 1;
    Rows rows;
    rows.load(m.rows, m.n, gl, smem);
    const uint64_t gchain = (uint64_t)(a.chain_offset + chain);
    const bool hi2 = (gl & 2) != 0, odd = (gl & 1) != 0;
    auto pick = [&](const float (&v)[8]) {  // element gl of a kernel-argument vector (uniform array, per-lane index: selects)
        const float a01 = odd ? v[1] : v[0], a23 = odd ? v[3] : v[2], a45 = odd ? v[5] : v[4], a67 = odd ? v[7] : v[6];
        const float lo = hi2 ? a23 : a01, hi = hi2 ? a67 : a45;
        return (gl & 4) ? hi : lo;
    };
    const float aq = pick(a.a), bq = pick(a.b), cq = pick(a.c), ivq = pick(m.prior.inv_var);
    float xq = gl < a.p ? a.state[chain * a.p + gl] : 0.0f, gq = 0.0f;

    // ll, lprior (replicated in the group) and -- GRAD -- the lpost gradient of the lane's coordinate, at the point whose gl-th
    // coordinate is pq
    auto evaluate = [&](auto want_value, auto want_grad, float pq, float& grad_q, double& ll, double& lpr) {
        constexpr bool VALUE = decltype(want_value)::value, GRAD = decltype(want_grad)::value;
        f32x2 bb[4], gpp[4];
        group8_allgather(pq * ExpScale<float>::k, bb);
        float v = 0.0f;
        rows.template eval<VALUE, GRAD>(bb, gpp, v);
        if constexpr (GRAD) {
            const float gv[8] = {gpp[0].x, gpp[0].y, gpp[1].x, gpp[1].y, gpp[2].x, gpp[2].y, gpp[3].x, gpp[3].y};
            grad_q = __builtin_fmaf(-pq, ivq, group8_reduce_scatter8(gv, hi2, odd));
        }
        if constexpr (VALUE) {
            ll = group_sum<G>((double)(v + rows.value_fixup()));
            lpr = m.prior.lprior_const - 0.5 * (double)group8_sum(pq * pq * ivq);
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
    for (int64_t it = 0; it < a.iters; ++it) {
        for (int64_t jt = 0; jt < a.thin; ++jt) {
            const uint64_t iter = (uint64_t)(a.iter_offset + it * a.thin + jt);
            float zq, logu_f;
            draw8_next(draws, a.seed, gchain, iter, gl, zq, logu_f);
            const double logu = (double)logu_f;
            float xp, gp = 0.0f;
            double llp = 0, lprp = 0, logr;
            if constexpr (KIND == KIND_RWMH) {
                xp = __builtin_fmaf(aq, zq, xq);  // prop = x + sd z                      fit-numpy.py:83-84
                evaluate(True{}, False{}, xp, gp, llp, lprp);
                logr = (llp + lprp) - lp;
            } else {
                const float advx = __builtin_fmaf(aq, gq, xq);  // advance(x)             fit-np-mala.py:76
                xp = __builtin_fmaf(bq, zq, advx);
                evaluate(True{}, True{}, xp, gp, llp, lprp);
                const float advp = __builtin_fmaf(aq, gp, xp);
                const float d1 = xq - advp, d2 = xp - advx;
                const float t = cq * __builtin_fmaf(-d2, d2, d1 * d1);
                logr = (llp + lprp) - lp - 0.5 * (double)group8_sum(t);
            }
            const bool acc = logu < logr;  // NaN -> reject
            if (acc) {
                ++nacc;
                lp = llp + lprp;
            }
            xq = acc ? xp : xq;
            if constexpr (KIND == KIND_MALA) gq = acc ? gp : gq;
        }
        if (live && gl < a.p) {  // the kept sample: every lane writes its own coordinate (32 contiguous bytes per chain)
            if (a.out) a.out[(it * a.C + chain) * a.p + gl] = xq;
            if (a.stats.buf) {
                const int64_t idx = a.stats.first + it, sb = idx / a.stats.batch, sk = idx - sb * a.stats.batch;
                double* s = a.stats.buf + ((sb * a.C + chain) * 2) * a.p;
                stats_fold(s + gl, s + a.p + gl, sk, 1.0 / (double)(sk + 1), (double)xq);
            }
        }
    }
    if (live) {
        if (gl < a.p) a.state[chain * a.p + gl] = xq;
        if (gl == 0) {
            if (a.accepts) a.accepts[chain] += nacc;
            a.lp_state[chain] = lp;
        }
    }
}

}  // namespace lr
