// lr_tall.h -- "stepwise" engine for data that do not fit on chip (tall n): the leapfrog /
// proposal loop is driven from the host as a sequence of two small kernels per log-posterior
// evaluation, so that the rows can be split across the WHOLE chip for every evaluation:
//
//   k_tall_partial  grid (chains/256, RS row slices): lane per chain, the slice's rows broadcast
//                   from the scalar unit (s_load -> SGPR operands of v_fmac); writes the partial
//                   gradient / value of its slice:  part_g[RS][C][P], part_v[RS][C].
//                   X is read C/256 times per evaluation in total (not once per chain).
//   k_tall_update   P lanes per chain (lane = coordinate): sums the RS partials in a fixed order
//                   (deterministic, no atomics), adds the prior, and advances the chain state
//                   machine by one phase (leapfrog kick/drift, proposal, accept/reject, sample).
//
// Launch boundaries provide the device-wide synchronisation between "all slices reduced" and
// "next position known"; at tall n one evaluation is tens of microseconds, so the two
// boundaries per evaluation (~1.5 us each) cost a few percent.  Same Philox stream, same
// arithmetic per row, same accept rule as the fused kernels (lr_kernels.h).
#pragma once
#include <type_traits>
#include "lr_kernels.h"

namespace lr {

enum TallPhase { PH_LOAD = 0, PH_INIT = 1, PH_MID = 2, PH_END = 3, PH_STORE = 4, PH_EVAL = 5 };

// sum over the P coordinate-lanes of one chain (lane = coordinate, 256-thread blocks).
// P <= 64: the chain sits inside one wave (DPP butterfly).  P > 64: the chain spans P/64 waves
// of the block; wave sums are combined through LDS in wave order (identical in every lane).
// Must be called by all threads of the block (it synchronises).
template <int P, typename T> __device__ __forceinline__ T chain_sum(T v) {
    if constexpr (P <= 64) {
        return group_sum<P>(v);
    } else {
        __shared__ double part[4];
        constexpr int WPC = P / 64;  // waves per chain
        const int wave = threadIdx.x >> 6;
        const double ws = (double)group_sum<64>(v);
        __syncthreads();  // previous use of `part` is over
        if ((threadIdx.x & 63) == 0) part[wave] = ws;
        __syncthreads();
        const int w0 = (wave / WPC) * WPC;
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < WPC; ++i) s += part[w0 + i];
        return (T)s;
    }
}

template <typename T, int P> struct TallArgs {
    const T* rows;  // [n][P] signed rows
    const float* rows_tw;  // float32: the rows as twisted row pairs (lr::ScalarRowPairs); slices are even
    int64_t n, slice_len;
    int RS;
    Prior<T, P> prior;
    // workspace (device), padded layout [C][P]
    T* x;        // current state
    T* g;        // gradient at x
    double* lp;  // log-density attached to x
    T* q1;       // point being evaluated (trajectory position / proposal)
    T* pm;       // HMC momentum | MALA advance(x)
    double* aux; // HMC: initial kinetic energy
    uint32_t* nacc;
    T* part_g;       // [RS][C][P]
    double* part_v;  // [RS][C]
    // caller's arrays (device)
    T* state;
    double* lp_state;
    T* out;
    uint32_t* accepts;
    // lr_eval outputs (PH_EVAL), unpadded; any may be null
    T* ev_ll;
    T* ev_lprior;
    T* ev_lpost;
    T* ev_grad;
    int64_t C, chain_offset;
    uint64_t seed;
    int wide_bf16;  // wide models: 0 = fp32 MFMA partial kernel, 1 = exact-split bf16 MFMA partial kernel
    const uint16_t* xblk;  // wide bf16: per-32-row-block LDS images of the split rows (lr_wide_bf16.h)
    const uint16_t* xblk1;  // wide bf16: single-piece (round-to-nearest) images for interior leapfrog steps
    const uint16_t* xblk1h; // ... and their half-precision (f16) twin; null when the rows do not fit f16
    const uint16_t* xmx;    // narrow models (P = 8, float32): two-piece bf16 tile images (lr_tall_mx.h), else null
    int interior;  // this launch is an interior HMC gradient evaluation that may run in reduced precision
    int part_f32;  // float64 models: part_g holds FLOAT32 partials [RS][C][P] (written by a reduced-precision interior kernel)
    // fused interior step (k_wide_partial_bf16r, fuse_mid = 1): the kernel first finishes the PREVIOUS leapfrog step
    // itself -- kick with the slice partials in part_in, drift -- from the state in (q1_in, pm_in), stores the new state
    // to (q1, pm) (slice 0 only; ping-pong buffers, so nobody reads what is being written) and then evaluates there
    const T* part_in;
    const T* q1_in;
    const T* pm_in;
    const T* cvec;          // [2][P]: drift factors b[j] = eps / dmm[j], prior precisions (device copy for per-lane reads)
    int fuse_mid;
    int RS_i;               // row-split interior kernel (k_wide_partial_bf16r): slices, 0 = not used for this run
    int64_t slice_len_i;    //   and rows per slice (a multiple of 32 * rowsplit_waves: whole 32-row blocks per wave)
    int rowsplit_waves;     //   4 or 8 waves per workgroup (wide); 16: the 16-wave tall kernel k_tall_partial_mx16 is in use
    int traj_tiles;         // wide models, trajectory kernel: chain tiles (16 chains) per workgroup of k_wide_traj2_bf16: 1 or 2
    int traj_fmt;           // wide models, operand format of the interior kernels: 0 = bf16 rows x two bf16 pieces of beta, 1 = x one piece
                            //     (LR_PREC_BF16, on the two-tile trajectory kernel only), 2 = f16 rows x one f16 piece of beta (xblk1h)
    int p, l;
    T step;
    T a[P], b[P], c[P];
    StatsArgs stats;  // streaming statistics of the kept samples (lr_device.h); buf = null: off
    LR_STAMP_FIELDS  // development builds only (lr_stamps.h)
};


// ---------------------------------------------------------------------------------------------
// Workgroup = NW waves x 64 chains: wave w of the group takes the w-th sub-slice of the group's
// row slice; the NW partial results are combined through LDS in wave order, so only RS partials
// per (chain, coordinate) reach memory.
template <typename T, int P> struct TallGeom {
    static constexpr int kRaw = 2048 / (P * (int)sizeof(T));
    // LDS = NW*64*P*sizeof(T) <= 128 KB.  16 waves (1024 threads: 128 VGPRs per lane) only while the lane's three P-vectors (position,
    // fp32 and fp64 gradient sums: 4 P registers) leave room for the row pass -- at P = 32 they are the whole budget and the kernel
    // spilled 13-36 registers to scratch (round 5: tools/kernel_resources.py; no kernel of the library may use scratch)
    static constexpr int NW = (kRaw >= 16 && P <= 16) ? 16 : (kRaw >= 8 ? 8 : 4);
};

template <typename T, int P, bool VALUE, bool GRAD>
__global__ void __launch_bounds__((64 * TallGeom<T, P>::NW)) k_tall_partial(TallArgs<T, P> a) {
    constexpr int NW = TallGeom<T, P>::NW;
    __shared__ T red_g[GRAD ? NW : 1][64][P];
    __shared__ double red_v[VALUE ? NW : 1][64];
    const int lane = threadIdx.x & 63;
    // wave-uniform by construction; readfirstlane makes it PROVABLY uniform so the row addresses
    // stay scalar and the rows are fetched with s_load (not 64-fold redundant vector loads)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // grid = (row slices, chain blocks): workgroups are dealt to the 8 XCDs round-robin by linear id, so with the
    // SLICE index fastest every XCD works on its own eighth of the slices for all chain blocks and its L2 holds an
    // eighth of the rows; with the chain block fastest (round 1) every XCD streamed all of X through its 4 MB L2
    // for every evaluation (FETCH_SIZE 25 MB per launch against 3.6 MB of rows)
    const int cb = blockIdx.y;
    int64_t chain = (int64_t)cb * 64 + lane;
    const bool live = chain < a.C;
    if (!live) chain = a.C - 1;
    const int rs = blockIdx.x;
    const int64_t sub = ((a.slice_len + NW - 1) / NW + 1) & ~(int64_t)1;  // even: row pairs never straddle waves
    const int64_t s0 = (int64_t)rs * a.slice_len, s1 = s0 + a.slice_len < a.n ? s0 + a.slice_len : a.n;
    T bs[P], g[P];
    double gd[P];  // blocked summation: fp32 over 16 rows, then fp64 -- a sub-slice can be thousands of
                   // rows of strongly cancelling terms; a plain fp32 running sum loses ~sqrt(rows) ulps
#pragma unroll
    for (int j = 0; j < P; ++j) {
        bs[j] = a.q1[chain * P + j] * ExpScale<T>::k;
        g[j] = T(0);
        gd[j] = 0.0;
    }
    double v = 0.0;
    const int64_t i0 = s0 + wave * sub, i1 = i0 + sub < s1 ? i0 + sub : s1;
    if constexpr (sizeof(T) == 4 && P <= 16) {
        // float32: twisted row pairs through the scalar unit (slices and sub-slices are even)
        ScalarRowPairs<P> rows;
        rows.base = a.rows_tw;
        rows.k0 = i0 / 2;
        rows.k1 = i1 > i0 ? (i1 + 1) / 2 : rows.k0;
        rows.zero_rows = i1 > i0 ? (int)(i1 & 1) : 0;
        f32x2 bb[P / 2], gp[P / 2], hp[P / 2], vacc = {0.0f, 0.0f};
#pragma unroll
        for (int j = 0; j < P / 2; ++j) {
            bb[j] = f32x2{bs[2 * j], bs[2 * j + 1]};
            gp[j] = hp[j] = f32x2{0.0f, 0.0f};
        }
        auto flush = [&]() {
#pragma unroll
            for (int j = 0; j < P / 2; ++j) {
                if constexpr (GRAD) {
                    gd[2 * j] += (double)(gp[j].x + hp[j].y);
                    gd[2 * j + 1] += (double)(gp[j].y + hp[j].x);
                }
                gp[j] = hp[j] = f32x2{0.0f, 0.0f};
            }
            if constexpr (VALUE) v += (double)(vacc.x + vacc.y);
            vacc = f32x2{0.0f, 0.0f};
        };
        // blocks of 8 pairs (16 rows) between fp64 flushes; inside a block PFP pairs (64 SGPRs) are fetched
        // per wait, so the SMEM latency (~500 cycles from L2) is paid once per PFP pairs and the pair terms
        // of a batch interleave
        typedef const __attribute__((address_space(4))) f32x2* cptr;
        cptr cb = (cptr)rows.base;
        constexpr int BLK = 8, PFP = P <= 4 ? 8 : (P <= 8 ? 4 : 2);
        int64_t k = rows.k0;
        for (; k + BLK <= rows.k1; k += BLK) {
#pragma unroll
            for (int u0 = 0; u0 < BLK; u0 += PFP) {
                f32x2 q[PFP][P];
#pragma unroll
                for (int u = 0; u < PFP; ++u)
#pragma unroll
                    for (int j = 0; j < P; ++j) q[u][j] = cb[(k + u0 + u) * P + j];
#pragma unroll
                for (int u = 0; u < PFP; ++u) pair_term<P, VALUE, GRAD>(q[u], bb, gp, hp, vacc);
                // one batch in flight: hoisting the next batch's loads above this point needs more than the
                // ~100 SGPRs there are (34 spilled to VGPR lanes, +17 % instructions); the other waves of
                // the SIMD cover the SMEM latency instead
                __builtin_amdgcn_sched_barrier(0);
            }
            flush();
        }
        for (; k < rows.k1; ++k) {
            f32x2 q[P];
#pragma unroll
            for (int j = 0; j < P; ++j) q[j] = cb[k * P + j];
            pair_term<P, VALUE, GRAD>(q, bb, gp, hp, vacc);
        }
        flush();
        if constexpr (VALUE) v = (v + (double)rows.zero_rows) * (double)ExpScale<float>::inv;  // log2 units -> nats
    } else {
        ScalarRows<T, P, (P * sizeof(T) <= 64 ? 4 : 2)> rows;
        rows.base = a.rows;
        rows.i0 = i0;
        rows.i1 = i1;
        int cnt = 0;
        rows.for_each([&](const T(&xs)[P]) {
            T vt = T(0);
            row_term<T, P, VALUE, GRAD>(xs, bs, g, vt);
            if constexpr (VALUE) v += (double)vt;
            if constexpr (GRAD && sizeof(T) == 4) {
                if (++cnt == 16) {
                    cnt = 0;
#pragma unroll
                    for (int j = 0; j < P; ++j) {
                        gd[j] += (double)g[j];
                        g[j] = T(0);
                    }
                }
            }
        });
    }
    if constexpr (GRAD) {
#pragma unroll
        for (int j = 0; j < P; ++j) red_g[wave][lane][j] = (T)(gd[j] + (double)g[j]);
    }
    if constexpr (VALUE) red_v[wave][lane] = v;
    __syncthreads();
    if constexpr (GRAD) {
        for (int e = threadIdx.x; e < 64 * P; e += 64 * NW) {  // element e = (chain c, coordinate j)
            const int c = e / P, j = e % P;
            double s = 0.0;
#pragma unroll
            for (int w = 0; w < NW; ++w) s += (double)red_g[w][c][j];
            const int64_t ch = (int64_t)cb * 64 + c;
            if (ch < a.C) a.part_g[((int64_t)rs * a.C + ch) * P + j] = (T)s;
        }
    }
    if constexpr (VALUE) {
        if (wave == 0 && live) {
            double s = 0.0;
#pragma unroll
            for (int w = 0; w < NW; ++w) s += red_v[w][lane];
            a.part_v[(int64_t)rs * a.C + chain] = s;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// one coordinate per lane; chain = (global thread) / P
template <typename T, int P, int KIND>
__global__ void __launch_bounds__(256) k_tall_update(TallArgs<T, P> a, int phase, int64_t iter, int64_t out_row,
                                                     int begin_next) {
    const int64_t gt = (int64_t)blockIdx.x * 256 + threadIdx.x;
    int64_t chain = gt / P;
    const int j = (int)(gt % P);
    const bool live = chain < a.C;
    if (!live) chain = a.C - 1;
    const int64_t ix = chain * P + j;
    const uint64_t gchain = (uint64_t)(a.chain_offset + chain);
    const T aj = a.a[j], bj = a.b[j], cj = a.c[j], ivj = a.prior.inv_var[j];

    auto reduced_grad = [&]() {  // likelihood partials in slice order + prior
        // slice order, 16 loads in flight (a plain loop issues them one L2 round trip at a time: this kernel is
        // nothing but that latency)
        auto slices = [&](auto* pg) {
            using E = std::remove_cv_t<std::remove_pointer_t<decltype(pg)>>;
            double s = 0.0;
            const int64_t stride = a.C * P;
            int r = 0;
            for (; r + 16 <= a.RS; r += 16) {
                E t[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) t[u] = pg[(r + u) * stride];
#pragma unroll
                for (int u = 0; u < 16; ++u) s += (double)t[u];
            }
            for (; r + 4 <= a.RS; r += 4) {
                E t[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) t[u] = pg[(r + u) * stride];
#pragma unroll
                for (int u = 0; u < 4; ++u) s += (double)t[u];
            }
            for (; r < a.RS; ++r) s += (double)pg[r * stride];
            return s;
        };
        // (float64 models after a reduced-precision interior kernel: that kernel's partials are float32 values, written as such)
        const double s = sizeof(T) == 8 && a.part_f32 ? slices(reinterpret_cast<const float*>(a.part_g) + chain * P + j) : slices(a.part_g + chain * P + j);
        return (T)s - a.q1[ix] * ivj;
    };
    auto reduced_value = [&]() {  // lpost(q1) = sum of slice values + lprior(q1)
        double s = 0.0;
        for (int r = 0; r < a.RS; ++r) s += a.part_v[(int64_t)r * a.C + chain];
        const T q = a.q1[ix];
        const T quad = chain_sum<P>(q * q * ivj);
        return s + a.prior.lprior_const - 0.5 * (double)quad;
    };
    auto normal_j = [&](uint64_t it) {  // coordinate j: Philox block j/4, word pair (j%4)/2, element j&1
        const U4 w = philox4x32_10((uint32_t)gchain, (uint32_t)it, (uint32_t)(it >> 32), (uint32_t)(j >> 2),
                                   (uint32_t)a.seed, (uint32_t)(a.seed >> 32));
        const uint32_t wa = (j & 2) ? w.z : w.x, wb = (j & 2) ? w.w : w.y;
        T z0, z1;
        box_muller(wa, wb, z0, z1);
        return (j & 1) ? z1 : z0;
    };
    // start iteration `it` from (x, g, lp): draw, set q1 (+pm, aux)
    auto begin = [&](uint64_t it, T x, T g) {
        const T z = normal_j(it);
        if constexpr (KIND == KIND_HMC) {
            T p = z * aj;
            const T k0 = chain_sum<P>(p * p * cj);
            p = fma_t(T(0.5) * a.step, g, p);
            if (live) {
                a.pm[ix] = p;
                a.q1[ix] = fma_t(bj, p, x);
                if (j == 0) a.aux[chain] = (double)k0;
            }
        } else if constexpr (KIND == KIND_MALA) {
            const T advx = fma_t(aj, g, x);
            if (live) {
                a.pm[ix] = advx;
                a.q1[ix] = fma_t(bj, z, advx);
            }
        } else if constexpr (KIND == KIND_UL) {
            const T xn = fma_t(bj, z, fma_t(aj, g, x));
            if (live) {
                a.x[ix] = xn;
                a.q1[ix] = xn;
            }
        } else {
            if (live) a.q1[ix] = fma_t(aj, z, x);
        }
    };

    if (phase == PH_LOAD) {
        const T x = j < a.p ? a.state[chain * a.p + j] : T(0);
        if (live) {
            a.x[ix] = x;
            a.q1[ix] = x;
            if (j == 0) a.nacc[chain] = 0;
        }
        if (a.cvec && chain == 0) {  // per-coordinate constants where a lane can fetch them with an ordinary load
            const_cast<T*>(a.cvec)[j] = bj;
            const_cast<T*>(a.cvec)[P + j] = ivj;
        }
        return;
    }
    if (phase == PH_INIT) {  // after the evaluation at x
        T g = T(0);
        if constexpr (KIND != KIND_RWMH) g = reduced_grad();
        double lp;
        if constexpr (KIND == KIND_HMC) lp = reduced_value();
        else if constexpr (KIND == KIND_UL) lp = 0.0;
        else lp = a.lp_state[chain];
        if (live) {
            a.g[ix] = g;
            if (j == 0) a.lp[chain] = lp;
        }
        begin((uint64_t)iter, a.x[ix], g);
        return;
    }
    if (phase == PH_MID) {  // HMC interior step: kick with the new gradient, drift
        const T g1 = reduced_grad();
        const T p = fma_t(a.step, g1, a.pm[ix]);
        if (live) {
            a.pm[ix] = p;
            a.q1[ix] = fma_t(bj, p, a.q1[ix]);
        }
        return;
    }
    if (phase == PH_END) {
        T x = a.x[ix], g = a.g[ix];
        if constexpr (KIND == KIND_UL) {
            g = reduced_grad();
            if (live) {
                a.g[ix] = g;
                if (j == 0) a.nacc[chain] += 1;
            }
        } else {
            const T q1 = a.q1[ix];
            T g1 = T(0);
            if constexpr (KIND != KIND_RWMH) g1 = reduced_grad();
            const double lp1 = reduced_value();
            const double lp = a.lp[chain];
            double logr;
            if constexpr (KIND == KIND_HMC) {
                const T p = fma_t(T(0.5) * a.step, g1, a.pm[ix]);
                const T k1 = chain_sum<P>(p * p * cj);
                logr = (lp1 - lp) - 0.5 * ((double)k1 - a.aux[chain]);
            } else if constexpr (KIND == KIND_MALA) {
                const T advp = fma_t(aj, g1, q1);
                const T d1 = x - advp, d2 = q1 - a.pm[ix];
                const T dq = chain_sum<P>(cj * (d1 * d1 - d2 * d2));
                logr = (lp1 - lp) - 0.5 * (double)dq;
            } else {
                logr = lp1 - lp;
            }
            const double logu = (double)draw_log_uniform<T>(a.seed, gchain, (uint64_t)iter);
            const bool acc = logu < logr;
            if (acc) {
                x = q1;
                g = g1;
            }
            if (live && acc) {
                a.x[ix] = x;
                if constexpr (KIND != KIND_RWMH) a.g[ix] = g;
                if (j == 0) {
                    a.lp[chain] = lp1;
                    a.nacc[chain] += 1;
                }
            }
        }
        if (out_row >= 0 && live && j < a.p && a.out) a.out[(out_row * a.C + chain) * a.p + j] = x;
        if (out_row >= 0 && live && j < a.p && a.stats.buf) {  // lane = coordinate: one (mean, M2) pair each
            const int64_t idx = a.stats.first + out_row, sb = idx / a.stats.batch, sk = idx - sb * a.stats.batch;
            double* s = a.stats.buf + ((sb * a.C + chain) * 2) * a.p;
            stats_fold(s + j, s + a.p + j, sk, 1.0 / (double)(sk + 1), (double)x);
        }
        if (begin_next) begin((uint64_t)iter + 1, x, g);
        return;
    }
    if (phase == PH_EVAL) {  // ll / lprior / lpost / glp at q1 (lr_eval through the stepwise engine)
        const T gq = reduced_grad();
        double s = 0.0;
        for (int r = 0; r < a.RS; ++r) s += a.part_v[(int64_t)r * a.C + chain];
        const T q = a.q1[ix];
        const double lpr = a.prior.lprior_const - 0.5 * (double)chain_sum<P>(q * q * ivj);
        if (live) {
            if (a.ev_grad && j < a.p) a.ev_grad[chain * a.p + j] = gq;
            if (j == 0) {
                if (a.ev_ll) a.ev_ll[chain] = (T)s;
                if (a.ev_lprior) a.ev_lprior[chain] = (T)lpr;
                if (a.ev_lpost) a.ev_lpost[chain] = (T)(s + lpr);
            }
        }
        return;
    }
    if (phase == PH_STORE) {
        if (live) {
            if (j < a.p) a.state[chain * a.p + j] = a.x[ix];
            if (j == 0) {
                if (a.accepts) a.accepts[chain] += a.nacc[chain];
                if constexpr (KIND == KIND_RWMH || KIND == KIND_MALA) a.lp_state[chain] = a.lp[chain];
            }
        }
    }
}

}  // namespace lr
