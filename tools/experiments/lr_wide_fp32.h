// lr_wide.h -- partial-evaluation kernel of the stepwise engine for WIDE models (32 < p <= 128):
// here X.beta over a block of chains is a genuine dense GEMM, done on the matrix cores with
// fp32-in / fp32-accumulate MFMA (v_mfma_f32_16x16x4_f32: bit-for-bit an fmaf chain, so the
// results stay in the fp32 parity class of the vector kernels).
//
// Workgroup = 4 waves = 64 chains; wave w owns chains 16w..16w+15 of the block.  Lane l = (c, k),
// c = l & 15 the chain, k = l >> 4; the lane OWNS coordinates {k + 4h : h < P/4} of its chain.
// The block walks its row slice in tiles of 16 rows, staged through LDS in two permutations:
//   LA[16][.]  eta operand :  lane reads LA[c][k*(P/4) + h]      = Xs[row c][k + 4h]       (A of eta MFMA h)
//   LG[16][.]  grad operand:  lane reads LG[4k+s][cidx*(P/16)+mb] = Xs[row 4k+s][16mb+cidx] (A of grad MFMA)
//              with cidx = (c >> 2) + 4 (c & 3): slot m = c of M-block mb holds coordinate 16 mb + cidx
//   eta  tile: E[16 rows x 16 chains]   = sum_h  A_h (16x4) . B_h (4x16),  B_h = the lane's own beta[k+4h]
//   grad tile: G_mb[16 coords x 16 chains] += sum_s A_{mb,s} (16x4) . W_s (4x16),  W_s = sigma(-E) register s
// so the eta accumulator registers ARE the B operand of the gradient GEMM (no transposes), and
// gradient register (mb, r) of lane (c, k) is coordinate 16 mb + k + 4 r of chain c.
#pragma once
#include "lr_mfma.h"
#include "lr_tall.h"

namespace lr {

template <int P> struct WideGeom {
    static constexpr int H = P / 4;    // coordinates per lane = eta MFMAs per tile
    static constexpr int MB = P / 16;  // gradient M-blocks
    // LDS layouts chosen so that every ds_read_b128 of a 16-lane service group touches 16 distinct
    // 4-bank slots (the first version, row-major with a +4 pad, had 2-way conflicts on every read):
    //   LA row = 4 k-blocks of 64 floats (H used) + 4 pad  -> bank(c, k, i) = 4c + 4i      (LDA = 260)
    //   LG row = MB/4 chunks x [16 cidx][4 mb]             -> bank(cidx, i) = 4 cidx        (LDG = 16 MB)
    static constexpr int LDA = 4 * 64 + 4;
    static constexpr int LDG = 16 * MB;
    static constexpr int SMEM = 2 * 16 * LDA + 2 * 16 * LDG;  // floats
};

template <int P, bool VALUE>
__global__ void __launch_bounds__(256) k_wide_partial(TallArgs<float, P> a) {
    using G = WideGeom<P>;
    // one LDS block: [LA buffers | LG buffers] during the row loop, then reused as the 64 x P
    // output tile so that the partial gradient leaves the CU as one contiguous 64*P*4-byte copy
    __shared__ __attribute__((aligned(16))) float smem[G::SMEM];
    float (*LA)[16][G::LDA] = reinterpret_cast<float (*)[16][G::LDA]>(smem);
    float (*LG)[16][G::LDG] = reinterpret_cast<float (*)[16][G::LDG]>(smem + 2 * 16 * G::LDA);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, k = lane >> 4;
    int64_t chain = (int64_t)blockIdx.x * 64 + 16 * wave + c;
    const bool live = chain < a.C;
    if (!live) chain = a.C - 1;
    const int rs = blockIdx.y;
    const int64_t s0 = (int64_t)rs * a.slice_len, s1 = s0 + a.slice_len < a.n ? s0 + a.slice_len : a.n;
    const int64_t ntiles = s1 > s0 ? (s1 - s0 + 15) / 16 : 0;

    // the lane's coordinates, pre-scaled by log2(e)
    float bs[G::H];
#pragma unroll
    for (int h = 0; h < G::H; ++h) bs[h] = a.q1[chain * P + k + 4 * h] * ExpScale<float>::k;
    f32x4 gacc[G::MB];
#pragma unroll
    for (int mb = 0; mb < G::MB; ++mb) gacc[mb] = f32x4{0, 0, 0, 0};
    double vsum = 0.0;
    const int cidx = (c >> 2) + 4 * (c & 3);

    // staging: thread t moves 16*P/256 consecutive floats of row t/16 of the tile
    constexpr int PER = 16 * P / 256;
    const int srow = tid >> 4, scol = (tid & 15) * PER;
    float stage[PER];
    auto fetch = [&](int64_t t) {  // branch-free: out-of-slice rows re-read the last row and are zeroed
        const int64_t r = s0 + 16 * t + srow;
        const int64_t rc = r < s1 ? r : s1 - 1;
        const float keep = r < s1 ? 1.0f : 0.0f;
#pragma unroll
        for (int i = 0; i < PER; ++i) stage[i] = a.rows[rc * P + scol + i] * keep;
    };
    auto deposit = [&](int buf) {
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            const int col = scol + i;
            LA[buf][srow][(col & 3) * 64 + (col >> 2)] = stage[i];
            const int mb = col >> 4;
            LG[buf][srow][(mb >> 2) * 64 + (col & 15) * 4 + (mb & 3)] = stage[i];
        }
    };
    if (ntiles > 0) {
        fetch(0);
        deposit(0);
    }
    __syncthreads();
    for (int64_t t = 0; t < ntiles; ++t) {
        const int buf = (int)(t & 1);
        if (t + 1 < ntiles) fetch(t + 1);  // global loads in flight under the MFMAs of tile t
        // ---- all LDS operand reads of the tile are issued up front (16 x ds_read_b128 in flight);
        //      the sched_barrier keeps hipcc from sinking them next to their consumers, where each
        //      group of 8 MFMAs would wait a full LDS round trip
        f32x4 av[G::H / 4], gv[4][G::MB / 4];
        {
            const f32x4* la4 = reinterpret_cast<const f32x4*>(&LA[buf][c][k * 64]);
#pragma unroll
            for (int i = 0; i < G::H / 4; ++i) av[i] = la4[i];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const f32x4* lg4 = reinterpret_cast<const f32x4*>(&LG[buf][4 * k + s][cidx * 4]);
#pragma unroll
                for (int i = 0; i < G::MB / 4; ++i) gv[s][i] = lg4[i * 16];  // chunk i is 64 floats further
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- eta = Xs . beta^T
        f32x4 e0 = {0, 0, 0, 0}, e1 = {0, 0, 0, 0};
#pragma unroll
        for (int h = 0; h < G::H; h += 2) {
            e0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[h / 4][h % 4], bs[h], e0, 0, 0, 0);
            e1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av[(h + 1) / 4][(h + 1) % 4], bs[h + 1], e1, 0, 0, 0);
        }
        float w[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float ts = e0[r] + e1[r];
            w[r] = fast_rcp(1.0f + __builtin_amdgcn_exp2f(ts));
            if constexpr (VALUE) {
                const int64_t row = s0 + 16 * t + 4 * k + r;
                const float ats = ts < 0.0f ? -ts : ts;
                const float lv = (ts < 0.0f ? ts * ExpScale<float>::inv : 0.0f) - log1p_unit(__builtin_amdgcn_exp2f(-ats));
                if (row < s1) vsum += (double)lv;
            }
        }
        // ---- grad += Xs^T . W
#pragma unroll
        for (int s = 0; s < 4; ++s) {
#pragma unroll
            for (int mb = 0; mb < G::MB; ++mb)
                gacc[mb] = __builtin_amdgcn_mfma_f32_16x16x4f32(gv[s][mb / 4][mb % 4], w[s], gacc[mb], 0, 0, 0);
        }
        if (t + 1 < ntiles) deposit(buf ^ 1);
        __syncthreads();
    }
    // epilogue: registers -> LDS tile [64 chains][P] -> coalesced 16-byte stores.  (Direct stores
    // from the MFMA layout are 16-byte fragments 512 B apart: measured ~20 us per launch for 16 MB.)
    static_assert(64 * P <= G::SMEM, "output tile must fit the staging block");
    float* otile = smem;
#pragma unroll
    for (int mb = 0; mb < G::MB; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) otile[(16 * wave + c) * P + 16 * mb + k + 4 * r] = gacc[mb][r];
    __syncthreads();
    {
        const int64_t chain0 = (int64_t)blockIdx.x * 64;
        const int64_t nlive = a.C - chain0 < 64 ? a.C - chain0 : 64;
        f32x4* dst = reinterpret_cast<f32x4*>(a.part_g + ((int64_t)rs * a.C + chain0) * P);
        const f32x4* src = reinterpret_cast<const f32x4*>(otile);
        for (int i = tid; i < (int)(nlive * P / 4); i += 256) dst[i] = src[i];
    }
    if constexpr (VALUE) {
        const double tot = ksum(vsum);
        if (live && k == 0) a.part_v[(int64_t)rs * a.C + chain] = tot;
    }
}

}  // namespace lr
