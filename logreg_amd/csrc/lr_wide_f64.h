// lr_wide_f64.h -- partial-evaluation kernel of the stepwise engine for WIDE models (32 < p <= 128) in FLOAT64, the
// arithmetic the reference computes in (Python/fit-np-hmc.py:17-19: X float64, NumPy float64 throughout).
//
// X.beta over a block of chains is a dense GEMM; here it runs on the float64 matrix pipe (v_mfma_f64_16x16x4_f64:
// D[16x16] += A[16x4] . B[4x16], an fma chain per output element in K order, so results are in the float64 rounding class
// of the oracle).  Every evaluation of a float64 wide model goes through this kernel under LR_PREC_FULL; under the default policy
// the interior leapfrog gradients of HMC run on the bf16 pipe instead (k_wide_partial_bf16i<P, 4, double>: float64 position and
// momentum, float32 slice partials) and this kernel evaluates the end points.
//
// Workgroup = 4 waves = 64 chains; wave w owns chains 16w .. 16w+15 of the block.  Lane l = (c, k): c = l & 15 the chain
// (and, for the A operands, the row / the coordinate inside a 16-block), k = l >> 4 the K slot.  The block walks its row
// slice in tiles of 16 rows staged through LDS (two buffers, row stride P + 4 doubles: the 16 rows of an eta operand read
// then fall 8 banks apart).
//   eta  tile  E[16 rows x 16 chains] = sum_h A_h (16x4) . B_h (4x16):
//                A_h: lane (row c, k) = Xs[row c][4h + k]        B_h: lane (chain c, k) = beta[c][4h + k]   (registers)
//                D  : register r of lane (c, k) = E[row 4r + k][chain c]      (the f64 16x16x4 layout: rows interleaved by 4)
//   w = sigma(-E) on the lane's four accumulator values: they ARE the B operand of the gradient MFMAs
//   grad tile  G_mb[16 coords x 16 chains] += sum_r A_{mb,r} (16x4) . W_r (4x16):
//                A_{mb,r}: lane (m = c, k) = Xs[row 4r + k][16 mb + c]       W_r: lane (chain c, k) = w[r]
//                D       : register r' of lane (c, k) = gradient of coordinate 16 mb + 4r' + k of chain c
#pragma once
#include "lr_mfma.h"
#include "lr_tall.h"

namespace lr {

typedef double f64x4 __attribute__((ext_vector_type(4)));

template <int P> struct WideF64Geom {
    static constexpr int H = P / 4;    // eta MFMAs per tile = coordinates per lane
    static constexpr int MB = P / 16;  // gradient M-blocks
    static constexpr int LD = P + 4;   // doubles per staged row
    static constexpr int STAGE = 2 * 16 * LD, OUT = 64 * P;
    static constexpr int SMEM = STAGE > OUT ? STAGE : OUT;  // doubles
};

template <int P, bool VALUE>
__global__ void __launch_bounds__(256) k_wide_partial_f64(TallArgs<double, P> a) {
    using G = WideF64Geom<P>;
    __shared__ __attribute__((aligned(16))) double smem[G::SMEM];  // [2][16][LD] during the row loop, then the 64 x P output tile
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, k = lane >> 4;
    int64_t chain = (int64_t)blockIdx.x * 64 + 16 * wave + c;
    const bool live = chain < a.C;
    if (!live) chain = a.C - 1;
    const int rs = blockIdx.y;
    const int64_t s0 = (int64_t)rs * a.slice_len, s1 = s0 + a.slice_len < a.n ? s0 + a.slice_len : a.n;
    const int64_t ntiles = s1 > s0 ? (s1 - s0 + 15) / 16 : 0;

    double bs[G::H];
#pragma unroll
    for (int h = 0; h < G::H; ++h) bs[h] = a.q1[chain * P + 4 * h + k];
    f64x4 gacc[G::MB];
#pragma unroll
    for (int mb = 0; mb < G::MB; ++mb) gacc[mb] = f64x4{0, 0, 0, 0};
    double vsum = 0.0;

    // staging: thread t moves 16 P / 256 consecutive doubles of row t / 16 of the tile
    constexpr int PER = 16 * P / 256;
    const int srow = tid >> 4, scol = (tid & 15) * PER;
    double stage[PER];
    auto fetch = [&](int64_t t) {  // branch-free: out-of-slice rows re-read the last row and are zeroed
        const int64_t r = s0 + 16 * t + srow;
        const int64_t rc = r < s1 ? r : s1 - 1;
        const double keep = r < s1 ? 1.0 : 0.0;
#pragma unroll
        for (int i = 0; i < PER; ++i) stage[i] = a.rows[rc * P + scol + i] * keep;
    };
    auto deposit = [&](int buf) {
#pragma unroll
        for (int i = 0; i < PER; ++i) smem[(buf * 16 + srow) * G::LD + scol + i] = stage[i];
    };
    if (ntiles > 0) {
        fetch(0);
        deposit(0);
    }
    __syncthreads();
    for (int64_t t = 0; t < ntiles; ++t) {
        const int buf = (int)(t & 1);
        if (t + 1 < ntiles) fetch(t + 1);  // global loads in flight under the MFMAs of tile t
        const double* X = smem + buf * 16 * G::LD;
        // ---- eta = Xs . beta^T  (two accumulators: independent MFMA chains)
        f64x4 e0 = {0, 0, 0, 0}, e1 = {0, 0, 0, 0};
#pragma unroll
        for (int h = 0; h < G::H; h += 2) {
            e0 = __builtin_amdgcn_mfma_f64_16x16x4f64(X[c * G::LD + 4 * h + k], bs[h], e0, 0, 0, 0);
            e1 = __builtin_amdgcn_mfma_f64_16x16x4f64(X[c * G::LD + 4 * (h + 1) + k], bs[h + 1], e1, 0, 0, 0);
        }
        double w[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const double ts = e0[r] + e1[r];
            // one exponential for both (lr_device.h row_term, float64): e = exp(-|t|); sigma(-t) = e / (1 + e) for t > 0, 1 / (1 + e) otherwise
            // (was exp(t) and an IEEE division, plus exp(-|t|) and the library log1p with the value)
            const double e = exp(-__builtin_fabs(ts)), d1 = 1.0 + e;
            double rc = __builtin_amdgcn_rcp(d1);
            rc = __builtin_fma(__builtin_fma(-d1, rc, 1.0), rc, rc);
            rc = __builtin_fma(__builtin_fma(-d1, rc, 1.0), rc, rc);
            w[r] = ts > 0.0 ? e * rc : rc;
            if constexpr (VALUE) {
                const int64_t row = s0 + 16 * t + 4 * r + k;
                const double lv = (ts < 0.0 ? ts : 0.0) - log1p_unit(e);  // log sigma(t), stable for both signs
                if (row < s1) vsum += lv;
            }
        }
        // ---- grad += Xs^T . W
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int mb = 0; mb < G::MB; ++mb)
                gacc[mb] = __builtin_amdgcn_mfma_f64_16x16x4f64(X[(4 * r + k) * G::LD + 16 * mb + c], w[r], gacc[mb], 0, 0, 0);
        }
        if (t + 1 < ntiles) deposit(buf ^ 1);
        __syncthreads();
    }
    // epilogue: registers -> LDS tile [64 chains][P] -> coalesced stores
    double* otile = smem;
#pragma unroll
    for (int mb = 0; mb < G::MB; ++mb)
#pragma unroll
        for (int r = 0; r < 4; ++r) otile[(16 * wave + c) * P + 16 * mb + 4 * r + k] = gacc[mb][r];
    __syncthreads();
    {
        const int64_t chain0 = (int64_t)blockIdx.x * 64;
        const int64_t nlive = a.C - chain0 < 64 ? a.C - chain0 : 64;
        double* dst = a.part_g + ((int64_t)rs * a.C + chain0) * P;
        for (int i = tid; i < (int)(nlive * P); i += 256) dst[i] = otile[i];
    }
    if constexpr (VALUE) {
        const double tot = ksum(vsum);
        if (live && k == 0) a.part_v[(int64_t)rs * a.C + chain] = tot;
    }
}

}  // namespace lr
