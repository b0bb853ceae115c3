// lr_stats.h -- chain-pooled reduction of the streaming statistics buffer (include/logreg_hip.h, "Streaming
// statistics"): stats [slots][C][2][p] (mean, M2 per batch of B kept samples) -> sums [LR_STATS_ROWS][p].
//
// Replaces, on the device, what the reference does on the full sample matrix afterwards: scipy.stats.describe
// (Python/fit-np-hmc.py:113-117: mean, ddof = 1 variance) and smfsb::mcmcSummary (Python/analyse.R:17-19: ESS);
// split-R-hat is the many-chain addition.  Deterministic: a fixed tree over the chains, no atomics.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lr {

constexpr int kStatsRows = 7;

struct Moments {  // count, mean, sum of squared deviations
    double n, mean, m2;
    __device__ __forceinline__ void merge(double nb, double mb, double m2b) {  // Chan et al. pairwise update
        if (nb <= 0.0) return;
        if (n <= 0.0) { n = nb; mean = mb; m2 = m2b; return; }
        const double tot = n + nb, d = mb - mean;
        mean += d * (nb / tot);
        m2 += m2b + d * d * (n * nb / tot);
        n = tot;
    }
};

// block = (PW, CY): x = coordinate (PW = power of two >= p), y = chain inside the block; grid.x = ceil(C / CY).
// part [gridDim.x][kStatsRows][p]
template <int PW>
__global__ void __launch_bounds__(256) k_stats_partial(const double* __restrict__ stats, int64_t C, int p, int64_t B,
                                                       int64_t kept, const double* __restrict__ pivot,
                                                       double* __restrict__ part) {
    constexpr int CY = 256 / PW;
    __shared__ double red[kStatsRows][CY][PW];
    const int j = threadIdx.x, cy = threadIdx.y;
    const int64_t chain = (int64_t)blockIdx.x * CY + cy;
    const int64_t nb = kept / B, rem = kept - nb * B;
    const bool halves = nb >= 2 && (nb & 1) == 0;
    double v[kStatsRows];
#pragma unroll
    for (int r = 0; r < kStatsRows; ++r) v[r] = 0.0;
    if (j < p && chain < C) {
        const double piv = pivot[j];
        const int64_t stride = C * 2 * p;  // slot to slot
        const double* s = stats + (chain * 2) * p + j;
        Moments h[2] = {{0, 0, 0}, {0, 0, 0}};
        for (int64_t b = 0; b < nb; ++b) {
            const int which = halves && b >= nb / 2 ? 1 : 0;
            h[which].merge((double)B, s[b * stride], s[b * stride + p]);
        }
        Moments full = h[0];
        full.merge(h[1].n, h[1].mean, h[1].m2);
        double bm = 0.0;
        for (int64_t b = 0; b < nb; ++b) {
            const double d = s[b * stride] - full.mean;
            bm += d * d;
        }
        Moments all = full;
        if (rem > 0) all.merge((double)rem, s[nb * stride], s[nb * stride + p]);
        const double dm = all.mean - piv;
        v[0] = all.n * dm;
        v[1] = all.n * dm * dm;
        v[2] = all.m2;
        if (halves) {
#pragma unroll
            for (int w = 0; w < 2; ++w) {
                const double dh = h[w].mean - piv;
                v[3] += dh;
                v[4] += dh * dh;
                v[5] += h[w].n > 1.0 ? h[w].m2 / (h[w].n - 1.0) : 0.0;
            }
        }
        v[6] = nb >= 2 ? bm : 0.0;
    }
#pragma unroll
    for (int r = 0; r < kStatsRows; ++r) red[r][cy][j] = v[r];
    __syncthreads();
    for (int half = CY / 2; half >= 1; half >>= 1) {  // fixed tree over the block's chains
        if (cy < half) {
#pragma unroll
            for (int r = 0; r < kStatsRows; ++r) red[r][cy][j] += red[r][cy + half][j];
        }
        __syncthreads();
    }
    if (cy == 0 && j < p) {
#pragma unroll
        for (int r = 0; r < kStatsRows; ++r) part[((int64_t)blockIdx.x * kStatsRows + r) * p + j] = red[r][0][j];
    }
}

// one thread per (row, coordinate): block partials summed in block order
__global__ void __launch_bounds__(256) k_stats_final(const double* __restrict__ part, int64_t nblocks, int p,
                                                     double* __restrict__ sums) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= kStatsRows * p) return;
    double s = 0.0;
    for (int64_t b = 0; b < nblocks; ++b) s += part[b * kStatsRows * p + e];
    sums[e] = s;
}

}  // namespace lr
