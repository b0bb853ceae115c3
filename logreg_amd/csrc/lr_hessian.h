// lr_hessian.h -- closed-form second-order information at ONE parameter vector, in one pass over the rows and in
// float64 arithmetic whatever the model's storage type:
//     value  ll(beta)            = sum_i log sigma(t_i)                 t_i = xs_i . beta, xs_i = (2 y_i - 1) x_i
//     grad   d ll / d beta       = sum_i sigma(-t_i) xs_i               (fit-np-hmc.py:44-47 without the prior term)
//     H      -d2 ll / d beta2    = sum_i sigma(t_i) sigma(-t_i) xs_i xs_i^T   (= X^T W X: the sign squares away)
// Serves the device MAP finder (logreg_amd/optimize.py): Newton's method with the exact Hessian, as the
// reference's JAX variant does (Python/fit-jax-hmc.py:61-79), in place of SciPy BFGS (fit-np-hmc.py:49).
//
// grid = row slices of kHessRows rows; block = 256 threads.  The slice is staged in LDS as doubles, one thread
// per row forms t, sigma(-t), the weight and the value term; then the threads own the p(p+1)/2 upper-triangle
// entries (+ p gradient entries + the value) and sum over the slice's rows.  Block partials are summed in block
// order by k_hess_final: deterministic, no atomics.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lr {

constexpr int kHessRows = 64;

// part [gridDim.x][NE + p + 1], NE = p (p + 1) / 2, entry order (0,0) (0,1) .. (0,p-1) (1,1) ..
template <typename T>
__global__ void __launch_bounds__(256) k_hess_partial(const T* __restrict__ rows, int64_t n, int P, int p,
                                                      const double* __restrict__ beta, double* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    double* xs = reinterpret_cast<double*>(smem_raw);  // [kHessRows][p]
    double* w = xs + (size_t)kHessRows * p;            // sigma(t) sigma(-t)
    double* sneg = w + kHessRows;                      // sigma(-t)
    double* val = sneg + kHessRows;                    // log sigma(t)
    double* b = val + kHessRows;                       // beta [p]
    const int tid = threadIdx.x;
    const int64_t r0 = (int64_t)blockIdx.x * kHessRows;
    for (int e = tid; e < kHessRows * p; e += 256) {
        const int r = e / p, j = e - r * p;
        xs[e] = r0 + r < n ? (double)rows[(r0 + r) * P + j] : 0.0;
    }
    for (int j = tid; j < p; j += 256) b[j] = beta[j];
    __syncthreads();
    if (tid < kHessRows) {
        double t = 0.0;
        for (int j = 0; j < p; ++j) t += xs[tid * p + j] * b[j];
        const bool live = r0 + tid < n;
        const double e = exp(-fabs(t));                    // in (0, 1]
        const double s_pos = t >= 0 ? 1.0 / (1.0 + e) : e / (1.0 + e);  // sigma(t)
        const double s_neg = t >= 0 ? e / (1.0 + e) : 1.0 / (1.0 + e);  // sigma(-t)
        w[tid] = live ? s_pos * s_neg : 0.0;
        sneg[tid] = live ? s_neg : 0.0;
        val[tid] = live ? fmin(t, 0.0) - log1p(e) : 0.0;
    }
    __syncthreads();
    const int NE = p * (p + 1) / 2;
    double* out = part + (size_t)blockIdx.x * (NE + p + 1);
    for (int e = tid; e < NE; e += 256) {
        int a = 0, rem = e;
        while (rem >= p - a) { rem -= p - a; ++a; }
        const int c = a + rem;
        double s = 0.0;
        for (int r = 0; r < kHessRows; ++r) s += w[r] * xs[r * p + a] * xs[r * p + c];
        out[e] = s;
    }
    for (int j = tid; j < p; j += 256) {
        double s = 0.0;
        for (int r = 0; r < kHessRows; ++r) s += sneg[r] * xs[r * p + j];
        out[NE + j] = s;
    }
    if (tid == 0) {
        double s = 0.0;
        for (int r = 0; r < kHessRows; ++r) s += val[r];
        out[NE + p] = s;
    }
}

__global__ void __launch_bounds__(256) k_hess_final(const double* __restrict__ part, int64_t nblocks, int width,
                                                    double* __restrict__ sums) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= width) return;
    double s = 0.0;
    for (int64_t b = 0; b < nblocks; ++b) s += part[b * width + e];
    sums[e] = s;
}

}  // namespace lr
