// tools/sigmoid_probe.hip -- VERDICT r4 item 4: is w = sigma(-eta) cheaper on PACKED f16 arithmetic (v_pk_fma_f16: two values per
// lane and issue, a clamped odd polynomial, no division) than on the transcendental unit (v_exp_f32 + v_rcp_f32, quarter rate)?
// The interior kernel of config 4 (lr_tall_mx.h mx_pairs) computes 8 such values per lane and tile pair and rounds them to bf16 for
// the gradient MFMA; its loop is bound by exactly this vector work (tools/mx_loop_probe.hip: VALU alone 79.6 ns per pair and SIMD,
// MFMAs + LDS alone 45.7).  This probe times, at 4 waves per SIMD on every CU, the sigmoid of 8 accumulator values -> 4 packed bf16
// registers in three forms and prints ns per 8 values and SIMD, plus the largest absolute error of each form against double:
//   A  the kernel's: 8 x (v_exp_f32, +1, v_rcp_f32), 4 x v_cvt_pk_bf16_f32
//   B  packed f16, degree-9 odd polynomial of tanh on the clamped argument: 4 x v_cvt_pkrtz, clamp, x^2, 4 pk_fma, 1 pk_fma, back through f32
//   C  packed f16, degree-5 (coarser)
// hipcc --offload-arch=gfx950 -O3 tools/sigmoid_probe.hip -o /tmp/sigmoid_probe && /tmp/sigmoid_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t pack_bf16(float a, float b) { return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2{a, b}, bf16x2)); }

// sigma(-t) for t in log2 units (the kernels carry eta * log2 e): 1 / (1 + 2^t)
template <int FORM> __device__ __forceinline__ void sig8(const float (&e)[8], uint32_t (&w)[4]) {
    if constexpr (FORM == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float a = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(e[2 * i]));
            const float b = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(e[2 * i + 1]));
            w[i] = pack_bf16(a, b);
        }
    } else {
        // sigma(-x) = 0.5 - 0.5 tanh(x / 2), x = t ln 2;  u = clamp(t * (ln 2 / 2), -U, U);  tanh(u) ~ u P(u^2)
        constexpr float U = FORM == 1 ? 4.0f : 3.0f;
        const f16x2 s = {(_Float16)0.34657359f, (_Float16)0.34657359f}, lo = {(_Float16)-U, (_Float16)-U}, hi = {(_Float16)U, (_Float16)U};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f16x2 u = __builtin_bit_cast(f16x2, __builtin_amdgcn_cvt_pkrtz(e[2 * i], e[2 * i + 1])) * s;
            u = __builtin_elementwise_max(__builtin_elementwise_min(u, hi), lo);
            const f16x2 z = u * u;
            f16x2 p;
            if constexpr (FORM == 1) {  // degree 9 (minimax fit on [-4, 4]): tanh(u) ~ u (c0 + c1 z + c2 z^2 + c3 z^3 + c4 z^4)
                const f16x2 c4 = {(_Float16)1.8454e-5f, (_Float16)1.8454e-5f}, c3 = {(_Float16)-7.8984e-4f, (_Float16)-7.8984e-4f}, c2 = {(_Float16)1.25933e-2f, (_Float16)1.25933e-2f},
                            c1 = {(_Float16)-9.6205e-2f, (_Float16)-9.6205e-2f}, c0 = {(_Float16)0.468287f, (_Float16)0.468287f};  // (0.5 tanh(u) ~ u P(u^2): minimax fit, max error 8.8e-3)
                p = __builtin_elementwise_fma(z, c4, c3);
                p = __builtin_elementwise_fma(z, p, c2);
                p = __builtin_elementwise_fma(z, p, c1);
                p = __builtin_elementwise_fma(z, p, c0);
            } else {  // degree 5 on [-3, 3]
                const f16x2 c2 = {(_Float16)3.69971e-3f, (_Float16)3.69971e-3f}, c1 = {(_Float16)-6.2661e-2f, (_Float16)-6.2661e-2f}, c0 = {(_Float16)0.436895f, (_Float16)0.436895f};  // (minimax fit, max error 2.0e-2)
                p = __builtin_elementwise_fma(z, c2, c1);
                p = __builtin_elementwise_fma(z, p, c0);
            }
            const f16x2 half = {(_Float16)0.5f, (_Float16)0.5f};
            const f16x2 r = __builtin_elementwise_fma(-u, p, half);  // 0.5 - u * (0.5 P)
            w[i] = pack_bf16((float)r[0], (float)r[1]);
        }
    }
}

template <int FORM> __global__ void __launch_bounds__(256) k_time(float* out, int iters, float seed) {
    float e[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) e[i] = seed * (float)(threadIdx.x % 37 - 18 + i) * 0.31f;
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
        uint32_t w[4];
        sig8<FORM>(e, w);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc ^= w[i];
#pragma unroll
        for (int i = 0; i < 8; ++i) e[i] += __builtin_bit_cast(float, (acc & 0x007FFFFFu) | 0x33000000u);  // a dependence the compiler cannot fold (tiny increment)
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = __builtin_bit_cast(float, acc) + e[0];
}
template <int FORM> __global__ void k_err(const float* t, float* w, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float e[8];
    uint32_t pk[4];
#pragma unroll
    for (int j = 0; j < 8; ++j) e[j] = t[i];
    sig8<FORM>(e, pk);
    w[i] = __builtin_bit_cast(float, pk[0] << 16);
}

int main() {
    float *d_out, *d_t, *d_w;
    const int n = 1 << 16, blocks = 256 * 4, iters = 20000;
    hipMalloc(&d_out, blocks * 256 * 4); hipMalloc(&d_t, n * 4); hipMalloc(&d_w, n * 4);
    std::vector<float> t(n), w(n);
    for (int i = 0; i < n; ++i) t[i] = -30.0f + 60.0f * i / (n - 1);  // log2 units: eta in [-20.8, 20.8]
    hipMemcpy(d_t, t.data(), n * 4, hipMemcpyHostToDevice);
    const char* names[3] = {"A  v_exp_f32 + v_rcp_f32 (the kernel's)", "B  packed f16, degree-9 odd polynomial, clamp 4", "C  packed f16, degree-5, clamp 3"};
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int f = 0; f < 3; ++f) {
        auto launch_t = [&](int it) { if (f == 0) k_time<0><<<blocks, 256>>>(d_out, it, 1.0f); else if (f == 1) k_time<1><<<blocks, 256>>>(d_out, it, 1.0f); else k_time<2><<<blocks, 256>>>(d_out, it, 1.0f); };
        launch_t(100); hipDeviceSynchronize();
        hipEventRecord(e0); launch_t(iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (f == 0) k_err<0><<<n / 256, 256>>>(d_t, d_w, n); else if (f == 1) k_err<1><<<n / 256, 256>>>(d_t, d_w, n); else k_err<2><<<n / 256, 256>>>(d_t, d_w, n);
        hipMemcpy(w.data(), d_w, n * 4, hipMemcpyDeviceToHost);
        double emax = 0, eat = 0;
        for (int i = 0; i < n; ++i) { const double ref = 1.0 / (1.0 + std::exp2((double)t[i])); const double d = std::fabs((double)w[i] - ref); if (d > emax) { emax = d; eat = t[i]; } }
        // 4 waves per SIMD (1024 blocks of 4 waves on 256 CUs x 4 SIMDs): ns per 8 values and SIMD = ms / iters / 4 waves
        printf("%-52s %7.2f ns per 8 values and SIMD (4 waves per SIMD)   max |error| %.2e at t = %.2f (bf16 rounding alone: 2e-3 relative)\n", names[f], ms * 1e6 / iters / 4.0, emax, eat);
    }
    return 0;
}
