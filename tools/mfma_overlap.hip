// mfma_overlap.hip -- does VALU work overlap with bf16 MFMA work issued by the same wave / by a
// co-resident wave on gfx950?  (tools only)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));

// NM bf16 MFMAs (16x16x32) + NV independent v_fmac per iteration, interleaved
template <int NM, int NV> __global__ void mixed(float* out, const float* in, int iters) {
    f4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = f4{0, 0, 0, 0};
    b8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)in[threadIdx.x + i]; b[i] = (__bf16)in[threadIdx.x + 8 + i]; }
    float v[8], x = in[threadIdx.x], y = in[threadIdx.x + 1];
    for (int i = 0; i < 8; ++i) v[i] = in[threadIdx.x + 32 + i];
    constexpr int STEPS = NM > 0 ? NM : 8;
    constexpr int VPER = NV / STEPS;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < STEPS; ++j) {
            if constexpr (NM > 0) acc[j & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[j & 3], 0, 0, 0);
#pragma unroll
            for (int q = 0; q < VPER; ++q) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[(j + q) & 7]) : "v"(x), "v"(y));
        }
    }
    float r = 0;
    for (int i = 0; i < 4; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) r += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <class F> float timeit(F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize(); hipEventRecord(e0); f(); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    float *out, *in; hipMalloc(&out, 1 << 24); hipMalloc(&in, 1 << 16); hipMemset(in, 0, 1 << 16);
    const int iters = 100000;
    for (int wps : {1, 2}) {
        dim3 g(256), b(256 * wps);
        float m = timeit([&] { hipLaunchKernelGGL((mixed<8, 0>), g, b, 0, 0, out, in, iters); });
        float v = timeit([&] { hipLaunchKernelGGL((mixed<0, 32>), g, b, 0, 0, out, in, iters); });
        float mv = timeit([&] { hipLaunchKernelGGL((mixed<8, 32>), g, b, 0, 0, out, in, iters); });
        float v2 = timeit([&] { hipLaunchKernelGGL((mixed<0, 64>), g, b, 0, 0, out, in, iters); });
        float mv2 = timeit([&] { hipLaunchKernelGGL((mixed<8, 64>), g, b, 0, 0, out, in, iters); });
        printf("waves/SIMD=%d: 8 bf16 mfma alone %.2f ms | 32 fmac alone %.2f | both %.2f | 64 fmac alone %.2f | 8 mfma + 64 fmac %.2f\n", wps, m, v, mv, v2, mv2);
    }
    return 0;
}
