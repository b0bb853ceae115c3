// mfma_rate.hip -- issue rate of v_mfma_f32_16x16x4_f32 with 2 / 8 round-robin accumulators,
// 1 or 2 waves per SIMD (tools only).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int NACC> __global__ void rate(float* out, const float* in, int iters) {
    f4 acc[NACC];
    float a[16], b[16];
    for (int i = 0; i < 16; ++i) { a[i] = in[threadIdx.x + 64 * i]; b[i] = in[threadIdx.x + 64 * i + 7]; }
    for (int i = 0; i < NACC; ++i) acc[i] = f4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 32; ++j) acc[j % NACC] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j & 15], b[(j + it) & 15], acc[j % NACC], 0, 0, 0);
    }
    float r = 0;
    for (int i = 0; i < NACC; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <class F> float timeit(F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize(); hipEventRecord(e0); f(); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
    float *out, *in; hipMalloc(&out, 1 << 24); hipMalloc(&in, 1 << 16); hipMemset(in, 0, 1 << 16);
    const int iters = 20000;
    for (int wps : {1, 2}) {
        float ms2 = timeit([&] { hipLaunchKernelGGL(rate<2>, dim3(256), dim3(256 * wps), 0, 0, out, in, iters); });
        float ms8 = timeit([&] { hipLaunchKernelGGL(rate<8>, dim3(256), dim3(256 * wps), 0, 0, out, in, iters); });
        double n = (double)iters * 32 * wps;
        printf("waves/SIMD=%d: 2 accumulators %.1f cycles@2.4GHz per MFMA per SIMD (%.1f TF), 8 accumulators %.1f (%.1f TF)\n", wps,
               ms2 * 1e-3 * 2.4e9 / n, n * 1024 * 2048 / (ms2 * 1e-3) / 1e12, ms8 * 1e-3 * 2.4e9 / n, n * 1024 * 2048 / (ms8 * 1e-3) / 1e12);
    }
    return 0;
}
