// mfma_f64_rate.hip -- v_mfma_f64_16x16x4_f64 on gfx950: issue rate with 1 / 2 / 8 round-robin accumulators (dependent chain, two
// chains, independent), 1 / 2 / 4 waves per SIMD, and how many independent v_fma_f64 fit in an MFMA's shadow (tools only; round 6:
// the float64 matrix-pipe chain kernel, lr_mm_f64.h).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
template <int NACC, int NVALU> __global__ void rate(double* out, const double* in, int iters) {
    d4 acc[NACC];
    double a[8], b[8], v[8];
    for (int i = 0; i < 8; ++i) { a[i] = in[threadIdx.x + 64 * i]; b[i] = in[threadIdx.x + 64 * i + 7]; v[i] = in[threadIdx.x + i]; }
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            acc[j % NACC] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[j & 7], b[(j + it) & 7], acc[j % NACC], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < NVALU; ++u) v[u & 7] = __builtin_fma(v[u & 7], a[u & 7], b[(u + 3) & 7]);  // independent of the MFMAs
        }
    }
    double r = 0;
    for (int i = 0; i < NACC; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 8; ++i) r += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <class F> float timeit(F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize(); hipEventRecord(e0); f(); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
template <int NACC, int NVALU> void run(int wps, int iters, double* out, const double* in) {
    float ms = timeit([&] { hipLaunchKernelGGL((rate<NACC, NVALU>), dim3(256), dim3(256 * wps), 0, 0, out, in, iters); });
    const double n = (double)iters * 16 * wps;  // MFMAs per SIMD
    printf("  waves/SIMD=%d accumulators=%d + %2d independent v_fma_f64 per MFMA: %6.1f cycles@2.4GHz per MFMA per SIMD (%.1f TF matrix)\n", wps, NACC, NVALU,
           ms * 1e-3 * 2.4e9 / n, n * 1024 * 2048 / (ms * 1e-3) / 1e12);
}
int main() {
    double *out, *in; hipMalloc(&out, 1 << 24); hipMalloc(&in, 1 << 16); hipMemset(in, 0, 1 << 16);
    const int iters = 5000;
    for (int wps : {1, 2, 4}) {
        run<1, 0>(wps, iters, out, in); run<2, 0>(wps, iters, out, in); run<8, 0>(wps, iters, out, in);
        run<8, 4>(wps, iters, out, in); run<8, 8>(wps, iters, out, in); run<8, 16>(wps, iters, out, in); run<8, 32>(wps, iters, out, in);
        run<2, 16>(wps, iters, out, in);
    }
    return 0;
}
