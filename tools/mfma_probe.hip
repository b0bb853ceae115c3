// mfma_probe.hip -- empirically derive the lane/register layout of v_mfma_f32_4x4x1_16b_f32 and
// its issue rate on gfx950 (tools only).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void probe(const float* A, const float* B, float* D) {
    const int l = threadIdx.x;
    f4 acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_f32_4x4x1f32(A[l], B[l], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[l * 4 + r] = acc[r];
}

template <int NACC> __global__ void rate(float* out, float s, int iters) {
    f4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f4{s, s, s, s};
    float a = s + threadIdx.x, b = s * 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
    }
    float r = 0;
    for (int i = 0; i < NACC; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

// mixed: NACC mfma + NV independent v_fmac per iteration: do they overlap inside one wave?
template <int NACC, int NV> __global__ void mixed(float* out, float s, int iters) {
    f4 acc[NACC];
    float v[16];
    for (int i = 0; i < NACC; ++i) acc[i] = f4{s, s, s, s};
    for (int i = 0; i < 16; ++i) v[i] = s + i;
    float a = s + threadIdx.x, b = s * 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            acc[i] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NV / NACC; ++j) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[(i * (NV / NACC) + j) & 15]) : "v"(a), "v"(b));
        }
    }
    float r = 0;
    for (int i = 0; i < NACC; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 16; ++i) r += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <class F> float timeit(F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    f(); hipDeviceSynchronize();
    hipEventRecord(e0); f(); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}

int main() {
    float *dA, *dB, *dD;
    hipMalloc(&dA, 64 * 4); hipMalloc(&dB, 64 * 4); hipMalloc(&dD, 256 * 4);
    // A[lane] = 1 + lane, B[lane] = 100 + lane  -> D[lane][r] = A[la]*B[lb]: decode la, lb
    std::vector<float> hA(64), hB(64), hD(256);
    for (int i = 0; i < 64; ++i) { hA[i] = 1 + i; hB[i] = 1000 + 7 * i; }
    hipMemcpy(dA, hA.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dB, hB.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD);
    hipMemcpy(hD.data(), dD, 1024, hipMemcpyDeviceToHost);
    printf("layout: D[lane][reg] = A[la]*B[lb]\n");
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int r = 0; r < 4; ++r) {
            int fa = -1, fb = -1;
            for (int a = 0; a < 64 && fa < 0; ++a) for (int b = 0; b < 64; ++b) if (hA[a] * hB[b] == hD[l * 4 + r]) { fa = a; fb = b; break; }
            printf("  r%d=(A%2d,B%2d)", r, fa, fb);
        }
        printf("\n");
        if (l == 7) l = 59;  // print first 8 and last 4 lanes
    }
    float* out; hipMalloc(&out, 1024 * 1024 * 4 * 4);
    const int iters = 200000;
    for (int wps : {1, 2}) {
        int blocks = 256, threads = 256 * wps;
        double n;
        float ms = timeit([&] { hipLaunchKernelGGL(rate<1>, dim3(blocks), dim3(threads), 0, 0, out, 1.0f, iters); });
        n = (double)iters * 1; printf("4x4x1 dependent x1   waves/SIMD=%d: %.2f ns/mfma/wave -> cycles@2.4GHz %.1f\n", wps, ms * 1e6 / n / 1, ms * 1e-3 * 2.4e9 / (n * wps));
        ms = timeit([&] { hipLaunchKernelGGL(rate<4>, dim3(blocks), dim3(threads), 0, 0, out, 1.0f, iters); });
        n = (double)iters * 4; printf("4x4x1 independent x4 waves/SIMD=%d: cycles@2.4GHz per mfma per SIMD %.1f\n", wps, ms * 1e-3 * 2.4e9 / (n * wps));
        ms = timeit([&] { hipLaunchKernelGGL(rate<8>, dim3(blocks), dim3(threads), 0, 0, out, 1.0f, iters); });
        n = (double)iters * 8; printf("4x4x1 independent x8 waves/SIMD=%d: cycles@2.4GHz per mfma per SIMD %.1f\n", wps, ms * 1e-3 * 2.4e9 / (n * wps));
        ms = timeit([&] { hipLaunchKernelGGL((mixed<8, 0>), dim3(blocks), dim3(threads), 0, 0, out, 1.0f, iters); });
        printf("mixed 8 mfma + 0 fmac  waves/SIMD=%d: %.3f ms\n", wps, ms);
        ms = timeit([&] { hipLaunchKernelGGL((mixed<8, 8>), dim3(blocks), dim3(threads), 0, 0, out, 1.0f, iters); });
        printf("mixed 8 mfma + 8 fmac  waves/SIMD=%d: %.3f ms\n", wps, ms);
        ms = timeit([&] { hipLaunchKernelGGL((mixed<8, 16>), dim3(blocks), dim3(threads), 0, 0, out, 1.0f, iters); });
        printf("mixed 8 mfma + 16 fmac waves/SIMD=%d: %.3f ms\n", wps, ms);
        ms = timeit([&] { hipLaunchKernelGGL((mixed<8, 32>), dim3(blocks), dim3(threads), 0, 0, out, 1.0f, iters); });
        printf("mixed 8 mfma + 32 fmac waves/SIMD=%d: %.3f ms\n", wps, ms);
    }
    return 0;
}
