// fp32_pipes_probe.hip -- are the fp32 vector FMAs (v_pk_fma_f32) and the fp32-input MFMA (v_mfma_f32_16x16x4_f32) one set of
// multipliers or two?  Both peak at 64 FLOP/clk/SIMD (MI355X_MICROARCH.md).  If a SIMD can run them side by side, a kernel that
// splits its multiply-adds between the two pipes has twice the fp32 roofline.  Measures TFLOP/s of: vector only, MFMA only,
// both interleaved in one wave, and two co-resident waves of which one does vector and one MFMA work.
//   hipcc --offload-arch=gfx950 -O3 tools/fp32_pipes_probe.hip -o /tmp/fp32_pipes && /tmp/fp32_pipes
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

// MODE 0: NV v_pk_fma per trip; 1: NM MFMAs per trip; 2: both interleaved; 3: wave parity decides (even waves vector, odd MFMA)
template <int MODE> __global__ void k(float* out, const float* in, int iters) {
    f4 acc[8];
    f2 v[16];
    for (int i = 0; i < 8; ++i) acc[i] = f4{0, 0, 0, 0};
    for (int i = 0; i < 16; ++i) v[i] = f2{in[threadIdx.x + i], in[threadIdx.x + i + 1]};
    const float a = in[threadIdx.x + 40], b = in[threadIdx.x + 41];
    const f2 x = {in[threadIdx.x + 42], in[threadIdx.x + 43]}, y = {in[threadIdx.x + 44], in[threadIdx.x + 45]};
    const bool vec = MODE == 0 || MODE == 2 || (MODE == 3 && ((threadIdx.x >> 6) & 1) == 0);
    const bool mat = MODE == 1 || MODE == 2 || (MODE == 3 && ((threadIdx.x >> 6) & 1) == 1);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (mat && (j & 1) == 0) acc[(j >> 1) & 7] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[(j >> 1) & 7], 0, 0, 0);
            if (vec) {
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(v[j]) : "v"(x), "v"(y));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(v[(j + 8) & 15]) : "v"(x), "v"(y));
            }
        }
    }
    float r = 0;
    for (int i = 0; i < 8; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 16; ++i) r += v[i].x + v[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
template <int MODE> void run(const char* what, int waves_per_simd, float* out, float* in) {
    const int iters = 4000, threads = 256 * waves_per_simd;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, in, 200);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(threads), 0, 0, out, in, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double waves = 1024.0 * waves_per_simd;
    // per trip and wave: 32 v_pk_fma (256 flop per lane-pair op: 64 lanes x 2 x 2) and 8 MFMAs (16x16x4x2 = 2048 flop)
    double vec_waves = MODE == 0 || MODE == 2 ? waves : (MODE == 3 ? waves / 2 : 0), mat_waves = MODE == 1 || MODE == 2 ? waves : (MODE == 3 ? waves / 2 : 0);
    const double flop = iters * (vec_waves * 32 * 256.0 + mat_waves * 8 * 2048.0);
    printf("%-58s %d waves/SIMD: %7.1f TFLOP/s (vector part %.1f, MFMA part %.1f)\n", what, waves_per_simd, flop / (ms * 1e-3) / 1e12,
           iters * vec_waves * 32 * 256.0 / (ms * 1e-3) / 1e12, iters * mat_waves * 8 * 2048.0 / (ms * 1e-3) / 1e12);
}
int main() {
    float *out, *in; (void)hipMalloc(&out, 1 << 24); (void)hipMalloc(&in, 1 << 16); (void)hipMemset(in, 0, 1 << 16);
    for (int w : {1, 2, 4}) {
        run<0>("vector only (v_pk_fma_f32)", w, out, in);
        run<1>("MFMA only (v_mfma_f32_16x16x4_f32)", w, out, in);
        run<2>("both, interleaved in every wave", w, out, in);
        if (w > 1) run<3>("even waves vector, odd waves MFMA", w, out, in);
    }
    return 0;
}
