// trans_rate.hip -- issue cost of the transcendental VALU ops the sigmoid needs, f32 vs f16 forms, one and four waves per SIMD.
//   hipcc --offload-arch=gfx950 -O2 tools/trans_rate.hip -o tools/bin/trans_rate && tools/bin/trans_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(X) X X X X X X X X X X X X X X X X
template <int WHICH> __global__ void k(float* out, int iters) {
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    for (int i = 0; i < iters; ++i) {
        if (WHICH == 0) { REP16(asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_exp_f32 %3, %3\n\tv_exp_f32 %4, %4\n\tv_exp_f32 %5, %5\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (WHICH == 1) { REP16(asm volatile("v_exp_f16 %0, %0\n\tv_exp_f16 %1, %1\n\tv_exp_f16 %2, %2\n\tv_exp_f16 %3, %3\n\tv_exp_f16 %4, %4\n\tv_exp_f16 %5, %5\n\tv_exp_f16 %6, %6\n\tv_exp_f16 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (WHICH == 2) { REP16(asm volatile("v_rcp_f32 %0, %0\n\tv_rcp_f32 %1, %1\n\tv_rcp_f32 %2, %2\n\tv_rcp_f32 %3, %3\n\tv_rcp_f32 %4, %4\n\tv_rcp_f32 %5, %5\n\tv_rcp_f32 %6, %6\n\tv_rcp_f32 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (WHICH == 3) { REP16(asm volatile("v_rcp_f16 %0, %0\n\tv_rcp_f16 %1, %1\n\tv_rcp_f16 %2, %2\n\tv_rcp_f16 %3, %3\n\tv_rcp_f16 %4, %4\n\tv_rcp_f16 %5, %5\n\tv_rcp_f16 %6, %6\n\tv_rcp_f16 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (WHICH == 4) { REP16(asm volatile("v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %1, %1, %1, %1\n\tv_fma_f32 %2, %2, %2, %2\n\tv_fma_f32 %3, %3, %3, %3\n\tv_fma_f32 %4, %4, %4, %4\n\tv_fma_f32 %5, %5, %5, %5\n\tv_fma_f32 %6, %6, %6, %6\n\tv_fma_f32 %7, %7, %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
template <int W> void run(const char* name, int wps) {
    float* d; hipMalloc(&d, 256 * 1024 * 64 * 4);
    const int iters = 2000, blocks = 256 * wps;  // 256 threads = 4 waves = 1 per SIMD per block
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<W>, dim3(blocks), dim3(256), 0, 0, d, 10);
    hipEventRecord(e0); hipLaunchKernelGGL(k<W>, dim3(blocks), dim3(256), 0, 0, d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double inst_per_simd = (double)iters * 128 * wps;
    printf("%-10s %d waves/SIMD: %.2f cycles@2.4GHz per wave-instruction per SIMD\n", name, wps, ms * 1e-3 * 2.4e9 / inst_per_simd);
    hipFree(d);
}
int main() {
    for (int wps : {1, 4}) { run<0>("v_exp_f32", wps); run<1>("v_exp_f16", wps); run<2>("v_rcp_f32", wps); run<3>("v_rcp_f16", wps); run<4>("v_fma_f32", wps); }
    return 0;
}
