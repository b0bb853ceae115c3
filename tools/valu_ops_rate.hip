// valu_ops_rate.hip -- issue cost of the non-transcendental VALU ops of the sigmoid / pack sequences (round 5): plain and packed f32 add / fma,
// the two 16-bit packs, at one and four waves per SIMD; independent chains of 8 registers (no dependent back-to-back issue).
//   hipcc --offload-arch=gfx950 -O2 tools/valu_ops_rate.hip -o tools/bin/valu_ops_rate && tools/bin/valu_ops_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(X) X X X X X X X X X X X X X X X X
typedef float f2 __attribute__((ext_vector_type(2)));
template <int WHICH> __global__ void k(float* out, int iters) {
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    unsigned u0 = 0, u1 = 0, u2 = 0, u3 = 0, u4 = 0, u5 = 0, u6 = 0, u7 = 0;
    for (int i = 0; i < iters; ++i) {
        if (WHICH == 0) { REP16(asm volatile("v_add_f32 %0, 1.0, %0\n\tv_add_f32 %1, 1.0, %1\n\tv_add_f32 %2, 1.0, %2\n\tv_add_f32 %3, 1.0, %3\n\tv_add_f32 %4, 1.0, %4\n\tv_add_f32 %5, 1.0, %5\n\tv_add_f32 %6, 1.0, %6\n\tv_add_f32 %7, 1.0, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (WHICH == 1) { REP16(asm volatile("v_pk_add_f32 %0, %0, 1.0\n\tv_pk_add_f32 %1, %1, 1.0\n\tv_pk_add_f32 %2, %2, 1.0\n\tv_pk_add_f32 %3, %3, 1.0\n\tv_pk_add_f32 %4, %4, 1.0\n\tv_pk_add_f32 %5, %5, 1.0\n\tv_pk_add_f32 %6, %6, 1.0\n\tv_pk_add_f32 %7, %7, 1.0" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7));) }
        if (WHICH == 2) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %0, %0\n\tv_pk_fma_f32 %1, %1, %1, %1\n\tv_pk_fma_f32 %2, %2, %2, %2\n\tv_pk_fma_f32 %3, %3, %3, %3\n\tv_pk_fma_f32 %4, %4, %4, %4\n\tv_pk_fma_f32 %5, %5, %5, %5\n\tv_pk_fma_f32 %6, %6, %6, %6\n\tv_pk_fma_f32 %7, %7, %7, %7" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7));) }
        if (WHICH == 3) { REP16(asm volatile("v_cvt_pk_bf16_f32 %0, %8, %9\n\tv_cvt_pk_bf16_f32 %1, %9, %10\n\tv_cvt_pk_bf16_f32 %2, %10, %11\n\tv_cvt_pk_bf16_f32 %3, %11, %12\n\tv_cvt_pk_bf16_f32 %4, %12, %13\n\tv_cvt_pk_bf16_f32 %5, %13, %14\n\tv_cvt_pk_bf16_f32 %6, %14, %15\n\tv_cvt_pk_bf16_f32 %7, %15, %8" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));) }
        if (WHICH == 4) { REP16(asm volatile("v_cvt_pk_f16_f32 %0, %8, %9\n\tv_cvt_pk_f16_f32 %1, %9, %10\n\tv_cvt_pk_f16_f32 %2, %10, %11\n\tv_cvt_pk_f16_f32 %3, %11, %12\n\tv_cvt_pk_f16_f32 %4, %12, %13\n\tv_cvt_pk_f16_f32 %5, %13, %14\n\tv_cvt_pk_f16_f32 %6, %14, %15\n\tv_cvt_pk_f16_f32 %7, %15, %8" : "+v"(u0), "+v"(u1), "+v"(u2), "+v"(u3), "+v"(u4), "+v"(u5), "+v"(u6), "+v"(u7) : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7));) }
        if (WHICH == 5) { REP16(asm volatile("v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %1, %1, %1, %1\n\tv_fma_f32 %2, %2, %2, %2\n\tv_fma_f32 %3, %3, %3, %3\n\tv_fma_f32 %4, %4, %4, %4\n\tv_fma_f32 %5, %5, %5, %5\n\tv_fma_f32 %6, %6, %6, %6\n\tv_fma_f32 %7, %7, %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y + (float)(u0 ^ u1 ^ u2 ^ u3 ^ u4 ^ u5 ^ u6 ^ u7);
}
template <int W> void run(const char* name, int wps) {
    float* d; hipMalloc(&d, 256 * 1024 * 64 * 4);
    const int iters = 2000, blocks = 256 * wps;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<W>, dim3(blocks), dim3(256), 0, 0, d, 10);
    hipEventRecord(e0); hipLaunchKernelGGL(k<W>, dim3(blocks), dim3(256), 0, 0, d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-20s %d waves/SIMD: %.2f cycles@2.4GHz per wave-instruction per SIMD\n", name, wps, ms * 1e-3 * 2.4e9 / ((double)iters * 128 * wps));
    hipFree(d);
}
int main() {
    for (int wps : {1, 2, 4}) {
        run<0>("v_add_f32", wps); run<5>("v_fma_f32", wps); run<1>("v_pk_add_f32", wps); run<2>("v_pk_fma_f32", wps);
        run<3>("v_cvt_pk_bf16_f32", wps); run<4>("v_cvt_pk_f16_f32", wps);
    }
    return 0;
}
