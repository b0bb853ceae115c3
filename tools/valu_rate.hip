// valu_rate.hip -- issue cost of the non-transcendental VALU ops of the sigmoid loops next to v_exp_f32 / v_rcp_f32, 4 waves per SIMD
// (development tool).   hipcc --offload-arch=gfx950 -O2 tools/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(X) X X X X X X X X
typedef float f2 __attribute__((ext_vector_type(2)));
template <int W> __global__ void k(float* out, int iters) {
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, b0 = a0 * 0.5f, b1 = a0 * 0.25f;
    f2 p0 = {a0, a1}, p1 = {a1, a2}, p2 = {a2, a3}, p3 = {a3, a4}, p4 = {a4, a5}, p5 = {a5, a6}, p6 = {a6, a7}, p7 = {a7, a0}, q0 = {b0, b1}, q1 = {b1, b0};
#define OP32(s) asm volatile(s : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1))
#define OP64(s) asm volatile(s : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(q0), "v"(q1))
    for (int i = 0; i < iters; ++i) {
        if (W == 0) { REP8(OP32("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7");) }
        if (W == 6) { REP8(OP32("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7");) }
        if (W == 3) { REP8(OP32("v_add_f32 %0, 1.0, %0\n v_add_f32 %1, 1.0, %1\n v_add_f32 %2, 1.0, %2\n v_add_f32 %3, 1.0, %3\n v_add_f32 %4, 1.0, %4\n v_add_f32 %5, 1.0, %5\n v_add_f32 %6, 1.0, %6\n v_add_f32 %7, 1.0, %7");) }
        if (W == 2) { REP8(OP32("v_cvt_pk_bf16_f32 %0, %0, %8\n v_cvt_pk_bf16_f32 %1, %1, %8\n v_cvt_pk_bf16_f32 %2, %2, %8\n v_cvt_pk_bf16_f32 %3, %3, %8\n v_cvt_pk_bf16_f32 %4, %4, %8\n v_cvt_pk_bf16_f32 %5, %5, %8\n v_cvt_pk_bf16_f32 %6, %6, %8\n v_cvt_pk_bf16_f32 %7, %7, %8");) }
        if (W == 5) { REP8(OP32("v_perm_b32 %0, %0, %8, %9\n v_perm_b32 %1, %1, %8, %9\n v_perm_b32 %2, %2, %8, %9\n v_perm_b32 %3, %3, %8, %9\n v_perm_b32 %4, %4, %8, %9\n v_perm_b32 %5, %5, %8, %9\n v_perm_b32 %6, %6, %8, %9\n v_perm_b32 %7, %7, %8, %9");) }
        if (W == 7) { REP8(OP32("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8");) }
        if (W == 1) { REP8(OP64("v_pk_add_f32 %0, %0, 1.0 op_sel_hi:[1,0]\n v_pk_add_f32 %1, %1, 1.0 op_sel_hi:[1,0]\n v_pk_add_f32 %2, %2, 1.0 op_sel_hi:[1,0]\n v_pk_add_f32 %3, %3, 1.0 op_sel_hi:[1,0]\n v_pk_add_f32 %4, %4, 1.0 op_sel_hi:[1,0]\n v_pk_add_f32 %5, %5, 1.0 op_sel_hi:[1,0]\n v_pk_add_f32 %6, %6, 1.0 op_sel_hi:[1,0]\n v_pk_add_f32 %7, %7, 1.0 op_sel_hi:[1,0]");) }
        if (W == 4) { REP8(OP64("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9");) }
        if (W == 9) { REP8(OP64("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8");) }
        if (W == 8) { REP8(OP32("v_exp_f32 %0, %0\n v_add_f32 %1, 1.0, %1\n v_exp_f32 %2, %2\n v_add_f32 %3, 1.0, %3\n v_exp_f32 %4, %4\n v_add_f32 %5, 1.0, %5\n v_exp_f32 %6, %6\n v_add_f32 %7, 1.0, %7");) }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
}
template <int W> void run(const char* name, int wps, float* d) {
    const int iters = 2000, blocks = 256 * wps;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<W>, dim3(blocks), dim3(256), 0, 0, d, 200);
    (void)hipEventRecord(e0); hipLaunchKernelGGL(k<W>, dim3(blocks), dim3(256), 0, 0, d, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %d waves/SIMD: %.2f ns per wave-instruction per SIMD (%.2f cycles at 2.4 GHz)\n", name, wps, ms * 1e6 / ((double)iters * 64 * wps), ms * 1e-3 * 2.4e9 / ((double)iters * 64 * wps));
}
int main() {
    float* d; (void)hipMalloc(&d, 256 * 1024 * 64 * 4);
    const int wps = 4;
    run<0>("v_exp_f32", wps, d); run<6>("v_rcp_f32", wps, d); run<1>("v_pk_add_f32 v, v, 1.0", wps, d); run<3>("v_add_f32 v, 1.0, v", wps, d);
    run<2>("v_cvt_pk_bf16_f32", wps, d); run<4>("v_pk_fma_f32", wps, d); run<9>("v_pk_mul_f32", wps, d); run<5>("v_perm_b32", wps, d); run<7>("v_add_u32", wps, d);
    run<8>("v_exp_f32 / v_add_f32 alternating", wps, d);
    return 0;
}
