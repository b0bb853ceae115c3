// int_rate.hip -- issue cost of the integer / special ops of the Philox + Box-Muller draw path (development tool), 2 and 4 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O2 tools/int_rate.hip -o /tmp/int_rate && /tmp/int_rate
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define REP8(X) X X X X X X X X
template <int W> __global__ void __launch_bounds__(256) k(uint32_t* out, int iters) {
    uint32_t a0 = threadIdx.x * 2654435761u + 1, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 9, a5 = a0 * 11, a6 = a0 * 13, a7 = a0 * 15, b0 = a0 | 1, b1 = 0xCD9E8D57u;
    uint64_t w0 = a0, w1 = a1, w2 = a2, w3 = a3;
    float f0 = threadIdx.x * 1e-3f + 1, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3;
#define OP8(s) asm volatile(s : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b0), "v"(b1));
#define OPW(s) asm volatile(s : "+v"(w0), "+v"(w1), "+v"(w2), "+v"(w3) : "v"(b0), "v"(b1) : "vcc");
#define OPF(s) asm volatile(s : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(b0), "v"(b1));
    for (int i = 0; i < iters; ++i) {
        if (W == 0) { REP8(OP8("v_mul_lo_u32 %0, %0, %9\n v_mul_lo_u32 %1, %1, %9\n v_mul_lo_u32 %2, %2, %9\n v_mul_lo_u32 %3, %3, %9\n v_mul_lo_u32 %4, %4, %9\n v_mul_lo_u32 %5, %5, %9\n v_mul_lo_u32 %6, %6, %9\n v_mul_lo_u32 %7, %7, %9")); }
        if (W == 1) { REP8(OP8("v_mul_hi_u32 %0, %0, %9\n v_mul_hi_u32 %1, %1, %9\n v_mul_hi_u32 %2, %2, %9\n v_mul_hi_u32 %3, %3, %9\n v_mul_hi_u32 %4, %4, %9\n v_mul_hi_u32 %5, %5, %9\n v_mul_hi_u32 %6, %6, %9\n v_mul_hi_u32 %7, %7, %9")); }
        if (W == 2) { REP8(OPW("v_mad_u64_u32 %0, vcc, %4, %5, 0\n v_mad_u64_u32 %1, vcc, %4, %5, 0\n v_mad_u64_u32 %2, vcc, %4, %5, 0\n v_mad_u64_u32 %3, vcc, %4, %5, 0\n v_mad_u64_u32 %0, vcc, %4, %5, 0\n v_mad_u64_u32 %1, vcc, %4, %5, 0\n v_mad_u64_u32 %2, vcc, %4, %5, 0\n v_mad_u64_u32 %3, vcc, %4, %5, 0")); }
        if (W == 3) { REP8(OP8("v_xor_b32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_xor_b32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %8\n v_xor_b32 %6, %6, %8\n v_xor_b32 %7, %7, %8")); }
        if (W == 4) { REP8(OP8("v_mul_u32_u24 %0, %0, %9\n v_mul_u32_u24 %1, %1, %9\n v_mul_u32_u24 %2, %2, %9\n v_mul_u32_u24 %3, %3, %9\n v_mul_u32_u24 %4, %4, %9\n v_mul_u32_u24 %5, %5, %9\n v_mul_u32_u24 %6, %6, %9\n v_mul_u32_u24 %7, %7, %9")); }
        if (W == 5) { REP8(OPF("v_log_f32 %0, %0\n v_log_f32 %1, %1\n v_log_f32 %2, %2\n v_log_f32 %3, %3\n v_sqrt_f32 %0, %0\n v_sqrt_f32 %1, %1\n v_sin_f32 %2, %2\n v_cos_f32 %3, %3")); }
        if (W == 6) { REP8(OPF("v_fma_f32 %0, %0, %0, %1\n v_fma_f32 %1, %1, %1, %2\n v_fma_f32 %2, %2, %2, %3\n v_fma_f32 %3, %3, %3, %0\n v_fma_f32 %0, %0, %0, %1\n v_fma_f32 %1, %1, %1, %2\n v_fma_f32 %2, %2, %2, %3\n v_fma_f32 %3, %3, %3, %0")); }
        if (W == 7) { REP8(OPF("v_fmac_f32 %0, %1, %2\n v_fmac_f32 %1, %2, %3\n v_fmac_f32 %2, %3, %0\n v_fmac_f32 %3, %0, %1\n v_fmac_f32 %0, %1, %2\n v_fmac_f32 %1, %2, %3\n v_fmac_f32 %2, %3, %0\n v_fmac_f32 %3, %0, %1")); }
        if (W == 8) { REP8(OPF("v_add_f64 %0, %0, %0\n v_add_f64 %1, %1, %1\n v_add_f64 %2, %2, %2\n v_add_f64 %3, %3, %3\n v_add_f64 %0, %0, %0\n v_add_f64 %1, %1, %1\n v_add_f64 %2, %2, %2\n v_add_f64 %3, %3, %3")); }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (uint32_t)(w0 + w1 + w2 + w3) + (uint32_t)(f0 + f1 + f2 + f3);
}
template <int W> void run(const char* name, int wps, uint32_t* d) {
    const int iters = 2000, blocks = 256 * wps;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<W>, dim3(blocks), dim3(256), 0, 0, d, 200);
    (void)hipEventRecord(e0); hipLaunchKernelGGL(k<W>, dim3(blocks), dim3(256), 0, 0, d, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-40s %d waves/SIMD: %.2f ns per wave-instruction per SIMD\n", name, wps, ms * 1e6 / ((double)iters * 64 * wps));
}
int main() {
    uint32_t* d; (void)hipMalloc(&d, 256 * 1024 * 64 * 4);
    for (int wps : {2, 4}) {
        run<0>("v_mul_lo_u32", wps, d); run<1>("v_mul_hi_u32", wps, d); run<2>("v_mad_u64_u32 (lo and hi in one)", wps, d); run<3>("v_xor_b32", wps, d);
        run<4>("v_mul_u32_u24", wps, d); run<5>("v_log / v_sqrt / v_sin / v_cos mix", wps, d); run<6>("v_fma_f32 (VOP3)", wps, d); run<7>("v_fmac_f32 (VOP2)", wps, d);
    }
    return 0;
}
