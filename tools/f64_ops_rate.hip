// f64_ops_rate.hip -- issue cost of the float64 vector instructions the sigmoid of lr_device.h row_term is made of, and of the cross-lane moves of
// lr_f64x.h, at 1 / 2 / 4 waves per SIMD: 8 independent register sets per instruction kind (no dependent back-to-back issue).
//   hipcc --offload-arch=gfx950 -O2 tools/f64_ops_rate.hip -o tools/bin/f64_ops_rate && tools/bin/f64_ops_rate      (round 6)
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP16(X) X X X X X X X X X X X X X X X X
template <int WHICH> __global__ void k(double* out, int iters) {
    double d0 = 1.0 + threadIdx.x * 1e-3, d1 = d0 + 1, d2 = d0 + 2, d3 = d0 + 3, d4 = d0 + 4, d5 = d0 + 5, d6 = d0 + 6, d7 = d0 + 7;
    int i0 = threadIdx.x & 3, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3, i4 = i0, i5 = i1, i6 = i2, i7 = i3;
    int j0 = 1, j1 = 2, j2 = 3, j3 = 4, j4 = 5, j5 = 6, j6 = 7, j7 = 8;
    __shared__ double lds[512];
    lds[threadIdx.x] = d0; lds[threadIdx.x + 256] = d1; __syncthreads();
    for (int i = 0; i < iters; ++i) {
        if (WHICH == 0) { REP16(asm volatile("v_fma_f64 %0, %0, %0, %0\n\tv_fma_f64 %1, %1, %1, %1\n\tv_fma_f64 %2, %2, %2, %2\n\tv_fma_f64 %3, %3, %3, %3\n\tv_fma_f64 %4, %4, %4, %4\n\tv_fma_f64 %5, %5, %5, %5\n\tv_fma_f64 %6, %6, %6, %6\n\tv_fma_f64 %7, %7, %7, %7" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7), "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7), "+v"(j0), "+v"(j1), "+v"(j2), "+v"(j3), "+v"(j4), "+v"(j5), "+v"(j6), "+v"(j7) : : "vcc");) }
        if (WHICH == 1) { REP16(asm volatile("v_add_f64 %0, %0, 1.0\n\tv_add_f64 %1, %1, 1.0\n\tv_add_f64 %2, %2, 1.0\n\tv_add_f64 %3, %3, 1.0\n\tv_add_f64 %4, %4, 1.0\n\tv_add_f64 %5, %5, 1.0\n\tv_add_f64 %6, %6, 1.0\n\tv_add_f64 %7, %7, 1.0" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7), "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7), "+v"(j0), "+v"(j1), "+v"(j2), "+v"(j3), "+v"(j4), "+v"(j5), "+v"(j6), "+v"(j7) : : "vcc");) }
        if (WHICH == 2) { REP16(asm volatile("v_mul_f64 %0, %0, 0.5\n\tv_mul_f64 %1, %1, 0.5\n\tv_mul_f64 %2, %2, 0.5\n\tv_mul_f64 %3, %3, 0.5\n\tv_mul_f64 %4, %4, 0.5\n\tv_mul_f64 %5, %5, 0.5\n\tv_mul_f64 %6, %6, 0.5\n\tv_mul_f64 %7, %7, 0.5" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7), "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7), "+v"(j0), "+v"(j1), "+v"(j2), "+v"(j3), "+v"(j4), "+v"(j5), "+v"(j6), "+v"(j7) : : "vcc");) }
        if (WHICH == 3) { REP16(asm volatile("v_max_f64 %0, %0, 1.0\n\tv_max_f64 %1, %1, 1.0\n\tv_max_f64 %2, %2, 1.0\n\tv_max_f64 %3, %3, 1.0\n\tv_max_f64 %4, %4, 1.0\n\tv_max_f64 %5, %5, 1.0\n\tv_max_f64 %6, %6, 1.0\n\tv_max_f64 %7, %7, 1.0" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7), "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7), "+v"(j0), "+v"(j1), "+v"(j2), "+v"(j3), "+v"(j4), "+v"(j5), "+v"(j6), "+v"(j7) : : "vcc");) }
        if (WHICH == 4) { REP16(asm volatile("v_rcp_f64 %0, %0\n\tv_rcp_f64 %1, %1\n\tv_rcp_f64 %2, %2\n\tv_rcp_f64 %3, %3\n\tv_rcp_f64 %4, %4\n\tv_rcp_f64 %5, %5\n\tv_rcp_f64 %6, %6\n\tv_rcp_f64 %7, %7" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7), "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7), "+v"(j0), "+v"(j1), "+v"(j2), "+v"(j3), "+v"(j4), "+v"(j5), "+v"(j6), "+v"(j7) : : "vcc");) }
        if (WHICH == 5) { REP16(asm volatile("v_rndne_f64 %0, %0\n\tv_rndne_f64 %1, %1\n\tv_rndne_f64 %2, %2\n\tv_rndne_f64 %3, %3\n\tv_rndne_f64 %4, %4\n\tv_rndne_f64 %5, %5\n\tv_rndne_f64 %6, %6\n\tv_rndne_f64 %7, %7" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7), "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7), "+v"(j0), "+v"(j1), "+v"(j2), "+v"(j3), "+v"(j4), "+v"(j5), "+v"(j6), "+v"(j7) : : "vcc");) }
        if (WHICH == 6) { REP16(asm volatile("v_ldexp_f64 %0, %0, %8\n\tv_ldexp_f64 %1, %1, %9\n\tv_ldexp_f64 %2, %2, %10\n\tv_ldexp_f64 %3, %3, %11\n\tv_ldexp_f64 %4, %4, %12\n\tv_ldexp_f64 %5, %5, %13\n\tv_ldexp_f64 %6, %6, %14\n\tv_ldexp_f64 %7, %7, %15" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7), "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7), "+v"(j0), "+v"(j1), "+v"(j2), "+v"(j3), "+v"(j4), "+v"(j5), "+v"(j6), "+v"(j7) : : "vcc");) }
        if (WHICH == 7) { REP16(asm volatile("v_cvt_i32_f64 %8, %0\n\tv_cvt_i32_f64 %9, %1\n\tv_cvt_i32_f64 %10, %2\n\tv_cvt_i32_f64 %11, %3\n\tv_cvt_i32_f64 %12, %4\n\tv_cvt_i32_f64 %13, %5\n\tv_cvt_i32_f64 %14, %6\n\tv_cvt_i32_f64 %15, %7" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7), "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7), "+v"(j0), "+v"(j1), "+v"(j2), "+v"(j3), "+v"(j4), "+v"(j5), "+v"(j6), "+v"(j7) : : "vcc");) }
        if (WHICH == 8) { REP16(asm volatile("v_cmp_lt_f64 vcc, %0, %0\n\tv_cmp_lt_f64 vcc, %1, %1\n\tv_cmp_lt_f64 vcc, %2, %2\n\tv_cmp_lt_f64 vcc, %3, %3\n\tv_cmp_lt_f64 vcc, %4, %4\n\tv_cmp_lt_f64 vcc, %5, %5\n\tv_cmp_lt_f64 vcc, %6, %6\n\tv_cmp_lt_f64 vcc, %7, %7" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7), "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7), "+v"(j0), "+v"(j1), "+v"(j2), "+v"(j3), "+v"(j4), "+v"(j5), "+v"(j6), "+v"(j7) : : "vcc");) }
        if (WHICH == 9) { REP16(asm volatile("v_cndmask_b32 %8, %8, %8, vcc\n\tv_cndmask_b32 %9, %9, %9, vcc\n\tv_cndmask_b32 %10, %10, %10, vcc\n\tv_cndmask_b32 %11, %11, %11, vcc\n\tv_cndmask_b32 %12, %12, %12, vcc\n\tv_cndmask_b32 %13, %13, %13, vcc\n\tv_cndmask_b32 %14, %14, %14, vcc\n\tv_cndmask_b32 %15, %15, %15, vcc" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7), "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7), "+v"(j0), "+v"(j1), "+v"(j2), "+v"(j3), "+v"(j4), "+v"(j5), "+v"(j6), "+v"(j7) : : "vcc");) }
        if (WHICH == 10) { REP16(asm volatile("v_mov_b64 %0, %0\n\tv_mov_b64 %1, %1\n\tv_mov_b64 %2, %2\n\tv_mov_b64 %3, %3\n\tv_mov_b64 %4, %4\n\tv_mov_b64 %5, %5\n\tv_mov_b64 %6, %6\n\tv_mov_b64 %7, %7" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7), "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7), "+v"(j0), "+v"(j1), "+v"(j2), "+v"(j3), "+v"(j4), "+v"(j5), "+v"(j6), "+v"(j7) : : "vcc");) }
        if (WHICH == 11) { REP16(asm volatile("v_lshl_add_u32 %8, %8, 20, %8\n\tv_lshl_add_u32 %9, %9, 20, %9\n\tv_lshl_add_u32 %10, %10, 20, %10\n\tv_lshl_add_u32 %11, %11, 20, %11\n\tv_lshl_add_u32 %12, %12, 20, %12\n\tv_lshl_add_u32 %13, %13, 20, %13\n\tv_lshl_add_u32 %14, %14, 20, %14\n\tv_lshl_add_u32 %15, %15, 20, %15" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7), "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7), "+v"(j0), "+v"(j1), "+v"(j2), "+v"(j3), "+v"(j4), "+v"(j5), "+v"(j6), "+v"(j7) : : "vcc");) }
        if (WHICH == 12) { REP16(asm volatile("v_permlane32_swap_b32 %8, %16\n\tv_permlane32_swap_b32 %9, %17\n\tv_permlane32_swap_b32 %10, %18\n\tv_permlane32_swap_b32 %11, %19\n\tv_permlane32_swap_b32 %12, %20\n\tv_permlane32_swap_b32 %13, %21\n\tv_permlane32_swap_b32 %14, %22\n\tv_permlane32_swap_b32 %15, %23" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7), "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7), "+v"(j0), "+v"(j1), "+v"(j2), "+v"(j3), "+v"(j4), "+v"(j5), "+v"(j6), "+v"(j7) : : "vcc");) }
        if (WHICH == 13) { REP16(asm volatile("v_mov_b32_dpp %8, %16 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %9, %17 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %10, %18 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %11, %19 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %12, %20 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %13, %21 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %14, %22 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %15, %23 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7), "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3), "+v"(i4), "+v"(i5), "+v"(i6), "+v"(i7), "+v"(j0), "+v"(j1), "+v"(j2), "+v"(j3), "+v"(j4), "+v"(j5), "+v"(j6), "+v"(j7) : : "vcc");) }
        if (WHICH == 14) { typedef double dd2 __attribute__((ext_vector_type(2))); dd2 q0, q1, q2, q3, q4, q5, q6, q7; unsigned a = 0; REP16(asm volatile("ds_read_b128 %0, %16 offset:0\n\tds_read_b128 %1, %16 offset:16\n\tds_read_b128 %2, %16 offset:32\n\tds_read_b128 %3, %16 offset:48\n\tds_read_b128 %4, %16 offset:64\n\tds_read_b128 %5, %16 offset:80\n\tds_read_b128 %6, %16 offset:96\n\tds_read_b128 %7, %16 offset:112\n\ts_waitcnt lgkmcnt(0)" : "=v"(q0), "=v"(q1), "=v"(q2), "=v"(q3), "=v"(q4), "=v"(q5), "=v"(q6), "=v"(q7), "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3), "+v"(d4), "+v"(d5), "+v"(d6), "+v"(d7) : "v"(a) : "memory");) d0 += q0.x + q1.x + q2.x + q3.x + q4.x + q5.x + q6.x + q7.x; }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = d0 + d1 + d2 + d3 + d4 + d5 + d6 + d7 + (double)(i0 ^ i1 ^ i2 ^ i3 ^ i4 ^ i5 ^ i6 ^ i7 ^ j0 ^ j1 ^ j2 ^ j3 ^ j4 ^ j5 ^ j6 ^ j7);
}
template <int W> void run(const char* name, int wps) {
    double* d; (void)hipMalloc(&d, 256 * 1024 * 64 * 8);
    const int iters = 1000, blocks = 256 * wps;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<W>, dim3(blocks), dim3(256), 0, 0, d, 10);
    (void)hipEventRecord(e0); hipLaunchKernelGGL(k<W>, dim3(blocks), dim3(256), 0, 0, d, iters); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s %d waves/SIMD: %6.2f cycles@2.4GHz per wave-instruction per SIMD\n", name, wps, ms * 1e-3 * 2.4e9 / ((double)iters * 128 * wps));
    (void)hipFree(d);
}
int main() {
    for (int wps : {1, 2, 4}) {
        run<0>("v_fma_f64", wps);
        run<1>("v_add_f64", wps);
        run<2>("v_mul_f64", wps);
        run<3>("v_max_f64", wps);
        run<4>("v_rcp_f64", wps);
        run<5>("v_rndne_f64", wps);
        run<6>("v_ldexp_f64", wps);
        run<7>("v_cvt_i32_f64", wps);
        run<8>("v_cmp_lt_f64", wps);
        run<9>("v_cndmask_b32", wps);
        run<10>("v_mov_b64", wps);
        run<11>("v_lshl_add_u32", wps);
        run<12>("v_permlane32_swap", wps);
        run<13>("v_mov_b32_dpp", wps);
        run<14>("ds_read_b128 (same address in all lanes)", wps);
    }
    return 0;
}
