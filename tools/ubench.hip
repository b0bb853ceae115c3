// ubench.hip -- VALU instruction-throughput microbenchmarks on gfx950 (tools only, not product).
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench.hip -o gpurun_out/ubench ; run on the GPU box.
// Each kernel runs ITER iterations of 16 independent instances of one instruction per wave and
// reports cycles per wave-instruction (s_memtime) for 1, 2, 4 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>

#define ITER 2048

#define REP16(S) S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7) S(8) S(9) S(10) S(11) S(12) S(13) S(14) S(15)

template <int KIND> __global__ void k(float* out, long long* cyc, float seed) {
    float a[16], b[16];
    for (int i = 0; i < 16; ++i) { a[i] = seed + threadIdx.x * 1e-3f + i; b[i] = seed * 0.5f + i * 1e-2f; }
    float s = seed * 1.0001f;
    long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; ++it) {
        if constexpr (KIND == 0) {  // v_fma_f32 independent
#define S(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(s));
            REP16(S)
#undef S
        } else if constexpr (KIND == 1) {  // v_pk_fma_f32
#define S(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(*(double*)&a[(i)&~1]) : "v"(*(double*)&b[(i)&~1]), "v"(*(double*)&b[((i)+2)&14]));
            REP16(S)
#undef S
        } else if constexpr (KIND == 2) {  // v_exp_f32
#define S(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]));
            REP16(S)
#undef S
        } else if constexpr (KIND == 3) {  // v_rcp_f32
#define S(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            REP16(S)
#undef S
        } else if constexpr (KIND == 4) {  // v_add_f32 dpp quad_perm
#define S(i) asm volatile("s_nop 1\n v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]));
            REP16(S)
#undef S
        } else if constexpr (KIND == 5) {  // v_mov_b32 dpp row_mirror (no nop; independent)
#define S(i) asm volatile("v_mov_b32_dpp %0, %1 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a[i]) : "v"(b[i]));
            REP16(S)
#undef S
        } else if constexpr (KIND == 6) {  // dependent fma chain (1 accumulator)
#define S(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[0]) : "v"(b[i]), "v"(s));
            REP16(S)
#undef S
        } else if constexpr (KIND == 7) {  // 2 interleaved dependent chains
#define S(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[(i)&1]) : "v"(b[i]), "v"(s));
            REP16(S)
#undef S
        } else if constexpr (KIND == 8) {  // v_fma with SGPR operand
            float ss = __builtin_amdgcn_readfirstlane(s);
#define S(i) asm volatile("v_fmac_f32 %0, %2, %1" : "+v"(a[i]) : "v"(b[i]), "s"(ss));
            REP16(S)
#undef S
        } else if constexpr (KIND == 9) {  // v_add_f32 dpp independent, no nop (src != dst written long ago)
#define S(i) asm volatile("v_add_f32_dpp %0, %1, %1 row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(a[i]) : "v"(b[i]));
            REP16(S)
#undef S
        } else if constexpr (KIND == 10) {  // v_mul_f32
#define S(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
            REP16(S)
#undef S
        } else if constexpr (KIND == 11) {  // v_pk_mul_f32
#define S(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*(double*)&a[(i)&~1]) : "v"(*(double*)&b[(i)&~1]));
            REP16(S)
#undef S
        } else if constexpr (KIND == 12) {  // 4 interleaved dependent chains
#define S(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[(i)&3]) : "v"(b[i]), "v"(s));
            REP16(S)
#undef S
        } else if constexpr (KIND == 15) {  // 12 v_pk_fma + 4 v_exp interleaved 3:1 (does the transcendental overlap?)
#define F(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(*(double*)&a[(i)&~1]) : "v"(*(double*)&b[(i)&~1]), "v"(*(double*)&b[((i)+2)&14]));
#define E(i) asm volatile("v_exp_f32 %0, %1" : "=v"(b[i]) : "v"(b[i]));
            F(0) F(2) F(4) E(12) F(6) F(8) F(10) E(13) F(0) F(2) F(4) E(14) F(6) F(8) F(10) E(15)
#undef F
#undef E
        } else if constexpr (KIND == 16) {  // v_pk_add_f32
#define S(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(double*)&a[(i)&~1]) : "v"(*(double*)&b[(i)&~1]));
            REP16(S)
#undef S
        } else if constexpr (KIND == 17) {  // v_add_f32
#define S(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(s));
            REP16(S)
#undef S
        } else if constexpr (KIND == 18) {  // 12 v_fmac + 4 v_exp interleaved
#define F(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(b[i]), "v"(s));
#define E(i) asm volatile("v_exp_f32 %0, %1" : "=v"(b[i]) : "v"(b[i]));
            F(0) F(1) F(2) E(12) F(3) F(4) F(5) E(13) F(6) F(7) F(8) E(14) F(9) F(10) F(11) E(15)
#undef F
#undef E
        } else if constexpr (KIND == 19) {  // v_pk_fma_f32 with an SGPR-pair operand
            float ss = __builtin_amdgcn_readfirstlane(s);
            double sd; { float t2[2] = {ss, ss}; sd = *(double*)t2; }
#define S(i) asm volatile("v_pk_fma_f32 %0, %2, %1, %0" : "+v"(*(double*)&a[(i)&~1]) : "v"(*(double*)&b[(i)&~1]), "s"(sd));
            REP16(S)
#undef S
        } else if constexpr (KIND == 20) {  // v_pk_fma_f32 with op_sel broadcast of one source
#define S(i) asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(*(double*)&a[(i)&~1]) : "v"(*(double*)&b[(i)&~1]), "v"(*(double*)&b[((i)+2)&14]));
            REP16(S)
#undef S
        } else if constexpr (KIND == 21) {  // v_permlane32_swap_b32 (8 swaps of register pairs)
#define S(i) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a[i]), "+v"(b[i]));
            REP16(S)
#undef S
        } else if constexpr (KIND == 22) {  // v_permlane16_swap_b32
#define S(i) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(a[i]), "+v"(b[i]));
            REP16(S)
#undef S
        } else if constexpr (KIND == 23) {  // v_mov_b64
#define S(i) asm volatile("v_mov_b64 %0, %1" : "=v"(*(double*)&a[(i)&~1]) : "v"(*(double*)&b[(i)&~1]));
            REP16(S)
#undef S
        } else if constexpr (KIND == 24) {  // swap + dependent add (the transposing reduction step)
#define S(i) asm volatile("v_permlane32_swap_b32 %0, %1\n\ts_nop 0\n\tv_add_f32 %0, %0, %1" : "+v"(a[i]), "+v"(b[i]));
            REP16(S)
#undef S
        } else if constexpr (KIND == 25) {  // v_mul_lo_u32
#define S(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            REP16(S)
#undef S
        } else if constexpr (KIND == 26) {  // v_mul_hi_u32
#define S(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(b[i]));
            REP16(S)
#undef S
        } else if constexpr (KIND == 27) {  // v_mad_u64_u32 (full 32x32 -> 64 product in one instruction)
#define S(i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, 0" : "=v"(*(double*)&a[(i)&~1]) : "v"(b[i]), "v"(b[(i+1)&15]) : "vcc");
            REP16(S)
#undef S
        } else if constexpr (KIND == 28) {  // v_log_f32
#define S(i) asm volatile("v_log_f32 %0, %0" : "+v"(a[i]));
            REP16(S)
#undef S
        } else if constexpr (KIND == 29) {  // v_sin_f32
#define S(i) asm volatile("v_sin_f32 %0, %0" : "+v"(a[i]));
            REP16(S)
#undef S
        } else if constexpr (KIND == 30) {  // v_add_f64
#define S(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(*(double*)&a[(i)&~1]) : "v"(*(double*)&b[(i)&~1]));
            REP16(S)
#undef S
        } else if constexpr (KIND == 13) {  // f32 MFMA 16x16x4
            typedef float f4 __attribute__((ext_vector_type(4)));
            f4 acc0 = {a[0], a[1], a[2], a[3]}, acc1 = {a[4], a[5], a[6], a[7]}, acc2 = {a[8], a[9], a[10], a[11]}, acc3 = {a[12], a[13], a[14], a[15]};
            for (int r = 0; r < 4; ++r) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(b[0], b[1], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b[2], b[3], acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_16x16x4f32(b[4], b[5], acc2, 0, 0, 0);
                acc3 = __builtin_amdgcn_mfma_f32_16x16x4f32(b[6], b[7], acc3, 0, 0, 0);
            }
            a[0] = acc0[0]; a[1] = acc1[1]; a[2] = acc2[2]; a[3] = acc3[3];
        } else if constexpr (KIND == 14) {  // mixed: mfma + 12 fma (do they overlap?)
            typedef float f4 __attribute__((ext_vector_type(4)));
            f4 acc0 = {a[0], a[1], a[2], a[3]}, acc1 = {a[4], a[5], a[6], a[7]};
            for (int r = 0; r < 2; ++r) {
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(b[0], b[1], acc0, 0, 0, 0);
#define S(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[8 + ((i)&7)]) : "v"(b[i]), "v"(s));
                REP16(S)
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b[2], b[3], acc1, 0, 0, 0);
                REP16(S)
#undef S
            }
            a[0] = acc0[0]; a[1] = acc1[1];
        }
    }
    long long t1 = __builtin_readcyclecounter();
    float r = 0;
    for (int i = 0; i < 16; ++i) r += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) cyc[(blockIdx.x * blockDim.x + threadIdx.x) / 64] = t1 - t0;
}

template <int KIND> void run(const char* name, int insts_per_iter) {
    float* out; long long* cyc;
    hipMalloc(&out, 256 * 2048 * 4 * sizeof(float));
    hipMalloc(&cyc, 256 * 32 * sizeof(long long) * 4);
    for (int wps : {1, 2, 4}) {
        int threads = 256 * wps;  // wps waves per SIMD (one block per CU if <= 1024 threads)
        int blocks = 256;
        if (threads > 1024) { blocks = 256 * (threads / 1024); threads = 1024; }
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.0f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 1.0f);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        int nw = blocks * threads / 64;
        std::vector<long long> h(nw);
        hipMemcpy(h.data(), cyc, nw * sizeof(long long), hipMemcpyDeviceToHost);
        double avg = 0; for (auto v : h) avg += v; avg /= nw;
        double total_inst = (double)nw * ITER * insts_per_iter;
        // wall-based: cycles per wave-instruction per SIMD at 2.4 GHz
        double simd_cyc_per_inst = ms * 1e-3 * 2.4e9 * 1024 / total_inst;
        printf("%-28s waves/SIMD=%d  ms=%.3f  memtime/inst(wave)=%.2f  SIMD-cycles/inst@2.4GHz=%.2f\n", name, wps, ms,
               avg / (ITER * (double)insts_per_iter), simd_cyc_per_inst);
    }
    hipFree(out); hipFree(cyc);
}

int main() {
    run<0>("v_fmac_f32 indep", 16);
    run<1>("v_pk_fma_f32 indep", 16);
    run<10>("v_mul_f32 indep", 16);
    run<11>("v_pk_mul_f32 indep", 16);
    run<8>("v_fmac_f32 sgpr operand", 16);
    run<2>("v_exp_f32", 16);
    run<3>("v_rcp_f32", 16);
    run<4>("v_add_f32_dpp dep+nop", 16);
    run<9>("v_add_f32_dpp indep", 16);
    run<5>("v_mov_b32_dpp indep", 16);
    run<6>("v_fmac dependent x1", 16);
    run<7>("v_fmac dependent x2", 16);
    run<12>("v_fmac dependent x4", 16);
    run<16>("v_pk_add_f32 indep", 16);
    run<17>("v_add_f32 indep", 16);
    run<15>("12 v_pk_fma + 4 v_exp mixed", 16);
    run<18>("12 v_fmac + 4 v_exp mixed", 16);
    run<19>("v_pk_fma_f32 sgpr operand", 16);
    run<20>("v_pk_fma_f32 op_sel bcast", 16);
    run<21>("v_permlane32_swap indep", 16);
    run<22>("v_permlane16_swap indep", 16);
    run<23>("v_mov_b64 indep", 16);
    run<24>("swap32 + nop + add (2 inst)", 32);
    run<25>("v_mul_lo_u32", 16);
    run<26>("v_mul_hi_u32", 16);
    run<27>("v_mad_u64_u32", 16);
    run<28>("v_log_f32", 16);
    run<29>("v_sin_f32", 16);
    run<30>("v_add_f64", 16);
    run<13>("mfma_f32_16x16x4f32", 16);
    run<14>("mfma + 16 fmac each (2+32..)", 2 * (2 + 32));
    return 0;
}
