// mfma4x4_probe.hip -- can the fp32 multiply-adds of the headline kernel (n = 200, p = 8, 4096 chains = one wave of 4 chains per
// SIMD) move to the matrix pipe in EXACT fp32 arithmetic, with the vector ALU keeping only the sigmoid?  The 16x16 MFMA shapes
// need 16 chains per wave; v_mfma_f32_4x4x1_16b_f32 is 16 independent 4x4 outer products: 4 chains x 4 rows per block, 16 blocks
// of rows per wave -- the 64-lanes-per-4-chains shape the register kernel already has.
//   lane l = (block b = l >> 2, j = l & 3); tile t, block b, slot i <-> row t 64 + 16 i + b
//   eta:   D_b[i][j] += A_b[i] B_b[j], A = x[row(t,b,i)][k] (the lane's own row), B = beta[chain j][k]   (8 MFMAs per tile, k = 0..7)
//          -> lane (b, j) reg i = eta[row(t,b,i)][chain j]
//   w = 1 / (1 + 2^eta) on the 4 registers (labels' sign and log2 e folded into the rows)
//   grad:  D_b[i][j] += A_b[i] B_b[j], A = x[row(t,b,m)][4 h + i], B = w reg m                            (8 MFMAs per tile: m = 0..3, h = 0, 1)
//          -> lane (b, j) acc[h][i] = sum over the block's rows of x[.][4 h + i] w[.][chain j]
//   sum over the 16 blocks: reduce-scatter over the 4 lane rows (v_permlane32_swap, v_permlane16_swap: coordinate k ends in lane
//   row k >> 1, register k & 1), all-reduce over the 4 blocks of a row (row_ror:4, row_ror:8); the position update touches the
//   lane's 2 coordinates; the eta MFMA of coordinate k takes B from lane row k >> 1 by the operand's own broadcast (blgp 4 + row).
// Prints cycles per gradient evaluation and wave, and checks the result against the host.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma4x4_probe.hip -o /tmp/mfma4x4 && /tmp/mfma4x4
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int NT = 4, P = 8, N = 200;

template <int CTRL> __device__ __forceinline__ float dpp(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
// (the builtins' result pair is mis-tracked by this hipcc: both halves come back as the first operand; inline asm instead)
__device__ __forceinline__ void swap32(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void swap16(float& a, float& b) { asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b)); }
template <int K> __device__ __forceinline__ f4 eta_mfma(float a, float b, f4 c) {
    return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 4 + (K >> 1));
}

// xo: [wave-invariant] own-row image [t][k][64 lanes]; xt: transposed image [t][m][h][64 lanes]; q0/p0: [chain][8]
// SKIP: the last tile holds rows 192..199 only (slot 0 of blocks 0..7): its slots 1..3 are not evaluated
template <bool SKIP>
__global__ void __launch_bounds__(256) k_leap(float* qout, const float* xo_g, const float* xt_g, const float* q0, const float* p0, float eps, float prec, int steps,
                                              long long* cyc) {
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int chain = wave * 4 + (lane & 3), lrow = lane >> 4;
    float xo[NT][P], xt[NT][4][2];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int k = 0; k < P; ++k) xo[t][k] = xo_g[(t * P + k) * 64 + lane];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int h = 0; h < 2; ++h) xt[t][m][h] = xt_g[((t * 4 + m) * 2 + h) * 64 + lane];
    }
    // the lane's two coordinates: 2 lrow, 2 lrow + 1
    float q[2] = {q0[chain * P + 2 * lrow], q0[chain * P + 2 * lrow + 1]}, p[2] = {p0[chain * P + 2 * lrow], p0[chain * P + 2 * lrow + 1]};
    const long long c0 = __builtin_readcyclecounter();
    for (int s = 0; s < steps; ++s) {
        f4 acc[2] = {f4{0, 0, 0, 0}, f4{0, 0, 0, 0}};
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            f4 e = {0, 0, 0, 0};
            e = eta_mfma<0>(xo[t][0], q[0], e); e = eta_mfma<1>(xo[t][1], q[1], e); e = eta_mfma<2>(xo[t][2], q[0], e); e = eta_mfma<3>(xo[t][3], q[1], e);
            e = eta_mfma<4>(xo[t][4], q[0], e); e = eta_mfma<5>(xo[t][5], q[1], e); e = eta_mfma<6>(xo[t][6], q[0], e); e = eta_mfma<7>(xo[t][7], q[1], e);
            const int live = (SKIP && t == NT - 1) ? 1 : 4;
#pragma unroll
            for (int m = 0; m < live; ++m) {
                const float w = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(e[m]));
                acc[0] = __builtin_amdgcn_mfma_f32_4x4x1f32(xt[t][m][0], w, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_4x4x1f32(xt[t][m][1], w, acc[1], 0, 0, 0);
            }
        }
        // reduce-scatter over the lane rows: (c_k, c_{k+4}) over the halves, then (s_k, s_{k+2}) over the row parity
        float c[8] = {acc[0][0], acc[0][1], acc[0][2], acc[0][3], acc[1][0], acc[1][1], acc[1][2], acc[1][3]};
        float sk[4];
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) { swap32(c[kk], c[kk + 4]); sk[kk] = c[kk] + c[kk + 4]; }
        float g[2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) { swap16(sk[kk], sk[kk + 2]); g[kk] = sk[kk] + sk[kk + 2]; }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            g[kk] += dpp<0x124>(g[kk]);  // row_ror:4
            g[kk] += dpp<0x128>(g[kk]);  // row_ror:8
            p[kk] += eps * (g[kk] - prec * q[kk]);
            q[kk] += eps * p[kk];
        }
    }
    const long long c1 = __builtin_readcyclecounter();
    if ((lane & 12) == 0) { qout[chain * P + 2 * lrow] = q[0]; qout[chain * P + 2 * lrow + 1] = q[1]; }
    if (lane == 0) cyc[wave] = c1 - c0;
}

int main() {
    const int chains = 4096, waves = chains / 4, steps = 50 * 20;
    std::vector<double> X(N * P);
    std::vector<float> xo(NT * P * 64, 0.f), xt(NT * 4 * 2 * 64, 0.f), q0(chains * P), p0(chains * P);
    srand(3);
    auto u = [] { return rand() / (double)RAND_MAX - 0.5; };
    for (auto& v : X) v = (float)(2 * u());
    for (auto& v : q0) v = (float)(0.6 * u());
    for (auto& v : p0) v = (float)(2 * u());
    const double log2e = 1.4426950408889634;
    for (int t = 0; t < NT; ++t)
        for (int l = 0; l < 64; ++l) {
            const int b = l >> 2, j = l & 3;
            const int own = t * 64 + 16 * j + b;
            for (int k = 0; k < P; ++k) xo[(t * P + k) * 64 + l] = own < N ? (float)(X[own * P + k] * log2e) : 0.f;
            for (int m = 0; m < 4; ++m)
                for (int h = 0; h < 2; ++h) {
                    const int r = t * 64 + 16 * m + b;
                    xt[((t * 4 + m) * 2 + h) * 64 + l] = r < N ? (float)X[r * P + 4 * h + j] : 0.f;
                }
        }
    float *dxo, *dxt, *dq0, *dp0, *dq; long long* dc;
    (void)hipMalloc(&dxo, xo.size() * 4); (void)hipMalloc(&dxt, xt.size() * 4); (void)hipMalloc(&dq0, q0.size() * 4); (void)hipMalloc(&dp0, p0.size() * 4);
    (void)hipMalloc(&dq, q0.size() * 4); (void)hipMalloc(&dc, waves * 8);
    (void)hipMemcpy(dxo, xo.data(), xo.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dxt, xt.data(), xt.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(dq0, q0.data(), q0.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(dp0, p0.data(), p0.size() * 4, hipMemcpyHostToDevice);
    const float eps = 0.02f, prec = 1.0f;
    // correctness: 3 steps against the host in double
    for (int skip = 0; skip < 2; ++skip) {
        if (skip) hipLaunchKernelGGL(k_leap<true>, dim3(waves / 4), dim3(256), 0, 0, dq, dxo, dxt, dq0, dp0, eps, prec, 3, dc);
        else hipLaunchKernelGGL(k_leap<false>, dim3(waves / 4), dim3(256), 0, 0, dq, dxo, dxt, dq0, dp0, eps, prec, 3, dc);
        std::vector<float> q(chains * P);
        (void)hipMemcpy(q.data(), dq, q.size() * 4, hipMemcpyDeviceToHost);
        double worst = 0;
        for (int ch = 0; ch < chains; ch += 37) {
            double qq[P], pp[P];
            for (int k = 0; k < P; ++k) { qq[k] = q0[ch * P + k]; pp[k] = p0[ch * P + k]; }
            for (int s = 0; s < 3; ++s) {
                double g[P] = {0};
                for (int r = 0; r < N; ++r) {
                    double e = 0;
                    for (int k = 0; k < P; ++k) e += X[r * P + k] * qq[k];
                    const double w = 1 / (1 + std::exp(e));
                    for (int k = 0; k < P; ++k) g[k] += w * X[r * P + k];
                }
                for (int k = 0; k < P; ++k) { pp[k] += eps * (g[k] - prec * qq[k]); qq[k] += eps * pp[k]; }
            }
            for (int k = 0; k < P; ++k) worst = std::fmax(worst, std::fabs(qq[k] - q[ch * P + k]));
        }
        printf("skip=%d: max |q_device - q_host| after 3 steps = %.3g\n", skip, worst);
    }
    for (int skip = 0; skip < 2; ++skip) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        for (int rep = 0; rep < 300; ++rep) {
            if (rep == 100) (void)hipEventRecord(e0);
            if (skip) hipLaunchKernelGGL(k_leap<true>, dim3(waves / 4), dim3(256), 0, 0, dq, dxo, dxt, dq0, dp0, 1e-4f, prec, steps, dc);
            else hipLaunchKernelGGL(k_leap<false>, dim3(waves / 4), dim3(256), 0, 0, dq, dxo, dxt, dq0, dp0, 1e-4f, prec, steps, dc);
        }
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> cyc(waves);
        (void)hipMemcpy(cyc.data(), dc, waves * 8, hipMemcpyDeviceToHost);
        const double per = ms * 1e-3 / 200 / steps;
        printf("skip=%d: %.1f ns per evaluation (4096 chains, one wave per SIMD), %.0f s_memtime ticks; %.3e chain-evaluations/s; %.1f TFLOP/s of 2 x 2 n p\n", skip,
               per * 1e9, (double)cyc[0] / steps, chains / per, chains / per * 4.0 * N * P / 1e12);
    }
    return 0;
}
