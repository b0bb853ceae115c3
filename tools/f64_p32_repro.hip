// tools/f64_p32_repro.hip -- stand-alone reproduction of round 4's float64 defect at padded p = 32.
//
// The replicated-state chain kernel k_chain<double, 32, G, MODE_LDS, 0, KIND> (lr_kernels.h; no longer instantiated by the library)
// keeps five to seven float64 32-vectors per lane: it spills VGPRs to scratch and SGPRs to VGPR lanes.  tests/fuzz_parity.py found
// MALA on 64 lanes per chain computing wrong states in every chain.  This program runs that kernel and the distributed-state kernel
// that replaced it (k_chain_dist, verified against the float64 oracle by tests/test_gpu_parity.py) on the same problem, same Philox
// stream, and prints how far their states are apart.  Built several ways by tools/gpu/f64_repro.sh:
//     default flags of the library | -mllvm -amdgpu-spill-sgpr-to-vgpr=0 | -O3 | -O1
// If the old kernel is wrong under one set of flags and right under another, the cause is the compiler's; if it is wrong under all, ours.
// -DREPRO_R4 (with -I tools/experiments/f64_p32_r4: lr_device.h / lr_kernels.h as they were when the fuzz found the defect, commit
// 1fee576^): the replicated-state kernel of THAT source alone; `--dump <file>` writes the final states in either build, and
// tools/gpu/f64_repro.sh compares the round-4 kernel's dumps under each flag set with the distributed-state kernel's.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <string>
#include <vector>

#include "lr_kernels.h"

using namespace lr;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)

static FILE* g_dump = nullptr;

template <int G, int KIND> int run(const char* name, int n, int p, int C) {
    constexpr int P = 32;
    std::vector<double> rows((size_t)n * P, 0.0), state((size_t)C * p), lp(C);
    unsigned long long s = 12345;
    auto u = [&]() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (double)(s >> 11) / 9007199254740992.0; };
    auto z = [&]() { return std::sqrt(-2.0 * std::log(u() + 1e-300)) * std::cos(6.283185307179586 * u()); };
    for (int i = 0; i < n; ++i) {
        const double sgn = u() < 0.5 ? -1.0 : 1.0;
        for (int j = 0; j < p; ++j) rows[(size_t)i * P + j] = sgn * (j == 0 ? 1.0 : z());
    }
    const double sc = 1.0 / std::sqrt((double)(n < 4 ? 4 : n));
    for (auto& v : state) v = 0.3 * sc * z();
    for (auto& v : lp) v = -INFINITY;  // the reference's start (fit-np-mala.py:82): the first proposal is always accepted
    ModelArgs<double, P> m{};
    ChainArgs<double, P> a{};
    double *d_rows, *d_state[2], *d_lp[2];
    uint32_t* d_acc[2];
    CK(hipMalloc(&d_rows, rows.size() * 8));
    CK(hipMemcpy(d_rows, rows.data(), rows.size() * 8, hipMemcpyHostToDevice));
    for (int k = 0; k < 2; ++k) {
        CK(hipMalloc(&d_state[k], state.size() * 8));
        CK(hipMalloc(&d_lp[k], C * 8));
        CK(hipMalloc(&d_acc[k], C * 4));
        CK(hipMemcpy(d_state[k], state.data(), state.size() * 8, hipMemcpyHostToDevice));
        CK(hipMemcpy(d_lp[k], lp.data(), C * 8, hipMemcpyHostToDevice));
        CK(hipMemset(d_acc[k], 0, C * 4));
    }
    m.rows = d_rows;
    m.n = n;
    m.prior.lprior_const = 0.0;
    const double dt = 0.05 * sc * sc, eps = 0.3 * sc;
    for (int j = 0; j < P; ++j) {
        const double pre = j < p ? 0.5 + 0.05 * j : 0.0, sd = 1.0 + 0.1 * j;
        m.prior.inv_var[j] = j < p ? 1.0 / (sd * sd) : 0.0;
        if (KIND == KIND_HMC) {  // dmm = pre
            a.a[j] = j < p ? std::sqrt(pre) : 0.0;
            a.b[j] = j < p ? eps / pre : 0.0;
            a.c[j] = j < p ? 1.0 / pre : 0.0;
            a.d[j] = a.b[j];
            a.e[j] = m.prior.inv_var[j];
        } else {
            a.a[j] = KIND == KIND_RWMH ? 0.3 * sc * pre : 0.5 * pre * dt;
            a.b[j] = std::sqrt(pre * dt);
            a.c[j] = j < p ? 1.0 / (pre * dt) : 0.0;
        }
    }
    a.C = C; a.first = 0; a.count = C; a.chain_offset = 0; a.iters = 2; a.thin = 2; a.iter_offset = 0; a.seed = 71; a.p = p; a.l = 3; a.step = eps;
#ifdef REPRO_R4
    const size_t lds = (size_t)n * P * 8;  // (round 4: rows unpadded in LDS)
#else
    const size_t lds = (size_t)n * (P + kLdsRowPad<double>) * 8;
#endif
    const dim3 grid((unsigned)(((size_t)C * G + 255) / 256)), block(256);
    auto k_old = &k_chain<double, P, G, MODE_LDS, 0, KIND>;
#ifdef REPRO_R4
    auto k_new = k_old;  // (no distributed-state kernel in that source: the old kernel twice)
#else
    auto k_new = &k_chain_dist<double, P, G, MODE_LDS, 0, KIND>;
#endif
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_old), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_new), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipFuncAttributes fo{}, fn{};
    CK(hipFuncGetAttributes(&fo, reinterpret_cast<const void*>(k_old)));
    CK(hipFuncGetAttributes(&fn, reinterpret_cast<const void*>(k_new)));
    for (int k = 0; k < 2; ++k) {
        a.state = d_state[k]; a.lp_state = d_lp[k]; a.accepts = d_acc[k]; a.out = nullptr;
        if (k == 0) hipLaunchKernelGGL(k_old, grid, block, lds, 0, m, a);
        else hipLaunchKernelGGL(k_new, grid, block, lds, 0, m, a);
        CK(hipGetLastError());
        CK(hipDeviceSynchronize());
    }
    std::vector<double> s0(state.size()), s1(state.size());
    std::vector<uint32_t> a0(C), a1(C);
    CK(hipMemcpy(s0.data(), d_state[0], s0.size() * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(s1.data(), d_state[1], s1.size() * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(a0.data(), d_acc[0], C * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(a1.data(), d_acc[1], C * 4, hipMemcpyDeviceToHost));
    if (g_dump) {  // the states of the SECOND kernel (distributed-state kernel; -DREPRO_R4: the round-4 replicated-state kernel)
        fprintf(g_dump, "case %s n=%d p=%d C=%d G=%d\n", name, n, p, C, G);
        for (size_t i = 0; i < s1.size(); ++i) fprintf(g_dump, "%a\n", s1[i]);
    }
    double dmax = 0;
    int bad_chains = 0, acc_diff = 0;
    for (int c = 0; c < C; ++c) {
        double dc = 0;
        for (int j = 0; j < p; ++j) dc = std::fmax(dc, std::fabs(s0[(size_t)c * p + j] - s1[(size_t)c * p + j]));
        if (!(dc < 1e-9 * sc)) ++bad_chains;
        if (a0[c] != a1[c]) ++acc_diff;
        dmax = std::fmax(dmax, dc);
    }
    printf("%-14s n=%d p=%d C=%d lanes/chain=%d: replicated-state kernel (scratch %zu B, %d regs) vs distributed-state kernel (scratch %zu B, %d regs): "
           "max |d state| %.3g, chains apart %d of %d, accept counts differing %d  -> %s\n",
           name, n, p, C, G, (size_t)fo.localSizeBytes, fo.numRegs, (size_t)fn.localSizeBytes, fn.numRegs, dmax, bad_chains, C, acc_diff,
           bad_chains ? "DIFFERENT" : "same");
    return bad_chains ? 1 : 0;
}

int main(int argc, char** argv) {
    if (argc > 2 && std::string(argv[1]) == "--dump") g_dump = fopen(argv[2], "w");
    int rc = 0;
    rc |= run<64, KIND_MALA>("MALA", 1, 17, 15);  // (the fuzz's first finding: every chain wrong)
    rc |= run<64, KIND_MALA>("MALA", 255, 32, 64);
    rc |= run<64, KIND_MALA>("MALA", 16, 24, 64);
#ifndef REPRO_ONLY_MALA64  // (tools/gpu/f64_bisect.sh: the one kernel that was wrong, one build per flag subset)
    rc |= run<64, KIND_RWMH>("RWMH", 255, 32, 64);
    rc |= run<64, KIND_HMC>("HMC l=3", 255, 32, 64);
    rc |= run<64, KIND_UL>("UL", 255, 32, 64);
    rc |= run<16, KIND_MALA>("MALA", 255, 32, 64);
    rc |= run<16, KIND_HMC>("HMC l=3", 100, 20, 130);
#endif
    if (g_dump) fclose(g_dump);
    return rc;
}
