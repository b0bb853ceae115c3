// lr_inst.hip -- instantiates the kernels for ONE (dtype, padded p) pair.
// Compiled several times:  hipcc -DLR_T=float -DLR_P=8 -DLR_SFX=f32_p8 -DLR_DTYPE=0 ...
#include "lr_inst.h"
#include "lr_kernels.h"
#include "lr_tall.h"
#if LR_DTYPE == 0 && LR_P >= 8
#include "lr_mfma.h"
#endif
#if LR_DTYPE == 1 && LR_P == 8
#include "lr_mfma_f64.h"
#include "lr_f64x.h"
#endif
#if LR_P >= 8
#include "lr_tall_mx.h"
#endif

#ifndef LR_T
#error "compile with -DLR_T=<float|double> -DLR_P=<4|8|16|32> -DLR_SFX=<suffix> -DLR_DTYPE=<0|1>"
#endif

namespace lr {
namespace {

using T = LR_T;
constexpr int P = LR_P;

// Variant table: X(mode, lanes-per-chain G, rows-per-lane R).  REG variants keep R*P <= 128
// VGPRs of data per lane; exact R for Pima's n = 200 (G*R >= 200: 64x4, 32x7, 16x13) plus
// larger fallbacks up to the 128-register budget (n <= 64 * 128 / P).
#if LR_DTYPE == 0 && LR_P == 8
#define LR_VARIANTS(X)                                                                                \
    X(MODE_REG, 64, 4) X(MODE_REG, 32, 7) X(MODE_REG, 32, 8) X(MODE_REG, 16, 13) X(MODE_REG, 16, 16) X(MODE_REG, 32, 16) \
    X(MODE_REG, 64, 8) X(MODE_REG, 64, 12) X(MODE_REG, 64, 16) \
    X(MODE_LDS, 1, 0) X(MODE_LDS, 8, 0) X(MODE_LDS, 64, 0) X(MODE_GLOBAL, 64, 0) X(MODE_GLOBAL, 1, 0)
#elif LR_DTYPE == 0 && LR_P == 4
#define LR_VARIANTS(X) \
    X(MODE_REG, 64, 8) X(MODE_REG, 16, 16) X(MODE_REG, 16, 32) X(MODE_REG, 32, 32) X(MODE_REG, 64, 32) X(MODE_LDS, 1, 0) \
    X(MODE_LDS, 8, 0) X(MODE_LDS, 64, 0) X(MODE_GLOBAL, 64, 0) X(MODE_GLOBAL, 1, 0)
#elif LR_DTYPE == 0 && LR_P == 16
#define LR_VARIANTS(X) \
    X(MODE_REG, 64, 8) X(MODE_REG, 32, 7) X(MODE_REG, 32, 8) X(MODE_LDS, 1, 0) X(MODE_LDS, 8, 0) X(MODE_LDS, 64, 0) X(MODE_GLOBAL, 64, 0) X(MODE_GLOBAL, 1, 0)
#elif LR_DTYPE == 0 && LR_P == 32
#define LR_VARIANTS(X) X(MODE_REG, 64, 4) X(MODE_LDS, 8, 0) X(MODE_LDS, 64, 0) X(MODE_GLOBAL, 64, 0) X(MODE_GLOBAL, 1, 0)
#elif LR_DTYPE == 1 && LR_P == 8
// (rows in LDS on 32 lanes per chain: built and measured in round 5 -- 3.0e7 it/s at 4096 chains against 4.75e7 on 16 lanes -- not kept)
// float64 at Pima's width: the rows in registers as well (7 rows x 8 doubles = 112 VGPRs at 32 lanes per chain; 16 lanes x 13 rows = 208 of the 256 addressable registers:
// tried again in round 6 on the distributed-state kernel, whose state is 2 coordinates per lane -- 14 spills to scratch, 77 with scheduling fences between the rows), plain (unpacked) v_fma_f64 arithmetic -- the reference computes in float64, so its rate is reported (bench.py extra.f64)
#define LR_VARIANTS(X) X(MODE_REG, 64, 4) X(MODE_REG, 32, 7) X(MODE_LDS, 1, 0) X(MODE_LDS, 8, 0) X(MODE_LDS, 16, 0) X(MODE_LDS, 64, 0) X(MODE_GLOBAL, 64, 0) X(MODE_GLOBAL, 1, 0)
#elif LR_DTYPE == 1 && LR_P == 32
// float64 at 17 <= p <= 32: chains on the distributed-state kernel (k_chain_dist: 16 lanes own the coordinates), so lane groups of 16 and
// 64; lane groups below 16 serve lr_eval only.  (No lane-per-chain variant with rows from the scalar unit: a row is 64 SGPRs and
// k_eval spilled to scratch.)
#define LR_VARIANTS(X) X(MODE_LDS, 1, 0) X(MODE_LDS, 8, 0) X(MODE_LDS, 16, 0) X(MODE_LDS, 64, 0) X(MODE_GLOBAL, 16, 0) X(MODE_GLOBAL, 64, 0)
#else  // float64: validation-grade path, no register-resident variants
#define LR_VARIANTS(X) X(MODE_LDS, 1, 0) X(MODE_LDS, 8, 0) X(MODE_LDS, 64, 0) X(MODE_GLOBAL, 64, 0) X(MODE_GLOBAL, 1, 0)
#endif

// matrix-core variants (fp32, padded p = 8 / 16 / 32): X(row-split ways S, tiles per wave NTW), ascending NTW per S;
// n <= 16*S*NTW.  (4, 8) and (4, 16): mid-size data, n <= 512 / 1024, for HMC with bf16 interior steps.  Operand registers
// per tile: 2.5 p (p/4 + p/4..p/2 fp32 end-point operands, p bf16 interior operands), so p = 32 stops at 8 tiles.
// (4, 0): the bf16 operands in LDS instead (MfmaRowsLds: 64 p/8 bytes per row): n <= 2400 at p = 8, 1200 at p = 16; at p = 32
// the LDS holds no more rows than the registers do.  (4, -1): the same images in device memory, built once per model.
// (1, 0): 16 chains per wave, all rows, the four waves of a workgroup sharing ONE LDS image (many chains, 208 < n <= 2400).
// (8, 0): the LDS variant with the rows split over 8 waves (two per SIMD hide each other's LDS and MFMA latencies: +15 %;
// with the operands in registers the doubled per-wave fixed work costs more than that: n=200 -18 %, n=1000 -3 %).
#if LR_DTYPE == 0 && LR_P == 8
#define LR_MFMA_VARIANTS(X) X(1, 13) X(1, 0) X(4, 4) X(4, 8) X(4, 16) X(4, 0) X(4, -1) X(8, 0) X(8, -1)
#elif LR_DTYPE == 0 && LR_P == 16
#define LR_MFMA_VARIANTS(X) X(1, 13) X(4, 4) X(4, 8) X(4, 16) X(4, 0) X(4, -1) X(8, 0)
#elif LR_DTYPE == 0 && LR_P == 32
#define LR_MFMA_VARIANTS(X) X(4, 4) X(4, 8) X(4, -1)
#else
#define LR_MFMA_VARIANTS(X)
#endif

// float64, padded p = 8: HMC with float32 interior gradients (k_chain_mixed / k_chain_mixed_rep): X(lanes per chain, rows per lane)
// (p = 8 on 16 lanes: the distributed-state kernel; every other row: the replicated-state one -- the float32 register variants' shapes)
#if LR_DTYPE == 1 && LR_P == 8
#define LR_MIXED_VARIANTS(X) X(16, 13) X(16, 16) X(64, 4) X(32, 7) X(32, 8) X(32, 16) X(64, 8) X(64, 12) X(64, 16)
#elif LR_DTYPE == 1 && LR_P == 4
#define LR_MIXED_VARIANTS(X) X(64, 8) X(16, 16) X(16, 32) X(32, 32) X(64, 32)
#elif LR_DTYPE == 1 && LR_P == 16
#define LR_MIXED_VARIANTS(X) X(64, 8) X(32, 7) X(32, 8)
#else  // (p > 16: the replicated float64 state of 32 coordinates does not fit the register file beside the rows -- 410 spills)
#define LR_MIXED_VARIANTS(X)
#endif

// float64, padded p = 8: HMC on the fused matrix-core kernel, 16 chains per wave (k_chain_mfma_f64): X(16-row tiles per wave)
#if LR_DTYPE == 1 && LR_P == 8
#define LR_MFMA64_VARIANTS(X) X(13)
#else
#define LR_MFMA64_VARIANTS(X)
#endif

#define LR_VARIANT_ROW(M_, G_, R_) {M_, G_, R_},
#define LR_MIXED_ROW(G_, R_) {MODE_MIXED, G_, R_},
#define LR_MFMA64_ROW(N_) {MODE_MFMA, 1, N_},
#define LR_MFMA_ROW(S_, N_) {MODE_MFMA, S_, N_},
const Variant kVariants[] = {LR_VARIANTS(LR_VARIANT_ROW) LR_MFMA_VARIANTS(LR_MFMA_ROW) LR_MIXED_VARIANTS(LR_MIXED_ROW) LR_MFMA64_VARIANTS(LR_MFMA64_ROW)};

inline int check(hipError_t e) { return e == hipSuccess ? 0 : -2; }

inline dim3 grid_for(int64_t C, int G) {
    const int64_t lanes = C * G;
    return dim3((unsigned)((lanes + 255) / 256));
}

template <int G, int MODE, int R>
int launch_eval_v(const LaunchCfg* cfg, int64_t C, const ModelArgs<T, P>& m, const EvalArgs<T>& a) {
    hipLaunchKernelGGL((k_eval<T, P, G, MODE, R>), grid_for(C, G), dim3(256), cfg->lds_bytes, cfg->stream, m, a);
    return check(hipGetLastError());
}

// one fused chain kernel, its LDS request padded to the residency cap (lr_inst.h: capped_lds)
template <auto Kernel, typename... Args>
int launch_capped(const LaunchCfg* cfg, dim3 grid, dim3 block, size_t lds_dynamic, const Args&... args) {
    const size_t lds = capped_lds((uint64_t)grid.x * grid.y, cfg->cus, static_lds<Kernel>(), lds_dynamic);
    if (allow_lds<Kernel>(lds) != hipSuccess) return -2;
    hipLaunchKernelGGL(Kernel, grid, block, lds, cfg->stream, args...);
    return check(hipGetLastError());
}

template <int G, int MODE, int R>
int launch_chain_v(const LaunchCfg* cfg, int64_t C, const ModelArgs<T, P>& m, const ChainArgs<T, P>& a) {
    const dim3 grid = grid_for(C, G), block(256);
#if LR_P == 32
    // Padded p = 32: replicated in every lane, the state of a chain (five to seven 32-vectors) takes the whole register file and
    // more.  float64: every such kernel spilled, and MALA on 64 lanes per chain computed wrong states beside its spills (round 4:
    // tests/fuzz_parity.py, tools/f64_p32_repro.py); float32: HMC with the rows in registers spilled to scratch.  Those run on
    // k_chain_dist (lr_kernels.h: the state distributed over the 16 lanes of a DPP row); lane groups below 16 have no chain kernel
    // in float64.
    constexpr bool kDist = G >= 16 && (LR_DTYPE == 1 || MODE == MODE_REG);
    if constexpr (LR_DTYPE == 1 && G < 16) {
        (void)grid; (void)block; (void)m; (void)a;
        return -3;
    } else if constexpr (kDist) {
        switch (cfg->kind) {
        case KIND_RWMH: if constexpr (LR_DTYPE == 1) return launch_capped<&k_chain_dist<T, P, G, MODE, R, KIND_RWMH>>(cfg, grid, block, cfg->lds_bytes, m, a); else break;
        case KIND_MALA: if constexpr (LR_DTYPE == 1) return launch_capped<&k_chain_dist<T, P, G, MODE, R, KIND_MALA>>(cfg, grid, block, cfg->lds_bytes, m, a); else break;
        case KIND_UL: if constexpr (LR_DTYPE == 1) return launch_capped<&k_chain_dist<T, P, G, MODE, R, KIND_UL>>(cfg, grid, block, cfg->lds_bytes, m, a); else break;
        case KIND_HMC: return launch_capped<&k_chain_dist<T, P, G, MODE, R, KIND_HMC>>(cfg, grid, block, cfg->lds_bytes, m, a);
        default: return -1;
        }
    }
#endif
#if LR_DTYPE == 1 && LR_P == 32
    return -1;  // (every float64 kind was dispatched above)
#else
#if LR_DTYPE == 0 && LR_P == 8
    if constexpr (G == 16 && MODE == MODE_REG) {  // state distributed over the 16 lanes of a chain (lr_kernels.h)
        if (cfg->kind == KIND_RWMH) return launch_capped<&k_chain_rs16<R, KIND_RWMH>>(cfg, grid, block, 0, m, a);
        if (cfg->kind == KIND_MALA) return launch_capped<&k_chain_rs16<R, KIND_MALA>>(cfg, grid, block, 0, m, a);
    }
#endif
#if LR_DTYPE == 1 && LR_P == 8
    if constexpr (G >= 8 && MODE != MODE_GLOBAL) {  // HMC: the chain state distributed over the group, the group across the DPP rows (lr_f64x.h)
        if (cfg->kind == KIND_HMC) return launch_capped<&k_chain_f64x<G, MODE, R>>(cfg, grid, block, cfg->lds_bytes, m, a);
    }
#endif
    switch (cfg->kind) {
    case KIND_RWMH: return launch_capped<&k_chain<T, P, G, MODE, R, KIND_RWMH>>(cfg, grid, block, cfg->lds_bytes, m, a);
    case KIND_MALA: return launch_capped<&k_chain<T, P, G, MODE, R, KIND_MALA>>(cfg, grid, block, cfg->lds_bytes, m, a);
    case KIND_HMC:
#if LR_P == 32 && LR_DTYPE == 0
        if constexpr (G >= 16 && MODE == MODE_REG) return -1;  // (dispatched to k_chain_dist above)
        else
#endif
        return launch_capped<&k_chain<T, P, G, MODE, R, KIND_HMC>>(cfg, grid, block, cfg->lds_bytes, m, a);
    case KIND_UL: return launch_capped<&k_chain<T, P, G, MODE, R, KIND_UL>>(cfg, grid, block, cfg->lds_bytes, m, a);
    default: return -1;
    }
#endif
}

#if LR_DTYPE == 0 && LR_P >= 8
template <int S, int NTW>
int launch_mfma_v(const LaunchCfg* cfg, int64_t C, const ModelArgs<T, P>& m, const ChainArgs<T, P>& a) {
    const int64_t per_block = S == 1 ? 64 : 16;
    const dim3 grid((unsigned)((C + per_block - 1) / per_block)), block(S == 1 ? 256 : 64 * S);
    const size_t dyn = NTW == 0 ? cfg->lds_bytes : 0;
    switch (cfg->kind) {
    case KIND_RWMH: return launch_capped<&k_chain_mfma<P, NTW, S, KIND_RWMH>>(cfg, grid, block, dyn, m, a);
    case KIND_MALA: return launch_capped<&k_chain_mfma<P, NTW, S, KIND_MALA>>(cfg, grid, block, dyn, m, a);
    case KIND_HMC: return launch_capped<&k_chain_mfma<P, NTW, S, KIND_HMC>>(cfg, grid, block, dyn, m, a);
    case KIND_UL: return launch_capped<&k_chain_mfma<P, NTW, S, KIND_UL>>(cfg, grid, block, dyn, m, a);
    default: return -1;
    }
}
#endif

#if LR_DTYPE == 0 && LR_P >= 8
// two images, one per row split: [S = 4][S = 8 (p = 8 only: at p = 16 the 8-wave split gained nothing)]
constexpr bool kMfmaImage8 = P == 8;
size_t mfma_image_bytes4(int64_t n) { return 4 * MfmaRowsLds<P, 4, true>::bytes_per_wave(((n + 15) / 16 + 3) / 4); }
size_t mfma_image_bytes(int64_t n) {
    return mfma_image_bytes4(n) + (kMfmaImage8 ? 8 * MfmaRowsLds<P, 8, true>::bytes_per_wave(((n + 15) / 16 + 7) / 8) : 0);
}
int launch_mfma_image(hipStream_t st, const void* rows, int64_t n, void* store) {
    hipLaunchKernelGGL((k_mfma_image_build<P, 4>), dim3(1), dim3(256), 0, st, static_cast<const float*>(rows), n,
                       static_cast<unsigned char*>(store));
    if constexpr (kMfmaImage8)
        hipLaunchKernelGGL((k_mfma_image_build<P, 8>), dim3(1), dim3(512), 0, st, static_cast<const float*>(rows), n,
                           static_cast<unsigned char*>(store) + mfma_image_bytes4(n));
    return check(hipGetLastError());
}
#define LR_MFMA_IMAGE_HOOKS &mfma_image_bytes, &launch_mfma_image
#else
#define LR_MFMA_IMAGE_HOOKS nullptr, nullptr
#endif

#if LR_DTYPE == 1
template <int G, int R> int launch_mixed_v(const LaunchCfg* cfg, int64_t C, const ModelArgs<T, P>& m, const ChainArgs<T, P>& a) {
    if constexpr (G == 16 && P == 8) return launch_capped<&k_chain_mixed<R>>(cfg, grid_for(C, G), dim3(256), cfg->lds_bytes, m, a);
    else return launch_capped<&k_chain_mixed_rep<P, G, R>>(cfg, grid_for(C, G), dim3(256), cfg->lds_bytes, m, a);
}
#endif

int launch_eval(const LaunchCfg* cfg, int64_t C, const void* model_args, const void* eval_args) {
    const auto& m = *static_cast<const ModelArgs<T, P>*>(model_args);
    const auto& a = *static_cast<const EvalArgs<T>*>(eval_args);
#define LR_DISPATCH_EVAL(M_, G_, R_) \
    if (cfg->mode == M_ && cfg->G == G_ && cfg->R == R_) return launch_eval_v<G_, M_, R_>(cfg, C, m, a);
    LR_VARIANTS(LR_DISPATCH_EVAL)
    return -3;
}

int launch_chain(const LaunchCfg* cfg, int64_t C, const void* model_args, const void* chain_args) {
    const auto& m = *static_cast<const ModelArgs<T, P>*>(model_args);
    const auto& a = *static_cast<const ChainArgs<T, P>*>(chain_args);
#define LR_DISPATCH_CHAIN(M_, G_, R_) \
    if (cfg->mode == M_ && cfg->G == G_ && cfg->R == R_) return launch_chain_v<G_, M_, R_>(cfg, C, m, a);
    LR_VARIANTS(LR_DISPATCH_CHAIN)
#define LR_DISPATCH_MFMA(S_, N_) \
    if (cfg->mode == MODE_MFMA && cfg->G == S_ && cfg->R == N_) return launch_mfma_v<S_, N_>(cfg, C, m, a);
    LR_MFMA_VARIANTS(LR_DISPATCH_MFMA)
#define LR_DISPATCH_MIXED(G_, R_)                                                              \
    if (cfg->mode == MODE_MIXED && cfg->G == G_ && cfg->R == R_ && cfg->kind == KIND_HMC) \
        return launch_mixed_v<G_, R_>(cfg, C, m, a);
    LR_MIXED_VARIANTS(LR_DISPATCH_MIXED)
#define LR_DISPATCH_MFMA64(N_)                                                                 \
    if (cfg->mode == MODE_MFMA && cfg->G == 1 && cfg->R == N_ && cfg->kind == KIND_HMC)   \
        return launch_capped<&k_chain_mfma_f64<N_>>(cfg, dim3((unsigned)((C + 63) / 64)), dim3(256), cfg->lds_bytes, m, a);
    LR_MFMA64_VARIANTS(LR_DISPATCH_MFMA64)
    return -3;
}

int launch_tall_partial(hipStream_t st, int want_value, int want_grad, const void* tall_args) {
    const auto& a = *static_cast<const TallArgs<T, P>*>(tall_args);
    const dim3 grid((unsigned)a.RS, (unsigned)((a.C + 63) / 64)), block(64 * TallGeom<T, P>::NW);  // slices fastest: XCD locality
#if LR_DTYPE == 0 && LR_P >= 8
    if (a.interior && !want_value && a.xmx && a.RS_i > 0) {  // reduced-precision interior leapfrog step on the bf16 matrix pipe
        const dim3 gridm((unsigned)a.RS_i, (unsigned)((a.C + 63) / 64));
        if (a.rowsplit_waves == 16) hipLaunchKernelGGL((k_tall_partial_mx16<P>), gridm, dim3(1024), 0, st, a);
        else hipLaunchKernelGGL((k_tall_partial_mx<P, 4>), gridm, dim3(256), 0, st, a);
        return check(hipGetLastError());
    }
#elif LR_P >= 8
    if (a.interior && !want_value && a.xmx && a.RS_i > 0) {  // float64 models: the same interior kernels, position rounded on the way in
        const dim3 gridm((unsigned)a.RS_i, (unsigned)((a.C + 63) / 64));
        if (a.rowsplit_waves == 16) hipLaunchKernelGGL((k_tall_partial_mx16<P, double>), gridm, dim3(1024), 0, st, a);
        else hipLaunchKernelGGL((k_tall_partial_mx<P, 4, double>), gridm, dim3(256), 0, st, a);
        return check(hipGetLastError());
    }
#endif
    if (want_value && want_grad) hipLaunchKernelGGL((k_tall_partial<T, P, true, true>), grid, block, 0, st, a);
    else if (want_grad) hipLaunchKernelGGL((k_tall_partial<T, P, false, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((k_tall_partial<T, P, true, false>), grid, block, 0, st, a);
    return check(hipGetLastError());
}

int launch_tall_update(hipStream_t st, int kind, int phase, int64_t iter, int64_t out_row, int begin_next,
                       const void* tall_args) {
    const auto& a = *static_cast<const TallArgs<T, P>*>(tall_args);
    const dim3 grid((unsigned)((a.C * P + 255) / 256)), block(256);
    switch (kind) {
    case KIND_RWMH: hipLaunchKernelGGL((k_tall_update<T, P, KIND_RWMH>), grid, block, 0, st, a, phase, iter, out_row, begin_next); break;
    case KIND_MALA: hipLaunchKernelGGL((k_tall_update<T, P, KIND_MALA>), grid, block, 0, st, a, phase, iter, out_row, begin_next); break;
    case KIND_HMC: hipLaunchKernelGGL((k_tall_update<T, P, KIND_HMC>), grid, block, 0, st, a, phase, iter, out_row, begin_next); break;
    case KIND_UL: hipLaunchKernelGGL((k_tall_update<T, P, KIND_UL>), grid, block, 0, st, a, phase, iter, out_row, begin_next); break;
    default: return -1;
    }
    return check(hipGetLastError());
}

const InstTable kTable = {LR_DTYPE, P, (int)(sizeof(kVariants) / sizeof(kVariants[0])), kVariants, &launch_eval,
                          &launch_chain, &launch_tall_partial, &launch_tall_update, nullptr, LR_MFMA_IMAGE_HOOKS};

}  // namespace
}  // namespace lr

#define LR_CAT2(a, b) a##b
#define LR_CAT(a, b) LR_CAT2(a, b)
extern "C" const lr::InstTable* LR_CAT(lr_inst_table_, LR_SFX)() { return &lr::kTable; }
