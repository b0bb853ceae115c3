// lr_inst_wide.hip -- kernels for WIDE models (padded p = 64 or 128): only the stepwise engine exists at these widths.
//   float32 (LR_DTYPE=0): the exact-split / chain-split / row-split / trajectory kernels on the bf16 matrix pipe (lr_wide_bf16.h)
//   float64 (LR_DTYPE=1): one exact partial kernel on the float64 matrix pipe (lr_wide_f64.h) + the chain-split bf16 interior kernel
// Compiled four times:  hipcc -DLR_P=64|128 -DLR_DTYPE=0|1 -DLR_SFX=f32_p64 ...
#include "lr_inst.h"
#include "lr_wide_bf16.h"
#if LR_DTYPE == 1
#include "lr_wide_f64.h"
#endif

namespace lr {
namespace {

constexpr int P = LR_P;
#if LR_DTYPE == 0
using T = float;
#else
using T = double;
#endif
inline int check(hipError_t e) { return e == hipSuccess ? 0 : -2; }

// the one-launch trajectory kernels (float32 and float64 models)
int launch_tall_traj(hipStream_t st, const void* tall_args) {
    const auto& a = *static_cast<const TallArgs<T, P>*>(tall_args);
    const dim3 g2((unsigned)((a.C + 31) / 32)), g1((unsigned)((a.C + 15) / 16)), block(512);
    if (a.traj_tiles == 2) {
        if (a.traj_fmt == 2) hipLaunchKernelGGL((k_wide_traj2_bf16<P, 2, T>), g2, block, 0, st, a);
        else if (a.traj_fmt == 1) hipLaunchKernelGGL((k_wide_traj2_bf16<P, 1, T>), g2, block, 0, st, a);
        else hipLaunchKernelGGL((k_wide_traj2_bf16<P, 0, T>), g2, block, 0, st, a);
    } else {
        if (a.traj_fmt == 2) hipLaunchKernelGGL((k_wide_traj2_bf16<P, 2, T, 1>), g1, block, 0, st, a);
        else hipLaunchKernelGGL((k_wide_traj2_bf16<P, 0, T, 1>), g1, block, 0, st, a);
    }
    return check(hipGetLastError());
}

#if LR_DTYPE == 0
int launch_tall_partial(hipStream_t st, int want_value, int /*want_grad*/, const void* tall_args) {
    const auto& a = *static_cast<const TallArgs<float, P>*>(tall_args);
    const dim3 grid((unsigned)((a.C + 63) / 64), (unsigned)a.RS), block(256);
    const dim3 gridb((unsigned)((a.C + 127) / 128), (unsigned)a.RS), blockb(512);
    if (a.interior && !want_value && a.xblk1) {  // reduced-precision interior leapfrog step
        const bool h = a.traj_fmt == 2;  // rows and beta in one f16 piece each (lr_engine.h)
        if (a.RS_i > 0) {  // few chains: one chain tile per workgroup, rows split over its waves
            const dim3 gridr((unsigned)((a.C + 15) / 16), (unsigned)a.RS_i);
            if (a.rowsplit_waves == 8) {
                if (h) hipLaunchKernelGGL((k_wide_partial_bf16r<P, 8, float, true>), gridr, dim3(512), 0, st, a);
                else hipLaunchKernelGGL((k_wide_partial_bf16r<P, 8, float, false>), gridr, dim3(512), 0, st, a);
            } else {
                if (h) hipLaunchKernelGGL((k_wide_partial_bf16r<P, 4, float, true>), gridr, block, 0, st, a);
                else hipLaunchKernelGGL((k_wide_partial_bf16r<P, 4, float, false>), gridr, block, 0, st, a);
            }
        } else if (a.wide_bf16 == 2) {
            if (h) hipLaunchKernelGGL((k_wide_partial_bf16i<P, 8, float, true>), gridb, blockb, 0, st, a);
            else hipLaunchKernelGGL((k_wide_partial_bf16i<P, 8, float, false>), gridb, blockb, 0, st, a);
        } else {
            if (h) hipLaunchKernelGGL((k_wide_partial_bf16i<P, 4, float, true>), grid, block, 0, st, a);
            else hipLaunchKernelGGL((k_wide_partial_bf16i<P, 4, float, false>), grid, block, 0, st, a);
        }
        return check(hipGetLastError());
    }
    if (a.wide_bf16 == 2) {  // 8 waves x 16 chains per workgroup
        if (want_value) hipLaunchKernelGGL((k_wide_partial_bf16<P, true, 8>), gridb, blockb, 0, st, a);
        else hipLaunchKernelGGL((k_wide_partial_bf16<P, false, 8>), gridb, blockb, 0, st, a);
    } else {  // 4 waves x 16 chains
        if (want_value) hipLaunchKernelGGL((k_wide_partial_bf16<P, true, 4>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((k_wide_partial_bf16<P, false, 4>), grid, block, 0, st, a);
    }
    return check(hipGetLastError());
}

#else
int launch_tall_partial(hipStream_t st, int want_value, int /*want_grad*/, const void* tall_args) {
    const auto& a = *static_cast<const TallArgs<double, P>*>(tall_args);
    const dim3 grid((unsigned)((a.C + 63) / 64), (unsigned)a.RS), block(256);
    if (a.interior && !want_value && a.xblk1) {  // interior leapfrog step of the default precision policy: the bf16 pipe
        const bool h = a.traj_fmt == 2;  // rows and beta in one f16 piece each (lr_engine.h)
        const dim3 gridr((unsigned)((a.C + 15) / 16), (unsigned)a.RS_i);
        if (a.RS_i > 0) {  // few chains: the row-split kernel (one chain tile per workgroup), its update a launch of its own
            if (h) hipLaunchKernelGGL((k_wide_partial_bf16r<P, 8, double, true>), gridr, dim3(512), 0, st, a);
            else hipLaunchKernelGGL((k_wide_partial_bf16r<P, 8, double, false>), gridr, dim3(512), 0, st, a);
        } else {
            if (h) hipLaunchKernelGGL((k_wide_partial_bf16i<P, 4, double, true>), grid, block, 0, st, a);
            else hipLaunchKernelGGL((k_wide_partial_bf16i<P, 4, double, false>), grid, block, 0, st, a);
        }
        return check(hipGetLastError());
    }
    if (want_value) hipLaunchKernelGGL((k_wide_partial_f64<P, true>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((k_wide_partial_f64<P, false>), grid, block, 0, st, a);
    return check(hipGetLastError());
}
#endif

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

const InstTable kTable = {LR_DTYPE, P, 0, nullptr, nullptr, nullptr, &launch_tall_partial, &launch_tall_update, &launch_tall_traj, nullptr, nullptr};

}  // namespace
}  // namespace lr

#define LR_CAT2(a, b) a##b
#define LR_CAT(a, b) LR_CAT2(a, b)
extern "C" const lr::InstTable* LR_CAT(lr_inst_table_, LR_SFX)() { return &lr::kTable; }
