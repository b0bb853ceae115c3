// lr_inst_wide.hip -- kernels for WIDE models (padded p = 64 or 128, float32): only the stepwise
// engine exists at these widths; its partial kernel is the MFMA GEMM of lr_wide.h.
// Compiled twice:  hipcc -DLR_P=64 -DLR_SFX=f32_p64 ... / -DLR_P=128 -DLR_SFX=f32_p128 ...
#include "lr_inst.h"
#include "lr_wide.h"
#include "lr_wide_bf16.h"
#include "lr_wide_persist.h"

namespace lr {
namespace {

constexpr int P = LR_P;
inline int check(hipError_t e) { return e == hipSuccess ? 0 : -2; }

int launch_tall_partial(hipStream_t st, int want_value, int /*want_grad*/, const void* tall_args) {
    const auto& a = *static_cast<const TallArgs<float, P>*>(tall_args);
    const dim3 grid((unsigned)((a.C + 63) / 64), (unsigned)a.RS), block(256);
    if (a.interior && !want_value && a.wide_bf16 >= 1 && a.xblk1) {  // reduced-precision interior leapfrog step
        if (a.RS_i > 0) {  // few chains: one chain tile per workgroup, rows split over its waves
            const dim3 gridr((unsigned)((a.C + 15) / 16), (unsigned)a.RS_i);
            if (a.rowsplit_waves == 8) hipLaunchKernelGGL((k_wide_partial_bf16r<P, 8>), gridr, dim3(512), 0, st, a);
            else hipLaunchKernelGGL((k_wide_partial_bf16r<P, 4>), gridr, block, 0, st, a);
        } else if (a.wide_bf16 == 2) {
            const dim3 gridb((unsigned)((a.C + 127) / 128), (unsigned)a.RS), blockb(512);
            hipLaunchKernelGGL((k_wide_partial_bf16i<P, 8>), gridb, blockb, 0, st, a);
        } else {
            hipLaunchKernelGGL((k_wide_partial_bf16i<P, 4>), grid, block, 0, st, a);
        }
        return check(hipGetLastError());
    }
    if (a.wide_bf16 == 2) {  // 8 waves x 16 chains per workgroup
        const dim3 gridb((unsigned)((a.C + 127) / 128), (unsigned)a.RS), blockb(512);
        if (want_value) hipLaunchKernelGGL((k_wide_partial_bf16<P, true, 8>), gridb, blockb, 0, st, a);
        else hipLaunchKernelGGL((k_wide_partial_bf16<P, false, 8>), gridb, blockb, 0, st, a);
    } else if (a.wide_bf16 == 1) {  // 4 waves x 16 chains
        if (want_value) hipLaunchKernelGGL((k_wide_partial_bf16<P, true, 4>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((k_wide_partial_bf16<P, false, 4>), grid, block, 0, st, a);
    } else {
        if (want_value) hipLaunchKernelGGL((k_wide_partial<P, true>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((k_wide_partial<P, false>), grid, block, 0, st, a);
    }
    return check(hipGetLastError());
}

int launch_tall_update(hipStream_t st, int kind, int phase, int64_t iter, int64_t out_row, int begin_next,
                       const void* tall_args) {
    const auto& a = *static_cast<const TallArgs<float, P>*>(tall_args);
    const dim3 grid((unsigned)((a.C * P + 255) / 256)), block(256);
    switch (kind) {
    case KIND_RWMH: hipLaunchKernelGGL((k_tall_update<float, P, KIND_RWMH>), grid, block, 0, st, a, phase, iter, out_row, begin_next); break;
    case KIND_MALA: hipLaunchKernelGGL((k_tall_update<float, P, KIND_MALA>), grid, block, 0, st, a, phase, iter, out_row, begin_next); break;
    case KIND_HMC: hipLaunchKernelGGL((k_tall_update<float, P, KIND_HMC>), grid, block, 0, st, a, phase, iter, out_row, begin_next); break;
    case KIND_UL: hipLaunchKernelGGL((k_tall_update<float, P, KIND_UL>), grid, block, 0, st, a, phase, iter, out_row, begin_next); break;
    default: return -1;
    }
    return check(hipGetLastError());
}

int launch_tall_traj(hipStream_t st, const void* tall_args) {
    const auto& a = *static_cast<const TallArgs<float, P>*>(tall_args);
    hipLaunchKernelGGL((k_wide_traj_bf16<P>), dim3((unsigned)((a.C + 15) / 16)), dim3(512), 0, st, a);
    return check(hipGetLastError());
}

// persistent row-split trajectory kernel (lr_wide_persist.h): a.traj_S slices per group of 32 chains, one workgroup per
// (group, slice); the step flags are zeroed before every launch (cdna_hip_programming.md Guideline 16: re-initialise every call)
int launch_tall_traj_rs(hipStream_t st, const void* tall_args) {
    const auto& a = *static_cast<const TallArgs<float, P>*>(tall_args);
    const int ngroups = (int)((a.C + kPersistChains - 1) / kPersistChains);
    const size_t lds = persist_lds_bytes<P>(a.traj_nbs);
    static size_t lds_set = 0;
    if (lds > lds_set) {  // dynamic LDS beyond 64 KB has to be asked for
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_wide_traj_rs<P>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return -2;
        lds_set = lds;
    }
    if (hipMemsetAsync(a.xflags, 0, (size_t)ngroups * a.traj_S * sizeof(uint32_t), st) != hipSuccess) return -2;
    hipLaunchKernelGGL((k_wide_traj_rs<P>), dim3((unsigned)(ngroups * a.traj_S)), dim3(64 * kPersistWaves), lds, st, a);
    return check(hipGetLastError());
}
size_t traj_rs_lds_bytes(int blocks_per_slice) { return persist_lds_bytes<P>(blocks_per_slice); }

const InstTable kTable = {0, P, 0, nullptr, nullptr, nullptr, &launch_tall_partial, &launch_tall_update, &launch_tall_traj, nullptr, nullptr,
                          &launch_tall_traj_rs, &traj_rs_lds_bytes};

}  // namespace
}  // namespace lr

#define LR_CAT2(a, b) a##b
#define LR_CAT(a, b) LR_CAT2(a, b)
extern "C" const lr::InstTable* LR_CAT(lr_inst_table_, LR_SFX)() { return &lr::kTable; }
