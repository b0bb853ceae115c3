// lr_inst.h -- type-erased launch interface between the C-ABI translation unit (lr_api.hip) and
// the per-(dtype, padded p) kernel instantiation units (lr_inst.hip compiled once per pair).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lr {

struct LaunchCfg {
    int mode, G, R, kind;
    hipStream_t stream;
    size_t lds_bytes;
    int cus;  // compute units of the model's device (0: unknown, no residency cap)
};

// Residency cap of the fused chain kernels.  A grid of k workgroups per CU finishes in k workgroup times only if the
// dispatcher spreads it evenly, and it is not obliged to when MORE than k workgroups of the kernel fit on a CU: the
// matrix-core kernel at 4096 chains (256 workgroups of 118 VGPRs and 6 KB of LDS: four fit) was measured, in about one
// process out of ten, at the launch time of 8192 or 12 288 chains for a whole run (0.548 / 0.694 ms instead of 0.381:
// two / three workgroups on one CU; profiles/r3_residency.txt).  Asking for enough LDS per workgroup that only
// ceil(blocks / CUs) of them fit on a CU leaves the even spread as the only placement.  Returns the DYNAMIC LDS bytes
// to launch with (>= lds_dynamic); grids of more than 4 workgroups per CU are left alone (the imbalance amortises).
constexpr size_t kLdsPerCu = 160 * 1024;
inline size_t capped_lds(uint64_t blocks, int cus, size_t lds_static, size_t lds_dynamic) {
    // only where the imbalance was observed: 1 .. 4 full rounds of workgroups.  A grid smaller than the chip (a single chain,
    // a few chains per launch) is left alone: nothing can double up that matters, and its small LDS footprint lets launches on
    // other streams share its CUs (two chain sets of one model on two streams).
    if (cus <= 0 || blocks < (uint64_t)cus) return lds_dynamic;
    const uint64_t cap = (blocks + (uint64_t)cus - 1) / (uint64_t)cus;
    if (cap > 4) return lds_dynamic;
    const size_t want = ((kLdsPerCu / (cap + 1) + 2048 + 1023) / 1024) * 1024;  // cap fit, cap + 1 do not
    return lds_static + lds_dynamic >= want ? lds_dynamic : want - lds_static;
}
// static LDS of a kernel, from its code object (asked once)
template <auto Kernel> inline size_t static_lds() {
    static long cached = -1;
    if (cached < 0) {
        hipFuncAttributes at;
        cached = hipFuncGetAttributes(&at, reinterpret_cast<const void*>(Kernel)) == hipSuccess ? (long)at.sharedSizeBytes : 0;
    }
    return (size_t)cached;
}
// dynamic LDS beyond 64 KB has to be asked for, per kernel and device
template <auto Kernel> inline hipError_t allow_lds(size_t dynamic_bytes) {
    if (dynamic_bytes <= 64 * 1024) return hipSuccess;
    static size_t granted[64] = {};  // per device ordinal (a race only repeats the call)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = -1;
    if (dev >= 0 && granted[dev] >= dynamic_bytes) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(Kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dynamic_bytes);
    if (e == hipSuccess && dev >= 0) granted[dev] = dynamic_bytes;
    return e;
}

struct Variant { int mode, G, R; };

struct InstTable {
    int dtype;  // LR_F32 / LR_F64
    int P;      // padded parameter width
    int nvariants;
    const Variant* variants;
    int (*launch_eval)(const LaunchCfg*, int64_t C, const void* model_args, const void* eval_args);
    int (*launch_chain)(const LaunchCfg*, int64_t C, const void* model_args, const void* chain_args);
    // stepwise (tall-data) engine, lr_tall.h; `tall_args` is a TallArgs<T,P>
    int (*launch_tall_partial)(hipStream_t, int want_value, int want_grad, const void* tall_args);
    int (*launch_tall_update)(hipStream_t, int kind, int phase, int64_t iter, int64_t out_row, int begin_next,
                              const void* tall_args);
    // wide models: all L - 1 interior leapfrog steps of an HMC trajectory in one launch (lr_wide_bf16.h); may be null
    int (*launch_tall_traj)(hipStream_t, const void* tall_args);
    // matrix-core chain kernel, operands in device memory: bytes of the bf16 operand images for n rows (0 = not available at
    // this width) and the one-off build into `store`; may be null
    size_t (*mfma_image_bytes)(int64_t n);
    int (*launch_mfma_image)(hipStream_t, const void* rows, int64_t n, void* store);
};

}  // namespace lr

#define LR_DECLARE_INST(sfx) extern "C" const lr::InstTable* lr_inst_table_##sfx();
