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
};

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
    // wide models: persistent row-split trajectory kernel (lr_wide_persist.h) and its LDS need per blocks-per-slice; may be null
    int (*launch_tall_traj_rs)(hipStream_t, const void* tall_args);
    size_t (*traj_rs_lds_bytes)(int blocks_per_slice);
};

}  // namespace lr

#define LR_DECLARE_INST(sfx) extern "C" const lr::InstTable* lr_inst_table_##sfx();
