// lr_model.h -- the model handle of the C ABI and the small host helpers around it (error text, tuning switches, kernel tables),
// shared by lr_api.hip and by tests/host/plan_harness.hip, which runs the planner (lr_plan.h) in the GPU-less build container.
// Host code only; included once per translation unit.
#pragma once
#include "../../include/logreg_hip.h"

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "lr_inst.h"

LR_DECLARE_INST(f32_p4)
LR_DECLARE_INST(f32_p8)
LR_DECLARE_INST(f32_p16)
LR_DECLARE_INST(f32_p32)
LR_DECLARE_INST(f64_p4)
LR_DECLARE_INST(f64_p8)
LR_DECLARE_INST(f64_p16)
LR_DECLARE_INST(f64_p32)
LR_DECLARE_INST(f32_p64)   // wide models: stepwise engine with the MFMA partial kernels only
LR_DECLARE_INST(f32_p128)
LR_DECLARE_INST(f64_p64)   // wide float64 models: stepwise engine on the float64 matrix pipe (lr_wide_f64.h)
LR_DECLARE_INST(f64_p128)

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    // HIP keeps the error of a failed runtime call until somebody reads it: left there, a failed hipMalloc (LR_ERR_NOMEM) made the NEXT
    // launch's hipGetLastError() check report "launch failed" for a launch that worked (found by tests/host/engine_harness.cpp on the stub
    // runtime, round 6).  An error this library has reported is consumed here.
    if (code == LR_ERR_HIP || code == LR_ERR_NOMEM) (void)hipGetLastError();
    return code;
}

#define LR_HIP(call)                                                                                    \
    do {                                                                                                \
        hipError_t e_ = (call);                                                                         \
        if (e_ != hipSuccess) return fail(LR_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_));   \
    } while (0)

const lr::InstTable* find_table(int dtype, int P) {
    const lr::InstTable* all[] = {lr_inst_table_f32_p4(),  lr_inst_table_f32_p8(), lr_inst_table_f32_p16(),
                                  lr_inst_table_f32_p32(), lr_inst_table_f64_p4(), lr_inst_table_f64_p8(),
                                  lr_inst_table_f64_p16(), lr_inst_table_f64_p32(), lr_inst_table_f32_p64(),
                                  lr_inst_table_f32_p128(), lr_inst_table_f64_p64(), lr_inst_table_f64_p128()};
    for (const lr::InstTable* t : all)
        if (t->dtype == dtype && t->P == P) return t;
    return nullptr;
}

// ---------------------------------------------------------------------------------------------------------------------------
// A/B switches of the run path.  ONE environment variable, LOGREG_DEBUG_OPTS="key=value,key=value", is parsed ONCE per model
// (lr_model_create) into this struct; nothing on the plan or launch path reads the environment.  Defaults are the measured
// best; lr_model_debug_opts() reports what a model runs with, so that a benchmark line can say it was a default run.
struct DebugOpts {
    int residency_cap = 1;  // 0: fused chain kernels without the LDS request that spreads a grid evenly over the CUs (lr_inst.h)
    int tall_mx16 = 1;      // 0: tall models run their interior leapfrog steps on the 4-wave kernel with separate update launches
                            //    instead of the 16-wave kernel that finishes the previous step in its prologue (lr_tall_mx.h)
    int wide_traj = -1;     // wide models: 1 / 2 force the one-launch trajectory kernel with 1 / 2 chain tiles per workgroup, 0 forbids it (-1: by chain count; lr_engine.h)
    int wide_waves = 0;     // wide models: 4 | 8 waves (64 | 128 chains) per workgroup of the exact and chain-split kernels (0: by chain count)
    int wide_f16 = 1;       // wide models: 0 keeps the bf16 two-piece interior where the rows fit the one-piece f16 format, 2 uses f16 also
                            //    where LR_PREC_BF16 would take bf16 x one piece (lr_engine.h)
    bool is_default() const { return residency_cap == 1 && tall_mx16 == 1 && wide_traj == -1 && wide_waves == 0 && wide_f16 == 1; }
};
// returns false (and names the culprit) on an unknown key or a value outside its range
inline bool parse_debug_opts(const char* text, DebugOpts* out, char* bad, size_t bad_len) {
    *out = DebugOpts{};
    if (!text) return true;
    std::string s(text);
    size_t pos = 0;
    while (pos < s.size()) {
        size_t end = s.find(',', pos);
        if (end == std::string::npos) end = s.size();
        const std::string item = s.substr(pos, end - pos);
        pos = end + 1;
        if (item.empty()) continue;
        const size_t eq = item.find('=');
        const std::string key = item.substr(0, eq);
        char* rest = nullptr;
        const long v = eq == std::string::npos ? 0 : std::strtol(item.c_str() + eq + 1, &rest, 10);
        bool ok = eq != std::string::npos && rest && *rest == 0 && eq + 1 < item.size();
        // key -> (field, the values it accepts)
        struct { const char* key; int* field; long a, b, c; } const table[] = {
            {"residency_cap", &out->residency_cap, 0, 1, 1}, {"tall_mx16", &out->tall_mx16, 0, 1, 1},
            {"wide_traj", &out->wide_traj, 0, 1, 2}, {"wide_waves", &out->wide_waves, 4, 8, 8}, {"wide_f16", &out->wide_f16, 0, 1, 2}};
        bool known = false;
        for (const auto& e : table)
            if (ok && key == e.key && (v == e.a || v == e.b || v == e.c)) {
                *e.field = (int)v;
                known = true;
            }
        ok = ok && known;
        if (!ok) {
            std::snprintf(bad, bad_len, "%s", item.c_str());
            return false;
        }
    }
    return true;
}

constexpr int kMaxP = 128;
constexpr size_t kLdsBudget = 160 * 1024;

}  // namespace

struct lr_model {
    int device = 0;
    int dtype = LR_F32;
    int64_t n = 0;
    int p = 0;   // real parameter count
    int P = 0;   // padded width (4, 8, 16, 32)
    int cus = 256;
    void* d_rows = nullptr;  // [n][P] signed rows, dtype
    void* d_rows_tw = nullptr;  // float32, P <= 32: [ceil(n/2)][P][2] twisted row pairs (lr::ScalarRowPairs)
    double inv_var[kMaxP];
    double lprior_const = 0;
    const lr::InstTable* table = nullptr;
    void* d_xblk = nullptr;  // wide float32 models: per-32-row-block bf16-piece images of the rows (lr_wide_bf16.h)
    void* d_xblk1 = nullptr;  // wide models: single-piece round-to-nearest images (interior leapfrog steps)
    void* d_xblk1h = nullptr; // wide models whose rows fit the f16 range: the same in half precision (the default interior format)
    void* d_xmx = nullptr;    // float32, P = 8: two-piece bf16 tile images for interior leapfrog steps (lr_tall_mx.h)
    void* d_xmf = nullptr;    // float32, P = 8 / 16, data beyond the register variants of the matrix-core chain kernel:
                              // fp32 MFMA operand images for the end-point evaluations (lr_mfma.h)
    void* d_xms = nullptr;    // ... and beyond its LDS variant: the bf16 operand images of the interior steps in device memory
    // stepwise-engine workspaces, one per stream (grow-only, owned by the handle): calls enqueued on ONE stream
    // run in order, so they may share a workspace; calls on different streams overlap on the device and get
    // disjoint ones (two ChainSets of one model on two streams, or an eval on the NULL stream beside a run)
    DebugOpts dbg;  // A/B switches, parsed once at creation (LOGREG_DEBUG_OPTS)
    // two-part plans (lr_plan.h): the remainder's launch runs beside the head's on a stream of the handle's own, forked from and
    // joined back into the caller's stream with these events (created on first use)
    // One {side stream, fork event, join event} per CALLER stream (as the workspaces below): two ChainSets of one model on two streams
    // each fork into a side stream of their own instead of serialising their remainders through one.  The handle is not thread-safe
    // (include/logreg_hip.h): callers serialise per handle.
    struct Side { hipStream_t caller; hipStream_t stream; hipEvent_t ev_fork, ev_join; };
    std::vector<Side> sides;
    struct Ws { hipStream_t stream; void* p; size_t bytes; };
    std::vector<Ws> ws;
    size_t esize() const { return dtype == LR_F32 ? 4 : 8; }
};


namespace {

// Which operand images a model of this shape carries besides its rows (one rule for lr_model_create, which builds them, and for
// the CPU planner harness, which only needs to know they would exist):
//   tall_mx   two-piece bf16 tile images for the interior steps of the stepwise engine (narrow models the planner would
//             ever send there by itself: rows beyond 64 KB)
//   mf_end    fp32 MFMA operand images for the end points of the matrix-core chain kernel, beyond its register variants
//   wide      three-piece and one-piece bf16 block images (32 < p <= 128)
// (the device-memory operand images of the matrix-core chain kernel, d_xms, additionally need "beyond the LDS variant", which
//  only lr_plan.h's mfma_lds_bytes can say: see model_wants_xms there)
//   wide1     the one-piece image alone: wide float64 models too (interior leapfrog gradients of the default precision policy)
struct ModelImages { bool tall_mx, mf_end, wide, wide1; };
inline ModelImages model_images(int64_t n, int P, int dtype) {
    constexpr int64_t kMfmaStreamMaxRows = 8192;  // rows the matrix-core chain kernel still takes with its operands streamed from device memory
    ModelImages im{};
    // (float64 models: from the rows rounded to float32; float64 at padded p = 32 gets the image whatever the row bytes -- the stepwise engine
    //  is what runs such a model once its rows no longer fit the LDS of k_chain_dist, and under a forced LR_MODE_STEPWISE: lr_plan.h)
    im.tall_mx = P >= 8 && P <= 32 && ((size_t)n * P * (dtype == LR_F32 ? 4 : 8) > 64 * 1024 || (dtype != LR_F32 && P == 32));
    // (p = 16 / 32: from half the register variants' rows -- at the largest tile counts HMC streams its end-point operands: lr_mfma.h END_MEM)
    im.mf_end = P >= 8 && P <= 32 && dtype == LR_F32 && n > (P == 32 ? 16 * 4 * 4 : (P == 8 ? 16 * 13 : 16 * 4 * 8)) && n <= kMfmaStreamMaxRows;
    im.wide = P > 32 && dtype == LR_F32;
    im.wide1 = P > 32;
    return im;
}
inline int padded_width(int p) { return p <= 4 ? 4 : p <= 8 ? 8 : p <= 16 ? 16 : p <= 32 ? 32 : p <= 64 ? 64 : 128; }

}  // namespace
