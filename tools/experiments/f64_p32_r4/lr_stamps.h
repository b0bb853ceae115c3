// lr_stamps.h -- DEVELOPMENT INSTRUMENTATION, compiled only into builds made with LOGREG_HIPCC_FLAGS=-DLR_STAMPS (tools/stamps.py):
// per-wave time stamps and cycle counters inside the stepwise interior kernels, per-phase cycle counters in the MALA / RWMH
// chain kernel, and timing experiments that knowingly break results (LOGREG_DEBUG_EXP bits).  In a production build every macro
// below is empty and the kernel argument structs carry none of the fields; the library refuses to be loaded as a development
// build unless the caller asks for one (logreg_amd/_lib.py).  The reference has no counterpart.
//
//   device side (kernels that take a TallArgs `a`):
//     LR_STAMP(a, k)        lane 0 of every wave records the 100 MHz wall clock at point k < 8
//     LR_STAMP_CLK(a, k)    ... the shader clock, 8 <= k < 16
//     LR_STAMP_AT(a, k)     the slot itself (for accumulated counters)
//     LR_DBG(a, bit)        experiment switch `bit` of LOGREG_DEBUG_EXP (0 in production)
//   k_chain_rs16:  LR_RS16_PHASES_BEGIN / LR_RS16_PHASE(k) / LR_RS16_PHASES_REPORT(kind, iterations)
//   host side (lr_api.hip / lr_engine.h):  LR_STAMPS_ARM(a) / LR_STAMPS_DISARM(a) around an interior-step launch,
//     lr_debug_read_stamps() to fetch the buffer.
#pragma once

#ifdef LR_STAMPS
#define LR_STAMP_FIELDS                                                                                     \
    unsigned long long* stamps; /* [launch slot][workgroup][16 waves][16] */                                \
    int stamp_slot;                                                                                         \
    int dbg; /* LOGREG_DEBUG_EXP: timing experiments that knowingly break the results */
#define LR_STAMP_AT(a, k)                                                                                                    \
    (a).stamps[((((size_t)(a).stamp_slot * (gridDim.x * gridDim.y) + blockIdx.y * gridDim.x + blockIdx.x) * 16) + (threadIdx.x >> 6)) * 16 + (k)]
#define LR_STAMP(a, k) do { if ((a).stamps && (threadIdx.x & 63) == 0) LR_STAMP_AT(a, k) = __builtin_amdgcn_s_memrealtime(); } while (0)
#define LR_STAMP_CLK(a, k) do { if ((a).stamps && (threadIdx.x & 63) == 0) LR_STAMP_AT(a, k) = __builtin_amdgcn_s_memtime(); } while (0)
#define LR_DBG(a, bit) (((a).dbg >> (bit)) & 1)
#define LR_RS16_PHASES_BEGIN unsigned long long ph[4] = {0, 0, 0, 0}, tp = __builtin_amdgcn_s_memtime();
#define LR_RS16_PHASE(k) do { const unsigned long long tn_ = __builtin_amdgcn_s_memtime(); ph[k] += tn_ - tp; tp = tn_; } while (0)
#define LR_RS16_PHASES_REPORT(kind, its)                                                                                                       \
    if (blockIdx.x == 0 && threadIdx.x == 0 && (its) >= 100)                                                                                   \
        printf("k_chain_rs16 kind %d, %lld iterations, shader cycles per iteration: draws %.1f, proposal + evaluation %.1f, accept %.1f, loop %.1f\n", \
               (kind), (long long)(its), (double)ph[0] / (its), (double)ph[1] / (its), (double)ph[2] / (its), (double)ph[3] / (its));
#else
#define LR_STAMP_FIELDS
#define LR_STAMP(a, k) do { } while (0)
#define LR_STAMP_CLK(a, k) do { } while (0)
#define LR_DBG(a, bit) 0
#define LR_RS16_PHASES_BEGIN
#define LR_RS16_PHASE(k) do { } while (0)
#define LR_RS16_PHASES_REPORT(kind, its)
#endif

// ---- host side: only where LR_STAMPS_HOST is defined before inclusion (lr_api.hip)
#if defined(LR_STAMPS_HOST)
#ifdef LR_STAMPS
namespace {
constexpr int kStampSlots = 64, kStampWgs = 512;
int g_stamp_slot = 0;
unsigned long long* g_stamp_buf = nullptr;
constexpr size_t kStampBytes = (size_t)kStampSlots * kStampWgs * 16 * 16 * sizeof(unsigned long long);
unsigned long long* stamp_buffer() {
    if (!g_stamp_buf && (hipMalloc(&g_stamp_buf, kStampBytes) != hipSuccess || hipMemset(g_stamp_buf, 0, kStampBytes) != hipSuccess))
        g_stamp_buf = nullptr;
    return g_stamp_buf;
}
}  // namespace
#define LR_STAMPS_ARM(a)                                                             \
    do {                                                                             \
        (a).stamps = g_stamp_slot < kStampSlots ? stamp_buffer() : nullptr;          \
        (a).stamp_slot = g_stamp_slot++;                                             \
        const char* e_ = getenv("LOGREG_DEBUG_EXP");                                 \
        (a).dbg = e_ ? atoi(e_) : 0;                                                 \
    } while (0)
#define LR_STAMPS_DISARM(a) do { (a).stamps = nullptr; } while (0)
// copy the stamp buffer [slots][512 workgroups][16 waves][16] to the host, restart the slot counter
extern "C" __attribute__((visibility("default"))) int lr_debug_read_stamps(unsigned long long* out, int* slots, int* wgs) {
    if (slots) *slots = kStampSlots;
    if (wgs) *wgs = kStampWgs;
    if (!g_stamp_buf || !out) return 0;
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    if (hipMemcpy(out, g_stamp_buf, kStampBytes, hipMemcpyDeviceToHost) != hipSuccess) return -2;
    if (hipMemset(g_stamp_buf, 0, kStampBytes) != hipSuccess) return -2;
    g_stamp_slot = 0;
    return 0;
}
#else
#define LR_STAMPS_ARM(a) do { } while (0)
#define LR_STAMPS_DISARM(a) do { } while (0)
#endif
#endif
