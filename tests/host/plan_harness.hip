// plan_harness.hip -- the planner of the C ABI (logreg_amd/csrc/lr_plan.h) as a host program: no GPU, no HIP call.
// Test infrastructure (tests/test_planner_cpu.py builds it with hipcc against the library's own instantiation objects, so the
// variant tables are the real ones): the host-side logic that picks kernels and sizes their LDS can then be exercised in the
// GPU-less build container instead of with metered GPU minutes.
//   stdin:  one request per line   dtype(0|1) p n chains kind(0 rwmh,1 mala,2 hmc,3 ul) precision(0 auto,1 full,2 bf16) group mode cus
//   stdout: one line per request   "<mode> <group> <rows_per_lane> <lds_bytes> <RS_i> <slice_len_i> <waves_i> <split> <tail_group> <tail_rows> <tail_mode>"   or   "ERR <status> <message>"
//           (the last three: lr_plan.h plan_interior -- row slices of the reduced-precision interior leapfrog steps of the stepwise
//            engine for this model and chain count; RS_i = 0: none)
#include "lr_model.h"

#include "lr_kernels.h"
#include "lr_mfma.h"
#include "lr_tall.h"

#include "lr_plan.h"
#include "lr_wide_bf16.h"

#include <cstring>
#include <vector>

// "f16": float32 bit patterns (hex, one per line) -> the f16 bit pattern lr_model_create writes into the half-precision image;
// "f16fit <n>": n x 64 float32 bit patterns -> "1 <sum of the image's 16-bit words>" or "0" (the range rule of wide_f16_prepare_rne)
static int f16_modes(const char* mode) {
    unsigned u;
    if (!std::strcmp(mode, "f16")) {
        while (std::scanf("%x", &u) == 1) {
            float x;
            std::memcpy(&x, &u, 4);
            std::printf("%04x\n", (unsigned)lr::f16_bits_rne(x));
        }
        return 0;
    }
    long long n = 0;
    if (std::scanf("%lld", &n) != 1 || n <= 0) return 2;
    std::vector<float> rows((size_t)n * 64);
    for (auto& x : rows) {
        if (std::scanf("%x", &u) != 1) return 2;
        std::memcpy(&x, &u, 4);
    }
    std::vector<uint16_t> img((size_t)((n + 31) / 32) * lr::WideBf16Geom<64>::BUF1);
    if (!lr::wide_f16_prepare_rne<64>(rows.data(), n, img.data())) {
        std::printf("0\n");
        return 0;
    }
    unsigned long long sum = 0;
    for (uint16_t w : img) sum += w;
    std::printf("1 %llu\n", sum);
    return 0;
}

int main(int argc, char** argv) {
    if (argc > 1) return f16_modes(argv[1]);
    int dtype, p, kind, prec, group, mode, cus;
    long long n, chains;
    while (std::scanf("%d %d %lld %lld %d %d %d %d %d", &dtype, &p, &n, &chains, &kind, &prec, &group, &mode, &cus) == 9) {
        lr_model m;
        m.dtype = dtype;
        m.n = n;
        m.p = p;
        m.P = padded_width(p);
        m.cus = cus;
        m.table = find_table(dtype, m.P);
        if (!m.table) {
            std::printf("ERR %d no kernels for dtype=%d padded p=%d\n", LR_ERR_UNSUPPORTED, dtype, m.P);
            continue;
        }
        // the images lr_model_create would have built for this shape (same rule: lr_model.h model_images / lr_plan.h model_wants_xms)
        void* const yes = reinterpret_cast<void*>(1);
        const ModelImages im = model_images(n, m.P, dtype);
        m.d_xmx = im.tall_mx ? yes : nullptr;
        m.d_xmf = im.mf_end ? yes : nullptr;
        m.d_xms = im.mf_end && model_wants_xms(&m) && m.table->mfma_image_bytes ? yes : nullptr;
        m.d_xblk = im.wide ? yes : nullptr;
        m.d_xblk1 = im.wide1 ? yes : nullptr;
        Plan pl{};
        int rc = check_group_for(&m, group, mode);
        if (rc == LR_OK) rc = make_plan(&m, chains, group, mode, &pl, false, kind == LR_KIND_HMC && prec != LR_PREC_FULL, kind, prec == LR_PREC_AUTO);
        if (rc != LR_OK) std::printf("ERR %d %s\n", rc, g_err);
        else {
            const InteriorPlan ip = plan_interior(&m, chains);
            std::printf("%d %d %d %zu %d %lld %d %lld %d %d %d\n", pl.mode, pl.G, pl.R, pl.lds_bytes, ip.RS_i, (long long)ip.slice_len_i, ip.waves,
                        (long long)pl.split, pl.G2, pl.R2, pl.mode2);
        }
    }
    return 0;
}
