// lr_plan.h -- which kernel variant runs a request: the planner of the C ABI (lr_plan, lr_plan_run, every lr_run_*).
// Part of lr_api.hip's translation unit (included behind `struct lr_model`); host code only.
//
// The measured crossovers live in DATA: kMfmaRules (when HMC with reduced-precision interior steps moves to the fused
// matrix-core chain kernel, by padded width, row count and chains per CU) and kPlanConst (the stepwise / scalar-row /
// wide-engine thresholds), each row with the measurement it came from (profiles/).  make_plan() applies them and checks what
// a table cannot know: which variants this build instantiates and whether a variant's operands fit its store (registers,
// LDS, the device-memory images the model carries).  tests/test_gpu_planner.py times AUTO against every forced alternative
// at shapes on both sides of the table's boundaries and fails when AUTO is more than 10 % off the best.
#pragma once

namespace {

struct Plan {
    int mode, G, R;
    size_t lds_bytes;
    // two-part plans (register family): chains [0, split) of the planned run on (G, R), [split, n) on (G2, R2); split = 0: one part
    int64_t split = 0;
    int mode2 = 0, G2 = 0, R2 = 0;
    bool corun = true;  // the remainder's launch beside the head's (false: after it)
};

// ---------------------------------------------------------------------------------------------------------------------------
// thresholds outside the matrix-core table
struct PlanConst {
    // tall data: LDS-resident rows lose to the stepwise engine once they take more than this many bytes (one workgroup per CU)
    // and there are this many chains (tools/midn_sweep.py, n = 4000 p = 8: 0.8e12 vs 1.1-1.8e12 chain-rows/s)
    // (round 3, tools/planner_bench.py, lds 64 lanes per chain | stepwise, chain-iterations/s: MALA n=4000 p=8: 2.32e8 | 1.48e8 at 2048
    //  chains, 2.33 | 2.43 at 4096, 2.33 | 3.23 at 8192; HMC all-fp32 n=4000: 1.64e7 | 1.00e7 at 2048, 1.64 | 1.54 at 4096: from 4096 chains)
    size_t lds_rows_prefer_stepwise_bytes = 64 * 1024;
    int64_t lds_rows_prefer_stepwise_chains = 4096;
    // register-resident rows, lanes per chain G: time of a launch ~ (R + reg_fixed_rows) f(w), R = rows per lane of the variant,
    // w = waves per SIMD (rounded up: the fullest SIMD sets the time), f(w) = 1 + reg_corun (w - 1).  Fitted to HMC n=200 p=8
    // (tools/planner_bench.py): exactly-filled launches give 8.88 / 6.69 / 5.2 us per iteration for G = 16 / 32 / 64 (R = 13 / 7 /
    // 4): fixed work (generator, reductions, update) worth 11 rows; two co-resident waves take 1.88x one.  Replaces "the
    // smallest G that still gives every SIMD a wave", which put 2560 chains on G = 32 (2.16e8 it/s) when G = 16 runs 2.88e8,
    // and MALA at 3072 chains on G = 32 (2.72e9) against 3.30e9.
    double reg_fixed_rows = 11.0;
    double reg_corun = 0.88;
    // All-fp32 families (MALA / RWMH / UL, HMC with precision FULL) whose register variant is ONE chain per wave (64 lanes per chain:
    // 64 copies of the state, a 6-level reduction per evaluation) -- round 4, tools/planner_bench.py, profiles/r4_planner_bench_many_chains.txt:
    //  * 9 <= p <= 32, rows within the register tiles of the fp32 matrix-core kernel (n <= 1024 at p <= 16, 512 beyond): from 16
    //    chains per CU the fp32-MFMA kernel (S = 4) beats every vector-ALU variant (MALA, 4096 chains, AUTO before | mfma, it/s:
    //    n=500 p=16 reg 64x8 0.88 | 1.43e9, n=200 p=12 reg 32x7 1.87 | 2.08, n=250 p=16 HMC all-fp32 1.47 | 1.91e8, n=200 p=24
    //    reg 64x4 0.47 | 1.27e9, n=400 p=30 lds 64 0.22 | 0.86e9; at 2048 chains n=200 p=12 1.85 | 1.04: not below).  At p <= 8
    //    it never does (n=800: 1.43 | 0.99).
    //  * from 64 chains per CU with rows within 28 KB: LDS rows with 8 lanes per chain (8 chains per wave) overtake one chain per
    //    wave (MALA, 16 384 chains, reg | lds 8: n=600 p=8 1.64 | 1.89e9, n=800 p=8 1.43 | 1.47 (1.44 | 1.64 at 65 536), n=300 p=12
    //    0.89 | 2.25, n=400 p=16 0.89 | 1.82; not at 32 KB: n=1000 p=8 1.42 | 1.19, n=500 p=16 mfma 1.69 | lds 8 1.52; not at
    //    p > 16: n=200 p=24, 4096 chains, lds 8 0.41 | mfma 1.27e9)
    //  * p <= 8 at 16 384 / 65 536 chains (several waves per SIMD), HMC all-fp32, reg 16x13 | mfma S=1: 5.44 | 3.72e8, 5.63 | 3.74e8 -- the
    //    fp32-input MFMAs alone bound that kernel (profiles/r4_planner_bench_full_many.txt): never for p <= 8
    // float64, p <= 8, rows that fit the register variants (n <= 256): 32 lanes x 7 rows pads 200 rows to 224 and pays a 5-level
    // f64 reduction (30 % of its leapfrog loop); rows in LDS with 16 lanes per chain from one wave per SIMD, 8 lanes per chain
    // (8 x 25 = 200 exactly) from two (round 4, HMC L=20 n=200, it/s, reg 32x7 | lds 16 | lds 8: 2048 chains 7.6 | 4.4 | 2.8e7,
    // 4096: 7.7 | 8.7 | 5.6, 8192: 7.7 | 8.7 | 11.2, 16 384: 7.7 | 8.8 | 11.3; MALA 8192: 5.3 | 7.2 | 7.2e8)
    // float64 HMC under LR_PREC_AUTO, p <= 8, n <= 256: k_chain_mixed from this many chains per CU
    int mixed_chains_per_cu = 0;
    int f64_mfma_chains_per_cu = 33;  // float64 HMC under LR_PREC_AUTO: k_chain_mfma_f64 beyond two rounds of k_chain_mixed (plan_mfma_hmc)
    int f64_lds16_chains_per_cu = 16;
    // ... and HMC from 256 chains per CU (one wave per SIMD of lane-per-chain work): rows in LDS, ONE lane per chain -- no reduction, no
    // gather at all (round 6, tools/f64_many_chains.py, every evaluation float64, lds 8 | lds 1, it/s: 32 768 chains 6.17 | 3.3e7, 65 536:
    // 6.30 | 6.44, 131 072: 6.36 | 6.82, 262 144: 6.37 | 6.92e7 = 0.33 of the fp64 vector peak -- the most ANY engine reaches at ANY chain count:
    // 43 instructions per (row, chain), 27 of them the sigmoid)
    int f64_lds1_chains_per_cu = 256;
    int f64_lds8_chains_per_cu = 32;
    int mfma_fp32_chains_per_cu = 16;
    int lds8_chains_per_cu = 64;
    size_t lds8_max_row_bytes = 28 * 1024;
    // a run is planned in two parts (exactly-filled head + remainder on its own variant) when the model prices the pair below
    // this fraction of the single launch (two launches, two sets of prologue row loads: not for a couple of per cent; the model's
    // sequential pricing is within a point or two of the measured gains: profiles/r4_chain_grid.txt)
    double split_gain = 0.97;
    // lane-per-chain with rows from the scalar unit: from this many waves per SIMD (HMC n=200 p=8: 2.20 / 2.46 / 2.64e8 it/s at
    // 2 / 4 / 8 waves per SIMD against 2.21e8 for rows in registers), rows within the 16 KB scalar cache
    int scalar_rows_waves_per_simd = 3;
    size_t scalar_rows_max_bytes = 16 * 1024;
    // wide models: the 8-wave (128-chain) exact kernel from this many chains, while its slices still hold this many rows
    // (tools/wide_nsweep.py: 4096 chains, n = 2048 35.6 vs 36.6 us per step for 4 vs 8 waves, n = 8192 102 vs 85; n = 512 18.9 vs 25.9)
    int64_t wide_8wave_chains = 4096;
    int64_t wide_8wave_min_slice_rows = 512;
};
constexpr PlanConst kPlanConst{};

// wide models, exact-split and chain-split bf16 kernels: 1 = 4 waves (64 chains) per workgroup, 2 = 8 waves (128 chains): the
// 128-chain workgroup halves the staging per chain but needs twice the row slices to fill the chip.  Float64 models: 0 (their one
// kernel, lr_wide_f64.h, takes 64 chains per workgroup).
int wide_engine(const lr_model* m, int64_t C) {
    if (m->dtype != LR_F32) return 0;
    if (m->dbg.wide_waves) return m->dbg.wide_waves == 8 ? 2 : 1;
    if (C < kPlanConst.wide_8wave_chains) return 1;
    const int64_t blocks2 = (C + 127) / 128, rs2 = (m->cus + blocks2 - 1) / blocks2;
    return m->n / rs2 >= kPlanConst.wide_8wave_min_slice_rows ? 2 : 1;
}
int64_t wide_chains_per_block(const lr_model* m, int64_t C) { return wide_engine(m, C) == 2 ? 128 : 64; }

// matrix-core chain kernel with its bf16 operands in LDS (lr_mfma.h MfmaRowsLds): bytes for a row split over S waves
// (the kernel's own layout function, so the two cannot drift apart: an earlier hand-written copy missed the even padding of
//  the eta images and under-allocated by 512 p/8 bytes per wave for odd tile counts)
template <int P> size_t mfma_lds_bytes_p(int64_t ntw, int S) {
    switch (S) {
    case 1: return 1 * lr::MfmaRowsLds<P, 1, false>::bytes_per_wave(ntw);
    case 4: return 4 * lr::MfmaRowsLds<P, 4, false>::bytes_per_wave(ntw);
    default: return 8 * lr::MfmaRowsLds<P, 8, false>::bytes_per_wave(ntw);
    }
}
size_t mfma_lds_bytes(const lr_model* m, int S) {
    const int64_t tiles = (m->n + 15) / 16, ntw = (tiles + S - 1) / S;
    return m->P == 8 ? mfma_lds_bytes_p<8>(ntw, S) : (m->P == 16 ? mfma_lds_bytes_p<16>(ntw, S) : mfma_lds_bytes_p<32>(ntw, S));
}
// 160 KB less the kernel's static exchange buffers (red: 2 x S x 64 x P/4 floats, redv: S x 64 doubles)
size_t mfma_lds_budget(const lr_model* m, int S = 4) { return 160 * 1024 - (size_t)128 * S * m->P - (size_t)512 * S; }
// beyond the LDS variant of the matrix-core chain kernel: the model also carries the bf16 operand images in device memory (d_xms)
bool model_wants_xms(const lr_model* m) { return model_images(m->n, m->P, m->dtype).mf_end && mfma_lds_bytes(m, 4) > mfma_lds_budget(m); }

// ---------------------------------------------------------------------------------------------------------------------------
// HMC whose interior leapfrog gradients may use the bf16 matrix pipe (LR_PREC_AUTO / LR_PREC_BF16), float32, padded p = 8 / 16 /
// 32: when the fused matrix-core chain kernel (lr_mfma.h, LR_MODE_MFMA) takes over from the vector-ALU kernels and the stepwise
// engine.  Rows are tried top to bottom; the first whose (row-split ways S, operand store) variant exists and whose operands fit
// wins.  cpc = chains per CU (chains / CUs: the thresholds scale with the chip).  Operand stores: registers (n <= 16 S R for
// the instantiated tile counts R), LDS (lr_mfma.h MfmaRowsLds within mfma_lds_budget), device memory (images built at model
// creation for 208 / 1024 / 512 < n <= 8192).  "S = 8 where it exists" = the 8-wave row split of the LDS / device variants.
enum Store { ST_REG, ST_LDS, ST_DEV };
struct MfmaRule {
    int P;                    // padded width
    Store store;
    int S;                    // row-split ways requested (LDS / device rows: upgraded to 8 where `s8` says so)
    int64_t n_lo, n_hi;       // n_lo < n <= n_hi
    int cpc_lo, cpc_hi;       // cpc_lo <= chains per CU < cpc_hi (0 = no upper bound)
    bool s8;                  // take the 8-wave split when that variant exists and fits
    const char* why;
};
constexpr int64_t kAnyN = INT64_MAX;
constexpr MfmaRule kMfmaRules[] = {
    // ---- p <= 8.  Bench workload (n = 200), chain-iterations/s, reg 16x13 | mfma S=4 | mfma S=1 (profiles/r2_mfma_chain_grid.txt):
    //      4096: 1.88e8 | 2.13e8 | 1.16e8    8192: 2.10e8 | 2.93e8 | 2.32e8    10240: 1.86e8 | 2.61e8 | 2.89e8    16384: 2.25e8 | 3.23e8 | 4.49e8
    {8, ST_REG, 1, 0, 208, 40, 0, false, "16 chains per wave, no LDS hand-off: from 40 chains per CU (10 240 chains: 2.89e8 vs 2.61e8 for S=4)"},
    {8, ST_LDS, 1, 208, kAnyN, 96, 0, false, "one LDS image shared by the workgroup's four chain tiles: n=300 2.24 | 2.30e8 at 16 384 chains, 2.33 | 2.68e8 at 32 768 (S=4 | this)"},
    {8, ST_REG, 4, 0, 1024, 16, 0, false, "rows in registers, split over 4 waves: from one workgroup per CU (4096 chains: 2.13e8 vs 1.88e8 reg 16x13; n=400: 2.82e8 vs 1.83e8); below, the register kernels win (n=400, 2048 chains: 1.45e8 vs 1.69e8)"},
    {8, ST_LDS, 4, 1024, kAnyN, 8, 0, true, "operands in LDS, 8-wave split: n=2000 HMC L=50 lane-group LDS kernel 56 TF | this 70 at 2048 chains (43 | 35 at 1024); 121 -> 139 TF over the 4-wave split"},
    {8, ST_LDS, 4, 1024, kAnyN, 16, 0, false, "operands in LDS, 4-wave split where the 8-wave one does not fit: n=2000 48 | 90 TF at 4096 chains (best other | this)"},
    {8, ST_DEV, 4, 1024, 6000, 8, 64, true, "operands streamed from device memory, 8-wave split below 64 chains per CU (n=3000 109 -> 120 TF at 4096 chains); from 8 chains per CU up to n = 6000 (round 3, 2048 chains, it/s, lds 64 | stepwise | this: n=3000 2.12 | 1.16 | 2.42e7, n=5000 1.35 | 1.08 | 1.57e7, n=6000 - | 1.33 | 1.33e7; at 1024 chains the lane-group LDS kernel wins, 2.11 | 0.65 | 1.21e7)"},
    {8, ST_DEV, 4, 6000, 8192, 16, 64, true, "... beyond n = 6000 from 16 chains per CU: n=8000 at 2048 chains stepwise 1.27e7 | this 1.04e7, at 4096 chains 118 -> 137 TF over the 4-wave split"},
    {8, ST_DEV, 4, 1024, 8192, 64, 0, false, "... 4-wave split from 64 chains per CU (16 384 chains: 135 vs 121 TF); n=8000: stepwise 130 | this 136"},
    // ---- 8 < p <= 16 (profiles/r2_midp_mfma.txt, HMC L=20, algorithmic TF, vector-ALU kernel | this): n=200 p=12: 13 | 19 at 1024 chains,
    //      33 | 72 at 4096, 34 | 138 (S=1) at 16 384; n=1000 p=12: 23 | 39, 23 | 150, 50 | 147
    {16, ST_REG, 1, 0, 208, 40, 0, false, "n=200 p=12: 98 (S=4) -> 138 TF at 16 384 chains"},
    {16, ST_REG, 4, 0, 256, 4, 0, false, "p > 8 moves to the matrix pipe from 4 chains per CU: n=200 p=12..16 +7..+40 % at 1024 chains (tools/planner_check.py)"},
    {16, ST_REG, 4, 256, 512, 8, 0, false, "the alternative is still a register-resident vector kernel (reg 64x8): n=500 p=16 at 1024 chains 3.17e7 | 2.87e7, at 2048 1.8x"},
    {16, ST_REG, 4, 512, 1024, 4, 0, false, "n=1000 p=12: 23 | 39 TF at 1024 chains"},
    {16, ST_LDS, 4, 1024, kAnyN, 16, 0, true, "operands in LDS: n=1150 p=16 30 | 32 TF at 1024 chains, 33 | 120 at 4096, 81 | 129 at 16 384: from one workgroup per CU"},
    {16, ST_DEV, 4, 1024, 4000, 4, 0, false, "device-memory operands: n=3000 p=16 stepwise | this 5.5 | 8.1e6 it/s at 1024 chains, 0.98 | 1.62e7 at 2048 (round 3), 66 | 121 TF at 4096, 130 | 146 at 16 384"},
    {16, ST_DEV, 4, 4000, 8192, 16, 64, false, "... beyond n = 4000 from 16 and below 64 chains per CU: n=8000 p=16 112 | 148 TF at 4096 chains but 187 | 177 at 16 384, and stepwise 4.9 | 3.3e6 it/s at 1024 chains, 9.6 | 6.7e6 at 2048; n=5000: 5.1 | 5.1e6 at 1024"},
    // ---- 16 < p <= 32 (8 tiles per wave at most: n <= 512 in registers; no LDS variant -- the LDS holds no more rows than the registers)
    {32, ST_REG, 4, 0, 512, 4, 0, false, "n=200 p=32: 25 | 29 TF at 1024 chains, 26 | 110 at 4096; n=500 p=32 (LDS kernel otherwise): 16 | 53, 16 | 199"},
    {32, ST_DEV, 4, 512, 2000, 4, 0, false, "n=700 p=30 at 1024 chains 16 -> 19 TF, n=2000 p=20 16 -> 20; at 4096 chains 31 | 74, 46 | 74"},
    {32, ST_DEV, 4, 2000, 2048, 4, 64, false, "(the 64-chains-per-CU row limit of 2000 below)"},
    {32, ST_DEV, 4, 2048, 8192, 16, 64, false, "n=5000 p=30: 105 | 132 TF at 4096 chains, 176 | 135 at 16 384: below 64 chains per CU only"},
};

// does the (S, store) variant exist in this build and do the model's operands fit it?  Fills the plan.
bool mfma_variant_fits(const lr_model* m, int S, Store st, Plan* out) {
    const lr::InstTable* t = m->table;
    for (int i = 0; i < t->nvariants; ++i) {
        const lr::Variant& v = t->variants[i];
        if (v.mode != lr::MODE_MFMA || v.G != S) continue;
        if (st == ST_REG && v.R > 0 && (int64_t)16 * v.G * v.R >= m->n) {  // (ascending R per S: the smallest that holds the rows)
            *out = Plan{v.mode, v.G, v.R, 0};
            return true;
        }
        // (S = 1 shares one image among the four chain tiles of a 4-wave workgroup: its statics are those of the 4-wave split)
        if (st == ST_LDS && v.R == 0 && mfma_lds_bytes(m, S) <= mfma_lds_budget(m, S == 1 ? 4 : S)) {
            *out = Plan{v.mode, v.G, 0, mfma_lds_bytes(m, S)};
            return true;
        }
        if (st == ST_DEV && v.R < 0 && m->d_xms != nullptr) {
            *out = Plan{v.mode, v.G, v.R, 0};
            return true;
        }
    }
    return false;
}

bool plan_mfma_hmc(const lr_model* m, int64_t C, Plan* out) {
    if (m->P < 8 || m->P > 32) return false;
    // float64 models: k_chain_mfma_f64 (lr_mfma_f64.h) -- p = 8, rows as register operands, 16 chains per wave: the table's first row
    // (its other variants are not instantiated for float64, so the rows that ask for them fit nothing); float64 rows in LDS
    if (m->dtype != LR_F32 && m->P != 8) return false;
    for (const MfmaRule& r : kMfmaRules) {
        if (r.P != m->P || m->n <= r.n_lo || m->n > r.n_hi) continue;
        // (float64: the alternative is the float32-interior kernel in rounds of 16 chains per CU -- beyond two rounds the one launch
        //  of k_chain_mfma_f64 wins: 9216 chains 1.30 ms in two parts | 0.96 ms; at 8192 0.943 | 0.96)
        const int cpc_lo = m->dtype != LR_F32 ? kPlanConst.f64_mfma_chains_per_cu : r.cpc_lo;
        if (C < (int64_t)cpc_lo * m->cus || (r.cpc_hi && C >= (int64_t)r.cpc_hi * m->cus)) continue;
        bool hit = r.s8 && mfma_variant_fits(m, 8, r.store, out);
        if (!hit && r.s8 && r.store == ST_LDS && r.cpc_lo < 16 && m->P == 8) continue;  // (the 8-chains-per-CU row is the 8-wave split's own)
        if (!hit) hit = mfma_variant_fits(m, r.S, r.store, out);
        if (!hit) continue;
        if (m->dtype != LR_F32) out->lds_bytes = (size_t)m->n * m->P * m->esize();
        return true;
    }
    return false;
}

// HMC on a float64 model whose interior leapfrog gradients may be cheaper (LR_PREC_AUTO / LR_PREC_BF16), p <= 16, rows within the
// instantiated (lanes per chain x rows per lane) shapes and within the LDS: k_chain_mixed / k_chain_mixed_rep (LR_MODE_MIXED) --
// float64 end points, Metropolis test, position and momentum; float32 force inside the trajectory.
double reg_cost(const lr_model* m, const lr::Variant& u, int64_t chains);
// dynamic LDS of k_chain_mixed: the float64 rows + its per-lane stash (lr_kernels.h)
// rows staged in LDS by the lane-group kernels: float64 rows are padded by two doubles (lr_device.h kLdsRowPad)
size_t lds_rows_bytes(const lr_model* m) { return (size_t)m->n * (m->P + (m->dtype == LR_F32 ? 0 : 2)) * m->esize(); }
size_t mixed_lds_bytes(const lr_model* m) { return lds_rows_bytes(m) + (size_t)lr::kMixedStashDoubles * 8 * 256; }
bool plan_mixed_hmc(const lr_model* m, int64_t C, Plan* out, int* whole = nullptr) {
    // (MIXED variants are instantiated for padded p <= 16 only -- lr_inst.hip LR_MIXED_VARIANTS: at p = 32 the loop below finds none;
    //  mixed_chains_per_cu = 0: from the first chain, i.e. the float64 model's DEFAULT policy is the float32 force at any chain count)
    if (m->dtype != LR_F64 || m->P > 32) return false;
    const size_t row_bytes = mixed_lds_bytes(m);
    if (row_bytes > kLdsBudget || C < (int64_t)kPlanConst.mixed_chains_per_cu * m->cus) return false;
    const lr::InstTable* t = m->table;
    int bi = -1;
    double cost = 0;
    for (int i = 0; i < t->nvariants; ++i) {  // lanes per chain by the register family's launch-time model (few chains: wide groups)
        const lr::Variant& v = t->variants[i];
        if (v.mode != lr::MODE_MIXED || (int64_t)v.G * v.R < m->n) continue;
        const double c = reg_cost(m, v, C) + 1e-3 * v.R;
        if (bi < 0 || c < cost) { bi = i; cost = c; }
    }
    if (bi < 0) return false;
    *out = Plan{t->variants[bi].mode, t->variants[bi].G, t->variants[bi].R, row_bytes};
    if (whole) *whole = bi;
    return true;
}

// ---------------------------------------------------------------------------------------------------------------------------
// make_plan and its steps.  Order: (1) HMC with reduced-precision interior steps -> kMfmaRules, then (float64, p = 8) k_chain_mixed;
// (2) wide models -> stepwise;
// (3) rows off chip (or forced) -> stepwise; (4) the best variant by residency tier and the register family's launch-time model;
// (5) measured overrides of vector-ALU plans; (6) a second part for the remainder between exactly-filled chain counts.
struct PlanReq {
    const lr_model* m;
    int64_t C;       // chains to plan for
    int group, mode; // the caller's request (0 / LR_MODE_AUTO: the planner's choice)
    bool for_eval;   // lr_eval: no matrix-core / stepwise / two-part plans
    int kind;        // LR_KIND_* of the run, -1 = not a run of a known family (lr_plan, lr_eval)
    bool automatic() const { return mode == LR_MODE_AUTO && group == 0 && !for_eval; }
};

// (2) wide models (32 < p <= 128): only the stepwise engine exists; its partial kernels are MFMA GEMMs over blocks of 64 / 128
// chains x row slices (lr_wide_bf16.h; float64 models: lr_wide_f64.h).  One workgroup per CU -- the fewest row slices -- measured
// fastest for the bf16 kernels (48-64 KB of LDS: 2-3 per CU would fit; 8192 chains: 188 / 170 / 155 TFLOP/s at 1 / 2 / 3 per CU).
int plan_wide(const PlanReq& q, Plan* out) {
    const lr_model* m = q.m;
    if (q.mode != LR_MODE_AUTO && q.mode != LR_MODE_STEPWISE)
        return fail(LR_ERR_UNSUPPORTED, "p=%d > 32 runs on the stepwise engine only (mode=%d requested)", m->p, q.mode);
    const int64_t cpb = wide_chains_per_block(m, q.C);
    const int64_t blocks = (q.C + cpb - 1) / cpb;
    int64_t RS = (m->cus + blocks - 1) / blocks;
    // (the float64 kernel, 64 KB of LDS: from 64 chain blocks two workgroups per CU -- 4096 chains, config 5's design, us per evaluation
    //  by slice count 4 | 8 | 16: 207 | 182 | 198; at 1024 chains 16 | 32 slices: 64.6 | 65.1 -- tools/f64_wide_slices_probe.py)
    if (m->dtype == LR_F64 && blocks >= 64) RS = (2 * m->cus + blocks - 1) / blocks;
    if (q.group > 0) RS = q.group;  // explicit slice count: pins the summation order whatever the chain count
    int64_t slice_len = (m->n + RS - 1) / RS;
    slice_len = (slice_len + 31) / 32 * 32;  // whole 32-row blocks (the bf16 kernel's K = 32)
    RS = (m->n + slice_len - 1) / slice_len;
    *out = Plan{lr::MODE_STEPWISE, (int)RS, (int)slice_len, 0};
    return LR_OK;
}

// (3) tall data: neither VGPRs nor LDS can hold the rows -> stepwise engine (lr_tall.h): the rows split into RS slices so that every
// evaluation occupies the whole chip with ~4 waves per SIMD.  Measured (tools/midn_sweep.py, HMC, p = 8, chain-rows/s): rows
// streamed from L2 by every group never beat the stepwise engine (n = 6000-8000: 0.4-1.0e12 vs 0.6-2.0e12).
void plan_tall(const PlanReq& q, Plan* out) {
    const lr_model* m = q.m;
    const int64_t want_waves = 4LL * m->cus;
    // a workgroup = NW waves x 64 chains working on one slice (NW as lr::TallGeom: LDS-limited)
    const int raw = 2048 / (m->P * (int)m->esize());
    const int64_t NW = (raw >= 16 && m->P <= 16) ? 16 : (raw >= 8 ? 8 : 4);
    const int64_t waves_per_slice = NW * ((q.C + 63) / 64);
    int64_t RS = (4 * want_waves + waves_per_slice - 1) / waves_per_slice;
    if (q.mode == LR_MODE_STEPWISE && q.group > 0) RS = q.group;  // explicit slice count (see plan_wide)
    int64_t slice_len = (m->n + RS - 1) / RS;
    if (slice_len < 16 * NW) slice_len = 16 * NW;  // at least 16 rows per wave
    slice_len = (slice_len + 1) & ~(int64_t)1;      // even: the float32 kernel walks row pairs
    if (m->d_xmx) slice_len = (slice_len + 31) / 32 * 32;  // whole tile pairs: the matrix-pipe interior kernel
    RS = (m->n + slice_len - 1) / slice_len;
    *out = Plan{lr::MODE_STEPWISE, (int)RS, (int)slice_len, 0};
}

// the register family's launch-time model (kPlanConst): predicted time of `chains` chains on variant u, in rows
double reg_cost(const lr_model* m, const lr::Variant& u, int64_t chains) {
    const int64_t want_waves = 4LL * m->cus, waves = (chains * u.G + 63) / 64, wps = (waves + want_waves - 1) / want_waves;
    return ((double)u.R + kPlanConst.reg_fixed_rows) * (1.0 + kPlanConst.reg_corun * (double)(wps - 1));
}

// (4) index of the best instantiated variant for the request, -1 if none: residency tier first (REG > LDS > GLOBAL), then group
// fitness (the register family by its launch-time model), then fewer padded rows; matrix-core variants only on request
int pick_variant(const PlanReq& q) {
    const lr_model* m = q.m;
    const lr::InstTable* t = m->table;
    const int64_t want_waves = 4LL * m->cus;
    const size_t row_bytes = (size_t)m->n * m->P * m->esize();
    int best = -1;
    long best_score = -1;
    for (int i = 0; i < t->nvariants; ++i) {
        const lr::Variant& v = t->variants[i];
        if (v.mode == lr::MODE_MFMA) {
            // fp32 matrix-core variants; G = row-split ways S, R = tiles per wave.  Here only on request (mode = LR_MODE_MFMA); the
            // planner's own uses are plan_mfma_hmc and measured_overrides.
            if (q.for_eval || q.mode != LR_MODE_MFMA) continue;
            if (m->dtype != LR_F32 && q.kind != LR_KIND_HMC) continue;  // (float64: k_chain_mfma_f64 is an HMC kernel)
            if (v.R < 0 ? m->d_xms == nullptr
                        : (v.R == 0 ? mfma_lds_bytes(m, v.G) > mfma_lds_budget(m, v.G) : (int64_t)16 * v.G * v.R < m->n)) continue;
            if (q.group != 0 && v.G != q.group) continue;
            const bool filled = q.C >= 16LL * want_waves;
            // operands in LDS / device memory: only when no register variant fits
            const long score = v.R <= 0 ? 0 : ((filled ? (v.G == 1) : (v.G == 4)) ? 2 : 1);
            if (score > best_score) { best_score = score; best = i; }
            continue;
        }
        if (v.mode == lr::MODE_MIXED) {
            // float64 HMC with float32 interior gradients (k_chain_mixed): here only on request; the planner's own use is plan_mixed_hmc
            if (q.for_eval || q.mode != LR_MODE_MIXED || q.kind != LR_KIND_HMC || (q.group != 0 && v.G != q.group)) continue;
            if ((int64_t)v.G * v.R < m->n || mixed_lds_bytes(m) > kLdsBudget) continue;
            const long score = 1000 - v.R;  // the fewest padded rows
            if (score > best_score) { best_score = score; best = i; }
            continue;
        }
        if (q.mode != LR_MODE_AUTO && v.mode != q.mode) continue;
        if (q.group != 0 && v.G != q.group) continue;
        // float64 at 17 <= p <= 32: chains run on the distributed-state kernel, whose state lives on the 16 lanes of a DPP row
        // (lr_kernels.h k_chain_dist); narrower lane groups serve lr_eval only
        if (!q.for_eval && m->dtype == LR_F64 && m->P == 32 && v.G < 16) continue;
        if (v.mode == lr::MODE_REG && (int64_t)v.G * v.R < m->n) continue;
        if (v.mode == lr::MODE_LDS && lds_rows_bytes(m) > kLdsBudget) continue;
        const int64_t waves = (q.C * v.G + 63) / 64;
        long score = (2 - v.mode) * 1000000L;
        if (waves >= want_waves) score += 100000L - 1000L * v.G;  // filled: prefer small groups
        else score += 10L * v.G;                                   // not filled: prefer large groups
        if (v.mode == lr::MODE_REG && q.group == 0) {
            // the lowest predicted time wins (ties: fewer padded rows); mapped into the tier's width (0, 900000], so that at very large
            // chain counts -- many waves per SIMD, large modelled cost -- a register variant still scores above the LDS tier
            score = 2000000L + (long)(900000.0 / (1.0 + reg_cost(m, v, q.C) / 64.0)) - v.R;
        } else if (v.mode == lr::MODE_REG) {
            score -= v.R;  // exact-fit R before padded R
        }
        // lane-per-chain with rows broadcast from the scalar unit has no replicated work and no reductions
        if (q.mode == LR_MODE_AUTO && q.group == 0 && v.mode == lr::MODE_GLOBAL && v.G == 1 &&
            waves >= kPlanConst.scalar_rows_waves_per_simd * want_waves && row_bytes <= kPlanConst.scalar_rows_max_bytes)
            score = 4000000L;
        if (score > best_score) { best_score = score; best = i; }
    }
    return best;
}

// (5) vector-ALU plans the measurements overrule (kPlanConst).  Returns true when *out is final (a matrix-core plan); else `best`
// may have been moved to another variant of the table.
bool measured_overrides(const PlanReq& q, int* best, Plan* out) {
    const lr_model* m = q.m;
    const lr::InstTable* t = m->table;
    const lr::Variant& b = t->variants[*best];
    const size_t row_bytes = (size_t)m->n * m->P * m->esize();
    auto move_to_lds = [&](int g) {
        for (int i = 0; i < t->nvariants; ++i)
            if (t->variants[i].mode == lr::MODE_LDS && t->variants[i].G == g) *best = i;
    };
    if (m->dtype == LR_F32 && b.mode != lr::MODE_GLOBAL) {
        // all-fp32 families (HMC under the default precision policy never gets here with enough chains: plan_mfma_hmc)
        const bool one_chain_per_wave = b.mode == lr::MODE_REG && b.G == 64;
        if (m->P <= 16 && one_chain_per_wave && q.C >= (int64_t)kPlanConst.lds8_chains_per_cu * m->cus && row_bytes <= kPlanConst.lds8_max_row_bytes)
            move_to_lds(8);
        else if (m->P >= 16 && q.C >= (int64_t)kPlanConst.mfma_fp32_chains_per_cu * m->cus && mfma_variant_fits(m, 4, ST_REG, out))
            return true;
    }
    // (... also where the scalar-row lane-per-chain variant would have been picked: 262 144 chains global 1 | lds 1: 6.43 | 6.92e7)
    if (m->dtype == LR_F64 && m->P == 8 && q.kind == LR_KIND_HMC && q.C >= (int64_t)kPlanConst.f64_lds1_chains_per_cu * m->cus &&
        (b.mode == lr::MODE_REG || (b.mode == lr::MODE_GLOBAL && b.G == 1)) && lds_rows_bytes(m) <= kLdsBudget) {
        move_to_lds(1);
    } else if (m->dtype == LR_F64 && m->P == 8 && b.mode == lr::MODE_REG) {
        if (q.C >= (int64_t)kPlanConst.f64_lds8_chains_per_cu * m->cus) move_to_lds(8);
        else if (q.C >= (int64_t)kPlanConst.f64_lds16_chains_per_cu * m->cus) move_to_lds(16);
    }
    // float64 at the other widths (LDS rows on 1 / 8 / 64 lanes per chain only): one chain per wave saturates at one wave per SIMD;
    // 8 lanes per chain overtake from 16 chains per CU (round 4, PLANNER_BENCH_DTYPE=float64 tools/planner_bench.py, lds 64 | lds 8,
    // it/s: MALA n=200 p=12: 2048 chains 2.61 | 2.58e8, 4096: 2.62 | 5.12e8; HMC all-float64 4096: 2.12 | 4.14e7; MALA p=3 4096: 6.1 | 8.5e8)
    if (m->dtype == LR_F64 && m->P != 8 && b.mode == lr::MODE_LDS && b.G == 64 && q.C >= (int64_t)kPlanConst.f64_lds16_chains_per_cu * m->cus)
        move_to_lds(m->P == 32 && !q.for_eval ? 16 : 8);  // (p > 16: the chain kernel needs 16 lanes per chain)
    return false;
}

// (6) Wave quantisation of the register family.  Between the chain counts that fill the chip exactly a launch takes the time of its
// fullest SIMD (5120 chains on 16 lanes per chain: 1.43e8 it/s where 4096 run 2.13e8, profiles/r3_chain_grid.txt).  A run left to the
// planner is then planned in two parts: the largest exactly-filled head on the variant the model prefers for that count, the
// remainder on whatever variant the model prefers for IT (usually a wider group that finishes in one short wave), when the model
// prices the two launches kPlanConst.split_gain below the single one.  *out holds the one-part plan on entry.
// `family`: MODE_REG, or MODE_MIXED (float64 models under LR_PREC_AUTO: the same quantisation, the same model; its parts run in turn --
// the head's waves hold 364 of a SIMD's 512 registers, nothing fits beside them)
void plan_second_part(const PlanReq& q, const lr::Variant& whole, Plan* out, int family = lr::MODE_REG) {
    const lr_model* m = q.m;
    const lr::InstTable* t = m->table;
    const int64_t want_waves = 4LL * m->cus;
    auto fits = [&](const lr::Variant& u) { return u.mode == family && (int64_t)u.G * u.R >= m->n; };
    auto best_reg = [&](int64_t chains, double* cost) {
        int bi = -1;
        for (int i = 0; i < t->nvariants; ++i) {
            if (!fits(t->variants[i])) continue;
            const double c = reg_cost(m, t->variants[i], chains) + 1e-3 * t->variants[i].R;
            if (bi < 0 || c < *cost) { bi = i; *cost = c; }
        }
        return bi;
    };
    double best_total = reg_cost(m, whole, q.C) * kPlanConst.split_gain;
    for (int i = 0; i < t->nvariants; ++i) {  // head variant: any register variant, its exactly-filling quantum
        const lr::Variant& a = t->variants[i];
        if (!fits(a)) continue;
        const int64_t quantum = want_waves * 64 / a.G;
        if (quantum <= 0 || q.C <= quantum || q.C % quantum == 0) continue;
        const int64_t head = q.C / quantum * quantum;
        double cb = 0;
        const int bi = best_reg(q.C - head, &cb);
        if (bi < 0) continue;
        const double total = reg_cost(m, a, head) + cb;
        if (total < best_total) {
            best_total = total;
            *out = Plan{a.mode, a.G, a.R, family == lr::MODE_MIXED ? out->lds_bytes : 0};
            out->split = head;
            out->mode2 = family;
            out->G2 = t->variants[bi].G;
            out->R2 = t->variants[bi].R;
            out->corun = family == lr::MODE_REG;
        }
    }
}

// (6b) ... and of the fused matrix-core kernel with its operands in registers (HMC under LR_PREC_AUTO; S = 4: 16 chains per workgroup,
// S = 1: 64): a remainder of at most a quarter of the exactly-filling count goes to the register kernel the model prefers for it and
// runs beside the head (profiles/r4_two_part_auto_probe.txt, n = 200, ms per 20 iterations, one launch | two parts: 5120 chains
// 0.538 | 0.440, 9216: 0.758 | 0.645, 20 480: 1.391 | 1.074; not for half a count: 6144: 0.539 | 0.666, 10 240: 0.697 | 0.701).  Its
// chains then run with exact interior gradients, which LR_PREC_AUTO permits (LR_PREC_BF16 asks for the matrix pipe: no second part).
void plan_second_part_mfma(const PlanReq& q, Plan* out) {
    const lr_model* m = q.m;
    const lr::InstTable* t = m->table;
    if (out->mode != lr::MODE_MFMA || out->R <= 0) return;
    const int64_t quantum = (int64_t)(out->G == 1 ? 64 : 16) * m->cus;
    const int64_t head = q.C / quantum * quantum, rem = q.C - head;
    if (head == 0 || rem == 0 || 4 * rem > quantum) return;
    // beside the head only while it fits lane groups of 32 or 64: a remainder of 4096 chains on 16 lanes per chain beside a 16 384-chain
    // head measured 1.43 ms against 1.39 for the single launch (its 256 uncapped workgroups double up on CUs) -- such a remainder runs
    // AFTER the head (0.70 + 0.38 ms)
    out->corun = rem <= 8LL * m->cus;
    // (float64 models: the remainder on the float32-interior kernels, after the head -- 18 432 chains: 0.96 + 0.41 ms against 1.92)
    const int family = m->dtype == LR_F32 ? lr::MODE_REG : lr::MODE_MIXED;
    int bi = -1;
    double cost = 0;
    for (int i = 0; i < t->nvariants; ++i) {
        const lr::Variant& u = t->variants[i];
        if (u.mode != family || (int64_t)u.G * u.R < m->n) continue;
        const double c = reg_cost(m, u, rem) + 1e-3 * u.R;
        if (bi < 0 || c < cost) { bi = i; cost = c; }
    }
    if (bi < 0) return;
    out->split = head;
    out->mode2 = family;
    out->G2 = t->variants[bi].G;
    out->R2 = t->variants[bi].R;
    if (family == lr::MODE_MIXED) {
        out->corun = false;
        if (mixed_lds_bytes(m) > out->lds_bytes) out->lds_bytes = mixed_lds_bytes(m);  // (one figure for both parts: lr_engine.h)
    }
}

// `hmc_bf16`: the run is HMC and its interior leapfrog gradients may use the bf16 matrix pipe (LR_PREC_AUTO / BF16)
// `exact_tail_ok`: ... and need not (LR_PREC_AUTO): a remainder may run on the all-fp32 register kernels
// `kind`: LR_KIND_* of the run, -1 = not a run of a known family (lr_plan, lr_eval)
int make_plan(const lr_model* m, int64_t C, int group, int mode, Plan* out, bool for_eval = false, bool hmc_bf16 = false, int kind = -1,
              bool exact_tail_ok = false) {
    if (for_eval && (mode == LR_MODE_MFMA || mode == LR_MODE_STEPWISE) && m->P <= 32) { mode = LR_MODE_AUTO; group = 0; }
    const PlanReq q{m, C, group, mode, for_eval, kind};
    if (hmc_bf16 && q.automatic() && plan_mfma_hmc(m, C, out)) {
        if (kind >= 0 && (exact_tail_ok || m->dtype != LR_F32)) plan_second_part_mfma(q, out);  // (float64: the tail's interior is reduced too)
        return LR_OK;
    }
    int mixed_whole = -1;
    if (hmc_bf16 && q.automatic() && plan_mixed_hmc(m, C, out, &mixed_whole)) {
        if (kind >= 0) plan_second_part(q, m->table->variants[mixed_whole], out, lr::MODE_MIXED);
        return LR_OK;
    }
    if (m->P > 32) return plan_wide(q, out);
    const size_t row_bytes = (size_t)m->n * m->P * m->esize();
    // (float64 at 17 <= p <= 32: the fused distributed-state kernel keeps its lead over the stepwise engine while the rows fit the LDS --
    //  n=500 p=20, HMC L=20, it/s, lds 16 | stepwise: 4096 chains 1.53 | 0.98e7, 16 384: 1.54 | 1.47e7; profiles/r5_f64_p32.txt)
    const bool f64_wide_fused = m->dtype == LR_F64 && m->P == 32;
    const bool prefer_stepwise = lds_rows_bytes(m) > kLdsBudget ||
                                 (!f64_wide_fused && row_bytes > kPlanConst.lds_rows_prefer_stepwise_bytes && C >= kPlanConst.lds_rows_prefer_stepwise_chains);
    if (!for_eval && (mode == LR_MODE_STEPWISE || (mode == LR_MODE_AUTO && group == 0 && prefer_stepwise))) {
        plan_tall(q, out);
        return LR_OK;
    }
    int best = pick_variant(q);
    if (best < 0)
        return fail(LR_ERR_UNSUPPORTED, "no kernel variant for dtype=%d p=%d (padded %d) n=%lld group=%d mode=%d",
                    m->dtype, m->p, m->P, (long long)m->n, group, mode);
    if (q.automatic() && measured_overrides(q, &best, out)) return LR_OK;
    const lr::Variant& v = m->table->variants[best];
    *out = Plan{v.mode, v.G, v.R, v.mode == lr::MODE_MIXED ? mixed_lds_bytes(m) : v.mode == lr::MODE_LDS ? lds_rows_bytes(m) : (v.mode == lr::MODE_MFMA && m->dtype != LR_F32) ? row_bytes : (v.mode == lr::MODE_MFMA && v.R == 0 ? mfma_lds_bytes(m, v.G) : 0)};
    if (v.mode == lr::MODE_REG && q.automatic() && kind >= 0) plan_second_part(q, v, out);
    return LR_OK;
}

// `group` means different things per mode (include/logreg_hip.h): lanes per chain (REG / LDS / GLOBAL / AUTO on narrow models:
// a power of two <= 64), row-split ways of the matrix-core chain kernel (MFMA: 1, 4, 8), or the slice count of the stepwise
// engine (STEPWISE, and every mode of a wide model: any positive count up to one slice per 32-row block -- ceil(n / slice_len)
// is arbitrary, e.g. 63 for n = 20 000 at 1024 chains, and lr_plan's group_out must round-trip)
int check_group_for(const lr_model* m, int group, int mode) {
    if (group == 0) return LR_OK;
    if (group < 0) return fail(LR_ERR_INVALID, "group must be >= 0 (got %d)", group);
    if (mode == LR_MODE_STEPWISE || m->P > 32) {
        const int64_t max_slices = (m->n + 31) / 32;
        if (group > max_slices) return fail(LR_ERR_INVALID, "stepwise slice count %d exceeds the %lld 32-row blocks of the data", group, (long long)max_slices);
        return LR_OK;
    }
    if (mode == LR_MODE_MFMA) {
        if (group != 1 && group != 4 && group != 8) return fail(LR_ERR_INVALID, "matrix-core mode: group (row-split ways) must be 0, 1, 4 or 8 (got %d)", group);
        return LR_OK;
    }
    if (group > 64 || (group & (group - 1))) return fail(LR_ERR_INVALID, "group (lanes per chain) must be 0 or a power of two <= 64 (got %d)", group);
    return LR_OK;
}

// Row slices of the reduced-precision INTERIOR leapfrog steps of the stepwise engine (HMC, precision policy permitting): how many
// slices RS_i of how many rows, and the workgroup shape of the kernel that takes them (waves: 4 / 8 row-split waves of the wide
// kernel, 4 or 16 waves of the tall matrix-pipe kernel).  RS_i = 0: no interior kernel for this model (the exact kernels run every
// step).  Cp: the chain count the run is planned for (lr_run_opts.plan_chains).
struct InteriorPlan { int RS_i; int64_t slice_len_i; int waves; };
InteriorPlan plan_interior(const lr_model* m, int64_t Cp) {
    int RS_i = 0, rs_waves = 4;
    int64_t slice_len_i = 0;
    // wide models, interior leapfrog steps with few chains: the row-split kernel (lr_wide_bf16.h) wants one chain
    // tile of 16 per workgroup and as many row slices as fill the chip; each slice a multiple of 128 rows
    if (m->P > 32 && m->d_xblk1) {
        const int64_t tiles = (Cp + 15) / 16;
        if (tiles <= (int64_t)m->cus) {
            int64_t want = m->cus / tiles;
            if (want < 1) want = 1;
            rs_waves = 8;  // (4 waves per workgroup measured slower at every chain count the row split is used for)
            const int64_t quantum = 32 * rs_waves;
            slice_len_i = ((m->n + want - 1) / want + quantum - 1) / quantum * quantum;
            RS_i = (int)((m->n + slice_len_i - 1) / slice_len_i);
        }
    }
    if (m->P <= 32 && m->d_xmx) {
        // narrow models, interior leapfrog steps on the matrix pipe (lr_tall_mx.h): 4-wave workgroups of 64 chains;
        // slices fine enough for ~4 waves per SIMD, each at least 256 rows, whole tile pairs
        const int64_t blocks = (Cp + 63) / 64;
        int64_t want = (4LL * 4 * m->cus + 4 * blocks - 1) / (4 * blocks);
        if (want < 1) want = 1;
        slice_len_i = ((m->n + want - 1) / want + 31) / 32 * 32;
        if (slice_len_i < 256) slice_len_i = 256;
        RS_i = (int)((m->n + slice_len_i - 1) / slice_len_i);
        // 16-wave workgroups (one per CU) need a quarter of the slices for the same waves per SIMD: few enough to fold
        // the update launch into the next launch's prologue (k_tall_partial_mx16).  Only with enough rows per slice to
        // keep the 4 row groups of a workgroup busy, and when the slice count fits the fused prologue.
        // (round 5: TWO such workgroups per CU -- 32 slices, 8 waves per SIMD, the fused prologue summing 32 partials -- measured slower:
        //  config 4's design 24.7 -> 27.4 us per evaluation at 1024 chains, 44.5 -> 47.4 at 2048; the vector ALU is 71 % busy at 4 waves per SIMD
        //  already and the extra slices cost more than the occupancy buys)
        int64_t want16 = (m->cus + blocks - 1) / blocks;
        if (want16 < 1) want16 = 1;
        int64_t len16 = ((m->n + want16 - 1) / want16 + 31) / 32 * 32;
        if (len16 < 1024) len16 = 1024;
        const int64_t rs16 = (m->n + len16 - 1) / len16;
        // (and only when those slices still fill the chip: n=5000 p=30 at 1024 chains would get 5 slices x 16 blocks = 80
        //  workgroups and ran 13.5 -> 17.9 us per step; n=20 000 p=12: 17.8 -> 13.8, config 4: 29.7 -> 27.5)
        //  at 4096 chains the steps are long enough that the saved launch no longer shows: -3 .. +5 %, so up to 2048 chains)
        if (m->dbg.tall_mx16 && rs16 >= 1 && rs16 <= 16 && RS_i > rs16 && 4 * rs16 * blocks >= 3LL * m->cus && blocks <= 32) {
            RS_i = (int)rs16;
            slice_len_i = len16;
            rs_waves = 16;
        }
    }
    return InteriorPlan{RS_i, slice_len_i, rs_waves};
}


}  // namespace
