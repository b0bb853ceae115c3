// lr_tall_mx.h -- INTERIOR-step partial kernel of the stepwise engine for NARROW tall models (padded p = 8, 16 or 32,
// float32; written out below for p = 8: wider models repeat the scheme per set of 8 coordinates) on the bf16 matrix pipe: the counterpart of lr_wide_bf16.h's interior kernels for data with many rows
// and few columns (BASELINE config 4: n = 100 000, p = 8).
//
// Why.  The fp32 vector kernel (k_tall_partial) spends 10.5 VALU instructions per row and 64 chains; it sits at
// ~0.5 of the fp32 vector peak and cannot go further.  Of those instructions only the two
// transcendentals of the sigmoid have to be VALU work: the 2 x 8 multiply-adds per row and chain are a GEMM with a
// tiny inner dimension.  For the L - 1 INTERIOR gradient evaluations of an HMC trajectory (any deterministic force
// keeps the leapfrog map volume-preserving and reversible; the Metropolis test uses the exact end-point values --
// see include/logreg_hip.h, LR_PREC_*) the GEMMs move to v_mfma_f32_16x16x32_bf16 and the VALU keeps exp, rcp and
// a bf16 pack: 26 instructions per 32 rows x 16 chains instead of 84.
//
// K = 32 of the MFMA is filled with PIECES, not coordinates (p = 8 would leave it three quarters empty):
//   x = xh + xl, beta = bh + bl (two round-to-nearest bf16 pieces each: 16 significand bits), and
//   eta = sum_j (xh_j + xl_j)(bh_j + bl_j) takes all four piece products of every coordinate from 2 x 16 K-slots.
// Lane l = (c, kg) of a wave (c = l & 15: chain, kg = l >> 4) OWNS coordinates a = kg and b = kg + 4.
//   eta tile (16 rows x 16 chains), ONE K = 32 MFMA (v_mfma_f32_16x16x32_bf16) per coordinate set:
//                                    A (lane (row, kg)) = [xh_a xl_a xh_b xl_b | xh_a xl_a xh_b xl_b]   (K-slots 8 kg .. 8 kg + 7): the lane's
//                                                         8 image bytes TWICE, by one ds_read2_b64 with equal offsets (mx_read_dup)
//                                    B (lane (c,   kg)) = [bh_a bh_a bh_b bh_b | bl_a bl_a bl_b bl_b]
//                                    D (lane (c, kg), r) = eta[row 4 kg + r][chain c]
//   (round 2 used two K = 16 MFMAs per tile and set; measured in isolation, tools/mx_loop_probe.hip, ns per tile pair and SIMD at
//    4 waves per SIMD: 117.6 -> 105.3 with the K = 32 form, -> 91.8 with two pairs per trip so that one pair's MFMAs issue among the
//    other's exp / rcp; the vector ALU alone -- 16 transcendentals, 4 packed adds, 4 packs -- takes 79.6, the MFMAs + LDS alone 45.7)
//   w = sigma(-eta) on the accumulator registers of two tiles, rounded to ONE bf16 piece = B of the gradient MFMA
//       (K-slot 8 kg + i <-> i < 4: tile 0 row 4 kg + i; i >= 4: tile 1 row 4 kg + i - 4: the lane's own outputs)
//   grad tile (K = 32 rows):  A (lane (m', kg)) = the 8 rows of slot group kg, element m' & 3 of coordinate group
//                             m' >> 2, i.e. M-row 4 kg'' + r' = (xh_a, xl_a, xh_b, xl_b)[r'] of group kg'' -- exactly
//                             what ds_read_b64_tr_b16 delivers from the SAME image the eta operand is read from
//                             D (lane (c, kg), r) = sum_rows w (xh_a, xl_a, xh_b, xl_b)[r] of the lane's own group:
//                             g_a = D0 + D1, g_b = D2 + D3 -- the lane's own coordinates, no data movement.
// Image (built once at model creation, 32 bytes per row like the fp32 rows): per 16-row tile
//   [kg''][row'][4 bf16],  row' = (row + 8 (kg'' >> 1)) & 15   (the swizzle that keeps the transposing reads of a
//   half-wave on distinct banks; the eta reads, 8 bytes per lane, are conflict-free with or without it).
// Workgroup = NW waves x 16 chains sharing one row slice; the slice streams L2 -> LDS in chunks of kChunkTiles
// tiles with the LDS-DMA load (inline asm: see lr_wide_bf16.h for why), two chunks of LDS (32 KB per workgroup:
// four workgroups per CU), one barrier per chunk.
#pragma once
#include <cstring>
#include <utility>

#include "lr_tall.h"

namespace lr {

typedef float mx_f32x4 __attribute__((ext_vector_type(4)));
typedef float mx_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 mx_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 mx_bf16x2 __attribute__((ext_vector_type(2)));
typedef uint32_t mx_u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t mx_u32x2 __attribute__((ext_vector_type(2)));
typedef short mx_s16x4 __attribute__((ext_vector_type(4)));

constexpr int kMxSetElems = 4 * 16 * 4;   // bf16 elements of one coordinate set (8 coordinates) of a 16-row tile: 512 bytes
constexpr int kMxChunkBytes = 16384;      // staged chunk: 512 rows at P = 8, 256 at P = 16, 128 at P = 32 (whole tile pairs)
// P = 8 s coordinates: lane (., kg) handles, for every set s < P / 8, the pair a = 8 s + kg, b = 8 s + kg + 4; a tile image is
// [s][kg][row'][4 bf16] and everything below runs once per set (eta: 2 more MFMAs into the same accumulator; gradient: one
// more MFMA and accumulator per set).
template <int P> struct MxGeom {
    static_assert(P == 8 || P == 16 || P == 32, "coordinate sets of 8");
    static constexpr int NS = P / 8;
    static constexpr int TILE = NS * kMxSetElems;              // bf16 elements per tile image
    static constexpr int CHUNK_TILES = kMxChunkBytes / (TILE * 2);
    static constexpr int PAIRS_PER_TRIP = NS <= 2 ? 2 : 1;     // tile pairs per call of mx_pairs (register budget: 4 waves per SIMD)
};

__host__ __device__ constexpr int mx_elem(int kg, int row) { return kg * 64 + ((row + 8 * (kg >> 1)) & 15) * 4; }

__device__ __forceinline__ uint32_t mx_pack_rne(float a, float b) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(mx_f32x2{a, b}, mx_bf16x2));
}
__device__ __forceinline__ mx_u32x2 mx_read_tr16(const uint16_t* p) {
    return __builtin_bit_cast(mx_u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) mx_s16x4*)(p)));
}

// the transposing read by LDS byte address (constant offsets fold into the instruction)
__device__ __forceinline__ mx_u32x2 mx_read_tr16_at(uint32_t lds_addr) {
    return __builtin_bit_cast(mx_u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) mx_s16x4*)(uintptr_t)lds_addr));
}
// The eta operands of a trip: the lane's 8 bytes [xh_a xl_a xh_b xl_b] of every (tile, set) image, each duplicated into 4 registers
// by ONE ds_read2_b64 whose two offsets are equal -- the duplicate comes out of the LDS unit, which has the slack, not out of v_mov
// on the vector ALU, which has none.  Inline asm (told about the two identical halves the compiler reads once and copies).  The
// compiler does not count an asm read, so the reads of a trip AND the wait for them are ONE asm statement: no copy or spill of a
// destination register can be scheduled between a read and its data (two statements -- reads, then an s_waitcnt naming the
// destinations -- only happened to stay adjacent).  Image I = tile * NS + set sits 512 I bytes behind `lds_addr`; the
// instruction's offsets are 8-bit in units of 8 bytes, so four images share an address register (one v_add per 2 KB window).
#define LR_MX_RD(o, a, imm) "ds_read2_b64 %" #o ", %" #a " offset0:" #imm " offset1:" #imm "\n\t"
#define LR_MX_RD4(o0, o1, o2, o3, a) LR_MX_RD(o0, a, 0) LR_MX_RD(o1, a, 64) LR_MX_RD(o2, a, 128) LR_MX_RD(o3, a, 192)
template <int N> __device__ __forceinline__ void mx_read_dup_all(uint32_t lds_addr, mx_u32x4* x) {
    static_assert(N == 2 || N == 4 || N == 8 || N == 16, "images per trip");
    if constexpr (N == 2)
        asm volatile(LR_MX_RD(0, 2, 0) LR_MX_RD(1, 2, 64) "s_waitcnt lgkmcnt(0)" : "=&v"(x[0]), "=&v"(x[1]) : "v"(lds_addr) : "memory");
    else if constexpr (N == 4)
        asm volatile(LR_MX_RD4(0, 1, 2, 3, 4) "s_waitcnt lgkmcnt(0)" : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]) : "v"(lds_addr) : "memory");
    else if constexpr (N == 8)
        asm volatile(LR_MX_RD4(0, 1, 2, 3, 8) LR_MX_RD4(4, 5, 6, 7, 9) "s_waitcnt lgkmcnt(0)"
                     : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]), "=&v"(x[4]), "=&v"(x[5]), "=&v"(x[6]), "=&v"(x[7])
                     : "v"(lds_addr), "v"(lds_addr + 2048u)
                     : "memory");
    else
        asm volatile(LR_MX_RD4(0, 1, 2, 3, 16) LR_MX_RD4(4, 5, 6, 7, 17) LR_MX_RD4(8, 9, 10, 11, 18) LR_MX_RD4(12, 13, 14, 15, 19) "s_waitcnt lgkmcnt(0)"
                     : "=&v"(x[0]), "=&v"(x[1]), "=&v"(x[2]), "=&v"(x[3]), "=&v"(x[4]), "=&v"(x[5]), "=&v"(x[6]), "=&v"(x[7]), "=&v"(x[8]), "=&v"(x[9]),
                       "=&v"(x[10]), "=&v"(x[11]), "=&v"(x[12]), "=&v"(x[13]), "=&v"(x[14]), "=&v"(x[15])
                     : "v"(lds_addr), "v"(lds_addr + 2048u), "v"(lds_addr + 4096u), "v"(lds_addr + 6144u)
                     : "memory");
}
#undef LR_MX_RD4
#undef LR_MX_RD

// NPAIR (1 or 2) adjacent tile pairs of one wave's 16 chains: eta of every tile (K = 32 MFMA per coordinate set), w = sigma(-eta)
// rounded to one bf16 piece, gradient MFMA per pair and set.  Two pairs per call leave the compiler one pair's MFMAs to place
// among the other pair's exp / rcp (tools/mx_loop_probe.hip: 105 -> 92 ns per pair and SIMD).
//   eta_lds / tr_lds   LDS byte address of the first tile image + the lane's offset for the eta read / the transposing read
//   b32[st]            B operand of the eta MFMA of coordinate set st
template <int P, int NPAIR>
__device__ __forceinline__ void mx_pairs(uint32_t eta_lds, uint32_t tr_lds, const mx_u32x4 (&b32)[MxGeom<P>::NS], mx_f32x4 (&gacc)[MxGeom<P>::NS]) {
    using G = MxGeom<P>;
    constexpr int NS = G::NS, TILE = G::TILE, NT = 2 * NPAIR;
    static_assert(kMxSetElems * 2 == 512 && TILE * 2 == 512 * NS, "image I = tile * NS + set sits 512 I bytes behind the first");
    mx_u32x2 xt[NT][NS];  // the gradient operands (transposing reads of the same images), requested first
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int st = 0; st < NS; ++st) xt[t][st] = mx_read_tr16_at(tr_lds + (uint32_t)(t * TILE + st * kMxSetElems) * 2);
    mx_u32x4 xa[NT][NS];
    mx_read_dup_all<NT * NS>(eta_lds, &xa[0][0]);  // (reads + lgkmcnt(0) in one statement; the LDS counter is in order: xt has landed too)
    mx_f32x4 e[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        e[t] = mx_f32x4{0, 0, 0, 0};
#pragma unroll
        for (int st = 0; st < NS; ++st)
            e[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mx_bf16x8, xa[t][st]), __builtin_bit_cast(mx_bf16x8, b32[st]), e[t], 0, 0, 0);
    }
#pragma unroll
    for (int pr = 0; pr < NPAIR; ++pr) {
        uint32_t wq[4];
#pragma unroll
        for (int T = 0; T < 2; ++T) {
            const mx_f32x4& et = e[2 * pr + T];
            // (the "1 +" as two packed adds: as four v_add_f32 -- packed f32 VALU being an anti-lever beside MFMAs elsewhere -- config 4 measured
            //  25.9 -> 26.4 us per evaluation at 1024 chains, 84.5 -> 87.7 at 4096: round 5)
            const mx_f32x2 d0 = mx_f32x2{__builtin_amdgcn_exp2f(et[0]), __builtin_amdgcn_exp2f(et[1])} + mx_f32x2{1.0f, 1.0f};
            const mx_f32x2 d1 = mx_f32x2{__builtin_amdgcn_exp2f(et[2]), __builtin_amdgcn_exp2f(et[3])} + mx_f32x2{1.0f, 1.0f};
            wq[2 * T] = mx_pack_rne(fast_rcp(d0.x), fast_rcp(d0.y));
            wq[2 * T + 1] = mx_pack_rne(fast_rcp(d1.x), fast_rcp(d1.y));
        }
        const mx_u32x4 wv = {wq[0], wq[1], wq[2], wq[3]};
#pragma unroll
        for (int st = 0; st < NS; ++st) {
            const mx_u32x4 xg = {xt[2 * pr][st][0], xt[2 * pr][st][1], xt[2 * pr + 1][st][0], xt[2 * pr + 1][st][1]};
            gacc[st] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mx_bf16x8, xg), __builtin_bit_cast(mx_bf16x8, wv), gacc[st], 0, 0, 0);
        }
    }
}

// B operands of the eta MFMAs from a chain's position: [bh_a bh_a bh_b bh_b | bl_a bl_a bl_b bl_b] of the lane's coordinates
// a = 8 st + kg, b = 8 st + kg + 4, times log2(e), in two round-to-nearest bf16 pieces
__device__ __forceinline__ mx_u32x4 mx_beta_operand(float qa_, float qb_) {
    const float qa = qa_ * ExpScale<float>::k, qb = qb_ * ExpScale<float>::k;
    const uint32_t ha = mx_pack_rne(qa, qa), hb = mx_pack_rne(qb, qb);
    const float la = qa - __builtin_bit_cast(float, ha << 16), lb = qb - __builtin_bit_cast(float, hb << 16);
    return mx_u32x4{ha, hb, mx_pack_rne(la, la), mx_pack_rne(lb, lb)};
}

// S = double: the same kernel on a FLOAT64 model (its default precision policy: lr_tall.h's float64 kernel runs the end points) --
// the position is read as float64 and rounded, the slice partials stay float32 (TallArgs::part_f32) for k_tall_update<double> to sum.
template <int P, int NW, typename S = float>
__global__ void __launch_bounds__((64 * NW)) k_tall_partial_mx(TallArgs<S, P> a) {
    using G = MxGeom<P>;
    constexpr int NS = G::NS, kMxTileElems = G::TILE, kMxChunkTiles = G::CHUNK_TILES;
    constexpr int CHUNK_BYTES = kMxChunkBytes;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * CHUNK_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, kg = lane >> 4;
    // grid = (row slices, chain blocks): slice index fastest, so that an XCD (workgroups are dealt round-robin by
    // linear id) streams only its own eighth of the rows, for all chain blocks, through its L2 (see k_tall_partial)
    int64_t chain = (int64_t)blockIdx.y * (16 * NW) + 16 * wave + c;
    const bool live = chain < a.C;
    if (!live) chain = a.C - 1;
    const int rs = blockIdx.x;
    // its own slicing (RS_i slices of slice_len_i rows, whole tile pairs): finer than the fp32 kernel's, so that
    // every SIMD holds ~4 of these 4-wave workgroups' waves and the LDS -> MFMA -> exp -> rcp -> MFMA chain of one
    // tile pair is covered by the other waves (one wave per SIMD: 350 cycles per pair; the VALU work is 170)
    const int64_t s0 = (int64_t)rs * a.slice_len_i, s1 = s0 + a.slice_len_i < a.n ? s0 + a.slice_len_i : a.n;
    const int64_t tile0 = s0 / 16;
    const int64_t ntile = s1 > s0 ? ((s1 - s0 + 31) / 32) * 2 : 0;  // tiles of this slice (even; the image is zero-padded)
    const int64_t nchunk = (ntile + kMxChunkTiles - 1) / kMxChunkTiles;
    const uint32_t smem_lds = (uint32_t)(uintptr_t)smem;

    auto issue = [&](int64_t g) {  // chunk g -> buffer g & 1; 1 KB per wave-instruction, dealt round-robin to the waves
        const int64_t t0 = g * kMxChunkTiles;
        const int nt = (int)(ntile - t0 < kMxChunkTiles ? ntile - t0 : kMxChunkTiles);
        const unsigned char* src = reinterpret_cast<const unsigned char*>(a.xmx + (tile0 + t0) * (int64_t)kMxTileElems) + lane * 16;
        const uint32_t dst = smem_lds + (uint32_t)((g & 1) * CHUNK_BYTES);
        const int nkb = nt * (kMxTileElems * 2) / 1024;  // 1 KB per wave-instruction (nt is even: whole KB at every P)
        for (int ch = wave; ch < nkb; ch += NW) {
            uint32_t keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep)
                         : "v"(src + ch * 1024), "s"(dst + ch * 1024)
                         : "memory");
        }
    };
    if (nchunk > 0) issue(0);

    mx_u32x4 b32[NS];
#pragma unroll
    for (int st = 0; st < NS; ++st) b32[st] = mx_beta_operand((float)a.q1[chain * P + 8 * st + kg], (float)a.q1[chain * P + 8 * st + kg + 4]);

    const int eta_off = mx_elem(kg, c);                                            // lane (row c, kg): its 4 elements
    const int tr_off = mx_elem(lane & 3, 4 * kg + ((lane & 15) >> 2));             // lane (kg, ri, ci): chunk ci, row 4 kg + ri
    mx_f32x4 gacc[NS];
#pragma unroll
    for (int st = 0; st < NS; ++st) gacc[st] = mx_f32x4{0, 0, 0, 0};

    for (int64_t g = 0; g < nchunk; ++g) {
        __builtin_amdgcn_s_waitcnt(0x0F70);  // own share of chunk g has landed ...
        __syncthreads();                     // ... everybody's has, and nobody still reads the other buffer
        if (g + 1 < nchunk) issue(g + 1);
        const int nt = (int)(ntile - g * kMxChunkTiles < kMxChunkTiles ? ntile - g * kMxChunkTiles : kMxChunkTiles);
        // running LDS addresses of the eta and the transposing reads (opaque to the optimiser: left to it, it keeps the chunk offset
        // apart and adds it back on every trip -- 4 address adds per trip instead of 2)
        uint32_t eta_lds = smem_lds + (uint32_t)((g & 1) * CHUNK_BYTES) + 2u * eta_off, tr_lds = smem_lds + (uint32_t)((g & 1) * CHUNK_BYTES) + 2u * tr_off;
        asm volatile("" : "+v"(eta_lds), "+v"(tr_lds));
        constexpr uint32_t kTrip = 2 * G::PAIRS_PER_TRIP * kMxTileElems * 2;
        int t = 0;
        for (; t + 2 * G::PAIRS_PER_TRIP <= nt; t += 2 * G::PAIRS_PER_TRIP, eta_lds += kTrip, tr_lds += kTrip) mx_pairs<P, G::PAIRS_PER_TRIP>(eta_lds, tr_lds, b32, gacc);
        for (; t < nt; t += 2, eta_lds += 2 * kMxTileElems * 2, tr_lds += 2 * kMxTileElems * 2) mx_pairs<P, 1>(eta_lds, tr_lds, b32, gacc);
    }
    if (live) {
        float* dst = reinterpret_cast<float*>(a.part_g) + ((int64_t)rs * a.C + chain) * P;  // (float64 models: TallArgs::part_f32)
#pragma unroll
        for (int st = 0; st < NS; ++st) {
            dst[8 * st + kg] = gacc[st][0] + gacc[st][1];
            dst[8 * st + kg + 4] = gacc[st][2] + gacc[st][3];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// 16-WAVE form (one workgroup per CU: 4 chain groups x 4 row groups): the same tile-pair work, but a workgroup covers
// 4x the rows of a slice, so the same four waves per SIMD need a quarter of the slices (16 instead of 64 for config 4)
// -- few enough that every launch but the first can finish the PREVIOUS leapfrog step in its own prologue (kick with
// the slice partials, drift: k_tall_update's PH_MID, as k_wide_partial_bf16r does for wide models) instead of a
// launch of its own.  Wave w: chain group cg = w & 3 (chains 16 cg .. 16 cg + 15 of the block's 64), row group
// rg = w >> 2 (every 4th tile pair of a chunk); the four row groups' gradients meet in LDS in row-group order.
// (2 chain groups x 8 row groups -- 32 chains per workgroup, two workgroups per CU, eight waves per SIMD -- was measured:
// config 4 unchanged (29.3 us), p = 12 / 24 / 32 slower (13.5 -> 16.1, 22.9 -> 27.9, 43.3 -> 45.6 us per step).)
constexpr int kMx16FuseSlices = 16;  // the host fuses only when RS_i <= this
// S = double: a FLOAT64 model's interior steps (position, momentum and the fused update float64; float32 slice partials).
template <int P, typename S = float>
__global__ void __launch_bounds__(1024) k_tall_partial_mx16(TallArgs<S, P> a) {
    using G = MxGeom<P>;
    constexpr int CHUNK_BYTES = 2 * kMxChunkBytes;  // 32 KB chunks: half the barriers of the 4-wave kernel (config 4: 30.2 -> 28.9 us per step; 48 KB: 29.5)
    constexpr int NS = G::NS, kMxTileElems = G::TILE, kMxChunkTiles = CHUNK_BYTES / (G::TILE * 2), NW = 16;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * CHUNK_BYTES];
    __shared__ __attribute__((aligned(16))) float qnew[64][P];
    __shared__ __attribute__((aligned(16))) float red[4][64][P];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cg = wave & 3, rg = wave >> 2;
    const int c = lane & 15, kg = lane >> 4;
    const int64_t chain0 = (int64_t)blockIdx.y * 64;
    const int rs = blockIdx.x;
    const int64_t s0 = (int64_t)rs * a.slice_len_i, s1 = s0 + a.slice_len_i < a.n ? s0 + a.slice_len_i : a.n;
    const int64_t tile0 = s0 / 16;
    const int64_t ntile = s1 > s0 ? ((s1 - s0 + 31) / 32) * 2 : 0;
    const int64_t nchunk = (ntile + kMxChunkTiles - 1) / kMxChunkTiles;
    const uint32_t smem_lds = (uint32_t)(uintptr_t)smem;
    LR_STAMP(a, 0);

    auto issue = [&](int64_t g) {  // chunk g -> buffer g & 1; 1 KB per wave-instruction, dealt round-robin to the 16 waves
        const int64_t t0 = g * kMxChunkTiles;
        const int nt = (int)(ntile - t0 < kMxChunkTiles ? ntile - t0 : kMxChunkTiles);
        const unsigned char* src = reinterpret_cast<const unsigned char*>(a.xmx + (tile0 + t0) * (int64_t)kMxTileElems) + lane * 16;
        const uint32_t dst = smem_lds + (uint32_t)((g & 1) * CHUNK_BYTES);
        const int nkb = nt * (kMxTileElems * 2) / 1024;
        for (int ch = wave; ch < nkb; ch += NW) {
            uint32_t keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep)
                         : "v"(src + ch * 1024), "s"(dst + ch * 1024)
                         : "memory");
        }
    };
    if (nchunk > 0) issue(0);

    // the position this launch evaluates at: either stored (first interior step) or finished here from the previous step's
    // slice partials (fuse_mid): g1 = sum over the slices (slice order, fp64) - q ivar;  p += eps g1;  q += (eps / m) p
    for (int e = tid; e < 64 * P; e += 64 * NW) {
        const int cc = e / P, j = e % P;
        int64_t chain = chain0 + cc;
        const bool live = chain < a.C;
        if (!live) chain = a.C - 1;
        const int64_t ix = chain * P + j;
        S q;
        if (a.fuse_mid) {
            float t[kMx16FuseSlices];  // (the partials are float32 whatever S)
            const float* part_f = reinterpret_cast<const float*>(a.part_in);
#pragma unroll
            for (int r = 0; r < kMx16FuseSlices; ++r) t[r] = part_f[((int64_t)(r < a.RS_i ? r : a.RS_i - 1) * a.C) * P + ix];
            const S q0 = a.q1_in[ix], p0 = a.pm_in[ix], bj = a.cvec[j], ivj = a.cvec[P + j];
            double sum = 0.0;
#pragma unroll
            for (int r = 0; r < kMx16FuseSlices; ++r) sum += r < a.RS_i ? (double)t[r] : 0.0;
            const S g1 = (S)sum - q0 * ivj;
            const S pn = fma_t(a.step, g1, p0);
            q = fma_t(bj, pn, q0);
            if (rs == 0 && live) {
                a.q1[ix] = q;
                a.pm[ix] = pn;
            }
        } else {
            q = a.q1[ix];
        }
        qnew[cc][j] = (float)q;
    }
    LR_STAMP(a, 1);
    __builtin_amdgcn_s_waitcnt(0x0F70);  // (the loads above and this wave's share of chunk 0)
    __syncthreads();
    LR_STAMP(a, 2);

    mx_u32x4 b32[NS];
#pragma unroll
    for (int st = 0; st < NS; ++st) b32[st] = mx_beta_operand(qnew[16 * cg + c][8 * st + kg], qnew[16 * cg + c][8 * st + kg + 4]);
    const int eta_off = mx_elem(kg, c);
    const int tr_off = mx_elem(lane & 3, 4 * kg + ((lane & 15) >> 2));
    mx_f32x4 gacc[NS];
#pragma unroll
    for (int st = 0; st < NS; ++st) gacc[st] = mx_f32x4{0, 0, 0, 0};

    LR_STAMP(a, 3);
    LR_STAMP_CLK(a, 8);
#ifdef LR_STAMPS
    unsigned long long bar_cyc = 0, dma_cyc = 0;  // shader cycles this wave spent in the chunk barriers / issuing the chunk loads
#endif
    for (int64_t g = 0; g < nchunk; ++g) {
#ifdef LR_STAMPS
        const unsigned long long tb0 = __builtin_amdgcn_s_memtime();
#endif
        if (g > 0 && !LR_DBG(a, 0)) {  // (dbg bit 0: no chunk barriers; bit 1: no chunk loads -- timing experiments)
            __builtin_amdgcn_s_waitcnt(0x0F70);
            __syncthreads();
        }
#ifdef LR_STAMPS
        const unsigned long long tb1 = __builtin_amdgcn_s_memtime();
#endif
        if (g + 1 < nchunk && !LR_DBG(a, 1)) issue(g + 1);
#ifdef LR_STAMPS
        bar_cyc += tb1 - tb0;
        dma_cyc += __builtin_amdgcn_s_memtime() - tb1;
#endif
        const int nt = (int)(ntile - g * kMxChunkTiles < kMxChunkTiles ? ntile - g * kMxChunkTiles : kMxChunkTiles);
        // this row group's tile pairs of the chunk: two ADJACENT pairs (tiles 16 k + 4 rg .. 16 k + 4 rg + 3) per round of 16 tiles,
        // so that one trip's images sit within the immediate offsets of one base address
        // running LDS addresses of the eta and the transposing reads: the row group's first tile of the chunk, 16 tiles further per
        // round (opaque to the optimiser: left to it, it keeps the chunk offset apart and adds it back on every trip)
        constexpr uint32_t kRound = 16 * kMxTileElems * 2;
        const uint32_t first = smem_lds + (uint32_t)((g & 1) * CHUNK_BYTES) + (uint32_t)(4 * rg) * (kMxTileElems * 2);
        uint32_t eta_lds = first + 2u * eta_off, tr_lds = first + 2u * tr_off;
        asm volatile("" : "+v"(eta_lds), "+v"(tr_lds));
        int t = 4 * rg;
        for (; t + 4 <= nt; t += 16, eta_lds += kRound, tr_lds += kRound) {
            if constexpr (G::PAIRS_PER_TRIP == 2) {
                mx_pairs<P, 2>(eta_lds, tr_lds, b32, gacc);
            } else {
                mx_pairs<P, 1>(eta_lds, tr_lds, b32, gacc);
                mx_pairs<P, 1>(eta_lds + 2 * kMxTileElems * 2, tr_lds + 2 * kMxTileElems * 2, b32, gacc);
            }
        }
        if (t < nt)  // the slice's last round may end after the first pair of a row group's two (tile counts are even)
            mx_pairs<P, 1>(eta_lds, tr_lds, b32, gacc);
    }
    LR_STAMP_CLK(a, 9);
    LR_STAMP(a, 4);
#ifdef LR_STAMPS
    if (a.stamps && lane == 0) {
        LR_STAMP_AT(a, 14) = bar_cyc;
        LR_STAMP_AT(a, 15) = dma_cyc;
    }
#endif
    // the four row groups' gradients of a chain, summed in row-group order, one slice partial per chain and coordinate
#pragma unroll
    for (int st = 0; st < NS; ++st) {
        red[rg][16 * cg + c][8 * st + kg] = gacc[st][0] + gacc[st][1];
        red[rg][16 * cg + c][8 * st + kg + 4] = gacc[st][2] + gacc[st][3];
    }
    __syncthreads();
    LR_STAMP(a, 5);
    for (int e = tid; e < 64 * P; e += 64 * NW) {
        const int cc = e / P, j = e % P;
        if (chain0 + cc < a.C)
            reinterpret_cast<float*>(a.part_g)[((int64_t)rs * a.C + chain0 + cc) * P + j] = (red[0][cc][j] + red[1][cc][j]) + (red[2][cc][j] + red[3][cc][j]);
    }
    LR_STAMP(a, 6);
}

// Host side: the tile images.  rows: [n][P] fp32 signed rows.  out: [ceil(n/32) * 2][MxGeom<P>::TILE] bf16 bit patterns.
inline uint16_t mx_bf16_rne(float x) {
    uint32_t b;
    memcpy(&b, &x, 4);
    return (uint16_t)((b + 0x7FFFu + ((b >> 16) & 1u)) >> 16);
}
inline float mx_bf16_to_f32(uint16_t h) {
    const uint32_t b = (uint32_t)h << 16;
    float x;
    memcpy(&x, &b, 4);
    return x;
}
template <int P> inline void tall_mx_prepare(const float* rows, int64_t n, uint16_t* out) {
    using G = MxGeom<P>;
    const int64_t ntile = (n + 31) / 32 * 2;
    for (int64_t t = 0; t < ntile; ++t) {
        uint16_t* base = out + t * (int64_t)G::TILE;
        for (int e = 0; e < G::TILE; ++e) base[e] = 0;
        for (int r = 0; r < 16; ++r) {
            const int64_t row = 16 * t + r;
            if (row >= n) continue;
            for (int st = 0; st < G::NS; ++st)
                for (int kg = 0; kg < 4; ++kg) {
                    const float xa = rows[row * P + 8 * st + kg], xb = rows[row * P + 8 * st + kg + 4];
                    const uint16_t ha = mx_bf16_rne(xa), hb = mx_bf16_rne(xb);
                    uint16_t* q = base + st * kMxSetElems + mx_elem(kg, r);
                    q[0] = ha;
                    q[1] = mx_bf16_rne(xa - mx_bf16_to_f32(ha));
                    q[2] = hb;
                    q[3] = mx_bf16_rne(xb - mx_bf16_to_f32(hb));
                }
        }
    }
}

}  // namespace lr
