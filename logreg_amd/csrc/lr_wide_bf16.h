// lr_wide_bf16.h -- wide models (32 < p <= 128) on the 16-bit matrix pipe: the partial kernel with fp32-EXACT inputs (trajectory end points,
// every evaluation of RWMH / MALA / UL and of precision = full) described first, and the INTERIOR-step kernels of HMC's default policy
// (chain-split, row-split, one-launch trajectory kernels) that share its operand layout with one-piece images -- rows, beta and sigmoid
// weights in IEEE half precision where the design fits f16, else bf16 rows x two bf16 pieces of beta (beta_operands / mfma16 below).
//
// v_mfma_f32_16x16x32_bf16 runs at 16x the fp32-MFMA rate on a pipe of its own.  To keep fp32
// numerics every fp32 operand is split by truncation into three bf16 pieces, x = h + m + l (8 + 8 + 8
// significand bits: the split is EXACT), and a product a.b is formed from the six piece products of
// weight >= 2^-24 (hh, hm, mh, hl, mm, lh; each exact in fp32, accumulated in fp32 by the MFMA):
// the result differs from the fp32 product by the three dropped terms (relative <= 2^-23), i.e. it
// is in the fp32 rounding class.  Per 32 rows x 16 chains: 96 bf16 MFMAs (17 cycles) instead of
// 128 fp32 MFMAs (32-44 cycles).
//
// Workgroup = 4 waves x 16 chains, row slice walked in blocks of 32 rows (two 16-row eta tiles
// T0, T1).  Lane l = (c, kg): chain c = l & 15, kg = l >> 4; the lane OWNS coordinates
// {32 m + 8 kg + i : m < P/32, i < 8}.
//   eta tile (16 rows x 16 chains), K = 32 per MFMA = coordinates 32 m + 8 kg + i:
//       A (lane (row, kg)) = 8 consecutive coordinates of an X piece      LA[q][T][m][kg][row][8]
//       B (lane (c,   kg)) = the same 8 coordinates of a beta piece (registers)
//       D (lane (c, kg), reg r) = eta[row 4 kg + r][chain c]
//   w = sigma(-eta) on the 8 accumulator values of T0, T1; split into 3 pieces; K = 32 rows per
//   gradient MFMA with K-slot 8 kg + i <-> (i < 4: T0 row 4 kg + i; i >= 4: T1 row 4 kg + i - 4),
//   so the B operand is built from the lane's OWN eta outputs (no data movement):
//       B (lane (c, kg))  = 8 w pieces                       (registers)
//       A (lane (m', kg)) = the 8 rows of slot group kg, coordinate mu(mb', m')
//       M-block mb' = (m, h): slot m' = 4 kg' + r' <-> coordinate 32 m + 8 kg' + 4 h + r'
//       D (lane (c, kg), reg r) = gradient of coordinate 32 m + 8 kg + 4 h + r = one the lane owns.
//   The gradient A operand is the TRANSPOSE of what the eta operand holds (4 rows of one coordinate
//   instead of 4 coordinates of one row); gfx950's ds_read_b64_tr_b16 does exactly that transposition
//   inside a 16-lane group (source lane (ri, ci) supplies the 8-byte chunk [row ri][4 coordinates of
//   kg' = ci]; result lane m' = 4 ci + r' receives [rows 0..3][coordinate r' of the chunk]), so ONE
//   LDS image serves both products.  A (kg', row, half) swizzle keeps the 16 lanes of every
//   ds_read_b128 service group on distinct 4-bank slots AND the 32 lanes of every transposing b64 read
//   on distinct 2-bank slots:  row -> (row + 8 (kg' >> 1)) & 15, half -> half ^ (kg' & 1)
//   (odd-kg lanes see their two 4-coordinate halves swapped in the eta read; their beta operands are
//   built with the same swap, once).
// The split of the rows is done ONCE at model creation (wide_bf16_prepare): HBM holds, per 32-row
// block, exactly the 24 KB LDS image (P = 128), so staging is a contiguous copy.  (A first version
// split the fp32 rows while staging them: 48 ds_write_b16 + ~300 VALU ops per thread per block cost
// as much as the MFMAs saved; a second kept a separate transposed image for the gradient: twice the
// staging traffic and LDS.)
#pragma once
#include <cstring>

#include "lr_mfma.h"
#include "lr_tall.h"

namespace lr {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

// ds_read_b64_tr_b16: see the header comment
__device__ __forceinline__ u32x2 lds_read_tr16(const uint16_t* p) {
    return __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(p)));
}

// truncation split x = h + m + l; returns the pieces as fp32 values whose low 16 bits are zero
__device__ __forceinline__ void split3(float x, float& h, float& m, float& l) {
    h = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, x) & 0xFFFF0000u);
    const float r1 = x - h;  // exact
    m = __builtin_bit_cast(float, __builtin_bit_cast(uint32_t, r1) & 0xFFFF0000u);
    l = r1 - m;              // exact, <= 8 significant bits
}
// pack the bf16 (high) halves of two fp32 values: low half <- a, high half <- b
__device__ __forceinline__ uint32_t pack_hi(float a, float b) {
    return __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, b), __builtin_bit_cast(uint32_t, a), 0x07060302u);
}
__device__ __forceinline__ bf16x8 as_bf16x8(const u32x4& v) { return __builtin_bit_cast(bf16x8, v); }

template <int P> struct WideBf16Geom {
    static constexpr int M32 = P / 32;    // coordinate chunks of 32 = eta MFMA K-chunks
    static constexpr int MBP = P / 16;    // gradient M-blocks (m, h)
    static constexpr int TILE = 4 * 16 * 8;         // bf16 elements of one (piece, tile, chunk): [kg'][row][8]
    static constexpr int BUF = 3 * 2 * M32 * TILE;  // bf16 elements per 32-row block image
    // element offset of (kg', row, i) inside a tile (the swizzle of the header comment)
    static constexpr int elem(int kgp, int row, int i) {
        return kgp * 128 + ((row + 8 * (kgp >> 1)) & 15) * 8 + 4 * ((i >> 2) ^ (kgp & 1)) + (i & 3);
    }
    static constexpr int tile(int q, int T, int m) { return ((q * 2 + T) * M32 + m) * TILE; }
    // single-piece (round-to-nearest bf16) image used by the interior-step kernel: [tile T][chunk m][kg'][row][8]
    static constexpr int BUF1 = 2 * M32 * TILE;
    static constexpr int tile1(int T, int m) { return (T * M32 + m) * TILE; }
};

typedef float f32x2w __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
// two fp32 -> packed bf16 pair, round to nearest even (v_cvt_pk_bf16_f32): low half <- a, high half <- b
__device__ __forceinline__ uint32_t pack_rne(float a, float b) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2w{a, b}, bf16x2));
}
// ... and the IEEE half-precision (f16: 11 significant bits, 2^-14 .. 65504) forms of the trajectory kernels' one-piece interior:
// v_cvt_pk_f16_f32 (round to nearest even), v_mfma_f32_16x16x32_f16 -- the same rate and operand layout as the bf16 instruction
typedef _Float16 f16x2w __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8w __attribute__((ext_vector_type(8)));
__device__ __forceinline__ uint32_t pack_f16(float a, float b) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2w{a, b}, f16x2w));
}
// (a position times log2 e beyond the f16 range saturates instead of becoming an infinity: the force stays finite and deterministic)
__device__ __forceinline__ float clamp_f16(float x) { return __builtin_amdgcn_fmed3f(x, -65504.0f, 65504.0f); }
template <bool F16> __device__ __forceinline__ uint32_t pack16(float a, float b) { return F16 ? pack_f16(a, b) : pack_rne(a, b); }
template <bool F16> __device__ __forceinline__ f32x4 mfma16(const u32x4& x, const u32x4& y, const f32x4& acc) {
    if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8w, x), __builtin_bit_cast(f16x8w, y), acc, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(x), as_bf16x8(y), acc, 0, 0, 0);
}
// The lane's eight coordinates 32 m + 8 kg + i of its chain's position, times log2(e), as the B operands of the eta MFMAs: two
// round-to-nearest bf16 pieces (hi, lo) -- or ONE f16 piece (F16: NB = 1).  Odd kg: halves swapped, as the eta read of the rows delivers them.
template <bool F16> __device__ __forceinline__ void beta_operands(const float (&x)[8], int kg, u32x4 (&out)[F16 ? 1 : 2]) {
    uint32_t hi[4], lo[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const float x0 = x[2 * i] * ExpScale<float>::k, x1 = x[2 * i + 1] * ExpScale<float>::k;
        if constexpr (F16) {
            hi[i] = pack_f16(clamp_f16(x0), clamp_f16(x1));
        } else {
            hi[i] = pack_rne(x0, x1);
            const float h0 = __builtin_bit_cast(float, hi[i] << 16), h1 = __builtin_bit_cast(float, hi[i] & 0xFFFF0000u);
            lo[i] = pack_rne(x0 - h0, x1 - h1);
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int j = (kg & 1) ? (i ^ 2) : i;
        out[0][i] = hi[j];
        if constexpr (!F16) out[1][i] = lo[j];
    }
}

// NW waves per workgroup (4 or 8): every wave owns 16 chains, all share the staged 32-row block.
// 8 waves halve the staging traffic per chain (191 vs 140 TF at 8192 chains); 4 waves give more
// workgroups when there are few chains (80 vs 73 TF at 1024 chains).
template <int P, bool VALUE, int NW>
__global__ void __launch_bounds__((64 * NW)) k_wide_partial_bf16(TallArgs<float, P> a) {
    using G = WideBf16Geom<P>;
    constexpr int NT = 64 * NW, CPB = 16 * NW;  // threads, chains per block
    // [buffer][piece q][tile T][chunk m][kg'][row][8]  (bf16), reused as the fp32 output tile
    constexpr int SMEM = 2 * G::BUF > CPB * (P + 4) * 2 ? 2 * G::BUF : CPB * (P + 4) * 2;
    __shared__ __attribute__((aligned(16))) uint16_t smem[SMEM];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c = lane & 15, kg = lane >> 4;
    int64_t chain = (int64_t)blockIdx.x * CPB + 16 * wave + c;
    const bool live = chain < a.C;
    if (!live) chain = a.C - 1;
    const int rs = blockIdx.y;
    const int64_t s0 = (int64_t)rs * a.slice_len, s1 = s0 + a.slice_len < a.n ? s0 + a.slice_len : a.n;
    const int64_t nblk = s1 > s0 ? (s1 - s0 + 31) / 32 : 0;

    // beta pieces of the lane's coordinates 32 m + 8 kg + i, pre-scaled by log2(e): B operands
    u32x4 bq[G::M32][3];
#pragma unroll
    for (int m = 0; m < G::M32; ++m) {
        float h[8], md[8], lo[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) split3(a.q1[chain * P + 32 * m + 8 * kg + i] * ExpScale<float>::k, h[i], md[i], lo[i]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int j = (kg & 1) ? (i ^ 2) : i;  // odd kg: halves swapped, as the eta read delivers them
            bq[m][0][i] = pack_hi(h[2 * j], h[2 * j + 1]);
            bq[m][1][i] = pack_hi(md[2 * j], md[2 * j + 1]);
            bq[m][2][i] = pack_hi(lo[2 * j], lo[2 * j + 1]);
        }
    }
    // per-lane offsets: eta read (lane = (row c, kg)); transposing read (lane = (kg, ri, ci))
    const int eta_off = G::elem(kg, c, 0) & ~7;
    const int ri = (lane & 15) >> 2, ci = lane & 3;
    const int tr_off[2] = {G::elem(ci, 4 * kg + ri, 0), G::elem(ci, 4 * kg + ri, 4)};
    f32x4 gacc[G::MBP];
#pragma unroll
    for (int mb = 0; mb < G::MBP; ++mb) gacc[mb] = f32x4{0, 0, 0, 0};
    double vsum = 0.0;

    // staging: the block image is contiguous in HBM: 256 threads x 16-byte chunks
    constexpr int NCH = G::BUF * 2 / 16;           // 16-byte chunks per block image
    constexpr int CHUNKS = (NCH + NT - 1) / NT;    // per thread
    u32x4 stage[CHUNKS];
    const int64_t blk0 = s0 / 32;  // slices are multiples of 32 rows
    auto fetch = [&](int64_t b) {
        const u32x4* src = reinterpret_cast<const u32x4*>(a.xblk + (blk0 + b) * (int64_t)G::BUF);
#pragma unroll
        for (int i = 0; i < CHUNKS; ++i)
            if (NT * (i + 1) <= NCH || tid + NT * i < NCH) stage[i] = src[tid + NT * i];
    };
    auto deposit = [&](int buf) {
        u32x4* dst = reinterpret_cast<u32x4*>(smem + buf * G::BUF);
#pragma unroll
        for (int i = 0; i < CHUNKS; ++i)
            if (NT * (i + 1) <= NCH || tid + NT * i < NCH) dst[tid + NT * i] = stage[i];
    };
    // the six piece products of weight >= 2^-24: (x piece, other piece)
    constexpr int QX[6] = {0, 0, 1, 0, 1, 2}, QO[6] = {0, 1, 0, 2, 1, 0};

    if (nblk > 0) {
        fetch(0);
        deposit(0);
    }
    __syncthreads();
    for (int64_t b = 0; b < nblk; ++b) {
        const int buf = (int)(b & 1);
        const uint16_t* base = smem + buf * G::BUF;
        if (b + 1 < nblk) fetch(b + 1);
        // ---- eta for the two tiles
        float w8[8];
#pragma unroll
        for (int T = 0; T < 2; ++T) {
            f32x4 e0 = {0, 0, 0, 0}, e1 = {0, 0, 0, 0};
#pragma unroll
            for (int m = 0; m < G::M32; ++m) {
                u32x4 xa[3];
#pragma unroll
                for (int q = 0; q < 3; ++q)
                    xa[q] = *reinterpret_cast<const u32x4*>(base + G::tile(q, T, m) + eta_off);
#pragma unroll
                for (int t = 0; t < 6; t += 2) {
                    e0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(xa[QX[t]]), as_bf16x8(bq[m][QO[t]]), e0, 0, 0, 0);
                    e1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(xa[QX[t + 1]]), as_bf16x8(bq[m][QO[t + 1]]), e1, 0, 0, 0);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float ts = e0[r] + e1[r];
                w8[4 * T + r] = fast_rcp(1.0f + __builtin_amdgcn_exp2f(ts));
                if constexpr (VALUE) {
                    const int64_t row = s0 + 32 * b + 16 * T + 4 * kg + r;
                    const float ats = ts < 0.0f ? -ts : ts;
                    const float lv = (ts < 0.0f ? ts * ExpScale<float>::inv : 0.0f) - log1p_unit(__builtin_amdgcn_exp2f(-ats));
                    if (row < s1) vsum += (double)lv;
                }
            }
        }
        // ---- w pieces: the B operand of the gradient MFMAs (K-slot 8 kg + i <-> w8[i])
        u32x4 wq[3];
        {
            float h[8], md[8], lo[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) split3(w8[i], h[i], md[i], lo[i]);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                wq[0][i] = pack_hi(h[2 * i], h[2 * i + 1]);
                wq[1][i] = pack_hi(md[2 * i], md[2 * i + 1]);
                wq[2][i] = pack_hi(lo[2 * i], lo[2 * i + 1]);
            }
        }
        // ---- grad += Xs^T . W
#pragma unroll
        for (int mb = 0; mb < G::MBP; ++mb) {
            u32x4 xg[3];
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const u32x2 t0 = lds_read_tr16(base + G::tile(q, 0, mb >> 1) + tr_off[mb & 1]);
                const u32x2 t1 = lds_read_tr16(base + G::tile(q, 1, mb >> 1) + tr_off[mb & 1]);
                xg[q] = u32x4{t0[0], t0[1], t1[0], t1[1]};
            }
#pragma unroll
            for (int t = 0; t < 6; ++t)
                gacc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(xg[QX[t]]), as_bf16x8(wq[QO[t]]), gacc[mb], 0, 0, 0);
        }
        if (b + 1 < nblk) deposit(buf ^ 1);
        __syncthreads();
    }
    // epilogue: gradient register (mb' = 2m + h, r) of lane (c, kg) is coordinate 32m + 8kg + 4h + r.
    // The LDS tile rows are P + 4 floats apart: with a stride of P (= 0 mod 64 banks) the 16 chains of a wave
    // would all write the same banks (16-way conflict on every ds_write_b128).
    constexpr int OT = P + 4;
    static_assert(CPB * OT * 2 <= SMEM, "padded output tile must fit");
    float* otile = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int mb = 0; mb < G::MBP; ++mb)
        *reinterpret_cast<f32x4*>(otile + (16 * wave + c) * OT + 32 * (mb >> 1) + 8 * kg + 4 * (mb & 1)) = gacc[mb];
    __syncthreads();
    {
        const int64_t chain0 = (int64_t)blockIdx.x * CPB;
        const int64_t nlive = a.C - chain0 < CPB ? a.C - chain0 : CPB;
        f32x4* dst = reinterpret_cast<f32x4*>(a.part_g + ((int64_t)rs * a.C + chain0) * P);
        for (int i = tid; i < (int)(nlive * P / 4); i += NT)  // consumed once, by the update kernel: non-temporal
            __builtin_nontemporal_store(*reinterpret_cast<const f32x4*>(otile + (i / (P / 4)) * OT + (i % (P / 4)) * 4), &dst[i]);
    }
    if constexpr (VALUE) {
        const double tot = ksum(vsum);
        if (live && kg == 0) a.part_v[(int64_t)rs * a.C + chain] = tot;
    }
}

// ---------------------------------------------------------------------------------------------------------------
// INTERIOR-STEP kernel (gradient only): the same GEMMs with  X in ONE bf16 piece (round-to-nearest image, 8 KB per
// 32-row block at P = 128 instead of 24 KB),  beta in TWO pieces (hi + lo: 16 significand bits),  w = sigma(-eta)
// in ONE piece: 24 MFMAs per block instead of 96, a third of the LDS traffic, no three-way splits on the VALU.
//
// Why this is legitimate.  HMC's leapfrog map is volume-preserving and reversible for ANY force that is a
// deterministic function of position (each kick/drift is a shear), so the interior gradient evaluations may be
// approximate without touching the exactness of the sampler: the Metropolis test compares the EXACT
// (three-piece, fp32-class) log-posterior at the two trajectory end points, evaluated by k_wide_partial_bf16, and
// the end-point half-kicks use the exact gradient as well.  What the approximation can cost is acceptance rate;
// with X rounded to bf16 the interior force is (to 2^-16) the exact gradient of the posterior of a data set
// perturbed by 2^-9 relative -- a smooth Hamiltonian next door -- and the acceptance rate at BASELINE config 5
// (n = 4096, p = 128, eps = 0.02, L = 50) moves from 0.757 to 0.75x (tests/test_gpu_fullsize.py measures it).
// The reference has no counterpart (fit-np-hmc.py:65-87 computes every glp in float64).
// Staging: the 8 KB block images go from L2/HBM straight into LDS with the LDS-DMA load (global_load_lds_dwordx4:
// wave-uniform LDS base + lane x 16 B, no staging VGPRs, no ds_write pass), a PASS of kPassBytes (4 blocks at
// P = 128) at a time into one of two buffers: the DMA of pass g + 1 runs under the arithmetic of pass g, and there
// is ONE barrier per pass instead of one per 32-row block (config 5's slices of 8 blocks: 2 barriers, was 8 --
// with the per-block barrier the loop paid the global-load latency of every block: 11.3 us per launch).
// S = double: the same kernel on a FLOAT64 model (the default precision policy there: lr_wide_f64.h runs the end points) -- the
// position is read as float64 and rounded, the slice partials stay float32 (TallArgs::part_f32) for k_tall_update<double> to sum.
template <int P, int NW, typename S = float, bool F16 = false>
__global__ void __launch_bounds__((64 * NW)) k_wide_partial_bf16i(TallArgs<S, P> a) {
    using G = WideBf16Geom<P>;
    constexpr int NB = F16 ? 1 : 2;  // (F16: the one-piece half-precision interior, as in the trajectory kernels)
    const uint16_t* const image = F16 ? a.xblk1h : a.xblk1;
    constexpr int NT = 64 * NW, CPB = 16 * NW;
    constexpr int kPassBytes = 32768, BLK_BYTES = G::BUF1 * 2;
    constexpr int SB = kPassBytes / BLK_BYTES;  // blocks per pass
    constexpr int PASS_EL = kPassBytes / 2;     // bf16 elements per buffer
    constexpr int SMEM = 2 * PASS_EL > CPB * (P + 4) * 2 ? 2 * PASS_EL : CPB * (P + 4) * 2;
    __shared__ __attribute__((aligned(1024))) uint16_t smem[SMEM];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, kg = lane >> 4;
    int64_t chain = (int64_t)blockIdx.x * CPB + 16 * wave + c;
    const bool live = chain < a.C;
    if (!live) chain = a.C - 1;
    const int rs = blockIdx.y;
    const int64_t s0 = (int64_t)rs * a.slice_len, s1 = s0 + a.slice_len < a.n ? s0 + a.slice_len : a.n;
    const int64_t nblk = s1 > s0 ? (s1 - s0 + 31) / 32 : 0;
    const int64_t npass = (nblk + SB - 1) / SB;
    const int64_t blk0 = s0 / 32;

    typedef __attribute__((address_space(1))) const void* gptr_t;
    typedef __attribute__((address_space(3))) void* lptr_t;
    auto issue = [&](int64_t g) {  // pass g -> buffer g & 1; 1 KB per wave-instruction, chunks dealt round-robin to the waves
        const int64_t b0 = g * SB;
        const int nb = (int)(nblk - b0 < SB ? nblk - b0 : SB);
        const char* src = reinterpret_cast<const char*>(image + (blk0 + b0) * (int64_t)G::BUF1);
        char* dst = reinterpret_cast<char*>(smem) + (g & 1) * kPassBytes;
        const int nchunk = nb * (BLK_BYTES / 1024);
        for (int ch = wave; ch < nchunk; ch += NW)
            __builtin_amdgcn_global_load_lds((gptr_t)(src + ch * 1024 + lane * 16), (lptr_t)(dst + ch * 1024), 16, 0, 0);
    };
    if (npass > 0) issue(0);  // in flight under the beta split below

    // beta = hi + lo (two round-to-nearest bf16 pieces) of the lane's coordinates 32 m + 8 kg + i, times log2(e)
    u32x4 bq[G::M32][NB];
#pragma unroll
    for (int m = 0; m < G::M32; ++m) {
        float x[8];
        if constexpr (sizeof(S) == 4) {
            const f32x4* src = reinterpret_cast<const f32x4*>(a.q1 + chain * P + 32 * m + 8 * kg);
            const f32x4 v0 = src[0], v1 = src[1];
#pragma unroll
            for (int i = 0; i < 4; ++i) x[i] = v0[i], x[4 + i] = v1[i];
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = (float)a.q1[chain * P + 32 * m + 8 * kg + i];
        }
        beta_operands<F16>(x, kg, bq[m]);
    }
    const int eta_off = G::elem(kg, c, 0) & ~7;
    const int ri = (lane & 15) >> 2, ci = lane & 3;
    const int tr_off[2] = {G::elem(ci, 4 * kg + ri, 0), G::elem(ci, 4 * kg + ri, 4)};
    f32x4 gacc[G::MBP];
#pragma unroll
    for (int mb = 0; mb < G::MBP; ++mb) gacc[mb] = f32x4{0, 0, 0, 0};

    for (int64_t g = 0; g < npass; ++g) {
        // pass g has landed (every wave waits for its own DMA, the barrier for everybody's), and all waves are done
        // reading the other buffer (pass g - 1), which the DMA of pass g + 1 may now overwrite
        // (builtin, not inline asm: a kernel containing inline asm makes the register allocator assume AGPRs may be
        // needed, selects the AGPR form of every MFMA and then shuffles the 32 gradient accumulators through VGPRs
        // on each trip -- 80 v_accvgpr moves per block)
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), lgkmcnt / expcnt untouched
        __syncthreads();
        if (g + 1 < npass) issue(g + 1);
        const int nb = (int)(nblk - g * SB < SB ? nblk - g * SB : SB);
        for (int bi = 0; bi < nb; ++bi) {
            const uint16_t* base = smem + (g & 1) * PASS_EL + bi * G::BUF1;
            // ---- eta for the two tiles: X (1 piece) x beta (hi, lo)
            uint32_t wq[4];
#pragma unroll
            for (int T = 0; T < 2; ++T) {
                f32x4 e0 = {0, 0, 0, 0}, e1 = {0, 0, 0, 0};
#pragma unroll
                for (int m = 0; m < G::M32; ++m) {
                    const u32x4 xa = *reinterpret_cast<const u32x4*>(base + G::tile1(T, m) + eta_off);
                    e0 = mfma16<F16>(xa, bq[m][0], e0);
                    if constexpr (!F16) e1 = mfma16<F16>(xa, bq[m][1], e1);
                }
                float w[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) w[r] = fast_rcp(1.0f + __builtin_amdgcn_exp2f(F16 ? e0[r] : e0[r] + e1[r]));
                wq[2 * T] = pack16<F16>(w[0], w[1]);  // K-slot 8 kg + 4 T + r <-> row 4 kg + r of tile T
                wq[2 * T + 1] = pack16<F16>(w[2], w[3]);
            }
            const u32x4 wv = {wq[0], wq[1], wq[2], wq[3]};
            // ---- grad += Xs^T . W
#pragma unroll
            for (int mb = 0; mb < G::MBP; ++mb) {
                const u32x2 t0 = lds_read_tr16(base + G::tile1(0, mb >> 1) + tr_off[mb & 1]);
                const u32x2 t1 = lds_read_tr16(base + G::tile1(1, mb >> 1) + tr_off[mb & 1]);
                const u32x4 xg = {t0[0], t0[1], t1[0], t1[1]};
                gacc[mb] = mfma16<F16>(xg, wv, gacc[mb]);
            }
        }
    }
    __syncthreads();  // the output tile aliases the staging buffers
    constexpr int OT = P + 4;
    static_assert(CPB * OT * 2 <= SMEM, "padded output tile must fit");
    float* otile = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int mb = 0; mb < G::MBP; ++mb)
        *reinterpret_cast<f32x4*>(otile + (16 * wave + c) * OT + 32 * (mb >> 1) + 8 * kg + 4 * (mb & 1)) = gacc[mb];
    __syncthreads();
    {
        const int64_t chain0 = (int64_t)blockIdx.x * CPB;
        const int64_t nlive = a.C - chain0 < CPB ? a.C - chain0 : CPB;
        // (float64 models: float32 partials all the same -- TallArgs::part_f32 tells k_tall_update<double>)
        f32x4* dst = reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.part_g) + ((int64_t)rs * a.C + chain0) * P);
        for (int i = tid; i < (int)(nlive * P / 4); i += NT)
            __builtin_nontemporal_store(*reinterpret_cast<const f32x4*>(otile + (i / (P / 4)) * OT + (i % (P / 4)) * 4), &dst[i]);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// ROW-SPLIT form of the interior-step kernel, for few chains (ceil(C / 16) <= CUs, i.e. C <= 4096 on MI355X).
// Measured on config 5 (1024 chains, 16 row slices x 16 chain blocks): of the 15 us per evaluation, 4.7 us were
// the 8 MB of slice partials leaving the chip (16 slices x 1024 chains x 128 coordinates x 4 B) and another ~2 us
// the update kernel reading them back -- more than the 3.9 us of matrix work.  Here a workgroup is ONE chain tile
// (16 chains) whose 4 waves take different ROWS of the slice and add their gradients through LDS before anything
// leaves the CU: 4 slices instead of 16, 2 MB of partials instead of 8.  The waves share nothing while they
// run, so the row loop has no barrier at all: every wave streams its own 8 KB block images L2 -> LDS with the
// LDS-DMA load into a private ring of 4 buffers (3 blocks in flight ahead of the one being consumed) and waits
// with counted vmcnt.  (The DMA is issued from inline asm: told about it, the compiler guards every
// ds_read_b64_tr_b16 with vmcnt(0) -- it cannot see that the transposing reads touch another ring slot -- and
// the prefetch would never run ahead.)  Price: the tile's rows are not shared between waves any more, 4x the
// L2 -> LDS traffic (64 MB per evaluation), which the DMA ring hides under the MFMAs.
template <int BYTES> __device__ __forceinline__ void wait_vm_blocks(int younger) {
    // wait until at most `younger` block images (BYTES / 1024 DMA instructions each) are still in flight
    constexpr int per = BYTES / 1024;
    static_assert(3 * per <= 63, "vmcnt is a 6-bit field");
    if (younger >= 3) __builtin_amdgcn_s_waitcnt(0x0F70 | ((3 * per) & 15) | (((3 * per) >> 4) << 14));
    else if (younger == 2) __builtin_amdgcn_s_waitcnt(0x0F70 | ((2 * per) & 15) | (((2 * per) >> 4) << 14));
    else if (younger == 1) __builtin_amdgcn_s_waitcnt(0x0F70 | (per & 15) | ((per >> 4) << 14));
    else __builtin_amdgcn_s_waitcnt(0x0F70);
}

// S = double: a FLOAT64 model's interior steps -- position, momentum and the fused update (kick, drift) are float64, the position
// enters the GEMM rounded to two bf16 pieces, the slice partials are float32 as for float32 models (TallArgs::part_f32).
template <int P, int NW, typename S = float, bool F16 = false>
__global__ void __launch_bounds__((64 * NW)) __attribute__((amdgpu_waves_per_eu(NW / 4, NW / 4))) k_wide_partial_bf16r(TallArgs<S, P> a) {
    constexpr bool kFusable = sizeof(S) == 4;
    using G = WideBf16Geom<P>;
    constexpr int NB = F16 ? 1 : 2;  // (F16: the one-piece half-precision interior, as in the trajectory kernels)
    const uint16_t* const image = F16 ? a.xblk1h : a.xblk1;
    // NW = 4: one wave per SIMD, ring of 4 block images per wave; NW = 8: two waves per SIMD (each hides the other's
    // LDS -> MFMA latency, which is what a block costs at one wave per SIMD), ring of 2
    constexpr int BLK_BYTES = G::BUF1 * 2, NBUF = NW == 4 ? 4 : 2, RING_BYTES = NBUF * BLK_BYTES;
    constexpr int OT = P + 4;
    static_assert(16 * OT * 4 <= RING_BYTES, "a wave's output tile lives in its own ring");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NW * RING_BYTES];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, kg = lane >> 4;
    const int64_t chain0 = (int64_t)blockIdx.x * 16;
    int64_t chain = chain0 + c;
    if (chain >= a.C) chain = a.C - 1;
    // (grid (tile, slice): the slice workgroups of a tile share an XCD, so the partials they exchange from launch to launch stay in its L2.
    //  Dealing the SLICES to the XCDs instead -- each L2 then streams one slice of the image, not all -- was measured in round 5 and lost:
    //  config 5 at 1024 chains 8.6 -> 10.9 us per evaluation, FETCH_SIZE 5.66 -> 7.12 MB per launch; profiles/r5_cfg5_1024.txt)
    const int rs = blockIdx.y;
    const int64_t s0 = (int64_t)rs * a.slice_len_i, s1 = s0 + a.slice_len_i < a.n ? s0 + a.slice_len_i : a.n;
    const int nblk_slice = s1 > s0 ? (int)((s1 - s0 + 31) / 32) : 0;
    const int per_wave = (nblk_slice + NW - 1) / NW;
    const int wb0 = wave * per_wave;
    const int wnb = nblk_slice - wb0 < 0 ? 0 : (nblk_slice - wb0 < per_wave ? nblk_slice - wb0 : per_wave);
    const int64_t blk0 = s0 / 32 + wb0;
    unsigned char* ring = smem + wave * RING_BYTES;
    const uint32_t ring_lds = (uint32_t)(uintptr_t)ring;  // LDS byte address (generic -> local keeps the low 32 bits)
    LR_STAMP(a, 0);
#ifdef LR_STAMPS
    if (a.stamps && lane == 0) LR_STAMP_AT(a, 13) = __builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 15;  // HW_REG_XCC_ID, bits 3:0
#endif

    auto issue = [&](int b) {  // block b of this wave -> ring slot b % NBUF
        // ONE M0 set-up per block: the instruction offset (13-bit signed) is added to the global AND to the LDS address, so the
        // block's 1 KB pieces are the offsets -4096 ... +3072 around its middle.  (A set-up per piece -- 4 scalar instructions
        // each -- made the 8 pieces of a block cost ~230 cycles of the wave's issue: 1.2 of the loop's 4.6 us at config 5.)
        static_assert(BLK_BYTES == 8192 || BLK_BYTES == 4096, "pieces addressed around the middle of the block");
        const unsigned char* src = reinterpret_cast<const unsigned char*>(image + (blk0 + b) * (int64_t)G::BUF1) + lane * 16 + BLK_BYTES / 2;
        const uint32_t dst = ring_lds + (uint32_t)((b & (NBUF - 1)) * BLK_BYTES) + BLK_BYTES / 2;
        uint32_t keep;
        if constexpr (BLK_BYTES == 8192)
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                         "global_load_lds_dwordx4 %1, off offset:-4096\n\tglobal_load_lds_dwordx4 %1, off offset:-3072\n\t"
                         "global_load_lds_dwordx4 %1, off offset:-2048\n\tglobal_load_lds_dwordx4 %1, off offset:-1024\n\t"
                         "global_load_lds_dwordx4 %1, off\n\tglobal_load_lds_dwordx4 %1, off offset:1024\n\t"
                         "global_load_lds_dwordx4 %1, off offset:2048\n\tglobal_load_lds_dwordx4 %1, off offset:3072\n\t"
                         "s_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
        else
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                         "global_load_lds_dwordx4 %1, off offset:-2048\n\tglobal_load_lds_dwordx4 %1, off offset:-1024\n\t"
                         "global_load_lds_dwordx4 %1, off\n\tglobal_load_lds_dwordx4 %1, off offset:1024\n\t"
                         "s_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
    };
    // (the first blocks are requested below, BEHIND the prologue's own loads: issuing the DMA first put its 60-185 cycles per
    //  1 KB piece in front of the one memory round trip the prologue waits for)

    // fused previous step (k_tall_update's PH_MID done here):  g1 = sum of the slice partials (slice order, fp64)
    // - q ivar;  p += eps g1;  q += (eps / m) p.  The four waves hold the SAME 16 chains, so the work is dealt out:
    // wave w does the 32-coordinate chunk m = w (every load issued before the first use: one memory round trip),
    // leaves the new position in LDS for the others and -- slice 0 only -- in the state buffers for the next launch.
    // Identical arithmetic in the RS_i workgroups of a tile.
    constexpr int kFuseSlices = 4;  // the host fuses only when RS_i <= kFuseSlices
    // the new position, laid out for the b128 accesses of lanes (c, kg): coordinate j of chain c at [j / 8][(j / 4) & 1][c][j & 3] -- the
    // 16 lanes of a service group (8 of one kg, 8 of the next: 512 bytes apart) then cover 16 distinct 16-byte bank slots.  As
    // [16][P] (round 2) the chains were 512 bytes = 0 banks apart: every access a 16-way conflict, 41 % of the kernel's LDS cycles.
    __shared__ __attribute__((aligned(16))) float qnew[P / 8][2][16][4];
    const bool fused = a.fuse_mid != 0;
    if (fused) {
        if (wave < G::M32) {
            const int m = wave;
            const int64_t at = chain * P + 32 * m + 8 * kg;
            // (the slice partials are float32 whatever the state's type S; state and constants in S: 16-byte loads)
            auto load8 = [](const S* src, S (&v)[8]) {
                if constexpr (sizeof(S) == 4) {
                    const f32x4 v0 = reinterpret_cast<const f32x4*>(src)[0], v1 = reinterpret_cast<const f32x4*>(src)[1];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = v0[e], v[4 + e] = v1[e];
                } else {
                    typedef double f64x2 __attribute__((ext_vector_type(2)));
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const f64x2 t = reinterpret_cast<const f64x2*>(src)[e];
                        v[2 * e] = t[0], v[2 * e + 1] = t[1];
                    }
                }
            };
            auto store8 = [](S* dst, const S (&v)[8]) {
                if constexpr (sizeof(S) == 4) {
                    reinterpret_cast<f32x4*>(dst)[0] = f32x4{v[0], v[1], v[2], v[3]};
                    reinterpret_cast<f32x4*>(dst)[1] = f32x4{v[4], v[5], v[6], v[7]};
                } else {
                    typedef double f64x2 __attribute__((ext_vector_type(2)));
#pragma unroll
                    for (int e = 0; e < 4; ++e) reinterpret_cast<f64x2*>(dst)[e] = f64x2{v[2 * e], v[2 * e + 1]};
                }
            };
            const float* part_f = reinterpret_cast<const float*>(a.part_in);
            f32x4 pg[kFuseSlices][2];
            S vq[8], vp[8], vb[8], vi[8];
#ifdef LR_STAMPS
            if (a.stamps && lane == 0) LR_STAMP_AT(a, 10) = __builtin_amdgcn_s_memrealtime();
            asm volatile("" ::: "memory");
#endif
            // (timing experiments of a -DLR_STAMPS build, results knowingly wrong: LOGREG_DEBUG_EXP bit 2 reads ONE slice partial
            //  instead of RS_i, bit 3 none and no momentum either -- is the prologue fetch bound by volume or by latency?)
            const int nread = LR_DBG(a, 3) ? 0 : (LR_DBG(a, 2) ? 1 : a.RS_i);
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int r = 0; r < kFuseSlices; ++r) {
                    pg[r][h] = f32x4{0, 0, 0, 0};
                    if (r < nread) pg[r][h] = *reinterpret_cast<const f32x4*>(part_f + ((int64_t)r * a.C) * P + at + 4 * h);
                }
            load8(a.q1_in + at, vq);
            if (LR_DBG(a, 3)) {
#pragma unroll
                for (int e = 0; e < 8; ++e) vp[e] = vq[e];
            } else {
                load8(a.pm_in + at, vp);
            }
            load8(a.cvec + 32 * m + 8 * kg, vb);
            load8(a.cvec + P + 32 * m + 8 * kg, vi);
#ifdef LR_STAMPS
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (a.stamps && lane == 0) LR_STAMP_AT(a, 11) = __builtin_amdgcn_s_memrealtime();
            asm volatile("" ::: "memory");
#endif
#pragma unroll
            for (int b = 0; b < NBUF - 1; ++b)
                if (b < wnb) issue(b);
#ifdef LR_STAMPS
            if (a.stamps && lane == 0) LR_STAMP_AT(a, 12) = __builtin_amdgcn_s_memrealtime();
#endif
            S xn[8], pn[8];
            f32x4 xf[2];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                double gs = 0.0;
#pragma unroll
                for (int r = 0; r < kFuseSlices; ++r)
                    if (r < a.RS_i) gs += (double)pg[r][e >> 2][e & 3];
                const S g1 = (S)gs - vq[e] * vi[e];
                const S pmn = fma_t(a.step, g1, vp[e]);
                pn[e] = pmn;
                xn[e] = fma_t(vb[e], pmn, vq[e]);
                xf[e >> 2][e & 3] = (float)xn[e];
            }
            *reinterpret_cast<f32x4*>(&qnew[4 * m + kg][0][c][0]) = xf[0];
            *reinterpret_cast<f32x4*>(&qnew[4 * m + kg][1][c][0]) = xf[1];
            if (rs == 0 && chain0 + c < a.C) {
                store8(a.q1 + at, xn);
                store8(a.pm + at, pn);
            }
        } else {
#pragma unroll
            for (int b = 0; b < NBUF - 1; ++b)
                if (b < wnb) issue(b);
        }
        LR_STAMP(a, 1);
        __syncthreads();
    } else {
#pragma unroll
        for (int b = 0; b < NBUF - 1; ++b)
            if (b < wnb) issue(b);
    }
    LR_STAMP(a, 2);
    // beta = hi + lo (two round-to-nearest bf16 pieces) of the lane's coordinates 32 m + 8 kg + i, times log2(e)
    u32x4 bq[G::M32][NB];
#pragma unroll
    for (int m = 0; m < G::M32; ++m) {
        const int64_t at = chain * P + 32 * m + 8 * kg;
        float x[8];
        if (fused) {
            const f32x4 v0 = *reinterpret_cast<const f32x4*>(&qnew[4 * m + kg][0][c][0]);
            const f32x4 v1 = *reinterpret_cast<const f32x4*>(&qnew[4 * m + kg][1][c][0]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                x[e] = v0[e];
                x[4 + e] = v1[e];
            }
        } else if constexpr (kFusable) {
            const f32x4* src = reinterpret_cast<const f32x4*>(a.q1 + at);
            const f32x4 v0 = src[0], v1 = src[1];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                x[e] = v0[e];
                x[4 + e] = v1[e];
            }
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = (float)a.q1[at + e];
        }
        beta_operands<F16>(x, kg, bq[m]);
    }
    const int eta_off = G::elem(kg, c, 0) & ~7;
    const int ri = (lane & 15) >> 2, ci = lane & 3;
    const int tr_off[2] = {G::elem(ci, 4 * kg + ri, 0), G::elem(ci, 4 * kg + ri, 4)};
    f32x4 gacc[G::MBP];
#pragma unroll
    for (int mb = 0; mb < G::MBP; ++mb) gacc[mb] = f32x4{0, 0, 0, 0};

    LR_STAMP(a, 3);
    LR_STAMP_CLK(a, 8);
    for (int b = 0; b < wnb; ++b) {
        // slot (b - 1) % NBUF was read during the previous trip and those reads have returned (their MFMAs
        // issued): refill it with block b + NBUF - 1, then wait for block b with the younger ones still in flight
        if (b + NBUF - 1 < wnb) issue(b + NBUF - 1);
        const int last = b + NBUF - 1 < wnb ? b + NBUF - 1 : wnb - 1;
        wait_vm_blocks<BLK_BYTES>(last - b);
        const uint16_t* base = reinterpret_cast<const uint16_t*>(ring + (b & (NBUF - 1)) * BLK_BYTES);
        // every operand of the block is requested before the first MFMA (16 + 16 registers of eta operands, 32 of transposed
        // ones): left to a 128-register budget the compiler re-used one register quad for the eta reads and each MFMA pair
        // waited for its own LDS round trip
        u32x4 xa[2][G::M32];
#pragma unroll
        for (int T = 0; T < 2; ++T)
#pragma unroll
            for (int m = 0; m < G::M32; ++m) xa[T][m] = *reinterpret_cast<const u32x4*>(base + G::tile1(T, m) + eta_off);
        u32x2 xt[G::MBP][2];
#pragma unroll
        for (int mb = 0; mb < G::MBP; ++mb) {
            xt[mb][0] = lds_read_tr16(base + G::tile1(0, mb >> 1) + tr_off[mb & 1]);
            xt[mb][1] = lds_read_tr16(base + G::tile1(1, mb >> 1) + tr_off[mb & 1]);
        }
        __builtin_amdgcn_sched_barrier(0);  // (the scheduler otherwise sinks every read back to its use)
        uint32_t wq[4];
#pragma unroll
        for (int T = 0; T < 2; ++T) {
            f32x4 e0 = {0, 0, 0, 0}, e1 = {0, 0, 0, 0};
#pragma unroll
            for (int m = 0; m < G::M32; ++m) {
                e0 = mfma16<F16>(xa[T][m], bq[m][0], e0);
                if constexpr (!F16) e1 = mfma16<F16>(xa[T][m], bq[m][1], e1);
            }
            float w[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) w[r] = fast_rcp(1.0f + __builtin_amdgcn_exp2f(F16 ? e0[r] : e0[r] + e1[r]));
            wq[2 * T] = pack16<F16>(w[0], w[1]);
            wq[2 * T + 1] = pack16<F16>(w[2], w[3]);
        }
        const u32x4 wv = {wq[0], wq[1], wq[2], wq[3]};
#pragma unroll
        for (int mb = 0; mb < G::MBP; ++mb) {
            const u32x4 xg = {xt[mb][0][0], xt[mb][0][1], xt[mb][1][0], xt[mb][1][1]};
            gacc[mb] = mfma16<F16>(xg, wv, gacc[mb]);
        }
    }
    LR_STAMP_CLK(a, 9);
    LR_STAMP(a, 4);
    // the four waves' gradients of the tile, summed in wave order (deterministic) and written once
    float* otile = reinterpret_cast<float*>(ring);
#pragma unroll
    for (int mb = 0; mb < G::MBP; ++mb)
        *reinterpret_cast<f32x4*>(otile + c * OT + 32 * (mb >> 1) + 8 * kg + 4 * (mb & 1)) = gacc[mb];
    __syncthreads();
    LR_STAMP(a, 5);
    {
        const int64_t nlive = a.C - chain0 < 16 ? a.C - chain0 : 16;
        f32x4* dst = reinterpret_cast<f32x4*>(reinterpret_cast<float*>(a.part_g) + ((int64_t)rs * a.C + chain0) * P);  // (float32 whatever S)
        for (int i = tid; i < (int)(nlive * P / 4); i += 64 * NW) {
            const int off = (i / (P / 4)) * OT + (i % (P / 4)) * 4;
            f32x4 acc = *reinterpret_cast<const f32x4*>(smem + off * 4);
#pragma unroll
            for (int w = 1; w < NW; ++w) acc += *reinterpret_cast<const f32x4*>(smem + w * RING_BYTES + off * 4);  // wave order
#ifdef LR_WIDE_NT_PARTIALS
            __builtin_nontemporal_store(acc, &dst[i]);
#else
            dst[i] = acc;  // plain: the 4 slice workgroups of the tile that read it back next launch sit on this XCD
#endif
        }
    }
    LR_STAMP(a, 6);
}

// ---------------------------------------------------------------------------------------------------------------
// TRAJECTORY kernels: all L - 1 interior leapfrog steps of a workgroup's chain tiles in ONE launch, no row slicing across workgroups.
// Why (round 3, config 5 at 1024 chains on the row-split kernel above, 10 us per step): a third of the time is the fused prologue
// re-reading the tile's slice partials and state in each of its 4 slice workgroups, a fifth the reduction + stores, and the 4-block row
// loop per wave never leaves its start-up transient; swapping the partials between resident workgroups instead (tools/xchg_probe.hip)
// costs 3.2 us per step on one XCD, 9.5 us across XCDs -- no cheaper than the launch boundary.  So here a tile's gradient never leaves
// its CU: the 8 waves of the workgroup split ALL the rows of the design (each streams its own 32-row block images L2 -> LDS through a
// private DMA ring, exactly the row loop above), the wave partials meet in LDS, and the thread that owns (chain, four coordinates)
// finishes the step: g = sum over the waves in wave order - q ivar, p += eps g, q += (eps / m) p, the new position's beta operand to
// LDS for the next step.  The first blocks of the next step are on their way while the reduction runs.  One workgroup streams the
// whole single-piece image every step (n p 2 bytes from L2: 1 MB for config 5, >= 6.8 us at 64 B/clk/CU), so a step costs the same from
// 16 to 16 x CUs chains.  Results do not depend on the chain count, the tile position, the tiles per workgroup or any slice plan.
// (Round 3's first form, k_wide_traj_bf16 -- one tile per workgroup, beta operands rebuilt by every lane, no software pipelining, 8
//  barriers per step -- was replaced in round 5 by the kernel below with NT2 = 1: the same trajectories bit for bit, 1.1 - 1.5 x faster.)
// ---------------------------------------------------------------------------------------------------------------
// k_wide_traj2_bf16 (round 5), written for TWO chain tiles (32 chains) per workgroup: config 5 as a whole -- 8192 chains on one GPU -- is
// 512 chain tiles on 256 CUs.  With one tile per workgroup every CU streams the whole one-piece image twice per leapfrog step
// (512 MB per step chip-wide from the L2s: 17 TB/s sustained at the 30 us per step measured, half the L2's aggregate peak; MFMA
// busy 38 %, profiles/r5_cfg5_whole_baseline.txt).  Here every wave carries the beta operands and gradient accumulators of TWO
// tiles and uses each block image it fetched -- and every LDS operand read of it -- for both: per chain, half the L2 -> LDS traffic,
// half the LDS operand reads, the same MFMAs; 256 workgroups of 8 waves = one per CU at 8192 chains.  Row partition over the
// waves, MFMA order per tile and the wave-order fp64 reduction do not depend on the tiles per workgroup: a chain's trajectory is
// bit-identical with one tile or two (tests/test_gpu_fullsize.py), so the choice between them is a matter of speed only.
//  * Block body, software-pipelined by hand (a scheduling barrier per MFMA slot): left to itself the scheduler issues the 32 eta MFMAs of both
//    tiles, then ALL 75 vector instructions of the two sigmoids with the matrix pipe idle, then the 16 gradient MFMAs (measured:
//    2200-2600 cycles per block and wave against 768 of MFMA).  Here tile 0's sigmoid issues between tile 1's eta MFMAs and tile 1's
//    between tile 0's gradient MFMAs.
//  * Reduction over the waves through the wave's OWN last ring slot, one tile (16 chains x P floats = one slot) at a time: 4
//    barriers per step instead of 16, 16-byte accesses only; thread (chain oc, chunk oq) owns coordinates 4 oq .. 4 oq + 3.
// LDS (p = 128, two tiles): 8 rings x 16 KB + beta operands 16 KB + momenta 16 KB = 160 KB.
// NB = pieces of beta in the eta MFMAs: 2 (hi + lo: the default policy) or 1 (LR_PREC_BF16, the caller's explicit request: a third of the
// MFMAs fewer -- the kernel is POWER-bound, 1300 W at 2.04 GHz, so the time follows the work: 24.2 -> 20.5 us per evaluation at config 5
// whole -- for 0.019 of acceptance, 0.756 -> 0.737; still an exact sampler: a deterministic force, exact end points).
// S = double (a FLOAT64 model): the owner thread's position and momentum are float64 and BOTH wait in global memory between the
// reductions (a.q1, a.pm: 32 bytes each per thread and tile; the LDS has no room for float64 momenta); kick and drift in float64.
// NT2 = chain tiles per workgroup: 2, or 1 -- the same kernel below one tile per CU, where a second tile per workgroup would leave CUs idle
// (config 5's design at 4096 chains: 15.3 -> 13.9 us per evaluation against round 3's one-tile kernel; small designs 1.2 - 1.5 x).
template <int P, int FMT = 0, typename S = float, int NT2 = 2>
__global__ void __launch_bounds__(512) k_wide_traj2_bf16(TallArgs<S, P> a) {
    using G = WideBf16Geom<P>;
    constexpr int NB = FMT == 0 ? 2 : 1;
    constexpr bool F16 = FMT == 2, F64 = sizeof(S) == 8;
    typedef double f64x2t __attribute__((ext_vector_type(2)));
    const uint16_t* const image = F16 ? a.xblk1h : a.xblk1;
    static_assert(NT2 == 1 || NT2 == 2, "one or two chain tiles per workgroup");
    constexpr int NW = 8, BLK_BYTES = G::BUF1 * 2, NBUF = P >= 128 ? 2 : 4, RING_BYTES = NBUF * BLK_BYTES;
    constexpr int NQ = P / 4;             // 16-byte chunks per chain
    static_assert(16 * P * 4 == BLK_BYTES, "a tile's gradients fill exactly one ring slot");
    static_assert(P == 128 || P == 64, "chunk ownership below");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NW * RING_BYTES];
    // The beta operands of the eta MFMAs, READY-MADE: piece h (hi, lo) of the 8 coordinates 32 m + 8 kg + .. of chain c of tile t as the
    // 16 bytes lane (c, kg) feeds the MFMA (odd kg: halves swapped, as the eta read of the rows delivers them), at
    // [t][m][kg][h][c ^ ((4 m + kg) & 15)] -- the XOR keeps the 16 chains of a reader's service group AND the 16 (m, kg) cells an owner
    // thread's chain is spread over on distinct bank groups.  Written once per step by the thread that owns the coordinates (8 values
    // per thread) instead of being rebuilt from fp32 positions by every lane of all 8 waves (64 values per lane: 3400 cycles per step).
    __shared__ __attribute__((aligned(16))) uint32_t qop[NT2][G::M32][4][2][16][4];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, kg = lane >> 4;
    const int nblk = (int)((a.n + 31) / 32);
    const int per_wave = (nblk + NW - 1) / NW;
    const int wb0 = wave * per_wave;
    const int wnb = nblk - wb0 < 0 ? 0 : (nblk - wb0 < per_wave ? nblk - wb0 : per_wave);
    unsigned char* ring = smem + wave * RING_BYTES;
    const uint32_t ring_lds = (uint32_t)(uintptr_t)ring;

    auto issue = [&](int b) {  // block b of this wave -> ring slot b % NBUF; one M0 set-up per block (see k_wide_partial_bf16r)
        static_assert(BLK_BYTES == 8192 || BLK_BYTES == 4096, "pieces addressed around the middle of the block");
        const unsigned char* src = reinterpret_cast<const unsigned char*>(image + (wb0 + b) * (int64_t)G::BUF1) + lane * 16 + BLK_BYTES / 2;
        const uint32_t dst = ring_lds + (uint32_t)((b & (NBUF - 1)) * BLK_BYTES) + BLK_BYTES / 2;
        uint32_t keep;
        if constexpr (BLK_BYTES == 8192)
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                         "global_load_lds_dwordx4 %1, off offset:-4096\n\tglobal_load_lds_dwordx4 %1, off offset:-3072\n\t"
                         "global_load_lds_dwordx4 %1, off offset:-2048\n\tglobal_load_lds_dwordx4 %1, off offset:-1024\n\t"
                         "global_load_lds_dwordx4 %1, off\n\tglobal_load_lds_dwordx4 %1, off offset:1024\n\t"
                         "global_load_lds_dwordx4 %1, off offset:2048\n\tglobal_load_lds_dwordx4 %1, off offset:3072\n\t"
                         "s_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
        else
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\t"
                         "global_load_lds_dwordx4 %1, off offset:-2048\n\tglobal_load_lds_dwordx4 %1, off offset:-1024\n\t"
                         "global_load_lds_dwordx4 %1, off\n\tglobal_load_lds_dwordx4 %1, off offset:1024\n\t"
                         "s_mov_b32 m0, %0"
                         : "=&s"(keep) : "v"(src), "s"(dst) : "memory");
    };

    // the thread's share of the state: chains 16 t + oc (t = 0, 1), coordinates 4 oq .. 4 oq + 3
    const int oc = tid / NQ, oq = tid % NQ;
    const bool owner = oc < 16;  // (P = 64: 256 of the 512 threads)
    const int occ = owner ? oc : 15;
    // (addresses are recomputed where they are used: kept live across the trajectory they spilled to scratch)
    auto state_at = [&](int t, bool& live) {
        int tix = threadIdx.x;
        asm volatile("" : "+v"(tix));  // (opaque: otherwise the two uses are merged and the addresses stay live -- in scratch)
        int64_t ch = (int64_t)blockIdx.x * (16 * NT2) + 16 * t + (tix / NQ < 16 ? tix / NQ : 15);
        live = tix / NQ < 16 && ch < a.C;
        if (ch >= a.C) ch = a.C - 1;
        return ch * P + 4 * (tix % NQ);
    };
    // Between the reductions the thread's momenta wait in LDS (pmom) and its positions in a.q1 itself -- the row loop needs every
    // register, and the LDS is full: 8 + 8 floats per thread.
    __shared__ __attribute__((aligned(16))) f32x4 pmom[NT2][F64 ? 1 : 512];
    // the two bf16 pieces of k * position for the thread's four coordinates -> qop
    auto put_ops = [&](int t, const f32x4& q) {
        const int m = oq >> 3, okg = (oq >> 1) & 3, ip = ((oq & 1) << 1) ^ ((okg & 1) << 1);  // pair slot after the odd-kg half swap
        uint32_t hi[2], lo[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float x0 = q[2 * i] * ExpScale<float>::k, x1 = q[2 * i + 1] * ExpScale<float>::k;
            if constexpr (F16) {
                hi[i] = pack_f16(clamp_f16(x0), clamp_f16(x1));
            } else {
                hi[i] = pack_rne(x0, x1);
                const float h0 = __builtin_bit_cast(float, hi[i] << 16), h1 = __builtin_bit_cast(float, hi[i] & 0xFFFF0000u);
                lo[i] = pack_rne(x0 - h0, x1 - h1);
            }
        }
        const int cs = occ ^ ((4 * m + okg) & 15);
        *reinterpret_cast<u32x2*>(&qop[t][m][okg][0][cs][ip]) = u32x2{hi[0], hi[1]};
        if constexpr (NB == 2) *reinterpret_cast<u32x2*>(&qop[t][m][okg][1][cs][ip]) = u32x2{lo[0], lo[1]};
    };
#pragma unroll
    for (int t = 0; t < NT2; ++t) {
        bool live;
        const int64_t at = state_at(t, live);
        if constexpr (!F64) {
            const f32x4 q0 = *reinterpret_cast<const f32x4*>(a.q1 + at);
            pmom[t][tid] = *reinterpret_cast<const f32x4*>(a.pm + at);
            if (owner) put_ops(t, q0);
        } else {
            const f64x2t lo = *reinterpret_cast<const f64x2t*>(a.q1 + at), hi = *reinterpret_cast<const f64x2t*>(a.q1 + at + 2);
            if (owner) put_ops(t, f32x4{(float)lo[0], (float)lo[1], (float)hi[0], (float)hi[1]});
        }
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);  // nothing but the DMA ring counts on vmcnt from here on
#pragma unroll
    for (int b = 0; b < NBUF - 1; ++b)
        if (b < wnb) issue(b);
    __syncthreads();

    const int eta_off = G::elem(kg, c, 0) & ~7;
    const int ri = (lane & 15) >> 2, ci = lane & 3;
    const int tr_off[2] = {G::elem(ci, 4 * kg + ri, 0), G::elem(ci, 4 * kg + ri, 4)};
    // the exchange slot (the wave's last ring slot, which no step-start prefetch touches), as [16 chains][NQ chunks of 16 bytes],
    // chunk index XOR-swizzled by the chain (8 consecutive chains write / 32 consecutive chunks read distinct bank groups)
    float* const xslot = reinterpret_cast<float*>(ring + (NBUF - 1) * BLK_BYTES);
    const int xw_row = c * P, xw_sw = c & 7;
    const int xr_off = occ * P + ((oq ^ (occ & 7)) << 2);
    const int nsteps = a.l - 1;
    LR_TRAJ_PHASES_BEGIN  // development builds: shader cycles per phase of a step (lr_stamps.h)
    for (int s = 0; s < nsteps; ++s) {
        // beta = hi + lo (two round-to-nearest bf16 pieces) of the lane's coordinates 32 m + 8 kg + i, times log2(e), per tile: ready in qop
        u32x4 bq[NT2][G::M32][NB];
#pragma unroll
        for (int t = 0; t < NT2; ++t)
#pragma unroll
            for (int m = 0; m < G::M32; ++m)
#pragma unroll
                for (int h = 0; h < NB; ++h) bq[t][m][h] = *reinterpret_cast<const u32x4*>(&qop[t][m][kg][h][c ^ ((4 * m + kg) & 15)][0]);
        f32x4 gacc[NT2][G::MBP];
#pragma unroll
        for (int t = 0; t < NT2; ++t)
#pragma unroll
            for (int mb = 0; mb < G::MBP; ++mb) gacc[t][mb] = f32x4{0, 0, 0, 0};
        LR_TRAJ_PHASE(0);
        // The row loop runs the gradient half ONE BLOCK BEHIND the eta half: trip b issues the eta MFMAs of block b with the
        // sigmoid of block b - 1 (independent work) in their shadow, then the gradient MFMAs of block b - 1 with the transposed
        // operand reads of block b in theirs.  The two waves of a SIMD leave every barrier in lockstep and stay there -- with the
        // matrix and the vector work of a trip depending on each other (eta -> sigmoid -> gradient of the SAME block) both waves
        // did their MFMAs together and their sigmoids together, the matrix pipe idle meanwhile: every exp / rcp cost its full issue
        // time (profiles/r5_traj2_marginal_costs.txt).
        u32x2 xt[G::MBP][2];  // transposed (gradient) operands of the block whose gradient MFMAs are still to come
        f32x4 e[NT2][2];      // its eta
        auto read_xt = [&](const uint16_t* base) {
#pragma unroll
            for (int mb = 0; mb < G::MBP; ++mb) {
                xt[mb][0] = lds_read_tr16(base + G::tile1(0, mb >> 1) + tr_off[mb & 1]);
                xt[mb][1] = lds_read_tr16(base + G::tile1(1, mb >> 1) + tr_off[mb & 1]);
            }
        };
        // One MFMA "slot" = the MFMA and the vector / LDS instructions that issue in its shadow; a scheduling barrier closes every slot, so
        // the emitted order is the source order (the group-barrier form left the sigmoid in one run in front of the MFMAs).
        constexpr int SPC = 2 * NB, NE = SPC * G::M32, NG = G::MBP;  // MFMA slots per 32-coordinate chunk; eta / gradient MFMAs per tile
        auto eta_mfma = [&](const u32x4 (&xa)[2][G::M32], int i, f32x4 (&en)[NT2][2]) {  // slot i of the NT2 NE: tile i / NE, then (m, h, T)
            // (the two row tiles T alternate, a tile's accumulator every second slot; one accumulator's MFMAs in consecutive slots, or the
            //  sigmoid behind every second slot only: 24.3 - 24.4 us against 24.4 -- the kernel is power-bound, not issue-order-bound)
            const int t = i / NE, r = i % NE, m = r / SPC, h = (r >> 1) % NB, T = r & 1;
            if (LR_TRAJ_EXP(2) && h) return;
            if (m == 0 && h == 0) en[t][T] = f32x4{0, 0, 0, 0};
            en[t][T] = mfma16<F16>(xa[T][m], bq[t][m][h], en[t][T]);
        };
        auto grad_mfma = [&](int t, int mb, const u32x4& wv) {
            if (LR_TRAJ_EXP(1) && mb) return;
            const u32x4 xg = {xt[mb][0][0], xt[mb][0][1], xt[mb][1][0], xt[mb][1][1]};
            gacc[t][mb] = mfma16<F16>(xg, wv, gacc[t][mb]);
        };
        // the sigmoid of the tiles of the pending block as four stages over NV = 8 NT2 values (v = 8 t + 4 T + r), spread over the eta slots
        constexpr int NV = 8 * NT2;
        float sx[NV];
        uint32_t spk[NV / 2];
        auto sig_stage = [&](int stage, int v) {
            const int t = v >> 3, T = (v >> 2) & 1, r = v & 3;
            if (stage == 0) sx[v] = LR_TRAJ_EXP(0) ? e[t][T][r] : __builtin_amdgcn_exp2f(e[t][T][r]);
            else if (stage == 1) sx[v] = LR_TRAJ_EXP(0) ? sx[v] : 1.0f + sx[v];
            else if (stage == 2) sx[v] = LR_TRAJ_EXP(0) ? sx[v] : fast_rcp(sx[v]);
            else if ((v & 1) == 0) spk[v >> 1] = pack16<F16>(sx[v], sx[v + 1]);
        };
        auto trip = [&](int b, auto first_tag) {
            constexpr bool FIRST = decltype(first_tag)::value;
            LR_TRAJ_PHASE(4);
            // (the transposed reads of block b - 1 have returned before the DMA may overwrite its slot)
            __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0)
            // (the requests stay in front of the wait: one per eta slot, in the MFMAs' shadow, cost 11 500 cycles more per step than the
            //  2900 they take here -- profiles/r5_cfg5_whole.txt)
            if (!LR_TRAJ_EXP(3) && b + NBUF - 1 < wnb) issue(b + NBUF - 1);
            const int last = b + NBUF - 1 < wnb ? b + NBUF - 1 : wnb - 1;
            LR_TRAJ_PHASE(1);
            if (!LR_TRAJ_EXP(3)) wait_vm_blocks<BLK_BYTES>(last - b);
            LR_TRAJ_PHASE(2);
            const uint16_t* base = reinterpret_cast<const uint16_t*>(ring + (b & (NBUF - 1)) * BLK_BYTES);
            __builtin_amdgcn_sched_barrier(0);
            // the eta operands of the block, requested two 32-coordinate chunks ahead of their MFMAs (all eight at once are 32
            // registers this kernel does not have); tile 1 reads them again
            u32x4 xa[2][G::M32];
            auto read_xa = [&](int m) {
                xa[0][m] = *reinterpret_cast<const u32x4*>(base + G::tile1(0, m) + eta_off);
                xa[1][m] = *reinterpret_cast<const u32x4*>(base + G::tile1(1, m) + eta_off);
            };
            auto xa_ahead = [&](int i) {  // in the shadow of eta slot i: the chunk two ahead in the (tile, chunk) order
                if (i % SPC != 0) return;
                const int seq = i / SPC + 2;  // chunk sequence number over the tiles
                if (seq < NT2 * G::M32) read_xa(seq % G::M32);
            };
            read_xa(0);
            if (G::M32 > 1) read_xa(1);
            __builtin_amdgcn_sched_barrier(0);
            f32x4 en[NT2][2];
            if constexpr (FIRST) {
#pragma unroll
                for (int i = 0; i < NT2 * NE; ++i) {
                    eta_mfma(xa, i, en);
                    xa_ahead(i);
                    __builtin_amdgcn_sched_barrier(0);
                }
                read_xt(base);
            } else {
                constexpr int Q = NT2 * NE / 4, VPS = NV / Q;  // slots per sigmoid stage, values per slot (P = 128, two tiles, two pieces: 8 and 2)
                static_assert(NT2 * NE % 4 == 0 && Q * VPS == NV, "the sigmoid values over a quarter of the eta slots per stage");
#pragma unroll
                for (int i = 0; i < NT2 * NE; ++i) {
                    eta_mfma(xa, i, en);
                    xa_ahead(i);
                    const int stage = i / Q, k = i % Q;        // values VPS k .. VPS k + VPS - 1 of this stage
#pragma unroll
                    for (int v = 0; v < VPS; ++v) sig_stage(stage, VPS * k + v);
                    __builtin_amdgcn_sched_barrier(0);
                }
                const u32x4 w0 = {spk[0], spk[1], spk[2], spk[3]}, w1 = {spk[NV / 2 - 4], spk[NV / 2 - 3], spk[NV / 2 - 2], spk[NV / 2 - 1]};
                // the tiles' gradient MFMAs of an operand back to back, then block b's operand takes its place: the transposed reads are
                // spread over the NT2 NG slots, and all but the last have returned when the next trip asks (it waits for them before its DMA
                // request may overwrite their slot)
#pragma unroll
                for (int mb = 0; mb < NG; ++mb) {
                    grad_mfma(0, mb, w0);
                    if constexpr (NT2 == 2) {
                        __builtin_amdgcn_sched_barrier(0);
                        grad_mfma(1, mb, w1);
                    }
                    xt[mb][0] = lds_read_tr16(base + G::tile1(0, mb >> 1) + tr_off[mb & 1]);
                    xt[mb][1] = lds_read_tr16(base + G::tile1(1, mb >> 1) + tr_off[mb & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
            for (int t = 0; t < NT2; ++t) e[t][0] = en[t][0], e[t][1] = en[t][1];
            __builtin_amdgcn_sched_barrier(0);
        };
        // (starting the second wave of every SIMD 256 .. 1024 cycles late, to break the lockstep of the pair: 24.4 .. 24.7 us against 24.5)
        if (wnb > 0) trip(0, std::integral_constant<bool, true>{});
        for (int b = 1; b < wnb; ++b) trip(b, std::integral_constant<bool, false>{});
        if (wnb > 0) {  // drain: sigmoid and gradient of the last block
#pragma unroll
            for (int stage = 0; stage < 4; ++stage)
#pragma unroll
                for (int v = 0; v < NV; ++v) sig_stage(stage, v);
            const u32x4 w0 = {spk[0], spk[1], spk[2], spk[3]}, w1 = {spk[NV / 2 - 4], spk[NV / 2 - 3], spk[NV / 2 - 2], spk[NV / 2 - 1]};
#pragma unroll
            for (int mb = 0; mb < NG; ++mb) grad_mfma(0, mb, w0);
            if constexpr (NT2 == 2) {
#pragma unroll
                for (int mb = 0; mb < NG; ++mb) grad_mfma(1, mb, w1);
            }
        }
        // Every ring slot has been read.  The step is finished from the thread's share of the state, which waits in global memory between
        // the reductions (positions in a.q1; float64 models: momenta in a.pm as well) together with the step's constants (drift factors,
        // prior precisions of the thread's coordinates: held in registers across the row loop all of it spilled to scratch).  Plain loads,
        // complete -- an explicit vmcnt(0): nothing else is in flight here -- BEFORE the next step's first blocks are requested, so that
        // vmcnt counts nothing but the DMA ring from then on.  (Round 5 first fetched them with inline-asm loads in front of the DMA
        // requests and a hand-counted s_waitcnt tied to the registers; the compiler, for whom an asm load's result is there when the
        // statement ends, placed register copies between load and wait: a rare wrong trajectory when a load was slow.)
        f32x4 sb, si, qg[NT2];          // float32 models
        f64x2t vbd[2], vid[2], qd[NT2][2], pd[NT2][2];  // float64 models: drift factors, prior precisions, positions, momenta as pairs
        bool qlive[NT2];
        int64_t qat[NT2];
        if constexpr (!F64) {
            sb = *reinterpret_cast<const f32x4*>(a.cvec + 4 * oq);
            si = *reinterpret_cast<const f32x4*>(a.cvec + P + 4 * oq);
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                vbd[h] = *reinterpret_cast<const f64x2t*>(a.cvec + 4 * oq + 2 * h);
                vid[h] = *reinterpret_cast<const f64x2t*>(a.cvec + P + 4 * oq + 2 * h);
            }
        }
#pragma unroll
        for (int t = 0; t < NT2; ++t) {
            qat[t] = state_at(t, qlive[t]);
            if constexpr (!F64) {
                qg[t] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(a.q1 + qat[t]));
            } else {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    qd[t][h] = __builtin_nontemporal_load(reinterpret_cast<const f64x2t*>(a.q1 + qat[t] + 2 * h));
                    pd[t][h] = __builtin_nontemporal_load(reinterpret_cast<const f64x2t*>(a.pm + qat[t] + 2 * h));
                }
            }
        }
        LR_TRAJ_PHASE(4);
        // the waves' gradients meet in every wave's own exchange slot, a tile at a time; wave-order fp64 sums, then the step's kick and drift
#pragma unroll
        for (int t = 0; t < (LR_TRAJ_EXP(5) ? 0 : NT2); ++t) {
#pragma unroll
            for (int mb = 0; mb < G::MBP; ++mb) {
                const int q = 8 * (mb >> 1) + 2 * kg + (mb & 1);  // chunk of coordinates 32 (mb >> 1) + 8 kg + 4 (mb & 1) ..
                *reinterpret_cast<f32x4*>(xslot + xw_row + ((q ^ xw_sw) << 2)) = gacc[t][mb];
            }
            if (t == 0) {  // (the loads travelled under the writes above) ... now the next step's first blocks may go
                __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
                if (!LR_TRAJ_EXP(3) && s + 1 < nsteps) {
#pragma unroll
                    for (int b = 0; b < NBUF - 1; ++b)
                        if (b < wnb) issue(b);
                }
            }
            __syncthreads();
            if (owner) {
                f32x4 pw[NW];
#pragma unroll
                for (int w = 0; w < NW; ++w) pw[w] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(smem + w * RING_BYTES + (NBUF - 1) * BLK_BYTES) + xr_off);
                if constexpr (!F64) {
                    f32x4 qn = qg[t], pn = pmom[t][tid];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        double gs = 0.0;
#pragma unroll
                        for (int w = 0; w < NW; ++w) gs += (double)pw[w][i];  // wave order
                        const float g1 = (float)gs - qn[i] * si[i];
                        pn[i] = fma_t(a.step, g1, pn[i]);
                        qn[i] = fma_t(sb[i], pn[i], qn[i]);
                    }
                    put_ops(t, qn);
                    pmom[t][tid] = pn;
                    if (qlive[t]) *reinterpret_cast<f32x4*>(a.q1 + qat[t]) = qn;
                } else {
                    S qn[4], pn[4];
                    f32x4 qf;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        qn[i] = qd[t][i >> 1][i & 1];
                        pn[i] = pd[t][i >> 1][i & 1];
                        double gs = 0.0;
#pragma unroll
                        for (int w = 0; w < NW; ++w) gs += (double)pw[w][i];  // wave order
                        const S g1 = (S)gs - qn[i] * vid[i >> 1][i & 1];
                        pn[i] = fma_t(a.step, g1, pn[i]);
                        qn[i] = fma_t((S)vbd[i >> 1][i & 1], pn[i], qn[i]);
                        qf[i] = (float)qn[i];
                    }
                    put_ops(t, qf);
                    if (qlive[t]) {
                        *reinterpret_cast<f64x2t*>(a.q1 + qat[t]) = f64x2t{(double)qn[0], (double)qn[1]};
                        *reinterpret_cast<f64x2t*>(a.q1 + qat[t] + 2) = f64x2t{(double)qn[2], (double)qn[3]};
                        *reinterpret_cast<f64x2t*>(a.pm + qat[t]) = f64x2t{(double)pn[0], (double)pn[1]};
                        *reinterpret_cast<f64x2t*>(a.pm + qat[t] + 2) = f64x2t{(double)pn[2], (double)pn[3]};
                    }
                }
            }
            __syncthreads();  // the exchange slots have been read (next tile / the next step's DMA may overwrite them); qop is complete
        }
        LR_TRAJ_PHASE(3);
    }
    LR_TRAJ_PHASES_REPORT(a)
#pragma unroll
    for (int t = 0; t < NT2; ++t) {
        bool live;
        const int64_t at = state_at(t, live);
        if constexpr (!F64) {
            if (live) *reinterpret_cast<f32x4*>(a.pm + at) = pmom[t][tid];  // (the position is in a.q1 already)
        } else {
            (void)at;  // (float64 models: position and momentum are in a.q1 / a.pm already)
        }
    }
}

// Host side: build the per-32-row-block LDS images (bf16 pieces, swizzled layout) from the signed rows.
// rows: [n][P] fp32 (host).  out: [ceil(n/32)][BUF] bf16 bit patterns.
template <int P> inline void wide_bf16_prepare(const float* rows, int64_t n, uint16_t* out) {
    using G = WideBf16Geom<P>;
    const int64_t nblk = (n + 31) / 32;
    for (int64_t b = 0; b < nblk; ++b) {
        uint16_t* base = out + b * (int64_t)G::BUF;
        for (int e = 0; e < G::BUF; ++e) base[e] = 0;
        for (int srow = 0; srow < 32; ++srow) {
            const int64_t r = 32 * b + srow;
            if (r >= n) continue;
            const int T = srow >> 4, rr = srow & 15;
            for (int cc = 0; cc < P; ++cc) {
                const float x = rows[r * P + cc];
                uint32_t xb, hb, mb_, lb;
                memcpy(&xb, &x, 4);
                hb = xb & 0xFFFF0000u;
                float h, r1, md, lo;
                memcpy(&h, &hb, 4);
                r1 = x - h;
                memcpy(&mb_, &r1, 4);
                mb_ &= 0xFFFF0000u;
                memcpy(&md, &mb_, 4);
                lo = r1 - md;
                memcpy(&lb, &lo, 4);
                const uint16_t pc[3] = {(uint16_t)(hb >> 16), (uint16_t)(mb_ >> 16), (uint16_t)(lb >> 16)};
                const int m = cc >> 5, w5 = cc & 31;
                for (int q = 0; q < 3; ++q) base[G::tile(q, T, m) + G::elem(w5 >> 3, rr, w5 & 7)] = pc[q];
            }
        }
    }
}

// The single-piece image of the interior-step kernel: bf16 round-to-nearest-even of the signed rows, same
// (kg', row, half) swizzle.  out: [ceil(n/32)][BUF1].
template <int P> inline void wide_bf16_prepare_rne(const float* rows, int64_t n, uint16_t* out) {
    using G = WideBf16Geom<P>;
    const int64_t nblk = (n + 31) / 32;
    for (int64_t b = 0; b < nblk; ++b) {
        uint16_t* base = out + b * (int64_t)G::BUF1;
        for (int e = 0; e < G::BUF1; ++e) base[e] = 0;
        for (int srow = 0; srow < 32; ++srow) {
            const int64_t r = 32 * b + srow;
            if (r >= n) continue;
            const int T = srow >> 4, rr = srow & 15;
            for (int cc = 0; cc < P; ++cc) {
                uint32_t xb;
                memcpy(&xb, &rows[r * P + cc], 4);
                const uint32_t rounded = xb + 0x7FFFu + ((xb >> 16) & 1u);  // finite inputs only (checked at model creation)
                const int m = cc >> 5, w5 = cc & 31;
                base[G::tile1(T, m) + G::elem(w5 >> 3, rr, w5 & 7)] = (uint16_t)(rounded >> 16);
            }
        }
    }
}


// ... and its half-precision twin (the trajectory kernels' f16 interior): IEEE binary16 round-to-nearest-even of the signed rows in
// the same layout.  Returns false -- no image, the bf16 pieces stay in use -- unless every |x| <= 2^15 and every non-zero column
// reaches 2^-10 somewhere (below 2^-14 an f16 loses bits one by one; a column that small as a whole would be carried worse than in bf16).
inline uint16_t f16_bits_rne(float x) {  // finite |x| <= 2^15
    uint32_t u;
    memcpy(&u, &x, 4);
    const uint32_t sign = (u >> 16) & 0x8000u;
    u &= 0x7FFFFFFFu;
    if (u < 0x33000001u) return (uint16_t)sign;  // <= 2^-25: rounds to zero (ties to even)
    if (u < 0x38800000u) {                       // subnormal result: shift the 24-bit significand into place, round to nearest even
        const int shift = 126 - (int)(u >> 23);  // 14 .. 24
        const uint32_t sig = (u & 0x7FFFFFu) | 0x800000u;
        const uint32_t q = sig >> shift, rem = sig & ((1u << shift) - 1u), half = 1u << (shift - 1);
        return (uint16_t)(sign | (q + ((rem > half || (rem == half && (q & 1u))) ? 1u : 0u)));
    }
    const uint32_t r = u + 0xFFFu + ((u >> 13) & 1u);  // normal: round the 13 dropped bits, a carry moves into the exponent
    return (uint16_t)(sign | ((r - 0x38000000u) >> 13));
}
template <int P> inline bool wide_f16_prepare_rne(const float* rows, int64_t n, uint16_t* out) {
    using G = WideBf16Geom<P>;
    float cmax[P];
    for (int cc = 0; cc < P; ++cc) cmax[cc] = 0.0f;
    for (int64_t r = 0; r < n; ++r)
        for (int cc = 0; cc < P; ++cc) {
            const float ax = rows[r * P + cc] < 0 ? -rows[r * P + cc] : rows[r * P + cc];
            cmax[cc] = ax > cmax[cc] ? ax : cmax[cc];
        }
    for (int cc = 0; cc < P; ++cc)
        if (cmax[cc] > 32768.0f || (cmax[cc] != 0.0f && cmax[cc] < 0x1p-10f)) return false;
    const int64_t nblk = (n + 31) / 32;
    for (int64_t b = 0; b < nblk; ++b) {
        uint16_t* base = out + b * (int64_t)G::BUF1;
        for (int e = 0; e < G::BUF1; ++e) base[e] = 0;
        for (int srow = 0; srow < 32; ++srow) {
            const int64_t r = 32 * b + srow;
            if (r >= n) continue;
            const int T = srow >> 4, rr = srow & 15;
            for (int cc = 0; cc < P; ++cc) {
                const int m = cc >> 5, w5 = cc & 31;
                base[G::tile1(T, m) + G::elem(w5 >> 3, rr, w5 & 7)] = f16_bits_rne(rows[r * P + cc]);
            }
        }
    }
    return true;
}

}  // namespace lr
