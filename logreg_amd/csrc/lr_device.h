// lr_device.h -- device-side building blocks of the fused many-chain MCMC kernels (gfx950 only).
//
// Mapping ("group per chain"): a chain is owned by G consecutive lanes of one wavefront
// (G in {1,2,4,8,16,32,64}); the n data rows are dealt round-robin to the G lanes; every lane of
// the group carries an identical copy of the chain state (q, p, gradient, log-density) in
// registers.  One log-posterior evaluation is: each lane accumulates value/gradient partials
// over its rows, then a butterfly ALL-reduce inside the group (DPP row_mirror / row_half_mirror
// / quad_perm inside a 16-lane row, v_permlane16_swap / v_permlane32_swap across rows) leaves
// bit-identical sums in every lane, so accept/reject decisions never diverge inside a group.
// G = 64 is "one wavefront per chain"; G = 1 is "one lane per chain".
//
// Rows are the SIGNED design rows  xs_i = (2 y_i - 1) * x_i : with t_i = xs_i . beta,
//     ll(beta)      = sum_i log sigma(t_i)              (== reference fit-np-hmc.py:23-24)
//     grad ll(beta) = sum_i sigma(-t_i) * xs_i          (== X^T (y - sigma(X beta)), fit-np-hmc.py:46)
// so y never has to be read in the hot loop and one pass over a row yields value and gradient.
//
// Row residency policies (template parameter `Rows`):
//     RegRows    -- the lane's rows live in VGPRs for the whole launch (Pima-scale n)
//     LdsRows    -- all rows staged once in LDS, lanes stride over them
//     GlobalRows -- rows streamed from global memory / L2 (tall data)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lr {

// ------------------------------------------------------------------------------------------
// Philox4x32-10 counter-based generator (Salmon et al. SC'11).  Stream layout:
// key = (seed_lo, seed_hi); counter = (chain, iter_lo, iter_hi, block | tag).
// Normal j of an iteration comes from block j/4: Box-Muller pairs (w0,w1)->(z0,z1),
// (w2,w3)->(z2,z3).  The accept uniform is word 0 of block tag 0x80000000.
// 24-bit uniforms u = ((w >> 8) + 0.5) * 2^-24 are exact in fp32 and fp64.
// ------------------------------------------------------------------------------------------
struct U4 { uint32_t x, y, z, w; };

__device__ __forceinline__ U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                            uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return U4{c0, c1, c2, c3};
}

constexpr uint32_t TAG_UNIFORM = 0x80000000u;

template <typename T> __device__ __forceinline__ T u01(uint32_t w) {
    return (T(w >> 8) + T(0.5)) * T(1.0 / 16777216.0);
}
// log of the 24-bit uniform in float: v_log_f32 (1 ulp) times ln 2
__device__ __forceinline__ float log_u01(uint32_t w) { return 0.693147180559945309f * __builtin_amdgcn_logf(u01<float>(w)); }

// float: on the hardware transcendentals -- v_log_f32 (log2), v_sqrt_f32, v_sin_f32 / v_cos_f32 (argument in REVOLUTIONS, which is
// what Box-Muller has: 2 pi u).  10 instructions per pair where logf / sqrtf / sincospif with their range handling took ~65; the
// normals agree with the float64 oracle's to ~1e-6 (tests/test_gpu_parity.py::test_device_normals_against_the_oracle), far inside
// every parity tolerance, and every kernel shares this function, so variants still agree with each other.
__device__ __forceinline__ void box_muller(uint32_t wa, uint32_t wb, float& z0, float& z1) {
    const float r = __builtin_amdgcn_sqrtf(-1.38629436111989062f * __builtin_amdgcn_logf(u01<float>(wa)));  // sqrt(-2 ln u), ln u = ln 2 * log2 u
    const float t = u01<float>(wb);
    z0 = r * __builtin_amdgcn_cosf(t);
    z1 = r * __builtin_amdgcn_sinf(t);
}
__device__ __forceinline__ void box_muller(uint32_t wa, uint32_t wb, double& z0, double& z1) {
    const double r = sqrt(-2.0 * log(u01<double>(wa)));
    double s, c;
    sincospi(2.0 * u01<double>(wb), &s, &c);
    z0 = r * c;
    z1 = r * s;
}

// z[0..P) standard normals for (seed, chain, iter).  P is the padded width; padded entries are
// generated but always multiplied by a zero scale by the callers.
template <typename T, int P>
__device__ __forceinline__ void draw_normals(uint64_t seed, uint64_t chain, uint64_t iter, T (&z)[P]) {
#pragma unroll
    for (int b = 0; 4 * b < P; ++b) {
        const U4 w = philox4x32_10((uint32_t)chain, (uint32_t)iter, (uint32_t)(iter >> 32), (uint32_t)b,
                                   (uint32_t)seed, (uint32_t)(seed >> 32));
        T a0, a1, a2, a3;
        box_muller(w.x, w.y, a0, a1);
        box_muller(w.z, w.w, a2, a3);
        if (4 * b + 0 < P) z[4 * b + 0] = a0;
        if (4 * b + 1 < P) z[4 * b + 1] = a1;
        if (4 * b + 2 < P) z[4 * b + 2] = a2;
        if (4 * b + 3 < P) z[4 * b + 3] = a3;
    }
}

// log(u) of the accept uniform for (seed, chain, iter)
template <typename T>
__device__ __forceinline__ T draw_log_uniform(uint64_t seed, uint64_t chain, uint64_t iter) {
    const U4 w = philox4x32_10((uint32_t)chain, (uint32_t)iter, (uint32_t)(iter >> 32), TAG_UNIFORM,
                               (uint32_t)seed, (uint32_t)(seed >> 32));
    if constexpr (sizeof(T) == 4) return log_u01(w.x);
    else return log(u01<double>(w.x));
}

// ------------------------------------------------------------------------------------------
// group all-reduce (sum) over G consecutive lanes; result bit-identical in all G lanes.
// ------------------------------------------------------------------------------------------
template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int CTRL> __device__ __forceinline__ double dpp_mov(double v) {
    const uint64_t b = __builtin_bit_cast(uint64_t, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)b, CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b >> 32), CTRL, 0xF, 0xF, true);
    return __builtin_bit_cast(double, ((uint64_t)(uint32_t)hi << 32) | (uint64_t)(uint32_t)lo);
}

// x-row exchange via v_permlane{16,32}_swap (gfx950).  ROCm 7.2's clang builtin for these
// returns element 0 twice, so they are issued as inline asm; the s_nop covers the
// VALU-write -> permlane-read hazard hipcc does not pad inside an asm statement.
__device__ __forceinline__ float swap16_sum(float v) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 0" : "+v"(a), "+v"(b));
    return a + b;
}
__device__ __forceinline__ float swap32_sum(float v) {
    float a = v, b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 0" : "+v"(a), "+v"(b));
    return a + b;
}
__device__ __forceinline__ double swap16_sum(double v) {
    const uint64_t bits = __builtin_bit_cast(uint64_t, v);
    uint32_t al = (uint32_t)bits, bl = al, ah = (uint32_t)(bits >> 32), bh = ah;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\tv_permlane16_swap_b32 %2, %3\n\ts_nop 0"
                 : "+v"(al), "+v"(bl), "+v"(ah), "+v"(bh));
    return __builtin_bit_cast(double, ((uint64_t)ah << 32) | al) + __builtin_bit_cast(double, ((uint64_t)bh << 32) | bl);
}
__device__ __forceinline__ double swap32_sum(double v) {
    const uint64_t bits = __builtin_bit_cast(uint64_t, v);
    uint32_t al = (uint32_t)bits, bl = al, ah = (uint32_t)(bits >> 32), bh = ah;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\tv_permlane32_swap_b32 %2, %3\n\ts_nop 0"
                 : "+v"(al), "+v"(bl), "+v"(ah), "+v"(bh));
    return __builtin_bit_cast(double, ((uint64_t)ah << 32) | al) + __builtin_bit_cast(double, ((uint64_t)bh << 32) | bl);
}

template <int G, typename T> __device__ __forceinline__ T group_sum(T v) {
    static_assert(G == 1 || G == 2 || G == 4 || G == 8 || G == 16 || G == 32 || G == 64, "bad group");
    if constexpr (G >= 16) v += dpp_mov<0x140>(v);  // row_mirror       : lane i <-> 15-i
    if constexpr (G >= 8) v += dpp_mov<0x141>(v);   // row_half_mirror  : lane i <-> 7-i (within 8)
    if constexpr (G >= 4) v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]
    if constexpr (G >= 2) v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]
    if constexpr (G >= 32) v = swap16_sum(v);       // rows 0<->1, 2<->3
    if constexpr (G >= 64) v = swap32_sum(v);       // halves
    return v;
}

// v_permlane{16,32}_swap on M register pairs in one statement (one hazard pad for all of them).
//   swap16: a <- [a.r0, b.r0, a.r2, b.r2], b <- [a.r1, b.r1, a.r3, b.r3]   (r = 16-lane row)
//   swap32: a <- [a.lo, b.lo],             b <- [a.hi, b.hi]               (lo/hi = 32-lane half)
template <int WIDTH, int M> __device__ __forceinline__ void swap_pairs(float (&a)[M], float (&b)[M]) {
    static_assert(WIDTH == 16 || WIDTH == 32, "row or half swap");
    static_assert(M == 1 || M == 2 || M == 4 || M % 8 == 0, "pairs per call");
    // the hazard pad must sit INSIDE the statement (hipcc neither sees the permlane nor keeps its own
    // instructions out from between two asm statements): one pad per statement of up to 8 swaps
#define LR_SW16 "v_permlane16_swap_b32 "
#define LR_SW32 "v_permlane32_swap_b32 "
#define LR_SWAP1(OP, i) asm volatile("s_nop 1\n\t" OP "%0, %1\n\ts_nop 0" : "+v"(a[i]), "+v"(b[i]))
#define LR_SWAP2(OP, i) \
    asm volatile("s_nop 1\n\t" OP "%0, %2\n\t" OP "%1, %3\n\ts_nop 0" : "+v"(a[i]), "+v"(a[i + 1]), "+v"(b[i]), "+v"(b[i + 1]))
#define LR_SWAP4(OP, i)                                                                                                  \
    asm volatile("s_nop 1\n\t" OP "%0, %4\n\t" OP "%1, %5\n\t" OP "%2, %6\n\t" OP "%3, %7\n\ts_nop 0"                 \
                 : "+v"(a[i]), "+v"(a[i + 1]), "+v"(a[i + 2]), "+v"(a[i + 3]), "+v"(b[i]), "+v"(b[i + 1]), "+v"(b[i + 2]), \
                   "+v"(b[i + 3]))
#define LR_SWAP8(OP, i)                                                                                                   \
    asm volatile("s_nop 1\n\t" OP "%0, %8\n\t" OP "%1, %9\n\t" OP "%2, %10\n\t" OP "%3, %11\n\t" OP "%4, %12\n\t" OP      \
                 "%5, %13\n\t" OP "%6, %14\n\t" OP "%7, %15\n\ts_nop 0"                                                  \
                 : "+v"(a[i]), "+v"(a[i + 1]), "+v"(a[i + 2]), "+v"(a[i + 3]), "+v"(a[i + 4]), "+v"(a[i + 5]),            \
                   "+v"(a[i + 6]), "+v"(a[i + 7]), "+v"(b[i]), "+v"(b[i + 1]), "+v"(b[i + 2]), "+v"(b[i + 3]),            \
                   "+v"(b[i + 4]), "+v"(b[i + 5]), "+v"(b[i + 6]), "+v"(b[i + 7]))
    if constexpr (WIDTH == 16) {
        if constexpr (M == 1) LR_SWAP1(LR_SW16, 0);
        else if constexpr (M == 2) LR_SWAP2(LR_SW16, 0);
        else if constexpr (M == 4) LR_SWAP4(LR_SW16, 0);
        else {
#pragma unroll
            for (int i = 0; i < M; i += 8) LR_SWAP8(LR_SW16, i);
        }
    } else {
        if constexpr (M == 1) LR_SWAP1(LR_SW32, 0);
        else if constexpr (M == 2) LR_SWAP2(LR_SW32, 0);
        else if constexpr (M == 4) LR_SWAP4(LR_SW32, 0);
        else {
#pragma unroll
            for (int i = 0; i < M; i += 8) LR_SWAP8(LR_SW32, i);
        }
    }
#undef LR_SWAP1
#undef LR_SWAP2
#undef LR_SWAP4
#undef LR_SWAP8
#undef LR_SW16
#undef LR_SW32
}

// N independent float values at once, LEVEL-major: between a value's write and its next DPP read there
// are N-1 other instructions, so no hazard padding and no dependency stall.  The scheduling barriers pin
// that order: left alone, the scheduler sometimes serialises the reduction value by value (32 dependent
// v_add_f32_dpp separated by s_nop: measured 10 % of the whole HMC kernel), depending on register
// pressure elsewhere in the kernel.
// Groups wider than a 16-lane row: the cross-row levels run FIRST, as a transposing reduce-scatter -- a
// permlane swap of (v[j], v[j + n/2]) followed by one add leaves the row sums of v[j] in one half of the
// rows and those of v[j + n/2] in the other, in ONE register -- so the four in-row DPP levels work on N/2
// (G = 32) or N/4 (G = 64) registers, and two (one) swap levels hand every lane all N totals back:
// 4 N instructions instead of 8 N (G = 64) / 6 N (G = 32).
template <int G, int N> __device__ __forceinline__ void group_sum_levels(float (&v)[N]) {
    static_assert(G == 1 || G == 2 || G == 4 || G == 8 || G == 16 || G == 32 || G == 64, "bad group");
#define LR_LEVEL(COND, CNT, EXPR)                  \
    if constexpr (COND) {                          \
        __builtin_amdgcn_sched_barrier(0);         \
        _Pragma("unroll") for (int j = 0; j < (CNT); ++j) v[j] = EXPR; \
    }
    constexpr bool X32 = G >= 64 && N % 4 == 0, X16 = G >= 32 && N % (G >= 64 ? 4 : 2) == 0 && (G < 64 || X32);
    constexpr int N1 = X32 ? N / 2 : N;   // registers after the half-swap level
    constexpr int N2 = X16 ? N1 / 2 : N1;  // registers after the row-swap level
    if constexpr (X32) {  // v[j] <- sum over the two halves of v[j] (lanes 0-31) | of v[j + N/2] (lanes 32-63)
        float a[N / 2], b[N / 2];
#pragma unroll
        for (int j = 0; j < N / 2; ++j) { a[j] = v[j]; b[j] = v[j + N / 2]; }
        swap_pairs<32>(a, b);
#pragma unroll
        for (int j = 0; j < N / 2; ++j) v[j] = a[j] + b[j];
    }
    if constexpr (X16) {  // v[j] <- sum over the row pair of v[j] (even rows) | of v[j + N1/2] (odd rows)
        float a[N1 / 2], b[N1 / 2];
#pragma unroll
        for (int j = 0; j < N1 / 2; ++j) { a[j] = v[j]; b[j] = v[j + N1 / 2]; }
        swap_pairs<16>(a, b);
#pragma unroll
        for (int j = 0; j < N1 / 2; ++j) v[j] = a[j] + b[j];
    }
    LR_LEVEL(G >= 16, N2, v[j] + dpp_mov<0x140>(v[j]))
    LR_LEVEL(G >= 8, N2, v[j] + dpp_mov<0x141>(v[j]))
    LR_LEVEL(G >= 4, N2, v[j] + dpp_mov<0x4E>(v[j]))
    LR_LEVEL(G >= 2, N2, v[j] + dpp_mov<0xB1>(v[j]))
    if constexpr (X16) {  // hand the totals back: even rows hold value j, odd rows value j + N1/2
        __builtin_amdgcn_sched_barrier(0);
        float a[N1 / 2], b[N1 / 2];
#pragma unroll
        for (int j = 0; j < N1 / 2; ++j) a[j] = b[j] = v[j];
        swap_pairs<16>(a, b);
#pragma unroll
        for (int j = 0; j < N1 / 2; ++j) { v[j] = a[j]; v[j + N1 / 2] = b[j]; }
    } else {
        LR_LEVEL(G >= 32, N, swap16_sum(v[j]))
    }
    if constexpr (X32) {
        float a[N / 2], b[N / 2];
#pragma unroll
        for (int j = 0; j < N / 2; ++j) a[j] = b[j] = v[j];
        swap_pairs<32>(a, b);
#pragma unroll
        for (int j = 0; j < N / 2; ++j) { v[j] = a[j]; v[j + N / 2] = b[j]; }
    } else {
        LR_LEVEL(G >= 64, N, swap32_sum(v[j]))
    }
#undef LR_LEVEL
    if constexpr (G >= 2) __builtin_amdgcn_sched_barrier(0);
}

// All draws of one iteration for a chain owned by G lanes: z[0..P) and log(u).
// With G >= P/4 + 1 lanes the work is SPLIT: lane gl computes ONE Philox block (normal blocks
// 0..NB-1, or the uniform block NB) and its two Box-Muller pairs, and the values are then
// broadcast inside the group with ds_swizzle (bit-mask mode: lane' = (lane & and) | or within 32
// lanes; v_readlane when the group is the whole wave).  Per lane: 1 Philox + 2 Box-Muller + 1 log
// instead of NB+1 Philox + 2 NB Box-Muller + 1 log -- the RNG is 60 % of a MALA/RWMH iteration.
// The VALUES are those of draw_normals/draw_log_uniform bit for bit.
template <int G, int SRC> __device__ __forceinline__ float group_bcast(float v) {
    static_assert(SRC < G, "source lane outside the group");
    if constexpr (G == 64) {
        return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), SRC));
    } else {
        constexpr int and_mask = 0x1F & ~(G - 1), pattern = and_mask | (SRC << 5);
        return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), pattern));
    }
}
template <int G, int SRC> __device__ __forceinline__ double group_bcast(double v) {
    const uint64_t b = __builtin_bit_cast(uint64_t, v);
    const float lo = group_bcast<G, SRC>(__builtin_bit_cast(float, (uint32_t)b));
    const float hi = group_bcast<G, SRC>(__builtin_bit_cast(float, (uint32_t)(b >> 32)));
    return __builtin_bit_cast(double, ((uint64_t)__builtin_bit_cast(uint32_t, hi) << 32) | __builtin_bit_cast(uint32_t, lo));
}

template <typename T, int P, int G, int B = 0>
__device__ __forceinline__ void bcast_blocks(const T (&mine)[4], T (&z)[P]) {
    if constexpr (4 * B < P) {
        if constexpr (4 * B + 0 < P) z[4 * B + 0] = group_bcast<G, B>(mine[0]);
        if constexpr (4 * B + 1 < P) z[4 * B + 1] = group_bcast<G, B>(mine[1]);
        if constexpr (4 * B + 2 < P) z[4 * B + 2] = group_bcast<G, B>(mine[2]);
        if constexpr (4 * B + 3 < P) z[4 * B + 3] = group_bcast<G, B>(mine[3]);
        bcast_blocks<T, P, G, B + 1>(mine, z);
    }
}

// pair-level split: lane 2b+h of the group owns Box-Muller pair h of block b
template <typename T, int P, int G, int J = 0>
__device__ __forceinline__ void bcast_pairs(const T (&mine)[2], T (&z)[P]) {
    if constexpr (2 * J < P) {
        z[2 * J] = group_bcast<G, J>(mine[0]);
        if constexpr (2 * J + 1 < P) z[2 * J + 1] = group_bcast<G, J>(mine[1]);
        bcast_pairs<T, P, G, J + 1>(mine, z);
    }
}

template <typename T, int P, int G>
__device__ __forceinline__ void draw_group(uint64_t seed, uint64_t chain, uint64_t iter, int gl, T (&z)[P], T& logu) {
    constexpr int NB = (P + 3) / 4;   // normal blocks
    constexpr int NP = (P + 1) / 2;   // Box-Muller pairs
    if constexpr (G >= NP + 1) {
        // lane j < NP: pair j (block j/2, words 2(j&1), 2(j&1)+1); lane NP: the accept uniform.
        // Per lane: 1 Philox + 1 Box-Muller + 1 log.  Lanes beyond NP recompute lane 0's work (never read).
        const int j = gl > NP ? 0 : gl;
        const bool is_u = j == NP;
        const uint32_t blk = is_u ? TAG_UNIFORM : (uint32_t)(j >> 1);
        const U4 w = philox4x32_10((uint32_t)chain, (uint32_t)iter, (uint32_t)(iter >> 32), blk, (uint32_t)seed,
                                   (uint32_t)(seed >> 32));
        const uint32_t wa = (j & 1) ? w.z : w.x, wb = (j & 1) ? w.w : w.y;
        T mine[2];
        box_muller(wa, wb, mine[0], mine[1]);
        T lu;
        if constexpr (sizeof(T) == 4) lu = log_u01(w.x);
        else lu = log(u01<double>(w.x));
        bcast_pairs<T, P, G>(mine, z);
        logu = group_bcast<G, NP>(lu);
    } else if constexpr (G >= NB + 1) {
        const bool is_u = gl == NB;  // lanes beyond NB recompute block 0; their values are never read
        const uint32_t blk = is_u ? TAG_UNIFORM : (uint32_t)(gl > NB ? 0 : gl);
        const U4 w = philox4x32_10((uint32_t)chain, (uint32_t)iter, (uint32_t)(iter >> 32), blk, (uint32_t)seed,
                                   (uint32_t)(seed >> 32));
        T mine[4];
        box_muller(w.x, w.y, mine[0], mine[1]);
        box_muller(w.z, w.w, mine[2], mine[3]);
        T lu;
        if constexpr (sizeof(T) == 4) lu = log_u01(w.x);
        else lu = log(u01<double>(w.x));
        bcast_blocks<T, P, G>(mine, z);
        logu = group_bcast<G, NB>(lu);
    } else {
        draw_normals<T, P>(seed, chain, iter, z);
        logu = draw_log_uniform<T>(seed, chain, iter);
    }
}

// Draws for SEVERAL consecutive iterations at once, in two kinds of generator pass.  An iteration needs NBn = ceil(P/4) NORMAL
// blocks (Philox + two Box-Muller pairs each) and one UNIFORM block (Philox + one log; only word 0 is used).  The G lanes of a group
// all execute the generator anyway, so
//   a normals pass:   lane gl computes normal block gl % NBn of iteration base + gl / NBn    -> NIn = G / NBn iterations per pass
//   a uniforms pass:  lane gl computes the uniform block of iteration base + gl               -> NIu = G iterations per pass
// and every iteration then only collects its values from the lanes that hold them (ds_bpermute: a runtime lane index, so the chain
// loop is not unrolled).  Roles are uniform across the wave inside a pass, so no lane runs a Box-Muller whose result nobody reads
// (round 3 mixed the two kinds in one pass: a third of the lanes did; per 16 iterations at G = 16, P = 8: 3.2 passes of ~220
// instructions then, 2 + 1 passes of ~135 / ~105 now).  The VALUES are those of draw_normals / draw_log_uniform bit for bit: only who
// computes them, and when, changes.
template <typename T, int P, int G> struct DrawBatch {
    static constexpr int NBn = (P + 3) / 4, NIn = G / NBn, NIu = G;
    static constexpr bool kEnabled = NIn >= 2;
    T mine[4];   // this lane's four normals of its normal block
    T lu;        // log(u) of this lane's uniform block
    int posn, posu;  // iterations already served from the current normals / uniforms pass
    __device__ __forceinline__ void reset() { posn = NIn; posu = NIu; }
    __device__ __forceinline__ void refill_normals(uint64_t seed, uint64_t chain, uint64_t iter_base, int gl) {
        const int io = gl / NBn, b = gl - io * NBn;  // (lanes beyond NIn * NBn run ahead of what is read: harmless)
        const uint64_t it = iter_base + (uint64_t)io;
        const U4 w = philox4x32_10((uint32_t)chain, (uint32_t)it, (uint32_t)(it >> 32), (uint32_t)b, (uint32_t)seed, (uint32_t)(seed >> 32));
        box_muller(w.x, w.y, mine[0], mine[1]);
        box_muller(w.z, w.w, mine[2], mine[3]);
        posn = 0;
    }
    __device__ __forceinline__ void refill_uniforms(uint64_t seed, uint64_t chain, uint64_t iter_base, int gl) {
        const uint64_t it = iter_base + (uint64_t)gl;
        const U4 w = philox4x32_10((uint32_t)chain, (uint32_t)it, (uint32_t)(it >> 32), TAG_UNIFORM, (uint32_t)seed, (uint32_t)(seed >> 32));
        if constexpr (sizeof(T) == 4) lu = log_u01(w.x);
        else lu = log(u01<double>(w.x));
        posu = 0;
    }
    static __device__ __forceinline__ float fetch(float v, int lane_byte) {
        return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(lane_byte, __builtin_bit_cast(int, v)));
    }
    static __device__ __forceinline__ double fetch(double v, int lane_byte) {
        const uint64_t bits = __builtin_bit_cast(uint64_t, v);
        const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute(lane_byte, (int)(uint32_t)bits);
        const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute(lane_byte, (int)(uint32_t)(bits >> 32));
        return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
    }
    // (iterations must be requested consecutively; both refills are wave-uniform)
    __device__ __forceinline__ void advance_to(uint64_t seed, uint64_t chain, uint64_t iter, int gl) {
        if (posn >= NIn) refill_normals(seed, chain, iter, gl);
        if (posu >= NIu) refill_uniforms(seed, chain, iter, gl);
    }
    // the two normals of coordinates 2q, 2q + 1 and log(u) only (state distributed over the group: k_chain_rs16)
    __device__ __forceinline__ void next_pair(uint64_t seed, uint64_t chain, uint64_t iter, int gl, int q, T& zx, T& zy, T& logu) {
        advance_to(seed, chain, iter, gl);
        const int gbase = (int)(threadIdx.x & 63) - gl;  // first lane of the group
        const int src = (gbase + posn * NBn + (q >> 1)) * 4;  // block (2q) / 4 of this iteration
        const T e0 = fetch(mine[0], src), e1 = fetch(mine[1], src), e2 = fetch(mine[2], src), e3 = fetch(mine[3], src);
        zx = (q & 1) ? e2 : e0;
        zy = (q & 1) ? e3 : e1;
        logu = fetch(lu, (gbase + posu) * 4);
        ++posn;
        ++posu;
    }
    // the normals of coordinates r + 16 k (k < NK) and log(u) only: state distributed over the 16 lanes of a DPP row (k_chain_dist)
    template <int NK>
    __device__ __forceinline__ void next_own(uint64_t seed, uint64_t chain, uint64_t iter, int gl, int r, T (&z)[NK], T& logu) {
        advance_to(seed, chain, iter, gl);
        const int gbase = (int)(threadIdx.x & 63) - gl;
#pragma unroll
        for (int k = 0; k < NK; ++k) {
            const int src = (gbase + posn * NBn + (r >> 2) + 4 * k) * 4;  // block (r + 16 k) / 4 of this iteration
            const T e0 = fetch(mine[0], src), e1 = fetch(mine[1], src), e2 = fetch(mine[2], src), e3 = fetch(mine[3], src);
            const T lo = (r & 1) ? e1 : e0, hi = (r & 1) ? e3 : e2;
            z[k] = (r & 2) ? hi : lo;
        }
        logu = fetch(lu, (gbase + posu) * 4);
        ++posn;
        ++posu;
    }
    // z[0..P) and log(u) of iteration `iter`
    __device__ __forceinline__ void next(uint64_t seed, uint64_t chain, uint64_t iter, int gl, T (&z)[P], T& logu) {
        advance_to(seed, chain, iter, gl);
        const int gbase = (int)(threadIdx.x & 63) - gl;
        const int base = (gbase + posn * NBn) * 4;  // byte address of the lane that holds block 0 of this iteration
#pragma unroll
        for (int j = 0; j < P; ++j) z[j] = fetch(mine[j & 3], base + 4 * (j >> 2));
        logu = fetch(lu, (gbase + posu) * 4);
        ++posn;
        ++posu;
    }
};

// ------------------------------------------------------------------------------------------
// fast scalar math for the hot loop
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float fma_t(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_t(double a, double b, double c) { return __builtin_fma(a, b, c); }
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
__device__ __forceinline__ double fast_exp(double x) { return exp(x); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
// 1 / x for x in [1, inf]: hardware seed + two Newton steps (6 instructions; the IEEE division the compiler expands `1.0 / x` to is
// 11: v_div_scale x2, v_rcp, 5 fma, v_div_fmas, v_div_fixup).  Error a few ulp -- far below what the tests ask of the float64 path
// (1e-11).  x = inf (exp overflowed) is clamped so that the Newton residual stays finite: the result is 1e-300 instead of 0.
__device__ __forceinline__ double fast_rcp(double x) {
    x = __builtin_fmin(x, 1e300);
    double r = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-x, r, 1.0);
    return __builtin_fma(r, e, r);
}
// exp(x) for x in [-745, 709] WITHOUT the range guards of the library function (its result select -- a compare and two v_cndmask per
// call -- and nothing else is what separates the two: the same reduction x = n ln 2 + r, |r| <= ln 2 / 2, the same degree-11 minimax
// polynomial, v_ldexp).  Below -745 the scaling underflows to 0 by itself as long as n fits an int and r stays a reduced argument: callers
// clamp at -750 where the argument is unbounded (row_term).  19 instructions, < 1 ulp.
__device__ __forceinline__ double exp_noguard(double x) {
    const double n = __builtin_rint(x * 1.4426950408889634);
    double r = __builtin_fma(n, -6.93147180559945286227e-01, x);
    r = __builtin_fma(n, -2.31904681384629955842e-17, r);
    double p = __builtin_bit_cast(double, 0x3e5ade156a5dcb37ull);
    p = __builtin_fma(p, r, __builtin_bit_cast(double, 0x3e928af3fca7ab0cull));
    p = __builtin_fma(p, r, __builtin_bit_cast(double, 0x3ec71dee623fde64ull));
    p = __builtin_fma(p, r, __builtin_bit_cast(double, 0x3efa01997c89e6b0ull));
    p = __builtin_fma(p, r, __builtin_bit_cast(double, 0x3f2a01a014761f6eull));
    p = __builtin_fma(p, r, __builtin_bit_cast(double, 0x3f56c16c1852b7b0ull));
    p = __builtin_fma(p, r, __builtin_bit_cast(double, 0x3f81111111122322ull));
    p = __builtin_fma(p, r, __builtin_bit_cast(double, 0x3fa55555555502a1ull));
    p = __builtin_fma(p, r, __builtin_bit_cast(double, 0x3fc5555555555511ull));
    p = __builtin_fma(p, r, __builtin_bit_cast(double, 0x3fe000000000000bull));
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return __builtin_amdgcn_ldexp(p, (int)n);
}
// log(1 + e) for e in [0, 1]
__device__ __forceinline__ float log1p_unit(float e) { return __builtin_amdgcn_logf(1.0f + e) * 0.693147180559945309f; }
// float64: the fdlibm log kernel (e_log.c: log(1 + f) = f - (f^2/2 - s (f^2/2 + R(s^2))), s = f / (2 + f), |f| < 0.4143) on 1 + e split
// as 2^k (1 + f), k in {0, 1}, with the rounding error of 1 + e carried as c / (1 + e).  ~45 instructions where the general-domain
// library log1p takes ~140; 0.84 ulp against 80-bit arithmetic over 4e6 points of (0, 1] (e = 0 and subnormal e included).
__device__ __forceinline__ double log1p_unit(double e) {
    const double s = 1.0 + e, c = e - (s - 1.0);  // s - 1 is exact (Sterbenz), so c is the rounding error of s exactly
    const bool big = s > 1.4142135623730951;
    const double m = big ? 0.5 * s : s, k = big ? 1.0 : 0.0;
    const double f = m - 1.0, hfsq = 0.5 * f * f, d = 2.0 + f;
    double r = __builtin_amdgcn_rcp(d);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    r = __builtin_fma(__builtin_fma(-d, r, 1.0), r, r);
    double q = f * r;
    q = __builtin_fma(__builtin_fma(-d, q, f), r, q);
    const double z = q * q, w = z * z;
    const double t1 = w * __builtin_fma(w, __builtin_fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
    const double t2 = z * __builtin_fma(w, __builtin_fma(w, __builtin_fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01), 2.857142874366239149e-01),
                                        6.666666666666735130e-01);
    const double R = t2 + t1;
    const double lo = __builtin_fma(k, 1.90821492927058770002e-10, c * __builtin_amdgcn_rcp(s));
    return k * 6.93147180369123816490e-01 - ((hfsq - __builtin_fma(q, hfsq + R, lo)) - f);
}

// elementwise helpers written on explicit 2-vectors (v_pk_fma_f32 / v_pk_mul_f32) for float: the
// library is built with -fno-slp-vectorize so that the DPP reduction adds stay fused
// (v_add_f32_dpp instead of v_mov_b32_dpp + v_pk_add_f32), hence packing is spelled out here.
typedef float f32x2 __attribute__((ext_vector_type(2)));
// y[j] = a[j] * x[j] + y[j]
template <typename T, int P> __device__ __forceinline__ void vfma_v(const T (&a)[P], const T (&x)[P], T (&y)[P]) {
    if constexpr (sizeof(T) == 4 && P % 2 == 0) {
#pragma unroll
        for (int j = 0; j < P; j += 2) {
            const f32x2 r = __builtin_elementwise_fma(f32x2{a[j], a[j + 1]}, f32x2{x[j], x[j + 1]}, f32x2{y[j], y[j + 1]});
            y[j] = r.x;
            y[j + 1] = r.y;
        }
    } else {
#pragma unroll
        for (int j = 0; j < P; ++j) y[j] = fma_t(a[j], x[j], y[j]);
    }
}
// y[j] = s * x[j] + y[j]
template <typename T, int P> __device__ __forceinline__ void vfma_s(T s, const T (&x)[P], T (&y)[P]) {
    if constexpr (sizeof(T) == 4 && P % 2 == 0) {
        const f32x2 s2 = {s, s};
#pragma unroll
        for (int j = 0; j < P; j += 2) {
            const f32x2 r = __builtin_elementwise_fma(s2, f32x2{x[j], x[j + 1]}, f32x2{y[j], y[j + 1]});
            y[j] = r.x;
            y[j + 1] = r.y;
        }
    } else {
#pragma unroll
        for (int j = 0; j < P; ++j) y[j] = fma_t(s, x[j], y[j]);
    }
}
// out[j] = c[j] - a[j] * b[j]
template <typename T, int P>
__device__ __forceinline__ void vnmsub(const T (&a)[P], const T (&b)[P], const T (&c)[P], T (&out)[P]) {
    if constexpr (sizeof(T) == 4 && P % 2 == 0) {
#pragma unroll
        for (int j = 0; j < P; j += 2) {
            const f32x2 r = __builtin_elementwise_fma(-f32x2{a[j], a[j + 1]}, f32x2{b[j], b[j + 1]}, f32x2{c[j], c[j + 1]});
            out[j] = r.x;
            out[j + 1] = r.y;
        }
    } else {
#pragma unroll
        for (int j = 0; j < P; ++j) out[j] = c[j] - a[j] * b[j];
    }
}
// out[j] = a[j] * x[j] + y[j]
template <typename T, int P>
__device__ __forceinline__ void vfma_o(const T (&a)[P], const T (&x)[P], const T (&y)[P], T (&out)[P]) {
    if constexpr (sizeof(T) == 4 && P % 2 == 0) {
#pragma unroll
        for (int j = 0; j < P; j += 2) {
            const f32x2 r = __builtin_elementwise_fma(f32x2{a[j], a[j + 1]}, f32x2{x[j], x[j + 1]}, f32x2{y[j], y[j + 1]});
            out[j] = r.x;
            out[j + 1] = r.y;
        }
    } else {
#pragma unroll
        for (int j = 0; j < P; ++j) out[j] = fma_t(a[j], x[j], y[j]);
    }
}
// sum_j c[j] * ((u[j] - v[j])^2 - (w[j] - z[j])^2)     (MALA proposal-density difference)
template <typename T, int P>
__device__ __forceinline__ T vdiffsq(const T (&c)[P], const T (&u)[P], const T (&v)[P], const T (&w)[P], const T (&z)[P]) {
    if constexpr (sizeof(T) == 4 && P % 2 == 0) {
        f32x2 acc = {0.0f, 0.0f};
#pragma unroll
        for (int j = 0; j < P; j += 2) {
            const f32x2 d1 = f32x2{u[j], u[j + 1]} - f32x2{v[j], v[j + 1]};
            const f32x2 d2 = f32x2{w[j], w[j + 1]} - f32x2{z[j], z[j + 1]};
            const f32x2 t = __builtin_elementwise_fma(-d2, d2, d1 * d1);
            acc = __builtin_elementwise_fma(f32x2{c[j], c[j + 1]}, t, acc);
        }
        return acc.x + acc.y;
    } else {
        T acc = T(0);
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const T d1 = u[j] - v[j], d2 = w[j] - z[j];
            acc = fma_t(c[j], d1 * d1 - d2 * d2, acc);
        }
        return acc;
    }
}
// MALA's proposal-density difference without keeping advance(x) and advance(prop) as arrays across the evaluation:
//   sum_j c[j] * ((x[j] - advp_j)^2 - (xp[j] - advx_j)^2),  advx_j = a[j] g[j] + x[j],  advp_j = a[j] gp[j] + xp[j]
// -- the same operations in the same order as vfma_o + vdiffsq on stored arrays (advx is recomputed from unchanged inputs: bit-
// identical), two P-vectors fewer live across the evaluation (float64 at padded p = 32: 128 VGPRs, the difference between spilling
// to scratch and not).
template <typename T, int P>
__device__ __forceinline__ T mala_dq(const T (&a)[P], const T (&c)[P], const T (&x)[P], const T (&g)[P], const T (&xp)[P], const T (&gp)[P]) {
    if constexpr (sizeof(T) == 4 && P % 2 == 0) {
        f32x2 acc = {0.0f, 0.0f};
#pragma unroll
        for (int j = 0; j < P; j += 2) {
            const f32x2 a2 = {a[j], a[j + 1]}, x2 = {x[j], x[j + 1]}, xp2 = {xp[j], xp[j + 1]};
            const f32x2 advx = __builtin_elementwise_fma(a2, f32x2{g[j], g[j + 1]}, x2);
            const f32x2 advp = __builtin_elementwise_fma(a2, f32x2{gp[j], gp[j + 1]}, xp2);
            const f32x2 d1 = x2 - advp, d2 = xp2 - advx;
            const f32x2 t = __builtin_elementwise_fma(-d2, d2, d1 * d1);
            acc = __builtin_elementwise_fma(f32x2{c[j], c[j + 1]}, t, acc);
        }
        return acc.x + acc.y;
    } else {
        T acc = T(0);
#pragma unroll
        for (int j = 0; j < P; ++j) {
            const T advx = fma_t(a[j], g[j], x[j]), advp = fma_t(a[j], gp[j], xp[j]);
            const T d1 = x[j] - advp, d2 = xp[j] - advx;
            acc = fma_t(c[j], d1 * d1 - d2 * d2, acc);
        }
        return acc;
    }
}
// sum_j a[j] * x[j]^2
template <typename T, int P> __device__ __forceinline__ T vquad(const T (&a)[P], const T (&x)[P]) {
    if constexpr (sizeof(T) == 4 && P % 2 == 0) {
        f32x2 acc = {0.0f, 0.0f};
#pragma unroll
        for (int j = 0; j < P; j += 2) {
            const f32x2 xx = f32x2{x[j], x[j + 1]};
            acc = __builtin_elementwise_fma(xx * xx, f32x2{a[j], a[j + 1]}, acc);
        }
        return acc.x + acc.y;
    } else {
        T acc = T(0);
#pragma unroll
        for (int j = 0; j < P; ++j) acc = fma_t(x[j] * x[j], a[j], acc);
        return acc;
    }
}
// out[j] = s * x[j]
template <typename T, int P> __device__ __forceinline__ void vscale(T s, const T (&x)[P], T (&out)[P]) {
    if constexpr (sizeof(T) == 4 && P % 2 == 0) {
        const f32x2 s2 = {s, s};
#pragma unroll
        for (int j = 0; j < P; j += 2) {
            const f32x2 r = s2 * f32x2{x[j], x[j + 1]};
            out[j] = r.x;
            out[j + 1] = r.y;
        }
    } else {
#pragma unroll
        for (int j = 0; j < P; ++j) out[j] = s * x[j];
    }
}

// 16 lanes per chain, 8 values: REDUCE-SCATTER instead of the all-reduce above.  The butterfly all-reduce costs
// 8 adds per level x 4 levels = 32 v_add_f32_dpp and leaves all 8 totals in all 16 lanes -- where the caller then
// repeats the same leapfrog update 16 times.  Here the first two levels halve the values a lane carries (bank-masked
// DPP adds: a lane only receives the values it keeps), the last two finish two values per lane:
//   level 1  row_mirror       lanes 0-7 keep v0..v3, lanes 8-15 keep v4..v7           8 adds (bank_mask 0x3 / 0xc)
//   level 2  row_half_mirror  lanes 0-3 | 4-7 | 8-11 | 12-15 keep (v0,v1) | (v2,v3) | (v4,v5) | (v6,v7)   4 adds
//   level 3, 4  quad_perm [2,3,0,1], [1,0,3,2] on the two remaining values                                 4 adds
// 16 adds; every lane of quad q ends with the totals of values 2q and 2q + 1 (bit-identical in the quad's 4 lanes).
// Inline asm: the bank-masked form (disabled lanes keep the destination) has no builtin that the DPP combiner would
// fuse, and the compiler pads no hazards inside asm -- the leading s_nop and the instruction order below keep two
// wait states between a VALU write of a register and its DPP read.
__device__ __forceinline__ void group16_reduce_scatter8(const float (&v)[8], float& u0, float& u1) {
    float r0, r1, r2, r3, s0, s1;
    asm volatile(
        "s_nop 1\n\t"
        "v_add_f32_dpp %0, %8, %8 row_mirror row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %1, %9, %9 row_mirror row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %2, %10, %10 row_mirror row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %3, %11, %11 row_mirror row_mask:0xf bank_mask:0x3\n\t"
        "v_add_f32_dpp %0, %12, %12 row_mirror row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %1, %13, %13 row_mirror row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %2, %14, %14 row_mirror row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %3, %15, %15 row_mirror row_mask:0xf bank_mask:0xc\n\t"
        "v_add_f32_dpp %4, %0, %0 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %5, %1, %1 row_half_mirror row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %4, %2, %2 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %5, %3, %3 row_half_mirror row_mask:0xf bank_mask:0xa\n\t"
        "s_nop 0\n\t"
        "v_add_f32_dpp %6, %4, %4 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %7, %5, %5 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_add_f32_dpp %6, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %7, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"  // the caller's next instruction may be a DPP read of u1 (the compiler pads nothing after asm)
        : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(s0), "=&v"(s1), "=&v"(u0), "=&v"(u1)
        : "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3]), "v"(v[4]), "v"(v[5]), "v"(v[6]), "v"(v[7]));
}
// sum over the four quads of a 16-lane row of a value that is identical inside each quad; symmetric exchanges
// (i <-> 7 - i, then i <-> 15 - i), so every lane adds the same two numbers at each level: bit-identical in all 16
__device__ __forceinline__ float group16_quad_sum(float v) {
    v += dpp_mov<0x141>(v);  // row_half_mirror: quads 0 <-> 1, 2 <-> 3
    v += dpp_mov<0x140>(v);  // row_mirror:      quads 0 <-> 3, 1 <-> 2
    return v;
}
// the reverse: every lane of a 16-lane row gets the pair held by quad q (its lane 4q) for q = 0..3: 8 v_mov_b32_dpp row_share
__device__ __forceinline__ void group16_allgather_pairs(const f32x2& mine, f32x2 (&all)[4]) {
    const float fx = mine.x, fy = mine.y;  // (bit_cast straight from a vector-element lvalue reads element 0 for both)
    const int mx = __builtin_bit_cast(int, fx), my = __builtin_bit_cast(int, fy);
    all[0] = f32x2{__builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, mx, 0x150, 0xF, 0xF, true)),
                   __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, my, 0x150, 0xF, 0xF, true))};
    all[1] = f32x2{__builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, mx, 0x154, 0xF, 0xF, true)),
                   __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, my, 0x154, 0xF, 0xF, true))};
    all[2] = f32x2{__builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, mx, 0x158, 0xF, 0xF, true)),
                   __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, my, 0x158, 0xF, 0xF, true))};
    all[3] = f32x2{__builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, mx, 0x15C, 0xF, 0xF, true)),
                   __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, my, 0x15C, 0xF, 0xF, true))};
}

// min(a, b) that keeps a NaN: v_minimum3_f32 (gfx950).  The clamps in front of an exponential use it -- fminf returns its OTHER operand
// for a NaN, which turned the logit of a beta with a NaN (or an infinity against a zero x entry) into 100 and the row's value term into
// 0, the best possible, where the reference (fit-np-hmc.py:23-24) gives NaN.  One instruction, as v_min_f32.
__device__ __forceinline__ float min_keep_nan(float a, float b) { return __builtin_elementwise_minimum(a, b); }

// exp(t) given ts = t * kScale<T> (float: the log2(e) factor is folded into beta once per
// evaluation instead of once per row; double: kScale = 1)
template <typename T> struct ExpScale;
template <> struct ExpScale<float> {
    static constexpr float k = 1.44269504088896341f, inv = 0.693147180559945309f;
    static __device__ __forceinline__ float exp_scaled(float ts) { return __builtin_amdgcn_exp2f(ts); }
};
template <> struct ExpScale<double> {
    static constexpr double k = 1.0, inv = 1.0;
    static __device__ __forceinline__ double exp_scaled(double ts) { return exp(ts); }
};

// one signed row: accumulate sigma(-t) * xs into g[], and (VALUE) log sigma(t) into v.
// `bs` is beta * ExpScale<T>::k.
// FASTW (float64, gradient only): the weight as rcp(1 + exp(t)) -- HMC's INTERIOR evaluations, whose gradients never meet a
// value + gradient evaluation of the same point (chunk boundaries fall between iterations): the sign-symmetric form below costs a
// compare, two selects and a multiply more per row.
template <typename T, int P, bool VALUE, bool GRAD, bool FASTW = false>
__device__ __forceinline__ void row_term(const T (&xs)[P], const T (&bs)[P], T (&g)[P], T& v) {
    T ts;
    if constexpr (sizeof(T) == 4 && P % 2 == 0 && P >= 4) {
        // two interleaved partial dot products on v_pk_fma_f32: P/2 packed ops + 1 add instead of P
        // serial v_fmac.  A single wave per SIMD is issue-limited (one VALU op per ~4-6 cycles whatever
        // its width), so at 4096 chains packed ops are nearly free extra work per issue slot.
        typedef float f2 __attribute__((ext_vector_type(2)));
        f2 acc = f2{xs[0], xs[1]} * f2{bs[0], bs[1]};
#pragma unroll
        for (int j = 2; j < P; j += 2) acc = __builtin_elementwise_fma(f2{xs[j], xs[j + 1]}, f2{bs[j], bs[j + 1]}, acc);
        ts = acc.x + acc.y;
    } else {
        ts = xs[0] * bs[0];
#pragma unroll
        for (int j = 1; j < P; ++j) ts = fma_t(xs[j], bs[j], ts);
    }
    if constexpr (sizeof(T) == 8) {
        // float64: ONE exponential, e = exp(-|t|) in (0, 1], serves value and gradient -- sigma(-t) = e / (1 + e) for t > 0, 1 / (1 + e)
        // otherwise; log sigma(t) = min(t, 0) - log1p(e) -- and 1 + e in (1, 2] needs no overflow guard in the reciprocal (was: exp(t),
        // exp(-|t|), a clamped reciprocal and the library log1p per row: 220 instructions per row with value and gradient, 114 now).
        // The gradient-only form uses the SAME weights (3 instructions more than rcp(1 + exp(t)) there): a chunked MALA run starts
        // each launch with a gradient-only evaluation of a state whose gradient the previous launch took from a value + gradient one.
        if constexpr (FASTW && GRAD && !VALUE) {
            // (clamped: exp stays finite -- 1 + e <= 1e304, no guard in the reciprocal -- and the reduction stays a reduction; fmax / fmin
            // swallow a NaN: the weight of a non-finite t is sigma(750) here, and the end point of the trajectory, evaluated on the
            // exact path below, is what rejects such a state)
            const T s1 = T(1) + exp_noguard(__builtin_fmin(__builtin_fmax(ts, T(-750)), T(700)));
            // 1 / s1 in three fma: v_rcp_f64 is good to ~2^-23, and r0 (1 + e + e^2) with e = 1 - s1 r0 converges cubically (2^-69:
            // below the final rounding) where two Newton steps take four
            const T r0 = __builtin_amdgcn_rcp(s1);
            const T err = __builtin_fma(-s1, r0, T(1));
            const T r = __builtin_fma(r0, __builtin_fma(err, err, err), r0);
#pragma unroll
            for (int j = 0; j < P; ++j) g[j] = fma_t(r, xs[j], g[j]);
            return;
        }
        // (the clamp is a select, not fmax: fmax returns its other operand for a NaN, which made e = 0, w = 1 and the value term 0 -- the
        // BEST possible log-likelihood and a finite gradient at a beta with a NaN, or an infinity against a zero x entry, where the
        // reference and the float32 path give NaN.  Two instructions per row on the value + gradient path only: HMC's interior force
        // above keeps fmax / fmin -- a non-finite state is judged at the trajectory's end point, on this path)
        const T na = -__builtin_fabs(ts);
        const T e = exp_noguard(na < T(-750) ? T(-750) : na);
        if constexpr (GRAD) {
            const T s1 = T(1) + e;
            T r = __builtin_amdgcn_rcp(s1);
            T err = __builtin_fma(-s1, r, T(1));
            r = __builtin_fma(r, err, r);
            err = __builtin_fma(-s1, r, T(1));
            r = __builtin_fma(r, err, r);
            const T w = ts > T(0) ? e * r : r;
#pragma unroll
            for (int j = 0; j < P; ++j) g[j] = fma_t(w, xs[j], g[j]);
        }
        if constexpr (VALUE) v += (ts < T(0) ? ts : T(0)) - log1p_unit(e);
        return;
    }
    if constexpr (GRAD) {
        const T w = fast_rcp(T(1) + ExpScale<T>::exp_scaled(ts));  // sigma(-t); exp overflow -> rcp(inf) = 0
        if constexpr (sizeof(T) == 4 && P % 2 == 0) {
            typedef float f2 __attribute__((ext_vector_type(2)));
            const f2 w2 = {w, w};
#pragma unroll
            for (int j = 0; j < P; j += 2) {
                const f2 r = __builtin_elementwise_fma(w2, f2{xs[j], xs[j + 1]}, f2{g[j], g[j + 1]});
                g[j] = r.x;
                g[j + 1] = r.y;
            }
        } else {
#pragma unroll
            for (int j = 0; j < P; ++j) g[j] = fma_t(w, xs[j], g[j]);
        }
    }
    if constexpr (VALUE) {
        // log sigma(t) = min(t,0) - log1p(exp(-|t|))  (stable for both signs)
        const T ats = ts < T(0) ? -ts : ts;
        v += (ts < T(0) ? ts * ExpScale<T>::inv : T(0)) - log1p_unit(ExpScale<T>::exp_scaled(-ats));
    }
}

// ------------------------------------------------------------------------------------------
// row residency policies.  for_each(f) calls f(xs[P]) for every row owned by this lane.
// ------------------------------------------------------------------------------------------
template <typename T, int P, int R, int G> struct RegRows {
    T x[R][P];
    int pad_rows;  // zero rows held by this lane (each contributes log sigma(0) = -log 2 to the value)
    __device__ __forceinline__ void load(const T* __restrict__ rows, int64_t n, int gl) {
        pad_rows = 0;
#pragma unroll
        for (int k = 0; k < R; ++k) {
            // branch-free (rows past the end read the last row and are masked): a guarded load is a basic block per element
            const int64_t i = gl + (int64_t)k * G, ic = i < n ? i : n - 1;
            if (i >= n) ++pad_rows;
#pragma unroll
            for (int j = 0; j < P; ++j) {
                const T v = rows[ic * P + j];
                x[k][j] = i < n ? v : T(0);
            }
        }
    }
    template <class F> __device__ __forceinline__ void for_each(F&& f) const {
#pragma unroll
        for (int k = 0; k < R; ++k) f(x[k]);
    }
    __device__ __forceinline__ T value_fixup() const { return T(pad_rows) * T(0.693147180559945309); }
};

// float rows in registers, stored as TWISTED ROW PAIRS.  For the row pair (A, B) = (2k, 2k+1) and the
// coordinate pair (j, j+1), j even, two 64-bit register pairs hold
//       q[k][j] = (A_j, B_{j+1})        q[k][j+1] = (A_{j+1}, B_j)
// so that BOTH contractions run on v_pk_fma_f32 without any horizontal add per row:
//   eta   (t_A, t_B) += q[k][j] * (b_j, b_{j+1})  +  q[k][j+1] * (b_{j+1}, b_j)      (op_sel swap of the beta pair)
//   grad  G[j/2] += q[k][j]   * (w_A, w_B)  =  (A_j w_A,     B_{j+1} w_B)  -> (g_j, g_{j+1})
//         H[j/2] += q[k][j+1] * (w_A, w_B)  =  (A_{j+1} w_A, B_j w_B)      -> (g_{j+1}, g_j)
//   once per evaluation:  (g_j, g_{j+1}) = G[j/2] + swap(H[j/2])                      (one v_pk_add per pair)
// and the two sigmoids of a pair share one v_pk_add: 10.5 VALU instructions per row instead of the 12
// of the coordinate-pair form (row_term: a horizontal add per row and a scalar +1 per sigmoid).  An odd
// R keeps its last row in coordinate-pair form (s[j/2] = (x_j, x_{j+1})); its gradient lands in G directly.
template <int P, int R, int G> struct RegRowPairs {
    typedef float f2 __attribute__((ext_vector_type(2)));
    static_assert(P % 2 == 0, "coordinate pairs");
    static constexpr int RP = R / 2;
    static constexpr bool ODD = (R & 1) != 0;
    f2 q[RP > 0 ? RP : 1][P];
    f2 s[P / 2];  // the unpaired last row (ODD)
    int pad_rows;
    // (S = double: a float64 model's rows rounded to float32 -- the interior leapfrog gradients of k_chain_mixed)
    template <typename S> __device__ __forceinline__ void load(const S* __restrict__ rows, int64_t n, int gl) {
        pad_rows = 0;
        auto at = [&](int k, int j) {  // element j of the lane's k-th row (zero beyond n); branch-free: clamped load, then mask
            const int64_t i = gl + (int64_t)k * G;
            const float v = (float)rows[(i < n ? i : n - 1) * P + j];
            return i < n ? v : 0.0f;
        };
#pragma unroll
        for (int k = 0; k < R; ++k)
            if (gl + (int64_t)k * G >= n) ++pad_rows;
#pragma unroll
        for (int k = 0; k < RP; ++k)
#pragma unroll
            for (int j = 0; j < P; j += 2) {
                q[k][j] = f2{at(2 * k, j), at(2 * k + 1, j + 1)};
                q[k][j + 1] = f2{at(2 * k, j + 1), at(2 * k + 1, j)};
            }
        if constexpr (ODD) {
#pragma unroll
            for (int j = 0; j < P; j += 2) s[j / 2] = f2{at(R - 1, j), at(R - 1, j + 1)};
        }
    }
    __device__ __forceinline__ float value_fixup() const { return float(pad_rows) * 0.693147180559945309f; }
};
template <class Rows> struct is_row_pairs { static constexpr bool value = false; };
template <int P, int R, int G> struct is_row_pairs<RegRowPairs<P, R, G>> { static constexpr bool value = true; };

// one twisted row pair (see RegRowPairs): q[j] = (A_j, B_{j+1}), q[j+1] = (A_{j+1}, B_j); bb[j/2] = (b_j, b_{j+1})
// is beta * ExpScale::k.  Gradient partial sums go to gp (straight) / hp (swapped), the value to vacc (log2 units).
//   value: log sigma(t) = ln2 * (ts - log2(1 + 2^ts)) with ts = t log2(e): the SAME 1 + 2^ts the gradient
//   needs, so the value costs one v_log per row on top.  ts is clamped at 100 (sigma(-t) < 2^-100 there:
//   nothing changes) so that 2^ts stays finite; for ts -> -inf the expression tends to ts exactly.
//   Cancellation at large ts costs an absolute ulp(ts) ~ 1e-6 per row, the size of the fp32 summation
//   error of the value itself.
// The row data may sit in VGPRs (RegRowPairs) or in SGPRs (ScalarRowPairs: scalar operands of the v_pk ops).
template <int P, bool VALUE, bool GRAD, bool PROD = false>
__device__ __forceinline__ void pair_term(const f32x2 (&q)[P], const f32x2 (&bb)[P / 2], f32x2 (&gp)[P / 2],
                                          f32x2 (&hp)[P / 2], f32x2& vacc, f32x2* pacc = nullptr) {
    typedef f32x2 f2;
    f2 ts = q[0] * bb[0];
    ts = __builtin_elementwise_fma(q[1], __builtin_shufflevector(bb[0], bb[0], 1, 0), ts);
#pragma unroll
    for (int j = 2; j < P; j += 2) {
        ts = __builtin_elementwise_fma(q[j], bb[j / 2], ts);
        ts = __builtin_elementwise_fma(q[j + 1], __builtin_shufflevector(bb[j / 2], bb[j / 2], 1, 0), ts);
    }
    // (the per-row form clamps ts so that 2^ts stays finite; the product form needs no clamp: an overflowing factor makes the lane's
    //  product infinite, which the caller detects and answers with the per-row form -- one v_min per row saved on the fast path)
    if constexpr (VALUE && !PROD) ts = f2{min_keep_nan(ts.x, 100.0f), min_keep_nan(ts.y, 100.0f)};
    const f2 d = f2{ExpScale<float>::exp_scaled(ts.x), ExpScale<float>::exp_scaled(ts.y)} + f2{1.0f, 1.0f};
    if constexpr (GRAD) {
        const f2 w = {fast_rcp(d.x), fast_rcp(d.y)};  // sigma(-t); exp overflow -> rcp(inf) = 0
#pragma unroll
        for (int j = 0; j < P; j += 2) {
            gp[j / 2] = __builtin_elementwise_fma(q[j], w, gp[j / 2]);
            hp[j / 2] = __builtin_elementwise_fma(q[j + 1], w, hp[j / 2]);
        }
    }
    if constexpr (VALUE && PROD) {  // sum of the ts, PRODUCT of the 1 + 2^ts: one v_log per lane instead of one per row (row_pairs_eval)
        vacc += ts;
        *pacc *= d;
    } else if constexpr (VALUE) {
        vacc += ts - f2{__builtin_amdgcn_logf(d.x), __builtin_amdgcn_logf(d.y)};
    }
}

// all rows of a RegRowPairs lane: gradient partial sums into gp[P/2] = (g_j, g_{j+1}) pairs, value into v.
// The value of a lane's rows is sum_i (ts_i - log2(1 + 2^ts_i)) = sum_i ts_i - log2(prod_i (1 + 2^ts_i)): the rows sit in registers, so
// the product of the lane's <= 16 factors replaces all but two of its quarter-rate v_log by full-rate multiplies (MALA at 13 rows
// per lane: 39 -> 28 transcendentals per lane and iteration).  Every factor is >= 1 (ts is NOT clamped on this path), so the
// product can only fail by OVERFLOW (positive logits summing beyond ~127 in one lane); a lane whose product overflowed takes the
// per-row form instead -- a per-LANE select, so a chain's value never depends on which chains share its wave; the per-row pass
// itself runs (wave-uniformly) only when some lane needs it.  Same cancellation class as the per-row form: ulp(sum ts) per lane.
template <int P, int R, int G, bool VALUE, bool GRAD>
__device__ __forceinline__ void row_pairs_eval(const RegRowPairs<P, R, G>& rows, const f32x2 (&bb)[P / 2],
                                               f32x2 (&gp)[P / 2], float& v) {
    typedef f32x2 f2;
    f2 hp[P / 2];
#pragma unroll
    for (int j = 0; j < P / 2; ++j) gp[j] = hp[j] = f2{0.0f, 0.0f};
    // (from 8 rows per lane: with 4 -- one chain per wave, a latency-bound launch -- the two logs saved do not pay for the
    //  overflow check: config 1 ran 6 % slower in the product form, MALA at 13 rows per lane 3 % faster)
    constexpr bool PROD = VALUE && R >= 8;
    f2 vacc = {0.0f, 0.0f}, pacc = {1.0f, 1.0f};
    float vs = 0.0f;
#pragma unroll
    for (int k = 0; k < RegRowPairs<P, R, G>::RP; ++k) pair_term<P, VALUE, GRAD, PROD>(rows.q[k], bb, gp, hp, vacc, &pacc);
    float ts_odd = 0.0f;
    if constexpr (RegRowPairs<P, R, G>::ODD) {
        f2 acc = rows.s[0] * bb[0];
#pragma unroll
        for (int j = 1; j < P / 2; ++j) acc = __builtin_elementwise_fma(rows.s[j], bb[j], acc);
        float ts = acc.x + acc.y;
        if constexpr (VALUE && !PROD) ts = min_keep_nan(ts, 100.0f);
        const float d = 1.0f + ExpScale<float>::exp_scaled(ts);
        if constexpr (GRAD) {
            const float w = fast_rcp(d);
#pragma unroll
            for (int j = 0; j < P / 2; ++j) gp[j] = __builtin_elementwise_fma(f2{w, w}, rows.s[j], gp[j]);
        }
        if constexpr (PROD) {
            vs = ts;
            pacc.y *= d;
            ts_odd = ts;
        } else if constexpr (VALUE) {
            vs = ts - __builtin_amdgcn_logf(d);
        }
    }
    if constexpr (VALUE && !PROD) v += ((vacc.x + vacc.y) + vs) * ExpScale<float>::inv;
    if constexpr (PROD) {
        float val = ((vacc.x + vacc.y) + vs) - (__builtin_amdgcn_logf(pacc.x) + __builtin_amdgcn_logf(pacc.y));
        const bool bad = !(val > -3.0e38f);  // a product overflowed (-inf), or NaN input
        if (__builtin_amdgcn_ballot_w64(bad) != 0) {  // rare: per-row logs, selected per lane
            f2 va = {0.0f, 0.0f}, g0[P / 2], h0[P / 2];
#pragma unroll
            for (int k = 0; k < RegRowPairs<P, R, G>::RP; ++k) pair_term<P, true, false, false>(rows.q[k], bb, g0, h0, va);
            float vo = 0.0f;
            if constexpr (RegRowPairs<P, R, G>::ODD) {
                const float tc = min_keep_nan(ts_odd, 100.0f);
                vo = tc - __builtin_amdgcn_logf(1.0f + ExpScale<float>::exp_scaled(tc));
            }
            const float safe = (va.x + va.y) + vo;
            val = bad ? safe : val;
        }
        v += val * ExpScale<float>::inv;
    }
    if constexpr (GRAD) {
#pragma unroll
        for (int j = 0; j < P / 2; ++j) gp[j] += __builtin_shufflevector(hp[j], hp[j], 1, 0);
    }
}

// The gradient-only row pass written BREADTH-FIRST: all dot products step by step across the row pairs, then all exponentials,
// all reciprocals, then the gradient accumulations.  Same arithmetic per row pair as pair_term (bit-identical sums: the order inside
// every accumulator chain is unchanged).  For kernels where the machine scheduler keeps the source order of the loop (k_chain_mixed:
// its max-ILP schedule of the leapfrog loop is discarded -- the float64 end-point code beside it puts the function over the register
// budget the scheduler reverts on -- and pair-by-pair source order costs 44 hazard no-ops and every dependent-issue stall per step).
template <int P, int R, int G>
__device__ __forceinline__ void row_pairs_grad_bf(const RegRowPairs<P, R, G>& rows, const f32x2 (&bb)[P / 2], f32x2 (&gp)[P / 2]) {
    typedef f32x2 f2;
    constexpr int RP = RegRowPairs<P, R, G>::RP;
    constexpr bool ODD = RegRowPairs<P, R, G>::ODD;
    f2 bs[P / 2];
#pragma unroll
    for (int j = 0; j < P / 2; ++j) bs[j] = __builtin_shufflevector(bb[j], bb[j], 1, 0);
    f2 ts[RP > 0 ? RP : 1], to = {0.0f, 0.0f};
#pragma unroll
    for (int k = 0; k < RP; ++k) ts[k] = rows.q[k][0] * bb[0];
    if constexpr (ODD) to = rows.s[0] * bb[0];
#pragma unroll
    for (int k = 0; k < RP; ++k) ts[k] = __builtin_elementwise_fma(rows.q[k][1], bs[0], ts[k]);
#pragma unroll
    for (int j = 2; j < P; j += 2) {
#pragma unroll
        for (int k = 0; k < RP; ++k) ts[k] = __builtin_elementwise_fma(rows.q[k][j], bb[j / 2], ts[k]);
        if constexpr (ODD) to = __builtin_elementwise_fma(rows.s[j / 2], bb[j / 2], to);
#pragma unroll
        for (int k = 0; k < RP; ++k) ts[k] = __builtin_elementwise_fma(rows.q[k][j + 1], bs[j / 2], ts[k]);
    }
    f2 d[RP > 0 ? RP : 1];
#pragma unroll
    for (int k = 0; k < RP; ++k) d[k] = f2{ExpScale<float>::exp_scaled(ts[k].x), ExpScale<float>::exp_scaled(ts[k].y)} + f2{1.0f, 1.0f};
    float wo = 0.0f;
    if constexpr (ODD) wo = fast_rcp(1.0f + ExpScale<float>::exp_scaled(to.x + to.y));
    f2 w[RP > 0 ? RP : 1];
#pragma unroll
    for (int k = 0; k < RP; ++k) w[k] = f2{fast_rcp(d[k].x), fast_rcp(d[k].y)};
    f2 hp[P / 2];
#pragma unroll
    for (int j = 0; j < P / 2; ++j) gp[j] = hp[j] = f2{0.0f, 0.0f};
#pragma unroll
    for (int k = 0; k < RP; ++k)
#pragma unroll
        for (int j = 0; j < P; j += 2) {
            gp[j / 2] = __builtin_elementwise_fma(rows.q[k][j], w[k], gp[j / 2]);
            hp[j / 2] = __builtin_elementwise_fma(rows.q[k][j + 1], w[k], hp[j / 2]);
        }
    if constexpr (ODD) {
#pragma unroll
        for (int j = 0; j < P / 2; ++j) gp[j] = __builtin_elementwise_fma(f2{wo, wo}, rows.s[j], gp[j]);
    }
#pragma unroll
    for (int j = 0; j < P / 2; ++j) gp[j] += __builtin_shufflevector(hp[j], hp[j], 1, 0);
}

// Rows staged in LDS sit kLdsRowPad<T> elements apart beyond P.  float64: a lane's 16-byte reads of a 64-byte row, 16 lanes of a group on
// 16 consecutive rows, hit the same 4 banks in every fourth lane (64 B = 16 banks: 4-way conflict; rocprofv3 on k_chain_mixed:
// SQ_LDS_BANK_CONFLICT = 61 % of SQ_LDS_IDX_ACTIVE); two doubles of padding put the 16 rows on 16 disjoint 4-bank windows (P = 4, 8,
// 16, 32: row pitch 12, 20, 36, 68 banks).
template <typename T> constexpr int kLdsRowPad = sizeof(T) == 8 ? 2 : 0;
template <typename T, int P, int G> struct StridedRows {  // LDS or global: same access code
    const T* base;  // row-major [n][ld]
    int64_t n;
    int gl;
    int ld = P;  // row pitch in elements (P in device memory, P + kLdsRowPad<T> in LDS)
    template <class F> __device__ __forceinline__ void for_each(F&& f) const { for_each_in(0, n, f); }
    // the lane's rows inside [lo, hi), lo a multiple of G
    template <class F> __device__ __forceinline__ void for_each_in(int64_t lo, int64_t hi, F&& f) const {
        // four rows are fetched before the first is used: with one load batch per row the loop is one LDS / L2
        // round trip per row, which nothing else in the wave covers
        // (256-byte rows -- float64 at padded p = 32 --: one at a time; two are 128 registers, beside the position and gradient
        //  32-vectors the whole file)
#ifndef LR_ROWS_AHEAD64
#define LR_ROWS_AHEAD64 4
#endif
        // (float64, rows of 64 bytes, HMC n = 200 p = 8 at 4096 chains on k_chain_f64x, rows ahead 1 / 2 / 3 / 4 / 8: 3.9 / 4.6 / 4.9 / 5.1 / 4.6e7
        //  it/s -- tools/gpu/r6_f64_rows_ahead.sh; two ping-pong buffers of two rows with the next pair's reads issued before the current
        //  pair's use measured 4.0e7: the compiler waits for the whole batch either way)
        constexpr int UB = sizeof(T) * P <= 64 ? (sizeof(T) == 8 ? LR_ROWS_AHEAD64 : 4) : (sizeof(T) * P <= 128 ? 2 : 1);
        const int64_t n = hi;
        int64_t i = lo + gl;
        for (; i + (UB - 1) * G < n; i += UB * G) {
            T xs[UB][P];
#pragma unroll
            for (int u = 0; u < UB; ++u)
#pragma unroll
                for (int j = 0; j < P; ++j) xs[u][j] = base[(i + u * G) * ld + j];
#pragma unroll
            for (int u = 0; u < UB; ++u) f(xs[u]);
        }
        for (; i < n; i += G) {
            T xs[P];
#pragma unroll
            for (int j = 0; j < P; ++j) xs[j] = base[i * ld + j];
            f(xs);
        }
    }
    __device__ __forceinline__ T value_fixup() const { return T(0); }
};

// Lane-per-chain (G = 1) with rows broadcast from the SCALAR unit: every lane of the wave needs
// the same row, so the row is fetched with s_load (constant address space -> SMEM through the
// scalar cache) into SGPRs and used as the scalar operand of v_fmac: no LDS, no VGPRs for data,
// no cross-lane reduction, nothing replicated.  [i0, i1) is the row slice this wave works on.
template <typename T, int P, int PF = 1> struct ScalarRows {
    typedef const __attribute__((address_space(4))) T* cptr;
    const T* base;
    int64_t i0, i1;
    // PF rows are fetched per wait (PF s_loads in flight): SMEM latency (~500 cycles from L2) is
    // paid once per PF rows.  The fused chain kernels use PF = 1 -- their uniform arguments already
    // fill the 102-SGPR budget and deeper prefetch spills to VGPR lanes (measured 2.5x slower);
    // the stepwise partial kernel has SGPRs to spare and uses PF = 4.
    template <class F> __device__ __forceinline__ void for_each(F&& f) const {
        cptr cb = (cptr)base;
        int64_t i = i0;
        if constexpr (PF > 1) {
            for (; i + PF <= i1; i += PF) {
                T xs[PF][P];
#pragma unroll
                for (int k = 0; k < PF; ++k)
#pragma unroll
                    for (int j = 0; j < P; ++j) xs[k][j] = cb[(i + k) * P + j];
#pragma unroll
                for (int k = 0; k < PF; ++k) f(xs[k]);
            }
        }
        for (; i < i1; ++i) {
            T xs[P];
#pragma unroll
            for (int j = 0; j < P; ++j) xs[j] = cb[i * P + j];
            f(xs);
        }
    }
    __device__ __forceinline__ T value_fixup() const { return T(0); }
};

// Lane-per-chain with the rows broadcast from the scalar unit as TWISTED ROW PAIRS: `base` is the pair image
// built at model creation ([ceil(n/2)][P] f32x2, layout of RegRowPairs; an odd n is closed with a zero row),
// [k0, k1) the pairs this wave works on.  Each pair is one s_load of 8 P bytes; its SGPR pairs are the
// scalar operands of the v_pk_fma: 10.5 instead of 12 VALU instructions per row (see RegRowPairs).
template <int P, int PF = 1> struct ScalarRowPairs {
    typedef const __attribute__((address_space(4))) f32x2* cptr;
    const float* base;
    int64_t k0, k1;
    int zero_rows;  // 1 when [k0, k1) ends with the zero row closing an odd n (value fix-up), else 0
    template <class F> __device__ __forceinline__ void for_each_pair(F&& f) const {
        cptr cb = (cptr)base;
        int64_t k = k0;
        if constexpr (PF > 1) {
            for (; k + PF <= k1; k += PF) {
                f32x2 q[PF][P];
#pragma unroll
                for (int u = 0; u < PF; ++u)
#pragma unroll
                    for (int j = 0; j < P; ++j) q[u][j] = cb[(k + u) * P + j];
#pragma unroll
                for (int u = 0; u < PF; ++u) f(q[u]);
            }
        }
        for (; k < k1; ++k) {
            f32x2 q[P];
#pragma unroll
            for (int j = 0; j < P; ++j) q[j] = cb[k * P + j];
            f(q);
        }
    }
    __device__ __forceinline__ float value_fixup() const { return float(zero_rows) * 0.693147180559945309f; }
};
template <class Rows> struct is_strided_rows { static constexpr bool value = false; };
template <typename T, int P, int G> struct is_strided_rows<StridedRows<T, P, G>> { static constexpr bool value = true; };
template <class Rows> struct is_scalar_rows { static constexpr bool value = false; };
template <typename T, int P, int PF> struct is_scalar_rows<ScalarRows<T, P, PF>> { static constexpr bool value = true; };
constexpr int kSeqRows = 512;  // rows one lane may sum sequentially in fp32 (lane-per-chain variants)
template <class Rows> struct is_scalar_pairs { static constexpr bool value = false; };
template <int P, int PF> struct is_scalar_pairs<ScalarRowPairs<P, PF>> { static constexpr bool value = true; };

// ------------------------------------------------------------------------------------------
// streaming posterior statistics (include/logreg_hip.h "Streaming statistics"): fold kept sample number `idx`
// of the statistics window into the running (mean, M2) of its batch slot.  stats [slots][C][2][p] doubles.
// Called once per kept sample by the lane(s) that own the chain's coordinates; `slot_of` returns the
// (mean, M2) pair of one coordinate.
struct StatsArgs {
    double* buf;       // null = off
    int64_t batch;     // kept samples per slot
    int64_t first;     // window index of the launch's first kept sample
};
__device__ __forceinline__ void stats_fold(double* mean_p, double* m2_p, int64_t k_in_slot, double inv_count, double x) {
    double mean = x, m2 = 0.0;
    if (k_in_slot != 0) {  // Welford
        mean = *mean_p;
        m2 = *m2_p;
        const double d = x - mean;
        mean += d * inv_count;
        m2 += d * (x - mean);
    }
    *mean_p = mean;
    *m2_p = m2;
}
template <typename T, int P>
__device__ __forceinline__ void stats_update(const StatsArgs& st, int64_t kept_in_launch, int64_t C, int64_t chain, int p,
                                             const T (&x)[P]) {
    const int64_t idx = st.first + kept_in_launch, b = idx / st.batch, k = idx - b * st.batch;
    const double inv = 1.0 / (double)(k + 1);
    double* s = st.buf + ((b * C + chain) * 2) * p;
#pragma unroll
    for (int j = 0; j < P; ++j)
        if (j < p) stats_fold(s + j, s + p + j, k, inv, (double)x[j]);
}

// ------------------------------------------------------------------------------------------
// log-posterior value / gradient of one chain, cooperatively over the G lanes of its group.
//   value = ll + lprior (double), grad = d/dbeta (T), both replicated in all lanes of the group.
// ------------------------------------------------------------------------------------------
template <typename T, int P> struct Prior {
    T inv_var[P];         // 1/sd^2 ; 0 for padded coordinates
    double lprior_const;  // sum_j ( -log sd_j - 0.5 log 2 pi ) over the real coordinates
};

// PRESCALED (gradient only): `beta` is already multiplied by ExpScale<T>::k and pr.inv_var divided by it
// (HMC carries the trajectory position in those units: no rescaling per evaluation); grad is d/dbeta.
template <typename T, int P, int G, bool VALUE, bool GRAD, bool PRESCALED = false, class Rows = void>
__device__ __forceinline__ void eval_lpost(const Rows& rows, const Prior<T, P>& pr, const T (&beta)[P], T (&grad)[P],
                                           double& ll, double& lprior) {
    static_assert(!(PRESCALED && VALUE), "the value pass takes the plain position");
    T g[P];
#pragma unroll
    for (int j = 0; j < P; ++j) g[j] = T(0);
    T v = T(0);
    double vwide = 0.0;  // value part already accumulated in fp64 (blocked lane-per-chain sums)
    T bs[P];
    if constexpr (PRESCALED) {
#pragma unroll
        for (int j = 0; j < P; ++j) bs[j] = beta[j];
    } else {
        vscale<T, P>(ExpScale<T>::k, beta, bs);
    }
    if constexpr (is_row_pairs<Rows>::value) {
        typedef float f2 __attribute__((ext_vector_type(2)));
        f2 bb[P / 2], gp[P / 2];
#pragma unroll
        for (int j = 0; j < P / 2; ++j) bb[j] = f2{bs[2 * j], bs[2 * j + 1]};
        row_pairs_eval<P, Rows::RP * 2 + (Rows::ODD ? 1 : 0), G, VALUE, GRAD>(rows, bb, gp, v);
#pragma unroll
        for (int j = 0; j < P / 2; ++j) {
            g[2 * j] = gp[j].x;
            g[2 * j + 1] = gp[j].y;
        }
    } else if constexpr (is_scalar_pairs<Rows>::value) {
        f32x2 bb[P / 2], gp[P / 2], hp[P / 2], vacc = {0.0f, 0.0f};
#pragma unroll
        for (int j = 0; j < P / 2; ++j) {
            bb[j] = f32x2{bs[2 * j], bs[2 * j + 1]};
            gp[j] = hp[j] = f32x2{0.0f, 0.0f};
        }
        // one lane sums ALL rows of its chain: beyond kSeqRows the fp32 running sums are flushed to fp64 block
        // by block (a forced lane-per-chain run at n = 9001 lost 3e-5 of lpost without it; the planner itself
        // picks this variant only for rows <= 16 KB)
        if (rows.k1 - rows.k0 <= kSeqRows / 2) {
            rows.for_each_pair([&](const f32x2(&q)[P]) { pair_term<P, VALUE, GRAD>(q, bb, gp, hp, vacc); });
            if constexpr (VALUE) v += (vacc.x + vacc.y) * ExpScale<float>::inv;
#pragma unroll
            for (int j = 0; j < P / 2; ++j) {
                g[2 * j] = gp[j].x + hp[j].y;
                g[2 * j + 1] = gp[j].y + hp[j].x;
            }
        } else {
            double gd[P], vd = 0.0;
#pragma unroll
            for (int j = 0; j < P; ++j) gd[j] = 0.0;
            Rows sub = rows;
            for (int64_t kb = rows.k0; kb < rows.k1; kb += kSeqRows / 2) {
                sub.k0 = kb;
                sub.k1 = kb + kSeqRows / 2 < rows.k1 ? kb + kSeqRows / 2 : rows.k1;
                sub.for_each_pair([&](const f32x2(&q)[P]) { pair_term<P, VALUE, GRAD>(q, bb, gp, hp, vacc); });
#pragma unroll
                for (int j = 0; j < P / 2; ++j) {
                    gd[2 * j] += (double)(gp[j].x + hp[j].y);
                    gd[2 * j + 1] += (double)(gp[j].y + hp[j].x);
                    gp[j] = hp[j] = f32x2{0.0f, 0.0f};
                }
                vd += (double)(vacc.x + vacc.y);
                vacc = f32x2{0.0f, 0.0f};
            }
#pragma unroll
            for (int j = 0; j < P; ++j) g[j] = (float)gd[j];
            if constexpr (VALUE) vwide = vd * (double)ExpScale<float>::inv;
        }
    } else if constexpr (G == 1 && sizeof(T) == 4 && is_scalar_rows<Rows>::value) {
        if (rows.i1 - rows.i0 <= kSeqRows) {
            rows.for_each([&](const T(&xs)[P]) { row_term<T, P, VALUE, GRAD, PRESCALED>(xs, bs, g, v); });
        } else {  // as above, for the row-at-a-time scalar form (P = 32)
            double gd[P], vd = 0.0;
#pragma unroll
            for (int j = 0; j < P; ++j) gd[j] = 0.0;
            Rows sub = rows;
            for (int64_t ib = rows.i0; ib < rows.i1; ib += kSeqRows) {
                sub.i0 = ib;
                sub.i1 = ib + kSeqRows < rows.i1 ? ib + kSeqRows : rows.i1;
                T vb = T(0);
                sub.for_each([&](const T(&xs)[P]) { row_term<T, P, VALUE, GRAD, PRESCALED>(xs, bs, g, vb); });
#pragma unroll
                for (int j = 0; j < P; ++j) {
                    gd[j] += (double)g[j];
                    g[j] = T(0);
                }
                vd += (double)vb;
            }
#pragma unroll
            for (int j = 0; j < P; ++j) g[j] = (T)gd[j];
            if constexpr (VALUE) vwide = vd;
        }
    } else if constexpr (sizeof(T) == 4 && is_strided_rows<Rows>::value) {
        if (rows.n <= (int64_t)kSeqRows * G) {
            rows.for_each([&](const T(&xs)[P]) { row_term<T, P, VALUE, GRAD, PRESCALED>(xs, bs, g, v); });
        } else {  // LDS / global rows with few lanes per chain (lds G = 1 reaches n = 4096 at p = 4): as above
            double gd[P], vd = 0.0;
#pragma unroll
            for (int j = 0; j < P; ++j) gd[j] = 0.0;
            for (int64_t ib = 0; ib < rows.n; ib += (int64_t)kSeqRows * G) {
                const int64_t ie = ib + (int64_t)kSeqRows * G < rows.n ? ib + (int64_t)kSeqRows * G : rows.n;
                T vb = T(0);
                rows.for_each_in(ib, ie, [&](const T(&xs)[P]) { row_term<T, P, VALUE, GRAD, PRESCALED>(xs, bs, g, vb); });
#pragma unroll
                for (int j = 0; j < P; ++j) {
                    gd[j] += (double)g[j];
                    g[j] = T(0);
                }
                vd += (double)vb;
            }
#pragma unroll
            for (int j = 0; j < P; ++j) g[j] = (T)gd[j];
            if constexpr (VALUE) vwide = vd;
        }
    } else {
        rows.for_each([&](const T(&xs)[P]) { row_term<T, P, VALUE, GRAD, PRESCALED>(xs, bs, g, v); });
    }
    if constexpr (GRAD) {
        if constexpr (sizeof(T) == 4) {
            group_sum_levels<G, P>(g);  // fused v_add_f32_dpp per level (-fno-slp-vectorize)
        } else {
#pragma unroll
            for (int j = 0; j < P; ++j) g[j] = group_sum<G>(g[j]);
        }
        vnmsub<T, P>(beta, pr.inv_var, g, grad);
    }
    if constexpr (VALUE) {
        ll = group_sum<G>((double)(v + rows.value_fixup()) + vwide);
        lprior = pr.lprior_const - 0.5 * (double)vquad<T, P>(pr.inv_var, beta);
    }
}

}  // namespace lr
