// lr_wide_persist.h -- PERSISTENT row-split trajectory kernel for wide models with few chains (BASELINE config 5:
// n = 4096, p = 128, 1024 chains per GPU): all L - 1 interior leapfrog steps of an HMC trajectory in ONE launch, the rows of
// the design resident in LDS for the whole launch, the row slices' partial gradients exchanged between the resident
// workgroups through device memory.
//
// Why.  With a launch per interior step (k_wide_partial_bf16r) a step costs 10.3 us at config 5 although its matrix work is
// 0.9 us (tools/stamps.py, round 3): 1.5 us kernel boundary, 2.0 us fused prologue (every one of a tile's 4 slice workgroups
// re-reads 48 KB of slice partials and state from beyond L2 at the ~15 B/clk a CU gets there), 4.6 us row loop (bound by the
// L2 -> LDS DMA of the slice, 256 KB per workgroup and step at ~55 GB/s per CU), 1.3 us reduction and stores.  None of that
// is arithmetic.  The 160 KB of LDS of the 256 CUs together hold the one-piece bf16 image of the design (1 MB) many times
// over, so here a workgroup loads ITS slice once per launch and keeps it; what is left per step is the matrix work, an
// in-workgroup reduction, and ONE hand-off per step between the S workgroups that share a set of chains.
//
// Geometry.  A workgroup (8 waves) = one GROUP of 32 chains (two 16-chain MFMA tiles) x one of S row SLICES; wave w takes
// the 32-row blocks w, w + 8, ... of the slice for both tiles (the eta operand read from LDS once serves both).  Per step:
//   1. row loop: per block and tile exactly the arithmetic of k_wide_partial_bf16r (X one bf16 piece, beta two, w one);
//   2. the 8 waves' gradients meet in LDS (4 rounds through a 32 KB scratch): thread t then OWNS the P / 16 consecutive
//      coordinates [(P / 16)(t % 16), ...) of chain t / 16 -- position, momentum and constants of those stay in its registers
//      for the whole trajectory;
//   3. publish: the workgroup's partial gradient -> xch[step parity][group][slice] with write-through (sc1) stores, every
//      wave drains its stores, barrier, ONE lane stores the flag flags[group][slice] = step + 1 (agent scope);
//   4. one wave polls the S flags of its group (relaxed agent loads, s_sleep between polls, BOUNDED), barrier;
//   5. every thread reads its coordinates of all S slices (sc1 loads), sums them in slice order (fp64, as k_tall_update does),
//      g = sum - q ivar, p += eps g, q += (eps / m) p: the S workgroups of a group do this redundantly and bit-identically,
//      so no second hand-off is needed; the new position goes to LDS, every wave rebuilds its beta operands.
// The exchange follows cdna_hip_programming.md Guideline 16 (R1): write-through payload + drained flag on the producer,
// relaxed poll + sc1 loads on the consumer; no fence, no reliance on dispatch order or XCD placement (the grid is laid
// out so that a group's S workgroups fall on one XCD -- blocks b, b + 8, ... -- which is a speed matter only).
// Two exchange buffers alternate by step parity: a workgroup publishes step k + 1 only after it has read step k from
// everybody, and step k + 2 (same buffer as k) only after everybody has published k + 1, i.e. finished reading k.
//
// Residency.  The hand-off needs all S workgroups of a group running at once.  The kernel takes > 80 KB of LDS, so there is
// one workgroup per CU, and the host launches it only with groups x S <= CUs on a device it has to itself; every spin is
// bounded all the same: on time-out the workgroup raises *xerr (which the next API call on the model reports), stops waiting
// and poisons its output with NaN, so a degraded run cannot pass for a result.
//
// Results depend on (n, p, S) only -- not on the chain count, the group a chain falls in, or timing.
//
// MEASURED (round 3, config 5, tools/stamps_persist.py): 12.8 us per evaluation against 10.3 us for the launch-per-step path it was
// to replace, so it is OPT-IN (LOGREG_WIDE_PERSIST=1).  Per step: row loop 3.5 us, the 8 waves' gradients through LDS 2.5 (128 KB
// of ds_write_b128 per step at ~79 B/clk/CU plus 8 barriers), publish 0.9, poll 1.2 (waiting for the slowest of the 8), gather
// 3.1 (128 KB per workgroup of write-through data, which a CU receives at ~15 B/clk -- the same fabric rate that makes the fused
// prologue of the launch-per-step kernel cost 2 us for 48 KB), operand build 0.5.  What the estimate behind it got wrong is those
// two rates: inside one launch or across launches, moving a 16 KB partial between 8 CUs costs more than the matrix work it feeds.
#pragma once
#include "lr_wide_bf16.h"

namespace lr {

constexpr int kPersistWaves = 8;
constexpr int kPersistChains = 32;            // chains per workgroup: two MFMA tiles
constexpr int kPersistScratchBytes = 32768;   // in-workgroup reduction scratch (aliases the position staging)
constexpr unsigned kPersistSpinLimit = 4u << 20;  // polls before giving up (each poll >= 100 ns: >= 0.4 s)

// bytes of LDS the kernel needs for `blocks_per_slice` 32-row blocks
template <int P> constexpr size_t persist_lds_bytes(int blocks_per_slice) {
    return (size_t)blocks_per_slice * WideBf16Geom<P>::BUF1 * 2 + kPersistScratchBytes;
}

template <int P>
__global__ void __launch_bounds__(64 * kPersistWaves) k_wide_traj_rs(TallArgs<float, P> a) {
    using G = WideBf16Geom<P>;
    constexpr int NW = kPersistWaves, NT = 64 * NW, BLK_EL = G::BUF1, BLK_BYTES = BLK_EL * 2;
    constexpr int EPT = kPersistChains * P / NT;  // elements (consecutive coordinates of one chain) a thread owns: 8 / 4
    static_assert(EPT == 4 || EPT == 8, "P = 64 or 128");
    constexpr int TPC = P / EPT;                  // threads per chain: 16
    constexpr int QROW = P + 4;                   // padded row of the position staging [32][P + 4] floats
    static_assert(kPersistChains * QROW * 4 <= kPersistScratchBytes, "position staging aliases the scratch");
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int S = a.traj_S, nbs = a.traj_nbs;     // slices per group, 32-row blocks per slice
    unsigned char* xs = lds;                                        // [nbs][BLK_BYTES]: the slice's block images
    float* scratch = reinterpret_cast<float*>(lds + (size_t)nbs * BLK_BYTES);
    // (the give-up flag of the polling wave, in the last word of the scratch: read right behind the barrier that follows the poll,
    //  before the scratch is used again; no static LDS -- config 5's slice + scratch are the whole 160 KB)
    volatile int* dead_s = reinterpret_cast<volatile int*>(lds + (size_t)nbs * BLK_BYTES + kPersistScratchBytes - 4);

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, kg = lane >> 4;
    // blocks b, b + 8, ... share a group (on this chip they also share an XCD and its L2: speed only)
    const int bid = blockIdx.x, ngroups = gridDim.x / S;
    int group = bid / S, slice = bid % S;
    if ((ngroups & 7) == 0 && !a.traj_scatter) {  // whole groups per XCD column
        const int xcd = bid & 7, idx = bid >> 3;
        group = xcd * (ngroups / 8) + idx / S;
        slice = idx % S;
    }
    const int64_t chain0 = (int64_t)group * kPersistChains;
    const int nblk = (int)((a.n + 31) / 32);
    const int b0 = slice * nbs, nb = nblk - b0 < 0 ? 0 : (nblk - b0 < nbs ? nblk - b0 : nbs);  // this slice's blocks

    // ---- the slice's block images -> LDS, once (LDS-DMA, 1 KB per wave-instruction)
    {
        const uint32_t xs_lds = (uint32_t)(uintptr_t)xs;
        const unsigned char* src = reinterpret_cast<const unsigned char*>(a.xblk1 + (int64_t)b0 * BLK_EL) + lane * 16;
        const int npiece = nb * (BLK_BYTES / 1024);
        for (int pc = wave; pc < npiece; pc += NW) {
            uint32_t keep;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(keep)
                         : "v"(src + (size_t)pc * 1024), "s"(xs_lds + (uint32_t)pc * 1024)
                         : "memory");
        }
    }
    // ---- the thread's share of the state: chain oc, coordinates oj .. oj + EPT - 1
    const int oc = tid / TPC, oj = (tid % TPC) * EPT;
    int64_t ochain = chain0 + oc;
    const bool olive = ochain < a.C;
    if (!olive) ochain = a.C - 1;
    float sq[EPT], sp[EPT], sb[EPT], si[EPT];
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        sq[e] = a.q1[ochain * P + oj + e];
        sp[e] = a.pm[ochain * P + oj + e];
        sb[e] = a.cvec[oj + e];
        si[e] = a.cvec[P + oj + e];
    }
    float* qst = scratch;  // position staging [32][QROW]
#pragma unroll
    for (int e = 0; e < EPT; ++e) qst[oc * QROW + oj + e] = sq[e];
    __builtin_amdgcn_s_waitcnt(0x0F70);  // the slice has landed (this wave's pieces; the barrier covers the others')
    __syncthreads();

    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(a.xch, 0, (int)a.xch_bytes, 0x00020000);
    const uint32_t grp_bytes = (uint32_t)S * kPersistChains * P * 4;        // one group's S partials
    const uint32_t par_bytes = (uint32_t)ngroups * grp_bytes;               // one parity's buffer
    const uint32_t own_off = (uint32_t)tid * (EPT * 4);                     // the thread's elements inside a partial
    typedef __attribute__((address_space(1))) uint32_t gu32;
    gu32* flags = (gu32*)(a.xflags + (size_t)group * S);

    const int eta_off = G::elem(kg, c, 0) & ~7;
    const int ri = (lane & 15) >> 2, ci = lane & 3;
    const int tr_off[2] = {G::elem(ci, 4 * kg + ri, 0), G::elem(ci, 4 * kg + ri, 4)};
    const int nsteps = a.l - 1;
    bool dead = false;
#ifdef LR_STAMPS  // development builds: 100 MHz ticks per phase, summed over the steps (tools/stamps.py)
    unsigned long long ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = __builtin_amdgcn_s_memrealtime();
#define LR_PHASE(k) do { const unsigned long long tn_ = __builtin_amdgcn_s_memrealtime(); ph[k] += tn_ - tprev; tprev = tn_; } while (0)
#else
#define LR_PHASE(k) do { } while (0)
#endif
    for (int s = 0; s < nsteps; ++s) {
        // ---- beta operands of the wave's two chain tiles from the staged position: hi + lo bf16 pieces, times log2(e)
        u32x4 bq[2][G::M32][2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int m = 0; m < G::M32; ++m) {
                const f32x4 v0 = *reinterpret_cast<const f32x4*>(&qst[(16 * t + c) * QROW + 32 * m + 8 * kg]);
                const f32x4 v1 = *reinterpret_cast<const f32x4*>(&qst[(16 * t + c) * QROW + 32 * m + 8 * kg + 4]);
                const float x[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                uint32_t hi[4], lo[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float x0 = x[2 * i] * ExpScale<float>::k, x1 = x[2 * i + 1] * ExpScale<float>::k;
                    hi[i] = pack_rne(x0, x1);
                    const float h0 = __builtin_bit_cast(float, hi[i] << 16), h1 = __builtin_bit_cast(float, hi[i] & 0xFFFF0000u);
                    lo[i] = pack_rne(x0 - h0, x1 - h1);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int j = (kg & 1) ? (i ^ 2) : i;  // odd kg: halves swapped, as the eta read delivers them
                    bq[t][m][0][i] = hi[j];
                    bq[t][m][1][i] = lo[j];
                }
            }
        __syncthreads();  // the staging area becomes the reduction scratch
        LR_PHASE(0);

        // ---- row loop: this wave's blocks of the resident slice, both chain tiles per block
        f32x4 gacc[2][G::MBP];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int mb = 0; mb < G::MBP; ++mb) gacc[t][mb] = f32x4{0, 0, 0, 0};
        for (int b = wave; b < nb; b += NW) {
            const uint16_t* base = reinterpret_cast<const uint16_t*>(xs + (size_t)b * BLK_BYTES);
            u32x4 wv[2];
#pragma unroll
            for (int T = 0; T < 2; ++T) {
                u32x4 xa[G::M32];
#pragma unroll
                for (int m = 0; m < G::M32; ++m) xa[m] = *reinterpret_cast<const u32x4*>(base + G::tile1(T, m) + eta_off);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    f32x4 e0 = {0, 0, 0, 0}, e1 = {0, 0, 0, 0};
#pragma unroll
                    for (int m = 0; m < G::M32; ++m) {
                        e0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(xa[m]), as_bf16x8(bq[t][m][0]), e0, 0, 0, 0);
                        e1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(xa[m]), as_bf16x8(bq[t][m][1]), e1, 0, 0, 0);
                    }
                    float w[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) w[r] = fast_rcp(1.0f + __builtin_amdgcn_exp2f(e0[r] + e1[r]));
                    wv[t][2 * T] = pack_rne(w[0], w[1]);  // K-slot 8 kg + 4 T + r <-> row 4 kg + r of row tile T
                    wv[t][2 * T + 1] = pack_rne(w[2], w[3]);
                }
            }
#pragma unroll
            for (int mb = 0; mb < G::MBP; ++mb) {
                const u32x2 t0 = lds_read_tr16(base + G::tile1(0, mb >> 1) + tr_off[mb & 1]);
                const u32x2 t1 = lds_read_tr16(base + G::tile1(1, mb >> 1) + tr_off[mb & 1]);
                const u32x4 xg = {t0[0], t0[1], t1[0], t1[1]};
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    gacc[t][mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(as_bf16x8(xg), as_bf16x8(wv[t]), gacc[t][mb], 0, 0, 0);
            }
        }

        LR_PHASE(1);
        // ---- the 8 waves' gradients meet in LDS: round (t, h) = chain tile t, coordinate half h; scratch [wave][16 chains][P / 2]
        // floats, the 16-byte slots of a row XOR-swizzled with the chain (unpadded rows are 0 banks apart: the 16 chains of a
        // b128 access would collide 16-fold; padding does not fit -- config 5's slice leaves exactly these 32 KB)
        constexpr int HW = P / 2, HMB = G::MBP / 2, SWZ = HW / 4 - 1;
        static_assert(NW * 16 * HW * 4 <= kPersistScratchBytes, "one round of all waves fits the scratch");
        auto sidx = [&](int w, int ch, int x) { return (w * 16 + ch) * HW + ((((x >> 2) ^ ch) & SWZ) << 2) + (x & 3); };
        float gs[EPT];
#pragma unroll
        for (int e = 0; e < EPT; ++e) gs[e] = 0.0f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int mh = 0; mh < HMB; ++mh) {  // gradient register (mb, r) of lane (c, kg) = coordinate 32 (mb >> 1) + 8 kg + 4 (mb & 1) + r
                    const int mb = HMB * h + mh;
                    *reinterpret_cast<f32x4*>(&scratch[sidx(wave, c, 32 * (mb >> 1) + 8 * kg + 4 * (mb & 1) - HW * h)]) = gacc[t][mb];
                }
                __syncthreads();
                if (oc / 16 == t && oj / HW == h) {  // the owners of this round's elements: wave order, deterministic
#pragma unroll
                    for (int e4 = 0; e4 < EPT; e4 += 4) {
                        f32x4 acc = *reinterpret_cast<const f32x4*>(&scratch[sidx(0, oc % 16, oj - HW * h + e4)]);
#pragma unroll
                        for (int w = 1; w < NW; ++w) acc += *reinterpret_cast<const f32x4*>(&scratch[sidx(w, oc % 16, oj - HW * h + e4)]);
#pragma unroll
                        for (int r = 0; r < 4; ++r) gs[e4 + r] = acc[r];
                    }
                }
                __syncthreads();
            }

        LR_PHASE(2);
        // ---- publish this workgroup's partial (write-through), drained, then the flag
        const uint32_t epoch = (uint32_t)s + 1u;
        const uint32_t pbase = (uint32_t)(s & 1) * par_bytes + (uint32_t)group * grp_bytes;
#pragma unroll
        for (int e4 = 0; e4 < EPT; e4 += 4)
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, f32x4{gs[e4], gs[e4 + 1], gs[e4 + 2], gs[e4 + 3]}), xrs,
                                                   pbase + (uint32_t)slice * (kPersistChains * P * 4) + own_off + e4 * 4, 0, 16);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // EVERY storing wave drains its write-through stores
        __syncthreads();
        if (tid == 0) __hip_atomic_store(flags + slice, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

        LR_PHASE(3);
        // ---- wait for the group's S partials of this step (one wave polls; bounded)
        if (tid == 0) *dead_s = dead ? 1 : 0;
        if (wave == 0 && !dead) {
            unsigned spins = 0;
            for (;;) {
                const uint32_t f = lane < S ? __hip_atomic_load(flags + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : epoch;
                // (a flag may already show a LATER step: that workgroup has read this step from everybody and moved on)
                if (__builtin_amdgcn_ballot_w64((int32_t)(f - epoch) < 0) == 0) break;
                if (++spins > kPersistSpinLimit) {
                    if (lane == 0) {
                        __hip_atomic_store((gu32*)a.xerr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);  // host memory
                        *dead_s = 1;
                    }
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        __syncthreads();
        dead = *dead_s != 0;
        LR_PHASE(4);

        // ---- gather: the thread's coordinates of all S partials, summed in slice order (fp64), then the leapfrog update
        {
            double sum[EPT];
#pragma unroll
            for (int e = 0; e < EPT; ++e) sum[e] = 0.0;
            constexpr int SB = 8;  // slices per batch of loads in flight
            for (int r0 = 0; r0 < S; r0 += SB) {
                u32x4 v[SB][EPT / 4];
#pragma unroll
                for (int r = 0; r < SB; ++r)
#pragma unroll
                    for (int e4 = 0; e4 < EPT / 4; ++e4)
                        v[r][e4] = __builtin_amdgcn_raw_buffer_load_b128(
                            xrs, pbase + (uint32_t)(r0 + r < S ? r0 + r : S - 1) * (kPersistChains * P * 4) + own_off + e4 * 16, 0, 16);
#pragma unroll
                for (int r = 0; r < SB; ++r)
                    if (r0 + r < S) {
#pragma unroll
                        for (int e4 = 0; e4 < EPT / 4; ++e4) {
                            const f32x4 f = __builtin_bit_cast(f32x4, v[r][e4]);
#pragma unroll
                            for (int q = 0; q < 4; ++q) sum[4 * e4 + q] += (double)f[q];
                        }
                    }
            }
#pragma unroll
            for (int e = 0; e < EPT; ++e) {
                const float g1 = (float)sum[e] - sq[e] * si[e];
                sp[e] = fma_t(a.step, g1, sp[e]);
                sq[e] = fma_t(sb[e], sp[e], sq[e]);
                qst[oc * QROW + oj + e] = sq[e];
            }
        }
        __syncthreads();
        LR_PHASE(5);
    }
#ifdef LR_STAMPS
    if (a.stamps && lane == 0)
        for (int k = 0; k < 6; ++k) a.stamps[((size_t)blockIdx.x * 16 + (threadIdx.x >> 6)) * 16 + k] = ph[k];
#endif
    if (olive && slice == 0) {
        const float poison = dead ? __builtin_nanf("") : 0.0f;
#pragma unroll
        for (int e = 0; e < EPT; ++e) {
            a.q1[ochain * P + oj + e] = sq[e] + poison;
            a.pm[ochain * P + oj + e] = sp[e] + poison;
        }
    }
}

}  // namespace lr
