// last_arriver_probe.hip -- config 5's interior step as a traffic skeleton: is it cheaper to let the LAST of a chain tile's four
// slice workgroups sum the tile's partials at the END of a launch (while they are L2-hot; the next launch's prologue then reads
// 8 KB of finished position instead of 48 KB of partials + state) than to do it in every workgroup's prologue as today?
// Same grid as k_wide_partial_bf16r at 1024 chains (64 tiles x 4 slices, 512 threads; a tile's slices share an XCD), same bytes:
//   variant 0 (today)         prologue: 4 slice partials + q + p (48 KB per workgroup), update arithmetic;  tail: store 8 KB partial
//   variant 1 (last arriver)  prologue: q (8 KB);  tail: write-through store of the 8 KB partial, drained, one relaxed agent-scope
//                             atomic per workgroup, the last arriver of the tile loads the other partials (sc1) + p (32 KB),
//                             updates, stores q, p (16 KB).  (A first version used __threadfence(): an agent-scope release
//                             writes the L2 back -- 38 us per launch.)
// Both spin T us in between (the row loop).  N dependent launches back to back; prints the period per launch minus T, and for
// variant 1 the tail's phases from the last arrivers' 100 MHz stamps.
//   hipcc --offload-arch=gfx950 -O3 tools/last_arriver_probe.hip -o tools/bin/last_arriver_probe && tools/bin/last_arriver_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int TILES = 64, SLICES = 4, P = 128, CH = 16, NT = 512;
constexpr int TILE_FLOATS = CH * P;  // 2048 floats = 8 KB per (tile, slice) partial; q and p per tile likewise

struct Bufs {
    float* part[2];  // [slice][tile][2048], ping-pong
    float* q[2];
    float* p[2];
    unsigned* count;  // [tile], monotonic
    unsigned long long* stamps;  // [tile][4]
};

typedef unsigned u4 __attribute__((ext_vector_type(4)));
// write-through store / L1-bypassing load (sc1) through a buffer resource, as the persistent kernel of round 3 used them
__device__ __forceinline__ void st_wt(__amdgpu_buffer_rsrc_t r, unsigned off, f4 v) { __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, v), r, off, 0, 16); }
__device__ __forceinline__ f4 ld_sc1(__amdgpu_buffer_rsrc_t r, unsigned off) { return __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 16)); }

template <int VARIANT>
__global__ void __launch_bounds__(NT) step(Bufs b, int cur, int ticks, unsigned launch_no) {
    const int tile = blockIdx.x, slice = blockIdx.y, tid = threadIdx.x;
    const int prev = cur ^ 1;
    // ---- prologue: every thread owns one f4 of the tile's 2048 floats
    f4 acc;
    if (VARIANT == 0) {
        f4 g[SLICES];
#pragma unroll
        for (int s = 0; s < SLICES; ++s) g[s] = reinterpret_cast<const f4*>(b.part[prev] + ((size_t)s * TILES + tile) * TILE_FLOATS)[tid];
        const f4 q = reinterpret_cast<const f4*>(b.q[prev] + (size_t)tile * TILE_FLOATS)[tid];
        f4 p = reinterpret_cast<const f4*>(b.p[prev] + (size_t)tile * TILE_FLOATS)[tid];
        f4 gs = g[0];
#pragma unroll
        for (int s = 1; s < SLICES; ++s) { gs.x += g[s].x; gs.y += g[s].y; gs.z += g[s].z; gs.w += g[s].w; }
        p.x += 0.01f * gs.x; p.y += 0.01f * gs.y; p.z += 0.01f * gs.z; p.w += 0.01f * gs.w;
        acc = f4{q.x + 0.02f * p.x, q.y + 0.02f * p.y, q.z + 0.02f * p.z, q.w + 0.02f * p.w};
        if (slice == 0) {
            reinterpret_cast<f4*>(b.q[cur] + (size_t)tile * TILE_FLOATS)[tid] = acc;
            reinterpret_cast<f4*>(b.p[cur] + (size_t)tile * TILE_FLOATS)[tid] = p;
        }
    } else {
        acc = reinterpret_cast<const f4*>(b.q[prev] + (size_t)tile * TILE_FLOATS)[tid];
    }
    // ---- the row loop (time only): the position must have arrived
    float sink = acc.x + acc.y + acc.z + acc.w;
    // (the spin starts when the prologue's data has arrived in every wave: the clock read depends on the loaded values)
    const unsigned long long ts = __builtin_amdgcn_s_memrealtime() + (unsigned long long)(__builtin_amdgcn_readfirstlane(sink == 1234.5f ? 1 : 0));
    __syncthreads();
    while (__builtin_amdgcn_s_memrealtime() - ts < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(1);
    // ---- tail: this workgroup's partial
    const f4 mine = {sink * 1e-3f, acc.y * 1e-3f, acc.z * 1e-3f, acc.w * 1e-3f + (float)slice};
    if (VARIANT == 0) {
        reinterpret_cast<f4*>(b.part[cur] + ((size_t)slice * TILES + tile) * TILE_FLOATS)[tid] = mine;
    } else {
        // no agent-scope FENCE: on this chip it writes the whole L2 back (measured: 38 us per launch).  The four workgroups of a
        // tile share an XCD, i.e. an L2: write-through stores, drained (vmcnt 0), then a relaxed agent-scope atomic; the last
        // arriver reads with L1-bypassing loads.
        const __amdgpu_buffer_rsrc_t pr = __builtin_amdgcn_make_buffer_rsrc(b.part[cur], 0, SLICES * TILES * TILE_FLOATS * 4, 0x00020000);
        __shared__ unsigned arrived;
        const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
        st_wt(pr, (unsigned)(((size_t)slice * TILES + tile) * TILE_FLOATS * 4 + tid * 16), mine);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) arrived = __hip_atomic_fetch_add(&b.count[tile], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const unsigned long long t2 = __builtin_amdgcn_s_memrealtime();
        if (arrived == launch_no * SLICES + SLICES - 1) {  // the last of the tile's four
            f4 gs = mine;
#pragma unroll
            for (int s = 0; s < SLICES; ++s)
                if (s != slice) {
                    const f4 g = ld_sc1(pr, (unsigned)(((size_t)s * TILES + tile) * TILE_FLOATS * 4 + tid * 16));
                    gs.x += g.x; gs.y += g.y; gs.z += g.z; gs.w += g.w;
                }
            f4 p = reinterpret_cast<const f4*>(b.p[prev] + (size_t)tile * TILE_FLOATS)[tid];
            p.x += 0.01f * gs.x; p.y += 0.01f * gs.y; p.z += 0.01f * gs.z; p.w += 0.01f * gs.w;
            const unsigned long long t3 = __builtin_amdgcn_s_memrealtime() + (unsigned long long)(__builtin_amdgcn_readfirstlane(p.x == 1234.5f ? 1 : 0));
            reinterpret_cast<f4*>(b.q[cur] + (size_t)tile * TILE_FLOATS)[tid] = f4{acc.x + 0.02f * p.x, acc.y + 0.02f * p.y, acc.z + 0.02f * p.z, acc.w + 0.02f * p.w};
            reinterpret_cast<f4*>(b.p[cur] + (size_t)tile * TILE_FLOATS)[tid] = p;
            if (tid == 0) {
                b.stamps[tile * 4 + 0] = t2 - t1;  // write-through store drained + barrier + atomic + barrier
                b.stamps[tile * 4 + 1] = t3 - t2;  // loads of the other partials + momentum, sums
                b.stamps[tile * 4 + 2] = __builtin_amdgcn_s_memrealtime() - t3;  // stores issued
            }
        }
    }
}

int main() {
    Bufs b;
    for (int i = 0; i < 2; ++i) {
        (void)hipMalloc(&b.part[i], (size_t)SLICES * TILES * TILE_FLOATS * 4);
        (void)hipMalloc(&b.q[i], (size_t)TILES * TILE_FLOATS * 4);
        (void)hipMalloc(&b.p[i], (size_t)TILES * TILE_FLOATS * 4);
        (void)hipMemset(b.part[i], 0, (size_t)SLICES * TILES * TILE_FLOATS * 4);
        (void)hipMemset(b.q[i], 0, (size_t)TILES * TILE_FLOATS * 4);
        (void)hipMemset(b.p[i], 0, (size_t)TILES * TILE_FLOATS * 4);
    }
    (void)hipMalloc(&b.count, TILES * 4);
    (void)hipMalloc(&b.stamps, TILES * 4 * 8);
    hipStream_t st;
    (void)hipStreamCreate(&st);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int N = 400;
    for (int us : {4, 6}) {
        float per[2] = {0, 0};
        for (int variant = 0; variant < 2; ++variant) {
            float best = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                (void)hipMemsetAsync(b.count, 0, TILES * 4, st);
                (void)hipEventRecord(e0, st);
                for (int i = 0; i < N; ++i) {
                    if (variant == 0) hipLaunchKernelGGL(step<0>, dim3(TILES, SLICES), dim3(NT), 0, st, b, i & 1, us * 100, (unsigned)i);
                    else hipLaunchKernelGGL(step<1>, dim3(TILES, SLICES), dim3(NT), 0, st, b, i & 1, us * 100, (unsigned)i);
                }
                (void)hipEventRecord(e1, st);
                (void)hipEventSynchronize(e1);
                float ms;
                (void)hipEventElapsedTime(&ms, e0, e1);
                best = std::min(best, ms * 1e3f / N);
            }
            per[variant] = best;
        }
        std::vector<unsigned long long> s(TILES * 4);
        (void)hipMemcpy(s.data(), b.stamps, TILES * 4 * 8, hipMemcpyDeviceToHost);
        double ph[3] = {0, 0, 0};
        for (int t = 0; t < TILES; ++t)
            for (int k = 0; k < 3; ++k) ph[k] += s[t * 4 + k] * 0.01 / TILES;
        printf("row loop %d us: period per launch  today (48 KB prologue) %.2f us   last arriver (8 KB prologue + tail) %.2f us   "
               "[last arriver's tail, mean over tiles: store drained + atomic %.2f, loads + sums %.2f, stores issued %.2f us]\n",
               us, per[0], per[1], ph[0], ph[1], ph[2]);
    }
    return 0;
}
