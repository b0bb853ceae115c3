// atomic_probe.hip -- cost of reducing slice partials with 64-bit integer (fixed-point, hence order-independent and
// bit-reproducible) atomic adds at L2 instead of writing [slices][C][P] partials and re-reading them.
//   hipcc --offload-arch=gfx950 -O2 tools/atomic_probe.hip -o tools/bin/atomic_probe && tools/bin/atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
// grid (slices, blocks); each workgroup adds `per_wg` values: address = block * per_wg + i  (slices-way contention)
__global__ void k_atomic(long long* acc, int per_wg, int reps) {
    long long* base = acc + (size_t)blockIdx.y * per_wg;
    for (int r = 0; r < reps; ++r)
        for (int i = threadIdx.x; i < per_wg; i += blockDim.x)
            __hip_atomic_fetch_add(base + i, (long long)(i + r), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void k_store(float* part, int per_wg, int reps, int nblocks) {
    float* base = part + ((size_t)blockIdx.x * nblocks + blockIdx.y) * per_wg;
    for (int r = 0; r < reps; ++r)
        for (int i = threadIdx.x; i < per_wg; i += blockDim.x) __builtin_nontemporal_store((float)(i + r), base + i);
}
int main() {
    struct Cfg { const char* name; int slices, blocks, per_wg, threads; } cfgs[] = {
        {"config 4: 64 slices x 16 blocks x (64 chains x 8)", 64, 16, 512, 256},
        {"config 5:  4 slices x 64 tiles  x (16 chains x 128)", 4, 64, 2048, 512},
        {"config 4 with 16 slices", 16, 16, 512, 256},
    };
    for (auto& c : cfgs) {
        long long* acc; float* part;
        hipMalloc(&acc, (size_t)c.blocks * c.per_wg * 8); hipMemset(acc, 0, (size_t)c.blocks * c.per_wg * 8);
        hipMalloc(&part, (size_t)c.slices * c.blocks * c.per_wg * 4);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int which = 0; which < 2; ++which) {
            const int launches = 200;
            for (int w = 0; w < 2; ++w) {
                if (w == 1) hipEventRecord(e0);
                for (int l = 0; l < launches; ++l) {
                    if (which == 0) hipLaunchKernelGGL(k_atomic, dim3(c.slices, c.blocks), dim3(c.threads), 0, 0, acc, c.per_wg, 1);
                    else hipLaunchKernelGGL(k_store, dim3(c.slices, c.blocks), dim3(c.threads), 0, 0, part, c.per_wg, 1, c.blocks);
                }
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("%-55s %s: %.2f us per launch (%d values)\n", c.name, which == 0 ? "int64 atomic add" : "nontemporal f32 store", ms * 1e3 / launches,
                   c.slices * c.blocks * c.per_wg);
        }
        hipFree(acc); hipFree(part);
    }
    return 0;
}
