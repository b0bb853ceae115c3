// xchg_probe.hip -- what does a per-step exchange of partial gradients between the RS workgroups of one chain tile cost
// when the workgroups stay resident (persistent trajectory kernel) instead of ending a launch per leapfrog step?
//   hipcc --offload-arch=gfx950 -O2 tools/xchg_probe.hip -o tools/bin/xchg_probe && tools/bin/xchg_probe
// Protocol per step and group of RS workgroups: every thread stores its share of the 8 KB partial with agent-scope
// relaxed atomic stores, waits for them (vmcnt 0), barrier, thread 0 publishes the step number in the workgroup's flag;
// thread r < RS polls peer r's flag (bounded), barrier, every thread loads its share of the RS partials (agent scope),
// sums them in slice order.  Buffers are double-buffered by step parity.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int RS, int NT, int WORDS /* 64-bit words per workgroup partial */>
__global__ void __launch_bounds__(NT) k(unsigned long long* part, unsigned* flags, float* out, int steps, int same_xcd, unsigned* err) {
    // group g: same_xcd -> the RS members are blocks g*8*... with equal (blockIdx % 8); else consecutive blocks
    const int nb = gridDim.x, b = blockIdx.x;
    int g, r;
    if (same_xcd) { const int x = b & 7, q = b >> 3; g = (q / RS) * 8 + x; r = q % RS; }  // members: same XCD (b % 8), consecutive q
    else { g = b / RS; r = b % RS; }
    const int ng = nb / RS;
    const int tid = threadIdx.x;
    float acc = 0.f;
    __shared__ int bad;
    if (tid == 0) bad = 0;
    __syncthreads();
    for (int s = 1; s <= steps; ++s) {
        unsigned long long* mine = part + (((size_t)(s & 1) * ng + g) * RS + r) * WORDS;
        for (int i = tid; i < WORDS; i += NT) {
            const float v0 = acc + (float)(i + r), v1 = (float)s;
            const unsigned long long w = (unsigned long long)__float_as_uint(v0) | ((unsigned long long)__float_as_uint(v1) << 32);
            __hip_atomic_store(&mine[i], w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): the stores are acknowledged
        __syncthreads();
        if (tid == 0) __hip_atomic_store(&flags[g * RS + r], (unsigned)s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid < RS) {
            int spins = 0;
            while (__hip_atomic_load(&flags[g * RS + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)s) {
                if (++spins > (1 << 22)) { bad = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
        if (bad) { if (tid == 0) atomicAdd(err, 1u); return; }
        float sum = 0.f;
        for (int i = tid; i < WORDS; i += NT) {
#pragma unroll
            for (int q = 0; q < RS; ++q) {
                const unsigned long long w = __hip_atomic_load(part + (((size_t)(s & 1) * ng + g) * RS + q) * WORDS + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((unsigned)(w >> 32) != __float_as_uint((float)s)) bad = 2;  // stale data would show here
                sum += __uint_as_float((unsigned)w);
            }
        }
        acc = sum * 1e-3f;
    }
    if (bad == 2 && tid == 0) atomicAdd(err, 1000u);
    out[(size_t)b * NT + tid] = acc;
}

// Round 5: the two-phase form a RESIDENT-image cluster kernel would need (DESIGN_HISTORY.md, "not built"): RS workgroups of one XCD
// hold one gradient partial each of WORDS 64-bit words (two floats); member r owns words [r WORDS / RS, (r + 1) WORDS / RS):
//   phase 1  every member stores its partial, flag, wait for all;   the owner sums the RS partials of its words (slice order)
//   phase 2  the owner stores the sums into the shared state, flag, wait for all;   every member loads the whole state
template <int RS, int NT, int WORDS>
__global__ void __launch_bounds__(NT) k2(unsigned long long* part, unsigned long long* state, unsigned* flags, float* out, int steps, unsigned* err) {
    const int nb = gridDim.x, b = blockIdx.x, x = b & 7, q = b >> 3, g = (q / RS) * 8 + x, r = q % RS, ng = nb / RS, tid = threadIdx.x;
    constexpr int OWN = WORDS / RS;
    float acc = 0.f;
    __shared__ int bad;
    if (tid == 0) bad = 0;
    __syncthreads();
    auto sync = [&](int phase, int s) {
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
        if (tid == 0) __hip_atomic_store(&flags[(phase * ng + g) * RS + r], (unsigned)s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tid < RS) {
            int spins = 0;
            while (__hip_atomic_load(&flags[(phase * ng + g) * RS + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)s) {
                if (++spins > (1 << 22)) { bad = 1; break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        __syncthreads();
    };
    for (int s = 1; s <= steps; ++s) {
        unsigned long long* mine = part + (((size_t)(s & 1) * ng + g) * RS + r) * WORDS;
        for (int i = tid; i < WORDS; i += NT) {
            const unsigned long long w = (unsigned long long)__float_as_uint(acc + (float)(i + r)) | ((unsigned long long)__float_as_uint((float)s) << 32);
            __hip_atomic_store(&mine[i], w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        sync(0, s);
        if (bad) { if (tid == 0) atomicAdd(err, 1u); return; }
        unsigned long long* st = state + ((size_t)(s & 1) * ng + g) * WORDS;
        for (int i = tid; i < OWN; i += NT) {
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < RS; ++k) {
                const unsigned long long w = __hip_atomic_load(part + (((size_t)(s & 1) * ng + g) * RS + k) * WORDS + r * OWN + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((unsigned)(w >> 32) != __float_as_uint((float)s)) bad = 2;
                sum += __uint_as_float((unsigned)w);
            }
            __hip_atomic_store(&st[r * OWN + i], (unsigned long long)__float_as_uint(sum * 1e-3f) | ((unsigned long long)__float_as_uint((float)s) << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        sync(1, s);
        if (bad == 1) { if (tid == 0) atomicAdd(err, 1u); return; }
        float tot = 0.f;
        for (int i = tid; i < WORDS; i += NT) {
            const unsigned long long w = __hip_atomic_load(&st[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned)(w >> 32) != __float_as_uint((float)s)) bad = 2;
            tot += __uint_as_float((unsigned)w);
        }
        acc = tot * 1e-3f;
    }
    if (bad == 2 && tid == 0) atomicAdd(err, 1000u);
    out[(size_t)b * NT + tid] = acc;
}

template <int RS, int NT, int WORDS> int run2(const char* name) {
    const int nb = 256, steps = 2000;
    unsigned long long *part, *state; unsigned *flags, *err; float* out;
    CK(hipMalloc(&part, (size_t)2 * nb * WORDS * 8)); CK(hipMalloc(&state, (size_t)2 * (nb / RS) * WORDS * 8)); CK(hipMalloc(&flags, 2 * nb * 4)); CK(hipMalloc(&err, 4));
    CK(hipMalloc(&out, (size_t)nb * NT * 4));
    CK(hipMemset(flags, 0, 2 * nb * 4)); CK(hipMemset(err, 0, 4)); CK(hipMemset(part, 0, (size_t)2 * nb * WORDS * 8)); CK(hipMemset(state, 0, (size_t)2 * (nb / RS) * WORDS * 8));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    int st = steps;
    void* args[] = {&part, &state, &flags, &out, &st, &err};
    CK(hipEventRecord(e0));
    CK(hipLaunchCooperativeKernel((const void*)k2<RS, NT, WORDS>, dim3(nb), dim3(NT), args, 0, 0));
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned herr; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    printf("two-phase %-24s RS=%d threads=%d bytes/partial=%d (same XCD): %.3f us per step  (err=%u)\n", name, RS, NT, WORDS * 8, ms * 1e3 / steps, herr);
    hipFree(part); hipFree(state); hipFree(flags); hipFree(err); hipFree(out);
    return 0;
}

template <int RS, int NT, int WORDS> int run(const char* name, int same_xcd) {
    const int nb = 256, steps = 2000;
    unsigned long long* part; unsigned *flags, *err; float* out;
    CK(hipMalloc(&part, (size_t)2 * nb * WORDS * 8)); CK(hipMalloc(&flags, nb * 4)); CK(hipMalloc(&err, 4)); CK(hipMalloc(&out, (size_t)nb * NT * 4));
    CK(hipMemset(flags, 0, nb * 4)); CK(hipMemset(err, 0, 4)); CK(hipMemset(part, 0, (size_t)2 * nb * WORDS * 8));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    void* args[] = {&part, &flags, &out, (void*)&steps, &same_xcd, &err};
    int st = steps;
    args[3] = &st;
    CK(hipEventRecord(e0));
    CK(hipLaunchCooperativeKernel((const void*)k<RS, NT, WORDS>, dim3(nb), dim3(NT), args, 0, 0));
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned herr; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    printf("%-28s RS=%d threads=%d bytes/partial=%d same_xcd=%d : %.3f us per step  (err=%u)\n", name, RS, NT, WORDS * 8, same_xcd, ms * 1e3 / steps, herr);
    hipFree(part); hipFree(flags); hipFree(err); hipFree(out);
    return 0;
}
int main() {
    for (int sx : {0, 1}) {
        run<4, 512, 1024>("wide tile 16x128 f32", sx);
        run<8, 512, 2048>("wide tile 32x128 f32", sx);
        run<4, 256, 1024>("4 waves", sx);
        run<16, 256, 64>("tall 64 chains x 8 f32 /4", sx);
        run<64, 256, 64>("tall, 64 slices", sx);
    }
    run2<16, 512, 4096>("4 tiles x 128, 16 members");   // 64 chains x 128 coordinates x 4 B = 32 KB partial per workgroup
    run2<8, 512, 2048>("2 tiles x 128, 8 members");     // 32 chains: 16 KB
    run2<4, 512, 1024>("1 tile x 128, 4 members");      // 16 chains: 8 KB
    run<16, 512, 4096>("one-phase, 16 members, 32 KB", 1);
    return 0;
}
