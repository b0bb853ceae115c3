// boundary_probe.hip -- what does a kernel boundary between two DEPENDENT launches cost, and does a captured graph shorten it?
// 256 workgroups x 512 threads spin for ~T us (s_memrealtime), N = 200 launches back to back: (a) plain stream launches,
// (b) the same N launches captured once into a hipGraph and replayed.  Prints the period per launch minus T.
//   hipcc --offload-arch=gfx950 -O3 tools/boundary_probe.hip -o /tmp/boundary && /tmp/boundary
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void __launch_bounds__(512) spin(float* buf, int ticks, int touch) {
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    float v = touch ? buf[blockIdx.x * 512 + threadIdx.x] : 0.f;  // a dependent read of what the previous launch wrote
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)ticks) __builtin_amdgcn_s_sleep(2);
    if (touch) buf[blockIdx.x * 512 + threadIdx.x] = v + 1.f;
}
int main() {
    float* buf; (void)hipMalloc(&buf, 256 * 512 * 4); (void)hipMemset(buf, 0, 256 * 512 * 4);
    hipStream_t st; (void)hipStreamCreate(&st);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int N = 200;
    for (int touch = 0; touch < 2; ++touch)
        for (int us : {2, 5, 10}) {
            const int ticks = us * 100;
            float ms;
            for (int rep = 0; rep < 3; ++rep) {
                (void)hipEventRecord(e0, st);
                for (int i = 0; i < N; ++i) hipLaunchKernelGGL(spin, dim3(256), dim3(512), 0, st, buf, ticks, touch);
                (void)hipEventRecord(e1, st); (void)hipEventSynchronize(e1);
                (void)hipEventElapsedTime(&ms, e0, e1);
            }
            const float plain = ms * 1e3f / N - us;
            hipGraph_t g; hipGraphExec_t ge;
            (void)hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
            for (int i = 0; i < N; ++i) hipLaunchKernelGGL(spin, dim3(256), dim3(512), 0, st, buf, ticks, touch);
            (void)hipStreamEndCapture(st, &g);
            (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
            for (int rep = 0; rep < 3; ++rep) {
                (void)hipEventRecord(e0, st);
                (void)hipGraphLaunch(ge, st);
                (void)hipEventRecord(e1, st); (void)hipEventSynchronize(e1);
                (void)hipEventElapsedTime(&ms, e0, e1);
            }
            printf("kernel %2d us, %s: boundary %.2f us with stream launches, %.2f us inside a captured graph\n", us,
                   touch ? "reads what the previous launch wrote" : "no memory traffic", plain, ms * 1e3f / N - us);
            (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
        }
    return 0;
}
