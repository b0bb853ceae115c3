// rs_probe.hip -- checks the 16-lane reduce-scatter used by the fused HMC kernel's interior loop (lr_device.h,
// group16_reduce_scatter8): 8 per-lane values summed over the 16 lanes of a DPP row with bank-masked v_add_f32_dpp,
// leaving in every lane of quad q the totals of values 2q and 2q+1; and the row_share all-gather back.
//   hipcc --offload-arch=gfx950 -O2 -I logreg_amd/csrc tools/rs_probe.hip -o tools/bin/rs_probe && tools/bin/rs_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include "lr_device.h"

__global__ void k(const float* in, float* out, float* gath) {
    const int lane = threadIdx.x;
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = in[lane * 8 + j];
    float u0, u1;
    lr::group16_reduce_scatter8(v, u0, u1);
    out[lane * 2] = u0;
    out[lane * 2 + 1] = u1;
    lr::f32x2 bb[4];
    lr::group16_allgather_pairs(lr::f32x2{u0, u1}, bb);
    for (int q = 0; q < 4; ++q) { gath[lane * 8 + 2 * q] = bb[q].x; gath[lane * 8 + 2 * q + 1] = bb[q].y; }
}

int main() {
    float h[64 * 8], o[128], g[512], *di, *dout, *dg;
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 8; ++j) h[l * 8 + j] = (float)((l * 37 + j * 11) % 97) + 0.25f * j;
    hipMalloc(&di, sizeof h); hipMalloc(&dout, sizeof o); hipMalloc(&dg, sizeof g);
    hipMemcpy(di, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dout, dg);
    hipMemcpy(o, dout, sizeof o, hipMemcpyDeviceToHost);
    hipMemcpy(g, dg, sizeof g, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        const int row = l / 16, q = (l % 16) / 4;
        float want[8];
        for (int j = 0; j < 8; ++j) { want[j] = 0; for (int i = 0; i < 16; ++i) want[j] += h[(row * 16 + i) * 8 + j]; }
        if (o[l * 2] != want[2 * q] || o[l * 2 + 1] != want[2 * q + 1]) { ++bad; printf("lane %d: got %g %g want %g %g\n", l, o[l*2], o[l*2+1], want[2*q], want[2*q+1]); }
        for (int j = 0; j < 8; ++j) if (g[l * 8 + j] != want[j]) { ++bad; if (bad < 20) printf("gather lane %d j %d: got %g want %g\n", l, j, g[l*8+j], want[j]); }
    }
    printf(bad ? "rs_probe: %d mismatches\n" : "rs_probe ok (%d)\n", bad);
    return bad != 0;
}
