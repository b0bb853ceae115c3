// tr_probe.hip -- empirical semantics of ds_read_b64_tr_b16 on gfx950 (tools only)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(short* o, int mode) {
  __shared__ short lds[2048];
  for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = (short)i;
  __syncthreads();
  int l = threadIdx.x;
  // mode 0: lane l -> element offset l*4 (consecutive 8-byte chunks)
  // mode 1: lane l -> offset (l>>4)*64 + (l&15)*4   (same thing), mode 2: rows of stride 32 elements: (l>>4)*128 + ((l&15)>>2)*32 + (l&3)*4
  int off = mode == 0 ? l * 4 : (mode == 1 ? (l >> 4) * 64 + (l & 15) * 4 : (l >> 4) * 128 + ((l & 15) >> 2) * 32 + (l & 3) * 4);
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(lds + off));
  for (int j = 0; j < 4; ++j) o[l * 4 + j] = v[j];
}
int main() {
  short* d; hipMalloc(&d, 64 * 4 * 2);
  short h[256];
  for (int mode = 0; mode < 3; ++mode) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, mode);
    hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    printf("mode %d\n", mode);
    for (int l = 0; l < 64; ++l) { if (l < 20 || l == 32 || l == 63) printf("  lane %2d: %4d %4d %4d %4d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]); }
  }
  return 0;
}
