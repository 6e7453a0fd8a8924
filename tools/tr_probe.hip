// Semantics probe for gfx950's LDS transpose read (ds_read_b64_tr_b16, __builtin_amdgcn_ds_read_tr16_b64_*): LDS holds lds[i] = i
// (16-bit), every lane hands in its own 8-byte-aligned address, the four 16-bit values each lane receives are printed.
// Pattern A: lane l reads at element 4*l (64 consecutive 8-byte chunks).  Pattern B: the attention-V use: a 16-lane group (g = l >> 4)
// covers a [4 keys][16 d] block of a row-major [key][PITCH] image - lane c hands in key (c >> 2), d chunk (c & 3) * 4 - and should
// receive column c: V[key 0..3][d = c].
// build: hipcc --offload-arch=gfx950 -O3 tools/tr_probe.hip -o tools/tr_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) short s16x4;
#define PITCH 72
__global__ void k(short* out, int pattern) {
  __shared__ __attribute__((aligned(16))) short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
  __syncthreads();
  const int l = threadIdx.x, c = l & 15, g = l >> 4;
  int e = 4 * l;
  if (pattern == 1) e = (g * 4 + (c >> 2)) * PITCH + (c & 3) * 4;
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + e));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = v[j];
}
int main() {
  short* d;
  hipMalloc(&d, 64 * 4 * 2);
  short h[256];
  for (int p = 0; p < 2; ++p) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, p);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("pattern %c\n", 'A' + p);
    for (int l = 0; l < 64; ++l) {
      printf("lane %2d:", l);
      for (int j = 0; j < 4; ++j) {
        if (p == 0) printf(" %4d", h[l * 4 + j]);
        else printf(" (k%2d,d%2d)", h[l * 4 + j] / PITCH, h[l * 4 + j] % PITCH);
      }
      printf("\n");
    }
  }
  return 0;
}
