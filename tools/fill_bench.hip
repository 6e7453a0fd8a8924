// Microbenchmark (MI355X): how fast can a CU pull L2-resident data (a) into VGPRs, (b) VGPRs + ds_write, (c) LDS-DMA.
// Build: hipcc --offload-arch=gfx950 -O3 fill_bench.hip -o fill_bench ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

// region: `bytes` of L2/MALL-resident data; every wave sweeps it with 1 KiB wave-loads; pattern: contig or tile rows
template <int MODE, int UNROLL>
__global__ void fill_kernel(const uint4* __restrict__ src, long n_vec, int iters, int row_stride_vec, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  uint4 acc = {0, 0, 0, 0};
  // each wave-load: 8 rows x 128 B (lane>>3 = row, lane&7 = 16B slot) when row_stride_vec > 0, else contiguous 1 KiB
  long base = ((long)blockIdx.x * nw + wave) * 64 * 17 % n_vec;
  for (int it = 0; it < iters; ++it) {
    uint4 v[UNROLL];
#pragma unroll
    for (int u = 0; u < UNROLL; ++u) {
      long idx = row_stride_vec > 0 ? (base + (long)(lane >> 3) * row_stride_vec + (lane & 7) + u * 8) % n_vec
                                    : (base + u * 64 + lane) % n_vec;
      if (MODE == 2) {
        __builtin_amdgcn_global_load_lds((gptr_t*)(src + idx), (lptr_t*)(smem + (wave * UNROLL + u) * 1024), 16, 0, 0);
      } else {
        v[u] = src[idx];
      }
    }
    if (MODE == 2) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      acc.x ^= ((unsigned*)smem)[threadIdx.x];
    } else {
#pragma unroll
      for (int u = 0; u < UNROLL; ++u) {
        if (MODE == 1) *reinterpret_cast<uint4*>(smem + ((wave * UNROLL + u) * 64 + lane) * 16) = v[u];
        acc.x ^= v[u].x; acc.y ^= v[u].y; acc.z ^= v[u].z; acc.w ^= v[u].w;
      }
    }
    base = (base + 64 * UNROLL * 3) % n_vec;
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = acc.x;
}

template <int MODE, int UNROLL>
void run(const char* name, const uint4* src, long n_vec, int blocks, int waves, int row_stride_vec, unsigned* sink) {
  const int iters = 2000 / UNROLL * 4;
  size_t smem = (size_t)waves * UNROLL * 1024;
  hipFuncSetAttribute((const void*)fill_kernel<MODE, UNROLL>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  fill_kernel<MODE, UNROLL><<<blocks, waves * 64, smem>>>(src, n_vec, 10, row_stride_vec, sink);
  hipEventRecord(e0);
  fill_kernel<MODE, UNROLL><<<blocks, waves * 64, smem>>>(src, n_vec, iters, row_stride_vec, sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double bytes = (double)blocks * waves * iters * UNROLL * 1024.0;
  printf("%-10s blocks=%4d waves=%2d unroll=%2d stride=%4d region=%6.1fMB : %7.2f TB/s  (%6.1f GB/s per block)\n", name, blocks, waves,
         UNROLL, row_stride_vec, n_vec * 16 / 1e6, bytes / ms / 1e9, bytes / ms / 1e6 / blocks);
}

int main(int argc, char** argv) {
  unsigned* sink; hipMalloc(&sink, 4);
  if (argc > 1) {   // "layout" mode: does a CONTIGUOUS 1 KiB per wave-load fill faster than 8 rows x 128 B at a 2 KiB pitch?
    for (long mb : {1L, 8L, 64L}) {
      long nv = mb * (1 << 20) / 16;
      uint4* s2; hipMalloc(&s2, nv * 16); hipMemset(s2, 1, nv * 16);
      for (int waves : {8, 16}) {
        run<2, 4>("dma rows", s2, nv, 256, waves, 128, sink);
        run<2, 4>("dma contig", s2, nv, 256, waves, 0, sink);
        run<0, 8>("vgpr rows", s2, nv, 256, waves, 128, sink);
        run<0, 8>("vgpr contig", s2, nv, 256, waves, 0, sink);
      }
      hipFree(s2);
    }
    return 0;
  }
  long n_vec = 1L * (1 << 20) / 16;
  uint4* src; hipMalloc(&src, n_vec * 16); hipMemset(src, 1, n_vec * 16);
  for (int waves : {4, 8}) {
    run<0, 4>("vgpr", src, n_vec, 256, waves, 128, sink);
    run<0, 8>("vgpr", src, n_vec, 256, waves, 128, sink);
    run<0, 16>("vgpr", src, n_vec, 256, waves, 128, sink);
    run<0, 32>("vgpr", src, n_vec, 256, waves, 128, sink);
    run<1, 16>("vgpr+ds", src, n_vec, 256, waves, 128, sink);
    run<2, 4>("lds-dma", src, n_vec, 256, waves, 128, sink);
    run<2, 8>("lds-dma", src, n_vec, 256, waves, 128, sink);
    run<2, 16>("lds-dma", src, n_vec, 256, waves, 128, sink);
    run<2, 32>("lds-dma", src, n_vec, 256, waves, 128, sink);
  }
  // two blocks per CU
  run<0, 8>("vgpr", src, n_vec, 512, 4, 128, sink);
  run<0, 16>("vgpr", src, n_vec, 512, 4, 128, sink);
  run<2, 8>("lds-dma", src, n_vec, 512, 4, 128, sink);
  run<0, 8>("vgpr", src, n_vec, 1024, 4, 128, sink);
  run<2, 8>("lds-dma", src, n_vec, 1024, 4, 128, sink);
  return 0;
}
