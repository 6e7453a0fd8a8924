// Microbenchmark (MI355X): does the LDS-DMA fill rate of a CU depend on WHO ELSE reads the same lines at the same time?
// GEMM-like streams: workgroup (i, j) of a TI x TJ grid walks K and per step stages rows of A_i (shared by the TJ workgroups of row i) and
// rows of W_j (shared by the TI workgroups of column j).  Modes: 0 = private regions (nobody shares), 1 = GEMM sharing with the hardware's
// round-robin workgroup -> XCD placement, 2 = GEMM sharing with an XCD-aware remap (an XCD owns a block of tile rows x tile columns),
// 3 = GEMM sharing, every workgroup starts at its own K offset (same bytes, no lockstep).
// build: hipcc --offload-arch=gfx950 -O3 tools/fill_share.hip -o /tmp/fill_share ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

template <int WAVES, int ROWS>   // ROWS = rows per operand tile (x 128 B per K step): 256 -> 64 KB per step for both operands
__global__ __launch_bounds__(64 * WAVES) void share_kernel(const char* __restrict__ A, const char* __restrict__ W, int TI, int TJ, int ksteps,
                                                            long row_pitch, int mode, int reps, unsigned* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int bid = blockIdx.x;
  if (mode == 2) {                       // XCD x (= bid % 8) owns tiles [x * n/8, (x+1) * n/8) of the row-major tile order
    const int n = gridDim.x, per = n / 8;
    bid = (bid & 7) * per + (bid >> 3);
  }
  const int ti = bid / TJ, tj = bid % TJ;
  const long a_row0 = mode == 0 ? (long)blockIdx.x * ROWS : (long)ti * ROWS;
  const long w_row0 = mode == 0 ? (long)blockIdx.x * ROWS : (long)tj * ROWS;
  constexpr int CH = 2 * ROWS / 8;       // 1 KiB chunks (8 rows x 128 B) per step
  constexpr int CPW = CH / WAVES;
  const int lr = lane >> 3, ls = (lane & 7) * 16;
  const int k_off = mode == 3 ? (blockIdx.x * 7) % ksteps : 0;
  unsigned acc = 0;
  for (int r = 0; r < reps; ++r) {
    for (int k = 0; k < ksteps; ++k) {
      const int kk = (k + k_off) % ksteps;
#pragma unroll
      for (int i = 0; i < CPW; ++i) {
        const int q = wave + i * WAVES;
        const int row = (q * 8 + lr) % ROWS;
        const char* src = (q * 8 < ROWS ? A + (a_row0 + row) * row_pitch : W + (w_row0 + row) * row_pitch) + (long)kk * 128 + ls;
        __builtin_amdgcn_global_load_lds((gptr_t*)src, (lptr_t*)(smem + ((k & 1) * CH + q) * 1024), 16, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CPW) : "memory");      // previous step landed (one step in flight)
      __builtin_amdgcn_s_barrier();
      acc ^= ((const unsigned*)smem)[threadIdx.x];
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (acc == 0x12345678u) sink[0] = acc;
}

template <int WAVES, int ROWS>
void run(const char* A, const char* W, int TI, int TJ, int K, int mode, unsigned* sink) {
  const int ksteps = K / 64;
  const long pitch = (long)K * 2;
  const int reps = 8;
  size_t smem = 2 * (2 * ROWS / 8) * 1024;
  hipFuncSetAttribute((const void*)share_kernel<WAVES, ROWS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  share_kernel<WAVES, ROWS><<<TI * TJ, 64 * WAVES, smem>>>(A, W, TI, TJ, ksteps, pitch, mode, 1, sink);
  hipEventRecord(e0);
  share_kernel<WAVES, ROWS><<<TI * TJ, 64 * WAVES, smem>>>(A, W, TI, TJ, ksteps, pitch, mode, reps, sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double bytes = (double)TI * TJ * reps * ksteps * (2.0 * ROWS * 128);
  printf("waves=%2d tile=%3d rows  grid %2dx%2d K=%5d mode=%d : %6.2f TB/s  (%5.1f GB/s per workgroup, %6.1f us per pass)\n", WAVES, ROWS, TI, TJ, K,
         mode, bytes / ms / 1e9, bytes / ms / 1e6 / (TI * TJ), 1e3 * ms / reps);
}

int main() {
  unsigned* sink; hipMalloc(&sink, 4);
  const int K = 1024;
  const long rows = 256L * 256;                        // enough rows for the private mode (256 workgroups x 256 rows)
  char *A, *W;
  hipMalloc(&A, rows * K * 2); hipMalloc(&W, rows * K * 2);
  hipMemset(A, 1, rows * K * 2); hipMemset(W, 2, rows * K * 2);
  for (int mode = 0; mode < 4; ++mode) {
    run<8, 256>(A, W, 16, 16, K, mode, sink);          // 256 workgroups of 8 waves, 256 + 256 rows per step (64 KB)
    run<16, 256>(A, W, 16, 16, K, mode, sink);
    run<8, 128>(A, W, 16, 16, K, mode, sink);          // 128 + 128 rows per step (32 KB)
    run<16, 128>(A, W, 16, 16, K, mode, sink);
  }
  for (int mode = 0; mode < 4; ++mode) {               // two workgroups per CU (512 tiles of 128 rows: LDS 64 KB each)
    run<8, 128>(A, W, 16, 32, K, mode, sink);
    run<16, 128>(A, W, 16, 32, K, mode, sink);
  }
  return 0;
}
