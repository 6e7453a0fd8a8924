// Tiled bf16 MFMA GEMM for the MFMA-bound part of the step (ViT-L/14, Perceiver, media K/V projection).
//
//   C[M,N] (+)= epi( A[M,K] (bf16, row-major) * W[N,K]^T (bf16, nn.Linear layout) + bias[N] )
//
// gfx950 design notes
//  * 256 threads = 4 waves in a 2x2 arrangement; each wave owns a (BM/2)x(BN/2) sub-tile built from
//    v_mfma_f32_16x16x32_bf16 with SWAPPED operands (W fragment as the MFMA "A", activation fragment as
//    "B").  The accumulator then holds 4 consecutive output COLUMNS per lane, so the epilogue writes
//    8-byte (bf16) / 16-byte (f32) contiguous pieces instead of 2-byte scattered ones.
//  * BK = 64, global -> register -> LDS staging, double-buffered LDS, ONE barrier per K step: the
//    next tile's global loads are issued before the MFMA block and written to the other LDS buffer
//    after it (async-stage split), so HBM/L2 latency hides under the MFMAs.
//  * LDS rows are padded by one 16-byte access (pitch 72 bf16) to spread ds_read_b128 over banks.
//  * Epilogues fuse bias, QuickGELU / exact GELU, and the residual update x += tanh(gate) * y.
#include "common.h"

enum { EPI_BF16 = 0, EPI_F32 = 1, EPI_QGELU_BF16 = 2, EPI_GELU_BF16 = 3, EPI_RESADD_F32 = 4 };

#define GT_BK 64
#define GT_PITCH 72

template <int BM, int BN>
__global__ __launch_bounds__(256) void gemm_tiled_kernel(const bf16_t* __restrict__ A, int lda, long strideA,
                                                         const bf16_t* __restrict__ W, int ldw,
                                                         const float* __restrict__ bias, void* __restrict__ Cv,
                                                         int ldc, long strideC, int M, int N, int K, int epi,
                                                         const float* __restrict__ gate, const int* ctl) {
  DEER_RETURN_IF_EXITED(ctl);
  constexpr int TM = BM / 32, TN = BN / 32;          // 16x16 tiles per wave in M / N
  constexpr int A_CH = BM * 8 / 256, W_CH = BN * 8 / 256;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* As = reinterpret_cast<bf16_t*>(smem_raw);                 // [2][BM][PITCH]
  bf16_t* Ws = As + 2 * BM * GT_PITCH;                              // [2][BN][PITCH]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int c = lane & 15, g = lane >> 4;
  const int m0 = blockIdx.y * BM, n0 = blockIdx.x * BN;
  A += (long)blockIdx.z * strideA;

  f32x4 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  uint4 ra[A_CH], rw[W_CH];
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
      const int id = tid + i * 256, row = id >> 3, seg = id & 7;
      const int m = m0 + row, k = k0 + seg * 8;
      ra[i] = (m < M && k < K) ? *reinterpret_cast<const uint4*>(A + (long)m * lda + k) : uint4{0, 0, 0, 0};
    }
#pragma unroll
    for (int i = 0; i < W_CH; ++i) {
      const int id = tid + i * 256, row = id >> 3, seg = id & 7;
      const int n = n0 + row, k = k0 + seg * 8;
      rw[i] = (n < N && k < K) ? *reinterpret_cast<const uint4*>(W + (long)n * ldw + k) : uint4{0, 0, 0, 0};
    }
  };
  auto swrite = [&](int buf) {
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
      const int id = tid + i * 256, row = id >> 3, seg = id & 7;
      *reinterpret_cast<uint4*>(As + (buf * BM + row) * GT_PITCH + seg * 8) = ra[i];
    }
#pragma unroll
    for (int i = 0; i < W_CH; ++i) {
      const int id = tid + i * 256, row = id >> 3, seg = id & 7;
      *reinterpret_cast<uint4*>(Ws + (buf * BN + row) * GT_PITCH + seg * 8) = rw[i];
    }
  };

  const int nk = (K + GT_BK - 1) / GT_BK;
  gload(0);
  swrite(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) gload((kt + 1) * GT_BK);
    const bf16_t* as = As + (buf * BM + wm * (BM / 2) + c) * GT_PITCH + g * 8;
    const bf16_t* ws = Ws + (buf * BN + wn * (BN / 2) + c) * GT_PITCH + g * 8;
#pragma unroll
    for (int kk = 0; kk < GT_BK / 32; ++kk) {
      bf16x8 af[TM], wf[TN];
#pragma unroll
      for (int j = 0; j < TM; ++j) af[j] = *reinterpret_cast<const bf16x8*>(as + j * 16 * GT_PITCH + kk * 32);
#pragma unroll
      for (int i = 0; i < TN; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(ws + i * 16 * GT_PITCH + kk * 32);
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], af[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) swrite(buf ^ 1);
    __syncthreads();
  }

  // ---- epilogue: lane holds C[m = ..+c][n = ..+g*4 .. +3] ----
  const float gs = (epi == EPI_RESADD_F32 && gate != nullptr) ? tanhf(*gate) : 1.f;
#pragma unroll
  for (int i = 0; i < TN; ++i) {
    const int n = n0 + wn * (BN / 2) + i * 16 + g * 4;
    if (n >= N) continue;
    float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
    if (bias != nullptr) {
      const float4 bv = *reinterpret_cast<const float4*>(bias + n);
      b0 = bv.x; b1 = bv.y; b2 = bv.z; b3 = bv.w;
    }
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int m = m0 + wm * (BM / 2) + j * 16 + c;
      if (m >= M) continue;
      float v0 = acc[i][j][0] + b0, v1 = acc[i][j][1] + b1, v2 = acc[i][j][2] + b2, v3 = acc[i][j][3] + b3;
      const long off = (long)blockIdx.z * strideC + (long)m * ldc + n;
      if (epi == EPI_RESADD_F32) {
        float4* p = reinterpret_cast<float4*>(reinterpret_cast<float*>(Cv) + off);
        float4 r = *p;
        r.x += gs * v0; r.y += gs * v1; r.z += gs * v2; r.w += gs * v3;
        *p = r;
      } else if (epi == EPI_F32) {
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(Cv) + off) = float4{v0, v1, v2, v3};
      } else {
        if (epi == EPI_QGELU_BF16) {
          v0 = quick_gelu(v0); v1 = quick_gelu(v1); v2 = quick_gelu(v2); v3 = quick_gelu(v3);
        } else if (epi == EPI_GELU_BF16) {
          v0 = gelu_erf(v0); v1 = gelu_erf(v1); v2 = gelu_erf(v2); v3 = gelu_erf(v3);
        }
        *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(Cv) + off) = uint2{pack2bf(v0, v1), pack2bf(v2, v3)};
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// Pipelined variant: register ring of THREE K-tiles in flight.  At M ~ 514 (two camera frames) every GEMM of the
// ViT is latency-bound: few workgroups per CU, 16-64 K-steps each, and with a 1-deep prefetch every K-step pays a
// full L2/HBM round trip (~1 us measured).  Here the loads of tile kt+3 are issued while tile kt is computed, tile
// kt+1 (loaded two steps ago) is written to the other LDS buffer, and the counted vmcnt the compiler derives from
// the static issue order never drains the younger loads.  Loads past the last K-tile are clamped (not branched
// around) so that the outstanding-load count stays compile-time known.
// ------------------------------------------------------------------------------------------------------------
template <int BM, int BN>
__global__ __launch_bounds__(256) void gemm_tiled_pipe_kernel(const bf16_t* __restrict__ A, int lda, long strideA,
                                                              const bf16_t* __restrict__ W, int ldw,
                                                              const float* __restrict__ bias, void* __restrict__ Cv,
                                                              int ldc, long strideC, int M, int N, int K, int epi,
                                                              const float* __restrict__ gate, const int* ctl) {
  DEER_RETURN_IF_EXITED(ctl);
  constexpr int TM = BM / 32, TN = BN / 32;
  constexpr int A_CH = BM * 8 / 256, W_CH = BN * 8 / 256;
  struct Regs { uint4 a[A_CH]; uint4 w[W_CH]; };
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* As = reinterpret_cast<bf16_t*>(smem_raw);                 // [2][BM][PITCH]
  bf16_t* Ws = As + 2 * BM * GT_PITCH;                              // [2][BN][PITCH]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int c = lane & 15, g = lane >> 4;
  // XCD-aware mapping: consecutive workgroup ids round-robin over the 8 XCDs; give each XCD a contiguous band of
  // N-tiles so that a W panel is fetched from HBM into ONE L2 instead of eight (A is small and shared by all).
  int bx = blockIdx.x;
  {
    const int nbx = gridDim.x;
    if ((nbx & 7) == 0) bx = (blockIdx.x & 7) * (nbx >> 3) + (blockIdx.x >> 3);
  }
  const int m0 = blockIdx.y * BM, n0 = bx * BN;
  A += (long)blockIdx.z * strideA;

  f32x4 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = (K + GT_BK - 1) / GT_BK;
  const int k_last = (nk - 1) * GT_BK;
  // per-thread source pointers, hoisted out of the K loop.  Rows beyond M / N are CLAMPED to the last valid row
  // instead of predicated: they only feed accumulator rows/columns that are never stored, and unconditional loads
  // keep hipcc from branching around every load with a vmcnt(0) behind it (which would serialise the pipeline).
  // K must be a multiple of GT_BK here (checked on the host), so there is no K tail either.
  const bf16_t* ap[A_CH];
  const bf16_t* wp[W_CH];
#pragma unroll
  for (int i = 0; i < A_CH; ++i) {
    const int id = tid + i * 256, row = id >> 3;
    ap[i] = A + (long)min(m0 + row, M - 1) * lda + (id & 7) * 8;
  }
#pragma unroll
  for (int i = 0; i < W_CH; ++i) {
    const int id = tid + i * 256, row = id >> 3;
    wp[i] = W + (long)min(n0 + row, N - 1) * ldw + (id & 7) * 8;
  }
  auto gload = [&](Regs& r, int k0) {
    k0 = min(k0, k_last);                                           // clamp: keeps the load count static
#pragma unroll
    for (int i = 0; i < A_CH; ++i) r.a[i] = *reinterpret_cast<const uint4*>(ap[i] + k0);
#pragma unroll
    for (int i = 0; i < W_CH; ++i) r.w[i] = *reinterpret_cast<const uint4*>(wp[i] + k0);
  };
  auto swrite = [&](const Regs& r, int buf) {
#pragma unroll
    for (int i = 0; i < A_CH; ++i) {
      const int id = tid + i * 256, row = id >> 3, seg = id & 7;
      *reinterpret_cast<uint4*>(As + (buf * BM + row) * GT_PITCH + seg * 8) = r.a[i];
    }
#pragma unroll
    for (int i = 0; i < W_CH; ++i) {
      const int id = tid + i * 256, row = id >> 3, seg = id & 7;
      *reinterpret_cast<uint4*>(Ws + (buf * BN + row) * GT_PITCH + seg * 8) = r.w[i];
    }
  };
  auto compute = [&](int buf) {
    const bf16_t* as = As + (buf * BM + wm * (BM / 2) + c) * GT_PITCH + g * 8;
    const bf16_t* ws = Ws + (buf * BN + wn * (BN / 2) + c) * GT_PITCH + g * 8;
#pragma unroll
    for (int kk = 0; kk < GT_BK / 32; ++kk) {
      bf16x8 af[TM], wf[TN];
#pragma unroll
      for (int j = 0; j < TM; ++j) af[j] = *reinterpret_cast<const bf16x8*>(as + j * 16 * GT_PITCH + kk * 32);
#pragma unroll
      for (int i = 0; i < TN; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(ws + i * 16 * GT_PITCH + kk * 32);
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[i], af[j], acc[i][j], 0, 0, 0);
    }
  };

  Regs r0, r1, r2;
  gload(r0, 0);
  gload(r1, GT_BK);
  gload(r2, 2 * GT_BK);
  swrite(r0, 0);
  __syncthreads();
  for (int kt = 0; kt < nk; kt += 3) {
    gload(r0, (kt + 3) * GT_BK);                 // tile kt in LDS[kt&1]; r1 = kt+1, r2 = kt+2 in flight
    compute(kt & 1);
    swrite(r1, (kt + 1) & 1);
    __syncthreads();
    if (kt + 1 >= nk) break;
    gload(r1, (kt + 4) * GT_BK);                 // tile kt+1; r2 = kt+2, r0 = kt+3 in flight
    compute((kt + 1) & 1);
    swrite(r2, kt & 1);
    __syncthreads();
    if (kt + 2 >= nk) break;
    gload(r2, (kt + 5) * GT_BK);                 // tile kt+2; r0 = kt+3, r1 = kt+4 in flight
    compute(kt & 1);
    swrite(r0, (kt + 1) & 1);
    __syncthreads();
  }

  const float gs = (epi == EPI_RESADD_F32 && gate != nullptr) ? tanhf(*gate) : 1.f;
#pragma unroll
  for (int i = 0; i < TN; ++i) {
    const int n = n0 + wn * (BN / 2) + i * 16 + g * 4;
    if (n >= N) continue;
    float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
    if (bias != nullptr) {
      const float4 bv = *reinterpret_cast<const float4*>(bias + n);
      b0 = bv.x; b1 = bv.y; b2 = bv.z; b3 = bv.w;
    }
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int m = m0 + wm * (BM / 2) + j * 16 + c;
      if (m >= M) continue;
      float v0 = acc[i][j][0] + b0, v1 = acc[i][j][1] + b1, v2 = acc[i][j][2] + b2, v3 = acc[i][j][3] + b3;
      const long off = (long)blockIdx.z * strideC + (long)m * ldc + n;
      if (epi == EPI_RESADD_F32) {
        float4* p = reinterpret_cast<float4*>(reinterpret_cast<float*>(Cv) + off);
        float4 r = *p;
        r.x += gs * v0; r.y += gs * v1; r.z += gs * v2; r.w += gs * v3;
        *p = r;
      } else if (epi == EPI_F32) {
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(Cv) + off) = float4{v0, v1, v2, v3};
      } else {
        if (epi == EPI_QGELU_BF16) {
          v0 = quick_gelu(v0); v1 = quick_gelu(v1); v2 = quick_gelu(v2); v3 = quick_gelu(v3);
        } else if (epi == EPI_GELU_BF16) {
          v0 = gelu_erf(v0); v1 = gelu_erf(v1); v2 = gelu_erf(v2); v3 = gelu_erf(v3);
        }
        *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(Cv) + off) = uint2{pack2bf(v0, v1), pack2bf(v2, v3)};
      }
    }
  }
}

template <int BM, int BN, bool PIPE>
static int launch_tiled(const bf16_t* A, int lda, long strideA, const bf16_t* W, int ldw, const float* bias, void* C,
                        int ldc, long strideC, int M, int N, int K, int batch, int epi, const float* gate,
                        const int* ctl, hipStream_t st) {
  constexpr int smem = 2 * (BM + BN) * GT_PITCH * (int)sizeof(bf16_t);
  static bool attr_set = false;
  auto kern = PIPE ? &gemm_tiled_pipe_kernel<BM, BN> : &gemm_tiled_kernel<BM, BN>;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
      return DEER_ERR_LAUNCH;
    attr_set = true;
  }
  dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM, batch);
  hipLaunchKernelGGL(kern, grid, dim3(256), smem, st, A, lda, strideA, W, ldw, bias, C, ldc, strideC, M, N, K, epi, gate, ctl);
  DEER_LAUNCH_CHECK();
  return DEER_OK;
}

// tile: 0 = auto; simple 1-deep-prefetch kernels: 1 = 64x64, 2 = 64x128, 3 = 128x128;
//       3-deep pipelined kernels: 4 = 64x64, 5 = 128x64, 6 = 64x128, 7 = 128x128
extern "C" int deer_gemm_bf16_nt(const void* A, int lda, long strideA, const void* W, int ldw, const float* bias,
                                 void* C, int ldc, long strideC, int M, int N, int K, int batch, int epi,
                                 const float* gate, int tile, const int* ctl, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0 || (K & 7) || (N & 15) || (lda & 7) || (ldw & 7) || (ldc & 3) || epi < 0 || epi > 4)
    return DEER_ERR_SHAPE;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if ((K % GT_BK) != 0 && tile >= 4) tile -= 3;         // pipelined kernels need K % 64 == 0 (5 -> 2 is a different
  if ((K % GT_BK) != 0 && tile == 0) tile = 1;          // shape but always valid)
  if (tile == 0) {
    // fill the 256 CUs first, then grow the tile (less L2->LDS traffic per flop)
    auto nblk = [&](int bm, int bn) { return (long)((M + bm - 1) / bm) * ((N + bn - 1) / bn) * batch; };
    if (nblk(128, 128) >= 256) tile = 7;
    else if (nblk(128, 64) >= 224) tile = 5;
    else if (nblk(64, 128) >= 224) tile = 6;
    else tile = 4;
  }
  const bf16_t* a = reinterpret_cast<const bf16_t*>(A);
  const bf16_t* w = reinterpret_cast<const bf16_t*>(W);
#define DEER_GT(BM_, BN_, P_) \
  return launch_tiled<BM_, BN_, P_>(a, lda, strideA, w, ldw, bias, C, ldc, strideC, M, N, K, batch, epi, gate, ctl, st)
  switch (tile) {
    case 1: DEER_GT(64, 64, false);
    case 2: DEER_GT(64, 128, false);
    case 3: DEER_GT(128, 128, false);
    case 4: DEER_GT(64, 64, true);
    case 5: DEER_GT(128, 64, true);
    case 6: DEER_GT(64, 128, true);
    case 7: DEER_GT(128, 128, true);
    default: return DEER_ERR_SHAPE;
  }
#undef DEER_GT
}
