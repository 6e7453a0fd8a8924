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

template <int BM, int BN>
static int launch_tiled(const bf16_t* A, int lda, long strideA, const bf16_t* W, int ldw, const float* bias, void* C,
                        int ldc, long strideC, int M, int N, int K, int batch, int epi, const float* gate,
                        const int* ctl, hipStream_t st) {
  constexpr int smem = 2 * (BM + BN) * GT_PITCH * (int)sizeof(bf16_t);
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_tiled_kernel<BM, BN>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
      return DEER_ERR_LAUNCH;
    attr_set = true;
  }
  dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM, batch);
  hipLaunchKernelGGL((gemm_tiled_kernel<BM, BN>), grid, dim3(256), smem, st, A, lda, strideA, W, ldw, bias, C, ldc,
                     strideC, M, N, K, epi, gate, ctl);
  DEER_LAUNCH_CHECK();
  return DEER_OK;
}

// tile: 0 = auto, 1 = 64x64, 2 = 64x128, 3 = 128x128
extern "C" int deer_gemm_bf16_nt(const void* A, int lda, long strideA, const void* W, int ldw, const float* bias,
                                 void* C, int ldc, long strideC, int M, int N, int K, int batch, int epi,
                                 const float* gate, int tile, const int* ctl, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0 || (K & 7) || (N & 15) || (lda & 7) || (ldw & 7) || (ldc & 3) || epi < 0 || epi > 4)
    return DEER_ERR_SHAPE;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (tile == 0) {
    // fill the 256 CUs first, then grow the tile: (M/BM)*(N/BN)*batch workgroups
    const long b128 = (long)((M + 127) / 128) * ((N + 127) / 128) * batch;
    const long b64x128 = (long)((M + 63) / 64) * ((N + 127) / 128) * batch;
    tile = (b128 >= 384) ? 3 : (b64x128 >= 256 ? 2 : 1);
  }
  const bf16_t* a = reinterpret_cast<const bf16_t*>(A);
  const bf16_t* w = reinterpret_cast<const bf16_t*>(W);
  switch (tile) {
    case 1: return launch_tiled<64, 64>(a, lda, strideA, w, ldw, bias, C, ldc, strideC, M, N, K, batch, epi, gate, ctl, st);
    case 2: return launch_tiled<64, 128>(a, lda, strideA, w, ldw, bias, C, ldc, strideC, M, N, K, batch, epi, gate, ctl, st);
    case 3: return launch_tiled<128, 128>(a, lda, strideA, w, ldw, bias, C, ldc, strideC, M, N, K, batch, epi, gate, ctl, st);
    default: return DEER_ERR_SHAPE;
  }
}
