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
#include <cstdlib>

enum { EPI_BF16 = DEER_E_16, EPI_F32 = DEER_E_F32, EPI_QGELU_BF16 = DEER_E_QGELU_16, EPI_GELU_BF16 = DEER_E_GELU_16, EPI_RESADD_F32 = DEER_E_RESADD_F32,
       EPI_BF16OUT = DEER_E_BF16OUT };   // "BF16" = the kernel family's 16-bit format (bf16, or fp16 when F16)

#define GT_BK 64
#define GT_PITCH 72

template <int BM, int BN, bool F16>
__global__ __launch_bounds__(256) void gemm_tiled_kernel(const bf16_t* __restrict__ A, int lda, long strideA,
                                                         const bf16_t* __restrict__ W, int ldw, long strideW,
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
  W += (long)blockIdx.z * strideW;                                  // split-K: z selects the K slice of both operands

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
          acc[i][j] = mfma16<F16>(wf[i], af[j], acc[i][j]);
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
          v0 = quick_gelu_bf(v0); v1 = quick_gelu_bf(v1); v2 = quick_gelu_bf(v2); v3 = quick_gelu_bf(v3);
        } else if (epi == EPI_GELU_BF16) {
          v0 = gelu_erf(v0); v1 = gelu_erf(v1); v2 = gelu_erf(v2); v3 = gelu_erf(v3);
        }
        *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(Cv) + off) = uint2{pack2_epi<F16>(v0, v1, epi), pack2_epi<F16>(v2, v3, epi)};
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------
// LDS-ring variant (the one the engine uses).  At M ~ 514 rows (two camera frames) every GEMM of the ViT is
// LATENCY-bound: 1-2 workgroups per CU, 16-64 K-steps each, and a K-step is only ~100 ns of MFMA work, so a
// shallow prefetch pays a full L2/HBM round trip per K-step (measured ~1 us/step, 4 % of the MFMA roof).
// Here tiles are streamed with the CDNA4 LDS-DMA path (global_load_lds_dwordx4: no VGPR round trip) into a ring of
// D stages; D-1 tiles (~100 KiB per CU) are always in flight and the only per-step synchronisation is ONE counted
// `s_waitcnt vmcnt((D-2)*loads_per_stage)` + ONE raw s_barrier (a __syncthreads() would drain vmcnt to 0).
//  * LDS image of a stage is lane-linear (DMA writes wave-base + lane*16): rows of 128 B (BK = 64 bf16), A rows
//    then W rows.  Bank conflicts of the ds_read_b128 fragment reads are removed by an XOR swizzle of the 16-byte
//    slot with (row & 7), applied on the SOURCE address of the DMA and again on the read (same involution).
//  * rows beyond M / N are clamped (they only feed never-stored accumulators); K % 64 == 0 is required.
//  * loads past the last K-tile are clamped, not branched around, so the outstanding-load count stays static.
// ------------------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(1))) const void gptr_t;
typedef __attribute__((address_space(3))) void lptr_t;

template <int N_>
__device__ __forceinline__ void wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory");
}

// WM x WN waves per workgroup (64*WM*WN threads); each wave owns a (BM/WM) x (BN/WN) sub-tile.
// Measured on MI355X (tools/fill_bench.hip): the rate at which a CU can pull L2-resident data is set by the number
// of WAVES issuing loads on it (~8-9 GB/s per wave: 4 waves 8 TB/s chip-wide, 8 waves 16 TB/s, 16 waves 20 TB/s),
// not by how many loads each wave keeps in flight - so these kernels run 8-16 waves per workgroup and are sized so
// that two workgroups fit on a CU.
#ifdef DEER_KTRACE
KT_DEFINE(gemm)
// the four ViT projections by shape: in_proj -> slots 0.., out_proj 8.., c_fc 16.., c_proj 24.. (stamps of workgroup (0, 0, 0))
#define GKT(slot) KT(gemm, blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && kt_base >= 0, kt_base + (slot))
#else
#define GKT(slot) do { } while (0)
#endif
template <int BM, int BN, int WM, int WN, int D, int DBG = 0, int U = 1, int PIPE = 0, bool F16 = false>   // DBG (ablation only): 1 = no MFMA/ds_read, 2 = no DMA in the loop
__global__ __launch_bounds__(64 * WM * WN) void gemm_tiled_ring_kernel(const bf16_t* __restrict__ A, int lda, long strideA,
                                                                        const bf16_t* __restrict__ W, int ldw, long strideW,
                                                                        const float* __restrict__ bias,
                                                                        void* __restrict__ Cv, int ldc, long strideC,
                                                                        int M, int N, int K, int epi,
                                                                        const float* __restrict__ gate, const int* ctl) {
  DEER_RETURN_IF_EXITED(ctl);
#ifdef DEER_KTRACE
  const int kt_base = (N == 3072 && K == 1024) ? 0 : (N == 4096 && K == 1024) ? 16 : (N == 1024 && (int)gridDim.z * K == 1024) ? 8 : (N == 1024 && (int)gridDim.z * K == 4096) ? 24 : -1;
#endif
  GKT(0);
  constexpr int NW = WM * WN;
  constexpr int TM = BM / WM / 16, TN = BN / WN / 16;               // 16x16 MFMA tiles per wave
  constexpr int CH = (BM + BN) / 8;                                 // 1 KiB DMA chunks (8 rows x 128 B) per stage
  constexpr int CPW = CH / NW;                                      // chunks per wave per stage
  static_assert(CH % NW == 0, "every wave must issue the same number of DMA loads (static vmcnt)");
  static_assert((BM / WM) % 16 == 0 && (BN / WN) % 16 == 0, "wave tile");
  static_assert((D - 2) * CPW <= 63, "vmcnt field");
  static_assert(U >= 1 && D >= 2 * U, "U stages are consumed per barrier: U free slots + U landing + the rest in flight");
  constexpr int STAGE = (BM + BN) * 128;                            // bytes per ring stage
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int c = lane & 15, g = lane >> 4;
  int bx = blockIdx.x;                                              // XCD-aware: contiguous band of N-tiles per XCD
  if ((gridDim.x & 7) == 0) bx = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
  const int m0 = blockIdx.y * BM, n0 = bx * BN;
  A += (long)blockIdx.z * strideA;
  W += (long)blockIdx.z * strideW;                                  // split-K: z selects the K slice of both operands

  // DMA chunk q = wave + i*NW: rows 8q..8q+7 of [A tile ; W tile]; lane: row 8q + (lane>>3), slot (lane&7) ^ (row&7)
  const int lr = lane >> 3, ls = ((lane & 7) ^ lr) * 8;
  const bf16_t* sp[CPW];
#pragma unroll
  for (int i = 0; i < CPW; ++i) {
    const int row = (wave + i * NW) * 8 + lr;                       // uniform per wave: A chunk or W chunk
    sp[i] = (row < BM) ? A + (long)min(m0 + row, M - 1) * lda + ls : W + (long)min(n0 + row - BM, N - 1) * ldw + ls;
  }
  const int nk = K / GT_BK;
  auto issue = [&](int tile) {
    const int k0 = min(tile, nk - 1) * GT_BK;
    unsigned char* st = smem + (tile % D) * STAGE;
#pragma unroll
    for (int i = 0; i < CPW; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t*)(sp[i] + k0), (lptr_t*)(st + (wave + i * NW) * 1024), 16, 0, 0);
  };

  f32x4 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // fragment read offsets inside a stage (bytes): row*128 + ((slot ^ (row&7)) << 4), slot = kk*4 + g, row&7 == c&7
  const int a_row = (wm * (BM / WM) + c) * 128, w_row = (BM + wn * (BN / WN) + c) * 128;
  const int sw0 = ((0 * 4 + g) ^ (c & 7)) << 4, sw1 = ((1 * 4 + g) ^ (c & 7)) << 4;

  if constexpr (PIPE) {
    // Register-pipelined K loop: the fragments of K-step k+1 are read from LDS WHILE the MFMAs of K-step k run (two
    // fragment sets, loop unrolled by two).  In the plain loop every wave leaves the barrier, reads its fragments, waits,
    // then issues MFMAs - with one workgroup per CU the LDS phase and the MFMA phase of a K-step never overlap (PMC: MFMA
    // 18 % busy, 55 % of wave cycles at s_waitcnt/s_barrier).  Here barrier k guarantees tile k+1 has landed, so the
    // prefetch distance is D-2 K-steps and D is chosen deep (K-steps get ~2x shorter, the L2 latency does not).
    static_assert(D >= 3 && U == 1, "pipelined loop: tile k+1 must be resident at barrier k");
    bf16x8 fa[2][2][TM], fw[2][2][TN];
    auto load_frags = [&](int set, int tile) {
      const unsigned char* st = smem + (tile % D) * STAGE;
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
        const int sw = kk ? sw1 : sw0;
#pragma unroll
        for (int j = 0; j < TM; ++j) fa[set][kk][j] = *reinterpret_cast<const bf16x8*>(st + a_row + j * 16 * 128 + sw);
#pragma unroll
        for (int i = 0; i < TN; ++i) fw[set][kk][i] = *reinterpret_cast<const bf16x8*>(st + w_row + i * 16 * 128 + sw);
      }
    };
    auto mma = [&](int set) {
#pragma unroll
      for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
          for (int j = 0; j < TM; ++j)
            acc[i][j] = mfma16<F16>(fw[set][kk][i], fa[set][kk][j], acc[i][j]);
    };
#pragma unroll
    for (int t = 0; t < D - 1; ++t) issue(t);
    wait_vmcnt<(D - 2) * CPW>();
    __builtin_amdgcn_s_barrier();
    load_frags(0, 0);
    for (int kt = 0; kt < nk; kt += 2) {            // nk is even (checked by the launcher)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int k = kt + h;
        wait_vmcnt<(D - 3) * CPW>();                // this wave's part of tile k+1 has landed
        __builtin_amdgcn_s_barrier();               // ... everybody's; and everybody has read tile k-1 (its slot is refilled now)
        issue(k + D - 1);
        if (k + 1 < nk) load_frags(h ^ 1, k + 1);
        mma(h);
      }
    }
  } else {
  // U K-steps per barrier: with <= ~1 workgroup per CU (M = 257 / 514) nothing hides the wait -> barrier -> ds_read -> MFMA
  // chain of a K-step, so U > 1 amortises it (D - U stages in flight at the loop top, U slots free behind the barrier).
  constexpr int INFLIGHT = (U == 1) ? D - 1 : D - U;
#pragma unroll
  for (int t = 0; t < INFLIGHT; ++t) issue(t);
  // Round 5 (tools/ktrace_gemm.py, profiles/r05_i_*): one loop body (U K-steps) takes 0.6 us at M = 257 - not a memory round trip (touching
  // the rest of both operand panels behind the prologue, so that every DMA hits this XCD's L2, left the loop at 5.1 us for K = 1024 and
  // cost 3 us in front of it) but the CU's fill rate: 16 KB per K-step and workgroup, two workgroups per CU = 107 GB/s per CU.
  GKT(1);
  for (int kt = 0; kt < nk; kt += U) {
    wait_vmcnt<(INFLIGHT - U) * CPW>();         // this wave's part of tiles kt .. kt+U-1 has landed
    __builtin_amdgcn_s_barrier();               // ... everybody's part; and everybody finished reading the tiles before kt
    if (kt == 0) GKT(2);
    if (kt == U) GKT(3);
    if (DBG != 2) {
#pragma unroll
      for (int u = 0; u < U; ++u) issue(kt + INFLIGHT + u);   // refill the slots freed by the previous iteration
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      if (U > 1 && kt + u >= nk) break;
      const unsigned char* st = smem + ((kt + u) % D) * STAGE;
#pragma unroll
      for (int kk = 0; kk < (DBG == 1 ? 0 : 2); ++kk) {
        const int sw = kk ? sw1 : sw0;
        bf16x8 af[TM], wf[TN];
#pragma unroll
        for (int j = 0; j < TM; ++j) af[j] = *reinterpret_cast<const bf16x8*>(st + a_row + j * 16 * 128 + sw);
#pragma unroll
        for (int i = 0; i < TN; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(st + w_row + i * 16 * 128 + sw);
#pragma unroll
        for (int i = 0; i < TN; ++i)
#pragma unroll
          for (int j = 0; j < TM; ++j)
            acc[i][j] = mfma16<F16>(wf[i], af[j], acc[i][j]);
      }
    }
  }

  }

  GKT(4);
  const float gs = (epi == EPI_RESADD_F32 && gate != nullptr) ? tanhf(*gate) : 1.f;
#pragma unroll
  for (int i = 0; i < TN; ++i) {
    const int n = n0 + wn * (BN / WN) + i * 16 + g * 4;
    if (n >= N) continue;
    float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
    if (bias != nullptr) {
      const float4 bv = *reinterpret_cast<const float4*>(bias + n);
      b0 = bv.x; b1 = bv.y; b2 = bv.z; b3 = bv.w;
    }
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int m = m0 + wm * (BM / WM) + j * 16 + c;
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
          v0 = quick_gelu_bf(v0); v1 = quick_gelu_bf(v1); v2 = quick_gelu_bf(v2); v3 = quick_gelu_bf(v3);
        } else if (epi == EPI_GELU_BF16) {
          v0 = gelu_erf(v0); v1 = gelu_erf(v1); v2 = gelu_erf(v2); v3 = gelu_erf(v3);
        }
        *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(Cv) + off) = uint2{pack2_epi<F16>(v0, v1, epi), pack2_epi<F16>(v2, v3, epi)};
      }
    }
  }
  GKT(5);
}

template <bool F16, int BM, int BN, int WM, int WN, int D, int DBG = 0, int U = 1, int PIPE = 0>
static int launch_ring(const bf16_t* A, int lda, long strideA, const bf16_t* W, int ldw, long strideW, const float* bias, void* C,
                       int ldc, long strideC, int M, int N, int K, int batch, int epi, const float* gate,
                       const int* ctl, hipStream_t st) {
  constexpr int smem = D * (BM + BN) * 128;
  static_assert(smem <= 160 * 1024, "LDS");
  static std::atomic<bool> attr_set{false};
  if (PIPE && ((K / GT_BK) & 1)) return DEER_ERR_SHAPE;   // pipelined loop is unrolled by two K-steps
  auto kern = &gemm_tiled_ring_kernel<BM, BN, WM, WN, D, DBG, U, PIPE, F16>;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
      return DEER_ERR_LAUNCH;
    attr_set = true;
  }
  dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM, batch);
  hipLaunchKernelGGL(kern, grid, dim3(64 * WM * WN), smem, st, A, lda, strideA, W, ldw, strideW, bias, C, ldc, strideC, M, N, K, epi, gate,
                     ctl);
  DEER_LAUNCH_CHECK();
  return DEER_OK;
}

template <bool F16, int BM, int BN>
static int launch_tiled(const bf16_t* A, int lda, long strideA, const bf16_t* W, int ldw, long strideW, const float* bias, void* C,
                        int ldc, long strideC, int M, int N, int K, int batch, int epi, const float* gate,
                        const int* ctl, hipStream_t st) {
  constexpr int smem = 2 * (BM + BN) * GT_PITCH * (int)sizeof(bf16_t);
  static std::atomic<bool> attr_set{false};
  auto kern = &gemm_tiled_kernel<BM, BN, F16>;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
      return DEER_ERR_LAUNCH;
    attr_set = true;
  }
  dim3 grid((N + BN - 1) / BN, (M + BM - 1) / BM, batch);
  hipLaunchKernelGGL(kern, grid, dim3(256), smem, st, A, lda, strideA, W, ldw, strideW, bias, C, ldc, strideC, M, N, K, epi, gate, ctl);
  DEER_LAUNCH_CHECK();
  return DEER_OK;
}

// csrc/gemm_bigm.hip
template <bool F16>
int deer_launch_gemm_ring32(int variant, const bf16_t* A, int lda, long strideA, const bf16_t* W, int ldw, long strideW, const float* bias,
                            void* C, int ldc, long strideC, int M, int N, int K, int batch, int epi, const float* gate, const int* ctl,
                            hipStream_t st);

// tile: 0 = auto; simple register-staged kernels (any K % 8 == 0): 1 = 64x64, 2 = 64x128, 3 = 128x128;
//       LDS-ring DMA kernels (K % 64 == 0): 4 = 64x64 / 8 waves / 4 stages, 5 = 128x64 / 8 waves / 3 stages,
//       6 = 64x64 / 16 waves / 6 stages, 7 = 128x128 / 16 waves / 3 stages, 8 = 64x128 / 8 waves / 3 stages,
//       9 = 32x64 / 4 waves... (see switch)
template <bool F16>
static int gemm_dispatch(const void* A, int lda, long strideA, const void* W, int ldw, long strideW, const float* bias,
                         void* C, int ldc, long strideC, int M, int N, int K, int batch, int epi,
                         const float* gate, int tile, const int* ctl, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0 || (K & 7) || (N & 15) || (lda & 7) || (ldw & 7) || (ldc & 3) || epi < 0 || epi > 5)
    return DEER_ERR_SHAPE;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const bool ring_ok = (K % GT_BK) == 0;
  if (!ring_ok && tile >= 4) return DEER_ERR_SHAPE;
  if (tile < 0 || tile > 79) return DEER_ERR_SHAPE;
  static const int big_tile = [] { const char* e = getenv("DEER_GEMM_BIG"); return e ? atoi(e) : 17; }();   // 17: 16 waves 32x32 (best in situ, tools/graph_time.py); 28: 8 waves 32x64 (+5 % in the microbenchmark only)
  static const bool big_sel = [] { const char* e = getenv("DEER_GEMM_SEL"); return e == nullptr || e[0] != '0'; }();
  static const bool u2_ok = [] { const char* e = getenv("DEER_GEMM_U2"); return e == nullptr || e[0] != '0'; }();
  if (tile == 0) {
    // fill the 256 CUs first, then grow the tile (less L2->LDS traffic per flop)
    auto nblk = [&](int bm, int bn) { return (long)((M + bm - 1) / bm) * ((N + bn - 1) / bn) * batch; };
    if (!ring_ok) tile = (nblk(64, 128) >= 256) ? 2 : 1;
    else if (nblk(128, 128) >= 256) {
      // big M (env batch / calibration window): 64-80 KB rings so that two workgroups share a CU and one's ds_read phase overlaps the
      // other's MFMAs.  M = 257 * images is always a power of two plus a little, so the choice between 128x128 (16 waves) and the
      // 192x128 / 128x192 tiles (8 waves, 48x64 wave tiles: 14 fragment reads per 24 MFMAs instead of 8 per 8) is made on the
      // number of half-rounds each needs on the 256 CUs; a 192-row/column tile costs ~1.35x a 128x128 one (tools/bench_vendor_gemm.py)
      tile = big_tile;
      // 256x256 tiles (16 waves in two staggered groups, 32-column K-steps, csrc/gemm_bigm.hip) when they fill the chip in exactly ONE
      // round - ViT in_proj at 16 frames: 17 x 12 = 204 workgroups, 36.4 us against 41.1 (r03, profiles/r03_c_bigm_gemm_4112_stag.txt);
      // with more tiles than CUs the second round costs more than the larger tile returns (c_fc at 16 frames: 272 tiles, 55 vs 45 us)
      static const bool sel256 = [] { const char* e = getenv("DEER_GEMM_256"); return e == nullptr || e[0] != '0'; }();
      const long n256 = nblk(256, 256);
      // frame tiles (one camera frame = 257 rows per row tile, the 17th MFMA row tile dealt out over the waves; bf16 results leave
      // through LDS as whole lines): M = 257 n tiles EXACTLY - 16 frames x 3072 / 192 = 16 frames x 4096 / 256 = 256 workgroups, one
      // round.  in_proj at 16 frames 38.0-39.4 -> 32.1-33.1 us, c_fc 43.3-44.7 -> 39.9 (hipBLASLt on the same box 30.3-30.7 / 43.6-44.7);
      // at 12 frames (192 workgroups) 31.9 -> 28.8 and 37.0 -> 34.1 (profiles/r03_k_frame_tiles_*.txt)
      const long frames = (M % 257 == 0) ? M / 257 : 0;
      // ... or exactly two rounds (16 environments per engine: 32 frames x 16 column tiles = 512 workgroups; the second round's prologue
      // runs while the first round's last workgroups drain)
      auto one_round = [&](int bn) {
        if (frames <= 0 || (N % bn) != 0) return false;
        const long wgs = frames * (N / bn) * batch;
        return (wgs >= 192 && wgs <= 256) || wgs == 512;
      };
      // DEER_GEMM_FRAME8=1: the same tiles on eight waves (csrc/gemm_bigm.hip: gemm_frame8_kernel)
      static const bool frame8 = [] { const char* e = getenv("DEER_GEMM_FRAME8"); return e != nullptr && e[0] == '1'; }();
      // DEER_GEMM_FRAME4=1 (or 2: only where the launch is TWO rounds of workgroups): the same tiles on four waves (gemm_frame4_kernel,
      // round 6: bit-identical results; 16-bit / QuickGELU epilogues and the f32 slabs of the 257 x 128 tile, K-steps in pairs)
      static const int frame4 = [] { const char* e = getenv("DEER_GEMM_FRAME4"); return e != nullptr ? atoi(e) : 0; }();
      auto f4 = [&](int bn) { return (K & 63) == 0 && epi != EPI_GELU_BF16 && (frame4 == 1 || (frame4 == 2 && frames * (N / bn) * batch == 512)); };
      if (sel256 && big_sel && (K & 31) == 0 && (epi == EPI_BF16 || epi == EPI_QGELU_BF16 || epi == EPI_GELU_BF16 || epi == EPI_BF16OUT) && one_round(192)) tile = f4(192) ? 77 : frame8 ? 75 : 64;
      else if (sel256 && big_sel && (K & 31) == 0 && (epi == EPI_BF16 || epi == EPI_QGELU_BF16 || epi == EPI_GELU_BF16 || epi == EPI_BF16OUT) && one_round(256)) tile = f4(256) ? 76 : frame8 ? 74 : 63;
      // ... and the K halves of c_proj (f32 slabs, N = 1024) as 257 x 128 tiles: 16 frames x 8 x 2 = 256 workgroups, 49.8 -> 43.1 us
      else if (sel256 && big_sel && (K & 31) == 0 && epi == EPI_F32 && one_round(128)) tile = f4(128) ? 78 : 67;
      else if (sel256 && big_sel && (N & 255) == 0 && (K & 31) == 0 && n256 >= 192 && n256 <= 256) tile = 61;
      else if (big_sel && N <= 1024 && batch == 1) tile = 17;   // out_proj at 16 frames: 264 128x128 tiles sit two per CU (17.5 us; 192-row tiles 19.9)
      else if (big_sel) {
        auto cost = [&](int bm, int bn, float rel) { return (float)((nblk(bm, bn) + 255) / 256) * rel; };
        float best = cost(128, 128, 1.0f);
        if (cost(192, 128, 1.35f) < best) { best = cost(192, 128, 1.35f); tile = 39; }
        if (cost(128, 192, 1.35f) < best) { best = cost(128, 192, 1.35f); tile = 45; }
      }
    }
    else if (nblk(64, 64) > 512) tile = 8;             // measured on MI355X at M = 257 / 514 (tools/bench_gemm.py):
    else if (nblk(64, 64) > 256 && u2_ok) tile = 16;   //   two co-resident workgroups per CU: two K-steps per barrier (-8..10 %)
    else tile = 4;                                     //   64x64 / 8 waves wins whenever it gives <= 2 workgroups per CU (the
                                                       //   register-pipelined loop, tile 26, is within noise of it end to end)
  }
  const bf16_t* a = reinterpret_cast<const bf16_t*>(A);
  const bf16_t* w = reinterpret_cast<const bf16_t*>(W);
#define DEER_ARGS a, lda, strideA, w, ldw, strideW, bias, C, ldc, strideC, M, N, K, batch, epi, gate, ctl, st
  switch (tile) {
    case 1: return launch_tiled<F16, 64, 64>(DEER_ARGS);
    case 2: return launch_tiled<F16, 64, 128>(DEER_ARGS);
    case 3: return launch_tiled<F16, 128, 128>(DEER_ARGS);
    case 4: return launch_ring<F16, 64, 64, 2, 4, 4>(DEER_ARGS);
    case 5: return launch_ring<F16, 128, 64, 4, 2, 3>(DEER_ARGS);
    case 6: return launch_ring<F16, 64, 64, 4, 4, 6>(DEER_ARGS);
    case 7: return launch_ring<F16, 128, 128, 4, 4, 3>(DEER_ARGS);
    case 8: return launch_ring<F16, 64, 128, 2, 4, 3>(DEER_ARGS);
    case 9: return launch_ring<F16, 32, 64, 2, 2, 6>(DEER_ARGS);
    case 10: return launch_ring<F16, 128, 128, 2, 4, 3>(DEER_ARGS);
    case 12: return launch_ring<F16, 64, 64, 2, 4, 6, 0, 2>(DEER_ARGS);   // two K-steps per barrier
    case 13: return launch_ring<F16, 64, 64, 2, 4, 8, 0, 2>(DEER_ARGS);
    case 14: return launch_ring<F16, 64, 64, 2, 4, 8, 0, 4>(DEER_ARGS);   // four K-steps per barrier
    case 15: return launch_ring<F16, 64, 128, 2, 4, 4, 0, 2>(DEER_ARGS);
    case 16: return launch_ring<F16, 64, 64, 2, 4, 4, 0, 2>(DEER_ARGS);
    case 17: return launch_ring<F16, 128, 128, 4, 4, 2>(DEER_ARGS);       // shallow ring, two workgroups per CU (64 KB LDS each)
    // 64x64 WAVE tiles (4x4 MFMA tiles per wave): one ds_read_b128 per two MFMAs instead of one per MFMA - the 32x32 wave tiles of
    // tiles 7/17 spend as many LDS cycles on fragment reads as the SIMDs spend on MFMAs (16 waves x 4 reads x 4 clk = 4 x 4 MFMAs x 16 clk)
    case 18: return launch_ring<F16, 128, 128, 2, 2, 2>(DEER_ARGS);       // 4 waves, 64 KB: two workgroups per CU
    case 19: return launch_ring<F16, 128, 128, 2, 2, 4>(DEER_ARGS);       // 4 waves, 128 KB ring
    case 20: return launch_ring<F16, 256, 128, 4, 2, 3>(DEER_ARGS);       // 8 waves, 144 KB ring
    case 21: return launch_ring<F16, 256, 128, 4, 2, 2>(DEER_ARGS);       // 8 waves, 96 KB
    case 22: return launch_ring<F16, 128, 256, 2, 4, 3>(DEER_ARGS);       // 8 waves, 144 KB ring
    case 23: return launch_ring<F16, 256, 256, 4, 4, 2>(DEER_ARGS);       // 16 waves, 128 KB
    case 25: return launch_ring<F16, 128, 128, 2, 2, 3>(DEER_ARGS);       // 4 waves, 96 KB
    case 27: return launch_ring<F16, 128, 128, 2, 4, 2>(DEER_ARGS);       // 8 waves (64x32 wave tiles), 64 KB: two workgroups per CU
    case 28: return launch_ring<F16, 128, 128, 4, 2, 2>(DEER_ARGS);       // 8 waves (32x64 wave tiles), 64 KB
    case 29: return launch_ring<F16, 128, 256, 2, 4, 2>(DEER_ARGS);       // 8 waves (64x64), 96 KB
    case 30: return launch_ring<F16, 256, 128, 4, 2, 2>(DEER_ARGS);       // 8 waves (64x64), 96 KB
    case 32: return launch_ring<F16, 256, 256, 2, 4, 2>(DEER_ARGS);       // 8 waves, 128x64 wave tiles (64 MFMAs per wave per barrier), 128 KB
    case 33: return launch_ring<F16, 256, 128, 2, 4, 3>(DEER_ARGS);       // 8 waves, 128x32 wave tiles, 144 KB ring
    // row tiles that are NOT a power of two: M = 257 * images is always "a power of two plus a bit" (2056, 4112), so 128-row tiles end
    // in a nearly empty extra row of workgroups (17 x 32 = 544 for the fc1 of 8 images: one more than the 512 co-resident slots)
    case 35: return launch_ring<F16, 160, 128, 2, 2, 2>(DEER_ARGS);       // 4 waves (80x64 wave tiles), 72 KB: two workgroups per CU
    case 36: return launch_ring<F16, 160, 128, 2, 2, 3>(DEER_ARGS);       // same, 108 KB ring
    case 37: return launch_ring<F16, 160, 64, 2, 2, 2>(DEER_ARGS);        // 4 waves (80x32), 56 KB
    case 38: return launch_ring<F16, 192, 128, 2, 2, 2>(DEER_ARGS);       // 4 waves (96x64), 80 KB
    case 39: return launch_ring<F16, 192, 128, 4, 2, 2>(DEER_ARGS);       // 8 waves (48x64), 80 KB
    case 40: return launch_ring<F16, 192, 128, 2, 4, 2>(DEER_ARGS);       // 8 waves (96x32), 80 KB
    case 41: return launch_ring<F16, 96, 128, 2, 2, 2>(DEER_ARGS);        // 4 waves (48x64), 56 KB
    case 42: return launch_ring<F16, 192, 64, 4, 1, 2>(DEER_ARGS);        // 4 waves (48x64), 64 KB
    case 43: return launch_ring<F16, 128, 192, 2, 4, 2>(DEER_ARGS);       // 8 waves (64x48), 80 KB
    case 45: return launch_ring<F16, 128, 192, 4, 2, 2>(DEER_ARGS);       // 8 waves (32x96), 80 KB
    case 46: return launch_ring<F16, 96, 128, 2, 2, 3>(DEER_ARGS);        // 4 waves (48x64), 84 KB
    case 51: case 52: case 53: case 54: case 55: case 56: case 57: case 58: case 59: case 60: case 61: case 62:            // 16 waves, 32-column K-steps, deep ring (csrc/gemm_bigm.hip)
    case 63: case 64: case 65: case 66: case 67: case 68: case 69: case 70: case 71:                                     // one camera frame per row tile, balanced (csrc/gemm_bigm.hip)
    case 72: case 73:                                                                                                    // half a frame per row tile
    case 74: case 75:                                                                                                    // frame8: one frame per row tile on eight waves (round 5)
    case 76: case 77: case 78: case 79:                                                                                  // frame4: one frame per row tile on four waves (round 6)
      return deer_launch_gemm_ring32<F16>(tile - 51, DEER_ARGS);
    case 26: return launch_ring<F16, 64, 64, 2, 4, 4, 0, 1, 1>(DEER_ARGS);    // register-pipelined K loop (fragments of k+1 read under the MFMAs of k)
    case 24: return launch_ring<F16, 64, 64, 2, 4, 4, 1>(DEER_ARGS);   // ablations (tools/bench_gemm.py)
    case 34: return launch_ring<F16, 64, 64, 2, 4, 4, 2>(DEER_ARGS);
    case 44: return launch_ring<F16, 64, 64, 2, 4, 8>(DEER_ARGS);      // deeper ring
    default: return DEER_ERR_SHAPE;
  }
#undef DEER_ARGS
}

extern "C" int deer_gemm_bf16_nt(const void* A, int lda, long strideA, const void* W, int ldw, const float* bias,
                                 void* C, int ldc, long strideC, int M, int N, int K, int batch, int epi,
                                 const float* gate, int tile, const int* ctl, void* stream) {
  return gemm_dispatch<false>(A, lda, strideA, W, ldw, 0, bias, C, ldc, strideC, M, N, K, batch, epi, gate, tile, ctl, stream);
}

// Batched form with a per-batch weight: C[z] = A[z] * W[z]^T (Perceiver to_kv of all layers on their own norm_media output).
extern "C" int deer_gemm_bf16_nt_wbatch(const void* A, int lda, long strideA, const void* W, int ldw, long strideW,
                                        const float* bias, void* C, int ldc, long strideC, int M, int N, int K, int batch,
                                        int epi, int tile, const int* ctl, void* stream) {
  return gemm_dispatch<false>(A, lda, strideA, W, ldw, strideW, bias, C, ldc, strideC, M, N, K, batch, epi, nullptr, tile, ctl, stream);
}

// Split-K form for the latency-bound shapes (few output tiles, long K: ViT c_proj 514x1024x4096 leaves 112 CUs idle
// for 64 K-steps): slab[s][M][N] (f32) = A[:, Ks] * W[:, Ks]^T, s = 0..splitk-1, reduced by the consumer
// (deer_resadd_ln, which also adds the bias) - deterministic, no atomics.
extern "C" int deer_gemm_bf16_nt_splitk(const void* A, int lda, const void* W, int ldw, float* slab, int M, int N, int K,
                                        int splitk, int tile, const int* ctl, void* stream) {
  if (splitk <= 0 || K % splitk != 0 || ((K / splitk) & 7)) return DEER_ERR_SHAPE;
  const int ks = K / splitk;
  return gemm_dispatch<false>(A, lda, ks, W, ldw, ks, nullptr, slab, N, (long)M * N, M, N, ks, splitk, EPI_F32, nullptr, tile, ctl,
                       stream);
}

// ---- the same three entry points on fp16 operands (round 6: the vision tower's fp16 arithmetic = the reference's amp run).  A, W fp16;
// epi DEER_EPI_BF16 / QGELU / GELU store fp16, DEER_EPI_BF16OUT stores bf16 (media K/V for the trunk's x-attn) -----------------------
extern "C" int deer_gemm_f16_nt(const void* A, int lda, long strideA, const void* W, int ldw, const float* bias,
                                 void* C, int ldc, long strideC, int M, int N, int K, int batch, int epi,
                                 const float* gate, int tile, const int* ctl, void* stream) {
  return gemm_dispatch<true>(A, lda, strideA, W, ldw, 0, bias, C, ldc, strideC, M, N, K, batch, epi, gate, tile, ctl, stream);
}

// Batched form with a per-batch weight: C[z] = A[z] * W[z]^T (Perceiver to_kv of all layers on their own norm_media output).
extern "C" int deer_gemm_f16_nt_wbatch(const void* A, int lda, long strideA, const void* W, int ldw, long strideW,
                                        const float* bias, void* C, int ldc, long strideC, int M, int N, int K, int batch,
                                        int epi, int tile, const int* ctl, void* stream) {
  return gemm_dispatch<true>(A, lda, strideA, W, ldw, strideW, bias, C, ldc, strideC, M, N, K, batch, epi, nullptr, tile, ctl, stream);
}

// Split-K form for the latency-bound shapes (few output tiles, long K: ViT c_proj 514x1024x4096 leaves 112 CUs idle
// for 64 K-steps): slab[s][M][N] (f32) = A[:, Ks] * W[:, Ks]^T, s = 0..splitk-1, reduced by the consumer
// (deer_resadd_ln, which also adds the bias) - deterministic, no atomics.
extern "C" int deer_gemm_f16_nt_splitk(const void* A, int lda, const void* W, int ldw, float* slab, int M, int N, int K,
                                        int splitk, int tile, const int* ctl, void* stream) {
  if (splitk <= 0 || K % splitk != 0 || ((K / splitk) & 7)) return DEER_ERR_SHAPE;
  const int ks = K / splitk;
  return gemm_dispatch<true>(A, lda, ks, W, ldw, ks, nullptr, slab, N, (long)M * N, M, N, ks, splitk, EPI_F32, nullptr, tile, ctl,
                       stream);
}
