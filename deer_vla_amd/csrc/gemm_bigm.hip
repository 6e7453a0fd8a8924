// Big-M bf16 MFMA GEMMs of the vision tower for env batches / calibration windows (M = 257 x 8..16 frames):
//
//   C[M,N] = epi( A[M,K] (bf16, row-major) * W[N,K]^T (bf16, nn.Linear layout) + bias[N] )
//
// What round 3 measured about these shapes (tools/fill_share.hip, tools/bench_vendor_gemm.py; an 8-wave 256x256 kernel on
// the guide's 8-phase schedule was also written, measured at 40 us per 256x256x1024 tile and removed again):
//  * a 256x256x1024 tile takes 30-32 us in these kernels for ring depth 2..5 and with or without the staggered two-barrier schedule
//    (~46 % of the matrix pipe inside a tile; the guide's best plain-HIP GEMM sits at 53-60 %); the LDS-DMA path is NOT what it waits
//    for - DMAs alone move 92 GB/s per CU on a GEMM-like shared stream (11.4 us per tile pass);
//  * what separates the shapes is tile quantisation on 256 CUs: M = 257 n is never a multiple of a power-of-two tile (a ragged tile of
//    8-16 valid rows costs 3/4 of a full one), in_proj at 16 frames is 204 256x256 tiles (one round), c_fc 272 (two rounds);
//    hipBLASLt's 29-32 us on in_proj is ONE 256x256 tile time plus a shorter prologue / epilogue.
//  * a launch has ~8 us that do not scale with K (K sweep): the epilogue was store-ISSUE bound (a lane holds 4 columns
//    of one row: a wave store touches 16 rows x 32 bytes) and the activation of 68 outputs per lane is serial time at the end of a
//    one-round launch - the frame kernel stages bf16 results through LDS and stores whole lines, and uses quick_gelu_bf.
// Three kernels, all the ring structure of gemm_tiled.hip with 32-column K-steps (so that 256-row tiles still leave room for a ring):
//  * gemm_ring32_kernel: BM x BN in {256x256, 128x128}, 16 waves as 4 x 4;
//  * gemm_ring272_kernel: ONE IMAGE PER ROW TILE - 257 valid rows computed as 17 MFMA row tiles (272 rows; the 15 extra rows are
//    the next image's first rows, computed and dropped): M = 257 n tiles exactly into n row tiles, 16 x 16 = 256 workgroups for the
//    c_fc GEMM of 16 frames (one round on 256 CUs instead of 272 tiles), 16 x 12 for in_proj; wave row 0 carries the 17th row tile
//    (5 | 4 | 4 | 4) - kept as the measured baseline of
//  * gemm_frame_kernel: the same row tile with the 17th MFMA row tile dealt out one 16 x 16 tile per wave (auto-selected by
//    gemm_dispatch when frames x N / BN is one round of 192-256 workgroups).
#include "common.h"
#include <type_traits>

// "BF16" = the family's 16-bit format (bf16, or fp16 when F16); P8_EPI_BF16OUT stores bf16 whatever the operands are
enum { P8_EPI_BF16 = DEER_E_16, P8_EPI_F32 = DEER_E_F32, P8_EPI_QGELU_BF16 = DEER_E_QGELU_16, P8_EPI_GELU_BF16 = DEER_E_GELU_16,
       P8_EPI_RESADD_F32 = DEER_E_RESADD_F32, P8_EPI_BF16OUT = DEER_E_BF16OUT };

typedef __attribute__((address_space(1))) const void p8_gptr_t;
typedef __attribute__((address_space(3))) void p8_lptr_t;

template <int N_>
__device__ __forceinline__ void p8_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory");
}

// ------------------------------------------------------------------------------------------------------------------------
// ring32: the ring structure of gemm_tiled.hip with 32-column K-steps, so that a 256x256 tile (128 FLOP per staged byte, twice
// the 128x128 kernels) still has 3-4 stages IN FLIGHT inside 160 KB of LDS (5 x 32 KB).
//  * 16 waves as 4 x 4, wave tile (BM/4) x (BN/4) (64x64: 4x4 MFMA tiles, 64 accumulator VGPRs, 8 ds_read_b128 per 16 MFMAs);
//  * LDS rows of 64 bytes: the 16-byte slot of a row is XOR-swizzled with f((row >> 2) & 3), f = {0, 2, 3, 1}, which makes every
//    16-lane group of a ds_read_b128 fragment read hit 16 distinct 16-byte bank units (rows c = 0..15 of a tile, slot g);
//  * one counted vmcnt + ONE raw s_barrier per K-step, every wave issues exactly CPW DMAs per step.
// a raw workgroup barrier that neither the IR optimiser nor the machine scheduler moves LDS reads / DMAs / MFMAs across
#define BIGM_SYNC()                             \
  do {                                          \
    asm volatile("" ::: "memory");              \
    __builtin_amdgcn_sched_barrier(0);          \
    __builtin_amdgcn_s_barrier();               \
    __builtin_amdgcn_sched_barrier(0);          \
    asm volatile("" ::: "memory");              \
  } while (0)

// STAG: the 16 waves run as TWO GROUPS of 8 (one wave row pair each; two waves of either group on every SIMD), group 1 one barrier
// behind group 0, and a K-step has two barriers (fragment reads + DMA issue | MFMAs): while one group issues its 16 MFMAs the other
// reads its fragments and issues its DMAs.  With ONE barrier per K-step every wave leaves the barrier in the same phase - 16 waves
// read fragments (LDS busy, matrix pipe idle), then 16 waves issue MFMAs (matrix pipe busy, LDS idle): 512 + 1024 cycles per step
// measured as ~2000 (ring32, 31 us per 256x256x1024 tile) although the same DMA pattern alone runs at 92 GB/s per CU = 11.4 us per
// tile (tools/fill_share.hip).  Hazards (interval = time between two barrier events; group 0 reads stage k in interval 2k and issues
// its MFMAs in 2k+1, group 1 one interval later): a slot read in interval Y may be refilled from Y+2 on - READ(k) refills the slot of
// stage k-2 with stage k+D-2; a wave's counted vmcnt in READ(k) retires stage k+1, which group 0 reads from interval 2k+2 on (after
// the barrier that follows group 1's wait).
template <int BM, int BN, int D, bool STAG = false, bool F16 = false>
__global__ __launch_bounds__(1024) void gemm_ring32_kernel(const bf16_t* __restrict__ A, int lda, long strideA,
                                                            const bf16_t* __restrict__ W, int ldw, long strideW,
                                                            const float* __restrict__ bias, void* __restrict__ Cv, int ldc,
                                                            long strideC, int M, int N, int K, int epi,
                                                            const float* __restrict__ gate, const int* ctl) {
  DEER_RETURN_IF_EXITED(ctl);
  constexpr int TM = BM / 64, TN = BN / 64;                 // MFMA tiles per wave (4 x 4 waves)
  constexpr int CH = (BM + BN) / 16;                        // 1 KiB DMA chunks (16 rows x 64 B) per stage
  constexpr int CPW = CH / 16;
  static_assert(CH % 16 == 0, "every wave issues the same number of DMAs");
  constexpr int STAGE = (BM + BN) * 64;
  static_assert(D * STAGE <= 160 * 1024 && (D - 2) * CPW <= 63, "LDS / vmcnt field");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int c = lane & 15, g = lane >> 4;

  // tile order: full row tiles first, ragged ones last in every XCD's dispatch order (thin tiles spread over the XCDs, dispatched after the full ones)
  constexpr int BMS = BM == 256 ? 8 : 7;
  static_assert(BM == 256 || BM == 128, "row tile");
  const int tiles_n = (N + BN - 1) / BN;
  const int nb = gridDim.x, bid = blockIdx.x;
  const int nfull = (M >> BMS) * tiles_n;
  const int xcd = bid & 7, idx = bid >> 3;
  int tile;
  {
    const int fq = nfull >> 3, fr = nfull & 7;
    const int fcount = fq + (xcd < fr ? 1 : 0), fstart = xcd * fq + min(xcd, fr);
    if (idx < fcount) tile = fstart + idx;
    else {
      const int bq = nb >> 3, br = nb & 7;
      tile = nfull + (xcd * bq + min(xcd, br) - fstart) + (idx - fcount);
    }
  }
  const int m0 = (tile / tiles_n) * BM, n0 = (tile % tiles_n) * BN;
  A += (long)blockIdx.z * strideA;
  W += (long)blockIdx.z * strideW;

  // DMA chunk q = wave + i*16: rows 16q .. 16q+15 of [A tile ; W tile]; lane: row 16q + (lane >> 2), source slot (lane & 3) ^ f
  const int lr = lane >> 2;
  const int fsw = (0x1320 >> (((lr >> 2) & 3) * 4)) & 3;    // f = {0, 2, 3, 1} for (row >> 2) & 3 = 0..3
  const int ls = ((lane & 3) ^ fsw) * 8;
  const bf16_t* sp[CPW];
#pragma unroll
  for (int i = 0; i < CPW; ++i) {
    const int row = (wave + i * 16) * 16 + lr;              // wave-uniform side: A chunk or W chunk
    sp[i] = (row < BM) ? A + (long)min(m0 + row, M - 1) * lda + ls : W + (long)min(n0 + row - BM, N - 1) * ldw + ls;
  }
  const int nk = K >> 5;
  auto issue = [&](int t) {
    const int k0 = min(t, nk - 1) << 5;
    unsigned char* st = smem + (t % D) * STAGE;
#pragma unroll
    for (int i = 0; i < CPW; ++i)
      __builtin_amdgcn_global_load_lds((p8_gptr_t*)(sp[i] + k0), (p8_lptr_t*)(st + (wave + i * 16) * 1024), 16, 0, 0);
  };

  f32x4 acc[TN][TM];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // fragment of MFMA tile t: row t*16 + c, slot g: byte = row * 64 + ((g ^ f((row >> 2) & 3)) << 4); (row >> 2) & 3 == (c >> 2) & 3
  const int fr_sw = (g ^ ((0x1320 >> (((c >> 2) & 3) * 4)) & 3)) << 4;
  const int a_off = (wm * (BM / 4) + c) * 64 + fr_sw;
  const int w_off = (BM + wn * (BN / 4) + c) * 64 + fr_sw;
  // valid MFMA row tiles of this wave (ragged last row tile)
  const int mv = max(0, min(TM, (M - (m0 + wm * (BM / 4)) + 15) >> 4));

  if constexpr (STAG) {
    static_assert(D >= 4, "staggered groups: stage k+1 must be resident while stage k is consumed and stage k-1 still read");
    constexpr int AHEAD = D - 2;
    const int grp = wave >> 3;
#pragma unroll
    for (int t = 0; t < AHEAD; ++t) issue(t);
    p8_wait_vmcnt<(AHEAD - 1) * CPW>();         // stage 0 landed
    BIGM_SYNC();
    if (grp == 1) BIGM_SYNC();                  // group 1 runs one barrier behind group 0
    for (int kt = 0; kt < nk; ++kt) {
      const unsigned char* st = smem + (kt % D) * STAGE;
      bf16x8 af[TM], wf[TN];
#pragma unroll
      for (int j = 0; j < TM; ++j) af[j] = *reinterpret_cast<const bf16x8*>(st + a_off + j * 16 * 64);
#pragma unroll
      for (int i = 0; i < TN; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(st + w_off + i * 16 * 64);
      issue(kt + AHEAD);                        // into the slot of stage kt-2
      p8_wait_vmcnt<(AHEAD - 1) * CPW>();       // stage kt+1 landed (this wave's part)
      BIGM_SYNC();
      __builtin_amdgcn_s_setprio(1);
#pragma unroll
      for (int j = 0; j < TM; ++j)
        if (j < mv) {
#pragma unroll
          for (int i = 0; i < TN; ++i) acc[i][j] = mfma16<F16>(wf[i], af[j], acc[i][j]);
        }
      __builtin_amdgcn_s_setprio(0);
      BIGM_SYNC();
    }
    if (grp == 0) BIGM_SYNC();
  } else {
#pragma unroll
  for (int t = 0; t < D - 1; ++t) issue(t);
  for (int kt = 0; kt < nk; ++kt) {
    p8_wait_vmcnt<(D - 2) * CPW>();             // this wave's part of stage kt has landed
    __builtin_amdgcn_s_barrier();               // ... everybody's; and everybody finished reading stage kt-1 (refilled now)
    issue(kt + D - 1);
    const unsigned char* st = smem + (kt % D) * STAGE;
    bf16x8 af[TM], wf[TN];
#pragma unroll
    for (int j = 0; j < TM; ++j) af[j] = *reinterpret_cast<const bf16x8*>(st + a_off + j * 16 * 64);
#pragma unroll
    for (int i = 0; i < TN; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(st + w_off + i * 16 * 64);
#pragma unroll
    for (int j = 0; j < TM; ++j)
      if (j < mv) {                              // wave-uniform: a ragged last row tile computes only the MFMA tiles that hold a valid row
#pragma unroll
        for (int i = 0; i < TN; ++i) acc[i][j] = mfma16<F16>(wf[i], af[j], acc[i][j]);
      }
  }
  }
  p8_wait_vmcnt<0>();

  const float gs = (epi == P8_EPI_RESADD_F32 && gate != nullptr) ? tanhf(*gate) : 1.f;
#pragma unroll
  for (int i = 0; i < TN; ++i) {
    const int n = n0 + wn * (BN / 4) + i * 16 + g * 4;
    if (n >= N) continue;
    float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
    if (bias != nullptr) {
      const float4 bv = *reinterpret_cast<const float4*>(bias + n);
      b0 = bv.x; b1 = bv.y; b2 = bv.z; b3 = bv.w;
    }
#pragma unroll
    for (int j = 0; j < TM; ++j) {
      const int m = m0 + wm * (BM / 4) + j * 16 + c;
      if (m >= M) continue;
      float v0 = acc[i][j][0] + b0, v1 = acc[i][j][1] + b1, v2 = acc[i][j][2] + b2, v3 = acc[i][j][3] + b3;
      const long off = (long)blockIdx.z * strideC + (long)m * ldc + n;
      if (epi == P8_EPI_RESADD_F32) {
        float4* p = reinterpret_cast<float4*>(reinterpret_cast<float*>(Cv) + off);
        float4 r = *p;
        r.x += gs * v0; r.y += gs * v1; r.z += gs * v2; r.w += gs * v3;
        *p = r;
      } else if (epi == P8_EPI_F32) {
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(Cv) + off) = float4{v0, v1, v2, v3};
      } else {
        if (epi == P8_EPI_QGELU_BF16) {
          v0 = quick_gelu_bf(v0); v1 = quick_gelu_bf(v1); v2 = quick_gelu_bf(v2); v3 = quick_gelu_bf(v3);
        } else if (epi == P8_EPI_GELU_BF16) {
          v0 = gelu_erf(v0); v1 = gelu_erf(v1); v2 = gelu_erf(v2); v3 = gelu_erf(v3);
        }
        *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(Cv) + off) = uint2{pack2_epi<F16>(v0, v1, epi), pack2_epi<F16>(v2, v3, epi)};
      }
    }
  }
}

// One image per row tile: `tile_rows` (257) valid rows per tile, computed as 17 MFMA row tiles; BN = 256 columns; 16 waves as 4 x 4 with
// row tiles 5 | 4 | 4 | 4 per wave row (one wave of every row on each SIMD) and 64 columns per wave column.  A stage is 17 A chunks (16
// rows x 64 B) + 16 W chunks: wave w issues A chunk w and W chunk w, wave 0 also A chunk 16 - its loop is instantiated with its own
// counted vmcnt (3 DMAs per stage instead of 2).
template <int D, bool F16>
__global__ __launch_bounds__(1024) void gemm_ring272_kernel(const bf16_t* __restrict__ A, int lda, long strideA,
                                                             const bf16_t* __restrict__ W, int ldw, long strideW,
                                                             const float* __restrict__ bias, void* __restrict__ Cv, int ldc,
                                                             long strideC, int M, int N, int K, int epi, int tile_rows,
                                                             const float* __restrict__ gate, const int* ctl) {
  DEER_RETURN_IF_EXITED(ctl);
  constexpr int TN = 4, TMX = 5;
  constexpr int A_BYTES = 17 * 1024, STAGE = 33 * 1024;
  static_assert(D * STAGE <= 160 * 1024 && (D - 2) * 3 <= 63, "LDS / vmcnt field");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int c = lane & 15, g = lane >> 4;
  const int t0 = wm == 0 ? 0 : 1 + 4 * wm;                   // first MFMA row tile of this wave: 0, 5, 9, 13
  const int tm = wm == 0 ? 5 : 4;

  // contiguous run of tiles (column index fastest) per XCD: the workgroups of an XCD share few A row tiles in its L2
  const int tiles_n = N >> 8;
  const int nb = gridDim.x, bid = blockIdx.x;
  const int xq = nb >> 3, xr = nb & 7, xcd = bid & 7;
  const int tile = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
  const int m0 = (tile / tiles_n) * tile_rows, n0 = (tile % tiles_n) << 8;
  const int rows_valid = min(tile_rows, M - m0);
  A += (long)blockIdx.z * strideA;
  W += (long)blockIdx.z * strideW;

  const int lr = lane >> 2;
  const int ls = ((lane & 3) ^ ((0x1320 >> (((lr >> 2) & 3) * 4)) & 3)) * 8;
  const bf16_t* spa = A + (long)min(m0 + wave * 16 + lr, M - 1) * lda + ls;
  const bf16_t* spw = W + (long)min(n0 + wave * 16 + lr, N - 1) * ldw + ls;
  const bf16_t* spx = A + (long)min(m0 + 256 + lr, M - 1) * lda + ls;      // A chunk 16 (wave 0 only)
  const int nk = K >> 5;

  f32x4 acc[TN][TMX];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TMX; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int fr_sw = (g ^ ((0x1320 >> (((c >> 2) & 3) * 4)) & 3)) << 4;
  const int a_off = (t0 * 16 + c) * 64 + fr_sw;
  const int w_off = A_BYTES + (wn * 64 + c) * 64 + fr_sw;

  auto run = [&](auto cpw_tag) {
    constexpr int CPW = decltype(cpw_tag)::value;
    auto issue = [&](int t) {
      const int k0 = min(t, nk - 1) << 5;
      unsigned char* st = smem + (t % D) * STAGE;
      __builtin_amdgcn_global_load_lds((p8_gptr_t*)(spa + k0), (p8_lptr_t*)(st + wave * 1024), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((p8_gptr_t*)(spw + k0), (p8_lptr_t*)(st + A_BYTES + wave * 1024), 16, 0, 0);
      if (CPW == 3) __builtin_amdgcn_global_load_lds((p8_gptr_t*)(spx + k0), (p8_lptr_t*)(st + 16 * 1024), 16, 0, 0);
    };
#pragma unroll
    for (int t = 0; t < D - 1; ++t) issue(t);
    for (int kt = 0; kt < nk; ++kt) {
      p8_wait_vmcnt<(D - 2) * CPW>();
      __builtin_amdgcn_s_barrier();
      issue(kt + D - 1);
      const unsigned char* st = smem + (kt % D) * STAGE;
      bf16x8 af[TMX];
#pragma unroll
      for (int j = 0; j < TMX; ++j)
        if (j < tm) af[j] = *reinterpret_cast<const bf16x8*>(st + a_off + j * 16 * 64);
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        const bf16x8 wf = *reinterpret_cast<const bf16x8*>(st + w_off + i * 16 * 64);
#pragma unroll
        for (int j = 0; j < TMX; ++j)
          if (j < tm) acc[i][j] = mfma16<F16>(wf, af[j], acc[i][j]);
      }
    }
    p8_wait_vmcnt<0>();
  };
  if (wave == 0) run(std::integral_constant<int, 3>{});
  else run(std::integral_constant<int, 2>{});

  const float gs = (epi == P8_EPI_RESADD_F32 && gate != nullptr) ? tanhf(*gate) : 1.f;
#pragma unroll
  for (int i = 0; i < TN; ++i) {
    const int n = n0 + wn * 64 + i * 16 + g * 4;
    if (n >= N) continue;
    float b0 = 0.f, b1 = 0.f, b2 = 0.f, b3 = 0.f;
    if (bias != nullptr) {
      const float4 bv = *reinterpret_cast<const float4*>(bias + n);
      b0 = bv.x; b1 = bv.y; b2 = bv.z; b3 = bv.w;
    }
#pragma unroll
    for (int j = 0; j < TMX; ++j) {
      const int r = (t0 + j) * 16 + c;                      // row inside the tile: rows >= rows_valid belong to the next tile / nobody
      if (j >= tm || r >= rows_valid) continue;
      const int m = m0 + r;
      float v0 = acc[i][j][0] + b0, v1 = acc[i][j][1] + b1, v2 = acc[i][j][2] + b2, v3 = acc[i][j][3] + b3;
      const long off = (long)blockIdx.z * strideC + (long)m * ldc + n;
      if (epi == P8_EPI_RESADD_F32) {
        float4* p = reinterpret_cast<float4*>(reinterpret_cast<float*>(Cv) + off);
        float4 rr = *p;
        rr.x += gs * v0; rr.y += gs * v1; rr.z += gs * v2; rr.w += gs * v3;
        *p = rr;
      } else if (epi == P8_EPI_F32) {
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(Cv) + off) = float4{v0, v1, v2, v3};
      } else {
        if (epi == P8_EPI_QGELU_BF16) {
          v0 = quick_gelu_bf(v0); v1 = quick_gelu_bf(v1); v2 = quick_gelu_bf(v2); v3 = quick_gelu_bf(v3);
        } else if (epi == P8_EPI_GELU_BF16) {
          v0 = gelu_erf(v0); v1 = gelu_erf(v1); v2 = gelu_erf(v2); v3 = gelu_erf(v3);
        }
        *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(Cv) + off) = uint2{pack2_epi<F16>(v0, v1, epi), pack2_epi<F16>(v2, v3, epi)};
      }
    }
  }
}

template <bool F16, int D>
static int launch_ring272(const bf16_t* A, int lda, long strideA, const bf16_t* W, int ldw, long strideW, const float* bias, void* C,
                          int ldc, long strideC, int M, int N, int K, int batch, int epi, const float* gate, const int* ctl,
                          hipStream_t st) {
  if ((N & 255) || (K & 31) || M <= 0 || batch <= 0) return DEER_ERR_SHAPE;
  const int tile_rows = (M % 257 == 0) ? 257 : 272;         // M = 257 n (n camera frames): one frame per row tile
  constexpr int smem_bytes = D * 33 * 1024;
  static std::atomic<bool> attr_set{false};
  auto kern = &gemm_ring272_kernel<D, F16>;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes) != hipSuccess)
      return DEER_ERR_LAUNCH;
    attr_set = true;
  }
  const int tiles = ((M + tile_rows - 1) / tile_rows) * (N >> 8);
  hipLaunchKernelGGL(kern, dim3(tiles, 1, batch), dim3(1024), smem_bytes, st, A, lda, strideA, W, ldw, strideW, bias, C, ldc, strideC, M, N,
                     K, epi, tile_rows, gate, ctl);
  DEER_LAUNCH_CHECK();
  return DEER_OK;
}

// One camera frame per row tile, BALANCED: the 16 waves (or 8) are 4 wave rows x WN wave columns, every wave owns 4 x TN MFMA tiles of
// the frame's first 256 rows and the 17th row tile (the frame's 257th row) is dealt out one 16x16 tile per wave - wave (wm, wn) takes
// column tile wm of its own column group, whose W fragment it already holds - so a K-step is 4*TN + 1 MFMAs on every wave with wm < TN
// instead of wave row 0 carrying 5*TN (ring272 above: +25 % on the critical path for +6 % work).  BN = 16 * WN * TN columns per
// workgroup: 16 frames x N / BN is exactly 256 workgroups for in_proj (BN = 192), c_fc (BN = 256) and the N = 1024 projections (BN = 64
// or BN = 128 with two K halves) - one round on the 256 CUs where M = 257 n never tiles into powers of two.
// A stage is 17 A chunks + BN / 16 W chunks of 1 KiB (16 rows x 64 B, the swizzle of ring32); chunk q = wave + i * NW; the first N_HI
// waves issue CPW DMAs per stage, the others CPW - 1 - each group runs the loop instantiated with its own counted vmcnt.
// TM = MFMA row tiles per wave: 4 = a whole frame per workgroup (16 + 1 row tiles); 2 = HALF a frame (8 + 1 row tiles: rows 0..128 of the
// frame, the 129th being the dealt-out row tile, or rows 129..256 with no extra tile) - 32 half frames x 1024 / 128 = 256 workgroups for
// the N = 1024 projections of 16 frames without splitting K.
#ifdef DEER_KTRACE
KT_DEFINE(frame)
// by shape: in_proj -> slots 0.., c_fc 8.., c_proj halves 16.. (stamps of workgroup 0)
#define FKT(slot) KT(frame, blockIdx.x == 0 && blockIdx.z == 0 && kt_base >= 0, kt_base + (slot))
#else
#define FKT(slot) do { } while (0)
#endif
template <int TM, int WN, int TN, int D, bool F16>
__global__ __launch_bounds__(WN * 256) void gemm_frame_kernel(const bf16_t* __restrict__ A, int lda, long strideA,
                                                               const bf16_t* __restrict__ W, int ldw, long strideW,
                                                               const float* __restrict__ bias, void* __restrict__ Cv, int ldc,
                                                               long strideC, int M, int N, int K, int epi, int tile_rows,
                                                               const float* __restrict__ gate, const int* ctl) {
  DEER_RETURN_IF_EXITED(ctl);
#ifdef DEER_KTRACE
  const int kt_base = (N == 3072) ? 0 : (N == 4096) ? 8 : (N == 1024) ? 16 : -1;
#endif
  FKT(0);
  constexpr int NW = 4 * WN, BN = 16 * WN * TN, RT = 4 * TM, CH = RT + 1 + BN / 16, STAGE = CH * 1024;
  constexpr int CPW = (CH + NW - 1) / NW, N_HI = CH - (CPW - 1) * NW;
  static_assert(TN >= 1 && TN <= 4 && (WN == 2 || WN == 4) && (TM == 2 || TM == 4), "wave grid");
  static_assert(D * STAGE <= 160 * 1024 && (D - 2) * CPW <= 63, "LDS / vmcnt field");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int c = lane & 15, g = lane >> 4;

  // workgroup b runs on XCD b % 8: every XCD gets a BLOCK of (row tiles / gr) x (column tiles / gc) tiles, gr x gc = 8 chosen so that
  // the rows of A plus the rows of W its L2 has to fetch are fewest (16 frames x 16 column tiles: 4 x 8 tiles per XCD = 12 operand
  // panels instead of the 2 x 16 = 18 of a contiguous run; PMC: 110 MB of L2 misses per c_fc launch against 17 MB of operands)
  const int tiles_n = N / BN;
  const int nb = gridDim.x, bid = blockIdx.x;
  const int rows_n = nb / tiles_n;
  int rt, ct;
  {
    int gr = 0, gc = 0;
    long best = 1L << 60;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int r_ = 1 << e, c_ = 8 >> e;
      if (rows_n % r_ == 0 && tiles_n % c_ == 0) {
        const long cost = (long)(rows_n / r_) * tile_rows + (long)(tiles_n / c_) * BN;
        if (cost < best) { best = cost; gr = r_; gc = c_; }
      }
    }
    if (gr != 0) {
      const int xcd = bid & 7, idx = bid >> 3, bc = tiles_n / gc, br = rows_n / gr;
      rt = (xcd / gc) * br + idx / bc;
      ct = (xcd % gc) * bc + idx % bc;
    } else {                                                  // contiguous run of tiles (column index fastest) per XCD
      const int xq = nb >> 3, xr = nb & 7, xcd = bid & 7;
      const int tile = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
      rt = tile / tiles_n;
      ct = tile % tiles_n;
    }
  }
  const int n0 = ct * BN;
  int m0, rows_here;
  if (TM == 4) { m0 = rt * tile_rows; rows_here = tile_rows; }
  else if (tile_rows == 257) { m0 = (rt >> 1) * 257 + (rt & 1) * 129; rows_here = (rt & 1) ? 128 : 129; }   // half frames
  else { m0 = rt * 128; rows_here = 128; }
  const int rows_valid = min(rows_here, M - m0);
  A += (long)blockIdx.z * strideA;
  W += (long)blockIdx.z * strideW;

  const int lr = lane >> 2;
  const int ls = ((lane & 3) ^ ((0x1320 >> (((lr >> 2) & 3) * 4)) & 3)) * 8;
  const bf16_t* sp[CPW];
#pragma unroll
  for (int i = 0; i < CPW; ++i) {
    const int q = min(wave + i * NW, CH - 1);                 // wave-uniform: an A chunk or a W chunk
    sp[i] = (q < RT + 1) ? A + (long)min(m0 + q * 16 + lr, M - 1) * lda + ls : W + (long)min(n0 + (q - RT - 1) * 16 + lr, N - 1) * ldw + ls;
  }
  const int nk = K >> 5;

  f32x4 acc[TN][TM], accx = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < TM; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int fr_sw = (g ^ ((0x1320 >> (((c >> 2) & 3) * 4)) & 3)) << 4;
  const int a_off = (wm * TM * 16 + c) * 64 + fr_sw;
  const int x_off = (RT * 16 + c) * 64 + fr_sw;
  const int w_off = (RT + 1) * 1024 + (wn * TN * 16 + c) * 64 + fr_sw;
  const bool has_x = wm < TN && rows_valid > RT * 16;         // the dealt-out row tile holds a valid row (not in the second half of a frame)

  auto run = [&](auto cpw_tag) {
    constexpr int CPWL = decltype(cpw_tag)::value;
    auto issue = [&](int t) {
      const int k0 = min(t, nk - 1) << 5;
      unsigned char* st = smem + (t % D) * STAGE;
#pragma unroll
      for (int i = 0; i < CPWL; ++i)
        __builtin_amdgcn_global_load_lds((p8_gptr_t*)(sp[i] + k0), (p8_lptr_t*)(st + (wave + i * NW) * 1024), 16, 0, 0);
    };
#pragma unroll
    for (int t = 0; t < D - 1; ++t) issue(t);
    FKT(1);
    for (int kt = 0; kt < nk; ++kt) {
      p8_wait_vmcnt<(D - 2) * CPWL>();
      __builtin_amdgcn_s_barrier();
      if (kt == 0) FKT(2);
      if (kt == 8) FKT(3);
      issue(kt + D - 1);
      const unsigned char* st = smem + (kt % D) * STAGE;
      bf16x8 af[TM], wf[TN], afx;
#pragma unroll
      for (int j = 0; j < TM; ++j) af[j] = *reinterpret_cast<const bf16x8*>(st + a_off + j * 1024);
#pragma unroll
      for (int i = 0; i < TN; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(st + w_off + i * 1024);
      if (has_x) afx = *reinterpret_cast<const bf16x8*>(st + x_off);
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TM; ++j) acc[i][j] = mfma16<F16>(wf[i], af[j], acc[i][j]);
#pragma unroll
      for (int e = 0; e < TN; ++e)
        if (has_x && wm == e) accx = mfma16<F16>(wf[e], afx, accx);
    }
    p8_wait_vmcnt<0>();
  };
  if (wave < N_HI) run(std::integral_constant<int, CPW>{});
  else run(std::integral_constant<int, CPW - 1>{});

  FKT(4);
  const float gs = (epi == P8_EPI_RESADD_F32 && gate != nullptr) ? tanhf(*gate) : 1.f;
  const bool to_bf16 = epi == P8_EPI_BF16 || epi == P8_EPI_QGELU_BF16 || epi == P8_EPI_GELU_BF16 || epi == P8_EPI_BF16OUT;
  // bf16 outputs leave through LDS: the accumulator layout gives a lane 4 columns of ONE row (8 bytes; a wave store touches 16 rows,
  // 32 bytes each - the epilogue of the direct form is store-ISSUE bound, ~5 us of a 40 us launch); staged, a lane stores 16 bytes
  // of a row and a wave 1 KiB of whole 128-byte lines, half as many instructions.  Row pitch BN*2 + 16 bytes: the ds_write_b64 of a
  // 16 x 16 tile (rows c, 8-byte slot g) and the ds_read_b128 rows are both conflict-free.
  constexpr int CPITCH = BN * 2 + 16, FPITCH = BN * 4 + 16;
  static_assert((RT + 1) * 16 * CPITCH <= 160 * 1024, "C staging tile");
  // f32 slabs take the same route when the tile fits (BN <= 128: 272 x 528 B = 140 KB): ds_write_b128 of a 16 x 16 tile and the row
  // reads are conflict-free at this pitch; a wave then stores two whole 512-byte rows instead of 16 rows x 64 bytes
  constexpr bool F32_STAGE = (RT + 1) * 16 * FPITCH <= 160 * 1024;
  const bool f32_staged = F32_STAGE && epi == P8_EPI_F32;
  auto store = [&](int r, int n, const f32x4& a) {            // row r of the frame, columns n0 + n .. n0 + n + 3
    float v0 = a[0], v1 = a[1], v2 = a[2], v3 = a[3];
    if (bias != nullptr) {
      const float4 bv = *reinterpret_cast<const float4*>(bias + n0 + n);
      v0 += bv.x; v1 += bv.y; v2 += bv.z; v3 += bv.w;
    }
    if (to_bf16) {
      if (epi == P8_EPI_QGELU_BF16) {
        v0 = quick_gelu_bf(v0); v1 = quick_gelu_bf(v1); v2 = quick_gelu_bf(v2); v3 = quick_gelu_bf(v3);
      } else if (epi == P8_EPI_GELU_BF16) {
        v0 = gelu_erf(v0); v1 = gelu_erf(v1); v2 = gelu_erf(v2); v3 = gelu_erf(v3);
      }
      *reinterpret_cast<uint2*>(smem + r * CPITCH + n * 2) = uint2{pack2_epi<F16>(v0, v1, epi), pack2_epi<F16>(v2, v3, epi)};
      return;
    }
    if (f32_staged) {
      *reinterpret_cast<float4*>(smem + r * FPITCH + n * 4) = float4{v0, v1, v2, v3};
      return;
    }
    if (r >= rows_valid) return;
    const long off = (long)blockIdx.z * strideC + (long)(m0 + r) * ldc + n0 + n;
    if (epi == P8_EPI_RESADD_F32) {
      float4* p = reinterpret_cast<float4*>(reinterpret_cast<float*>(Cv) + off);
      float4 rr = *p;
      rr.x += gs * v0; rr.y += gs * v1; rr.z += gs * v2; rr.w += gs * v3;
      *p = rr;
    } else {
      *reinterpret_cast<float4*>(reinterpret_cast<float*>(Cv) + off) = float4{v0, v1, v2, v3};
    }
  };
  const bool staged = to_bf16 || f32_staged;
  if (staged) BIGM_SYNC();                                    // every wave has read its last fragments: the ring becomes the C tile
#pragma unroll
  for (int i = 0; i < TN; ++i) {
    const int n = (wn * TN + i) * 16 + g * 4;
#pragma unroll
    for (int j = 0; j < TM; ++j) store((wm * TM + j) * 16 + c, n, acc[i][j]);
  }
  if (has_x) store(RT * 16 + c, (wn * TN + wm) * 16 + g * 4, accx);
  if (staged) BIGM_SYNC();
  FKT(5);
  if (to_bf16) {
    constexpr int PPR = BN / 8;                               // 16-byte pieces per row
    const int pieces = rows_valid * PPR;
    bf16_t* Cb = reinterpret_cast<bf16_t*>(Cv) + (long)blockIdx.z * strideC + (long)m0 * ldc + n0;
    for (int p = tid; p < pieces; p += NW * 64) {
      const int r = p / PPR, cp = p - r * PPR;
      *reinterpret_cast<uint4*>(Cb + (long)r * ldc + cp * 8) = *reinterpret_cast<const uint4*>(smem + r * CPITCH + cp * 16);
    }
  } else if (f32_staged) {
    constexpr int PPR = BN / 4;
    const int pieces = rows_valid * PPR;
    float* Cf = reinterpret_cast<float*>(Cv) + (long)blockIdx.z * strideC + (long)m0 * ldc + n0;
    for (int p = tid; p < pieces; p += NW * 64) {
      const int r = p / PPR, cp = p - r * PPR;
      *reinterpret_cast<uint4*>(Cf + (long)r * ldc + cp * 4) = *reinterpret_cast<const uint4*>(smem + r * FPITCH + cp * 16);
    }
  }
  FKT(6);
}

template <bool F16, int TM, int WN, int TN, int D>
static int launch_frame(const bf16_t* A, int lda, long strideA, const bf16_t* W, int ldw, long strideW, const float* bias, void* C,
                        int ldc, long strideC, int M, int N, int K, int batch, int epi, const float* gate, const int* ctl,
                        hipStream_t st) {
  constexpr int BN = 16 * WN * TN;
  if ((N % BN) || (K & 31) || M <= 0 || batch <= 0) return DEER_ERR_SHAPE;
  const int tile_rows = (M % 257 == 0) ? 257 : 256;         // M = 257 n (n camera frames): one frame per row tile
  constexpr int ring_bytes = D * (4 * TM + 1 + BN / 16) * 1024, c_bytes = (4 * TM + 1) * 16 * (BN * 2 + 16);
  constexpr int f_bytes = (4 * TM + 1) * 16 * (BN * 4 + 16) <= 160 * 1024 ? (4 * TM + 1) * 16 * (BN * 4 + 16) : 0;
  constexpr int rc_bytes = ring_bytes > c_bytes ? ring_bytes : c_bytes;
  constexpr int smem_bytes = rc_bytes > f_bytes ? rc_bytes : f_bytes;   // the ring, then the staged C tile (bf16, or f32 where it fits)
  static std::atomic<bool> attr_set{false};
  auto kern = &gemm_frame_kernel<TM, WN, TN, D, F16>;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes) != hipSuccess)
      return DEER_ERR_LAUNCH;
    attr_set = true;
  }
  const int row_tiles = TM == 4 ? (M + tile_rows - 1) / tile_rows : (tile_rows == 257 ? 2 * (M / 257) : (M + 127) / 128);
  const int tiles = row_tiles * (N / BN);
  hipLaunchKernelGGL(kern, dim3(tiles, 1, batch), dim3(WN * 256), smem_bytes, st, A, lda, strideA, W, ldw, strideW, bias, C, ldc, strideC, M,
                     N, K, epi, tile_rows, gate, ctl);
  DEER_LAUNCH_CHECK();
  return DEER_OK;
}

// ------------------------------------------------------------------------------------------------------------------------
// frame8 (round 5; prototyped in round 3 as tools/frame8.hip): the frame tile on EIGHT waves.  The 16-wave kernel above spends as many
// LDS cycles on fragment reads as its SIMDs spend on MFMAs (tools/ktrace_frame.py: 1.0 us per K-step of 32 against 0.52 us of MFMA
// issue; 16 waves x 9 ds_read_b128 = 147 KB per K-step - read in round 5 as an LDS rate bound at 128 B/clk; ds_read_b128 moves 256 B/clk
// and the four-wave tile below, with 2.1 x fewer reads, has a 19 % faster loop, not 2 x); 16 waves leave 128 VGPRs per wave, which excludes
// bigger wave tiles.  Here 4 x 2 waves, wave tile 64 rows x (TN x 16) columns (BN = 32 TN: 256 or 192): 34 (26) MFMAs behind 13 (11)
// fragment reads per K-step and wave instead of 17 behind 9, the reads running one W fragment ahead of the MFMAs
// (sched_group_barrier), the LDS-DMA in its MUBUF form so that the compiler counts lgkmcnt for the fragment reads alone (32-bit byte
// offsets: operands below 4 GiB).  The 17th MFMA row tile (the frame's 257th row) is dealt out XT 16 x 16 tiles per wave.  bf16
// epilogues only (bias, QuickGELU / GELU), staged through LDS like the 16-wave kernel.
__device__ __forceinline__ void f8_dma16(const void* base, unsigned voff, unsigned soff, void* lds) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(__builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, -1, 0x00020000), (p8_lptr_t*)lds, 16, voff,
                                           soff, 0, 0);
}

template <int TN, int D, bool F16>
__global__ __launch_bounds__(512) void gemm_frame8_kernel(const bf16_t* __restrict__ A, int lda, long strideA, const bf16_t* __restrict__ W,
                                                           int ldw, long strideW, const float* __restrict__ bias, void* __restrict__ Cv, int ldc,
                                                           long strideC, int M, int N, int K, int epi, const int* ctl) {
  DEER_RETURN_IF_EXITED(ctl);
  constexpr int NW = 8, BN = 32 * TN, CH = 17 + BN / 16, STAGE = CH * 1024, XT = BN / 16 / 8 + ((BN / 16) % 8 ? 1 : 0);
  constexpr int CPW = (CH + NW - 1) / NW, N_HI = CH - (CPW - 1) * NW;
  static_assert(D * STAGE <= 160 * 1024 && (D - 2) * CPW <= 63, "LDS / vmcnt field");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int c = lane & 15, g = lane >> 4;
  const int tiles_n = N / BN, rows_n = gridDim.x / tiles_n;
  int rt, ct;
  {                                                          // XCD blocks, as in gemm_frame_kernel
    int gr = 0, gc = 0;
    long best = 1L << 60;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int r_ = 1 << e, c_ = 8 >> e;
      if (rows_n % r_ == 0 && tiles_n % c_ == 0) {
        const long cost = (long)(rows_n / r_) * 257 + (long)(tiles_n / c_) * BN;
        if (cost < best) { best = cost; gr = r_; gc = c_; }
      }
    }
    const int bid = blockIdx.x, nb = gridDim.x;
    if (gr != 0) {
      const int xcd = bid & 7, idx = bid >> 3, bc = tiles_n / gc, br = rows_n / gr;
      rt = (xcd / gc) * br + idx / bc;
      ct = (xcd % gc) * bc + idx % bc;
    } else {
      const int xq = nb >> 3, xr = nb & 7, xcd = bid & 7;
      const int tile = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
      rt = tile / tiles_n;
      ct = tile % tiles_n;
    }
  }
  const int m0 = rt * 257, n0 = ct * BN;
  const int rows_valid = min(257, M - m0);
  A += (long)blockIdx.z * strideA;
  W += (long)blockIdx.z * strideW;

  const int lr = lane >> 2;
  const int ls = ((lane & 3) ^ ((0x1320 >> (((lr >> 2) & 3) * 4)) & 3)) * 8;
  const bf16_t* base[CPW];
  unsigned vo[CPW];
#pragma unroll
  for (int i = 0; i < CPW; ++i) {
    const int q = min(wave + i * NW, CH - 1);
    const bool is_a = q < 17;
    base[i] = is_a ? A : W;
    vo[i] = is_a ? (unsigned)(((long)min(m0 + q * 16 + lr, M - 1) * lda + ls) * 2) : (unsigned)(((long)min(n0 + (q - 17) * 16 + lr, N - 1) * ldw + ls) * 2);
  }
  const int nk = K >> 5;
  f32x4 acc[TN][4], accx[XT];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int x = 0; x < XT; ++x) accx[x] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int fr_sw = (g ^ ((0x1320 >> (((c >> 2) & 3) * 4)) & 3)) << 4;
  const int a_off = (wm * 64 + c) * 64 + fr_sw;
  const int x_off = (256 + c) * 64 + fr_sw;
  const int w_off = 17 * 1024 + (wn * TN * 16 + c) * 64 + fr_sw;
  // the dealt-out row tile: column tiles xi = wm * XT + x of this wave's column group (x < XT), if xi < TN
  auto run = [&](auto cpw_tag) {
    constexpr int CPWL = decltype(cpw_tag)::value;
    auto issue = [&](int t) {
      const int k0 = min(t, nk - 1) << 5;
      unsigned char* st = smem + (t % D) * STAGE;
#pragma unroll
      for (int i = 0; i < CPWL; ++i) f8_dma16(base[i], vo[i], k0 * 2, st + (wave + i * NW) * 1024);
    };
#pragma unroll
    for (int t = 0; t < D - 1; ++t) issue(t);
    for (int kt = 0; kt < nk; ++kt) {
      p8_wait_vmcnt<(D - 2) * CPWL>();
      __builtin_amdgcn_s_barrier();
      issue(kt + D - 1);
      const unsigned char* st = smem + (kt % D) * STAGE;
      bf16x8 af[4], wf[TN];
#pragma unroll
      for (int j = 0; j < 4; ++j) af[j] = *reinterpret_cast<const bf16x8*>(st + a_off + j * 1024);
#pragma unroll
      for (int i = 0; i < TN; ++i) wf[i] = *reinterpret_cast<const bf16x8*>(st + w_off + i * 1024);
      const bf16x8 afx = *reinterpret_cast<const bf16x8*>(st + x_off);
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = mfma16<F16>(wf[i], af[j], acc[i][j]);
#pragma unroll
      for (int x = 0; x < XT; ++x) {
        bf16x8 wx = wf[0];
#pragma unroll
        for (int e = 1; e < TN; ++e) wx = (wm * XT + x == e) ? wf[e] : wx;
        accx[x] = mfma16<F16>(wx, afx, accx[x]);
      }
      // A fragments + the first TWO W fragments, then per W fragment: its 4 MFMAs, the read of the fragment after next (reads run one
      // group of MFMAs ahead); the dealt-out tiles last
      __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
      for (int i = 0; i < TN; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // W fragment i+2 ... and finally the extra A fragment (no-op once exhausted)
      }
      __builtin_amdgcn_sched_group_barrier(0x008, XT, 0);
    }
    p8_wait_vmcnt<0>();
  };
  if (wave < N_HI) run(std::integral_constant<int, CPW>{});
  else run(std::integral_constant<int, CPW - 1>{});

  constexpr int CPITCH = BN * 2 + 16;
  static_assert(272 * CPITCH <= 160 * 1024, "C staging");
  auto put = [&](int r, int n, const f32x4& a) {               // row r of the frame, columns n0 + n .. n0 + n + 3
    float v0 = a[0], v1 = a[1], v2 = a[2], v3 = a[3];
    if (bias != nullptr) {
      const float4 bv = *reinterpret_cast<const float4*>(bias + n0 + n);
      v0 += bv.x; v1 += bv.y; v2 += bv.z; v3 += bv.w;
    }
    if (epi == P8_EPI_QGELU_BF16) {
      v0 = quick_gelu_bf(v0); v1 = quick_gelu_bf(v1); v2 = quick_gelu_bf(v2); v3 = quick_gelu_bf(v3);
    } else if (epi == P8_EPI_GELU_BF16) {
      v0 = gelu_erf(v0); v1 = gelu_erf(v1); v2 = gelu_erf(v2); v3 = gelu_erf(v3);
    }
    *reinterpret_cast<uint2*>(smem + r * CPITCH + n * 2) = uint2{pack2_epi<F16>(v0, v1, epi), pack2_epi<F16>(v2, v3, epi)};
  };
  BIGM_SYNC();                                                 // every wave has read its last fragments: the ring becomes the C tile
#pragma unroll
  for (int i = 0; i < TN; ++i) {
    const int n = (wn * TN + i) * 16 + g * 4;
#pragma unroll
    for (int j = 0; j < 4; ++j) put(wm * 64 + j * 16 + c, n, acc[i][j]);
  }
#pragma unroll
  for (int x = 0; x < XT; ++x) {
    const int xi = wm * XT + x;
    if (xi < TN) put(256 + c, (wn * TN + xi) * 16 + g * 4, accx[x]);
  }
  BIGM_SYNC();
  constexpr int PPR = BN / 8;
  const int pieces = rows_valid * PPR;
  bf16_t* Cb = reinterpret_cast<bf16_t*>(Cv) + (long)blockIdx.z * strideC + (long)m0 * ldc + n0;
  for (int p = tid; p < pieces; p += 512) {
    const int r = p / PPR, cp = p - r * PPR;
    *reinterpret_cast<uint4*>(Cb + (long)r * ldc + cp * 8) = *reinterpret_cast<const uint4*>(smem + r * CPITCH + cp * 16);
  }
}

template <bool F16, int TN, int D>
static int launch_frame8(const bf16_t* A, int lda, long strideA, const bf16_t* W, int ldw, long strideW, const float* bias, void* C, int ldc,
                         long strideC, int M, int N, int K, int batch, int epi, const float* gate, const int* ctl, hipStream_t st) {
  constexpr int BN = 32 * TN;
  const bool to_bf16 = epi == P8_EPI_BF16 || epi == P8_EPI_QGELU_BF16 || epi == P8_EPI_GELU_BF16 || epi == P8_EPI_BF16OUT;
  if ((N % BN) || (K & 31) || M <= 0 || (M % 257) || batch <= 0 || !to_bf16) return DEER_ERR_SHAPE;
  // MUBUF byte offsets are 32 bits: every operand (one batch slice) has to end below 4 GiB
  if ((long)M * lda * 2 >= (1L << 32) || (long)N * ldw * 2 >= (1L << 32)) return DEER_ERR_SHAPE;
  constexpr int ring_bytes = D * (17 + BN / 16) * 1024, c_bytes = 272 * (BN * 2 + 16);
  constexpr int smem_bytes = ring_bytes > c_bytes ? ring_bytes : c_bytes;
  static std::atomic<bool> attr_set{false};
  auto kern = &gemm_frame8_kernel<TN, D, F16>;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes) != hipSuccess)
      return DEER_ERR_LAUNCH;
    attr_set = true;
  }
  const int tiles = (M / 257) * (N / BN);
  hipLaunchKernelGGL(kern, dim3(tiles, 1, batch), dim3(512), smem_bytes, st, A, lda, strideA, W, ldw, strideW, bias, C, ldc, strideC, M, N, K, epi,
                     ctl);
  DEER_LAUNCH_CHECK();
  return DEER_OK;
}

// ------------------------------------------------------------------------------------------------------------------------
// frame4 (round 6; prototyped as tools/frame4.hip): the frame tile on FOUR waves, 2 x 2, wave tile 128 rows x (TN x 16) columns
// (BN = 32 TN: 256, 192 or 128) - 17 fragment reads per 8 TN + TN / 2 MFMAs and wave.  What the prototype measured about this shape:
//  * with 256 threads per workgroup hipcc selects the AGPR form of the MFMA with dst != src C and copies every accumulator around every
//    MFMA (2.5 v_accvgpr_* per MFMA in the loop): the MFMAs are asm statements with the accumulator TIED in the accumulator file
//    ("+a"; TN = 8: all 256 of them).  Each accumulator is touched once per K-step (>= 34 MFMAs apart: no dependent-MFMA hazard);
//    `s_nop 15` separates the last MFMA from the epilogue's reads.  The dealt-out tiles of the 17th row tile accumulate in VGPRs ("+v"),
//    their W fragment chosen by v_cndmask (a branch on the wave row makes hipcc copy those accumulators with v_mov right behind the asm
//    MFMAs, whose latency it does not know: stale values), s_nop 1 between the select and the MFMA;
//  * one wave per SIMD: nothing else hides a wait.  The fragments of K-step kt+1 are read into a second register set while the MFMAs of
//    K-step kt issue, in two batches (a wave has at most 15 LDS operations in flight), the batch the head of the next K-step needs LAST
//    so that the compiler's lgkmcnt there is 0 and never lands behind new reads;
//  * an LDS-DMA piece costs its wave ~60 issue cycles (MI355X_MICROARCH.md): as a burst behind the barrier they run while the matrix
//    pipe is empty - dealt out one per MFMA group: c_fc at 16 frames 43.2 -> 40.4 us.
// Same K order per output element as the 16-wave tile: bit-identical results (tests/test_hip_ops.py compares every element in both 16-bit
// formats).  Measured (profiles/r06_g_*): the K loop is 19 % faster (26.1 against 32.2 us per 257 x 256 x 1024 tile); plain launches on
// cold weights, 16 frames: c_fc 39.8 us against 44.4, in_proj 31.4 against 33.8, K = 4096 117 against 140; 32 frames (two rounds), graph
// replay: c_fc 83.2 against 94.0, in_proj 61.4 against 66.8.  IN THE STEP it loses: with bias + QuickGELU on fp16 operands c_fc takes 53.0
// us against 46.6, in_proj 38.7 against 35.8, the c_proj halves 48.8 against 47.0 (rocprofv3 kernel trace of the 8-environment step) and
// `batched` goes from 1052-1061 to 1023-1024 env-steps/s: a lane holds 4 x the output tiles of the 16-wave kernel, so the epilogue was
// straight-line code over 68 tiles on ONE wave per SIMD (the transcendentals of QuickGELU and the LDS staging are latency-, not
// rate-bound there) and what the loop gains the epilogue gave back.  With the rolled epilogue below (f32 half tiles through LDS) the
// deficit halves: c_fc 52.3 against 49.2, in_proj 39.7 against 37.5, c_proj halves 49.1 against 48.8 (another box).  Shipped as tiles
// 76-79, selected only by DEER_GEMM_FRAME4=1 (or 2: two-round launches only); the selector keeps the 16-wave tiles.
template <bool F16>
__device__ __forceinline__ void f4_mma_a(f32x4& acc, const bf16x8& w, const bf16x8& a) {
  if constexpr (F16) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(w), "v"(a));
  else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(w), "v"(a));
}
template <bool F16>
__device__ __forceinline__ void f4_mma_v(f32x4& acc, const bf16x8& w, const bf16x8& a) {
  if constexpr (F16) asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(a));
  else asm volatile("s_nop 1\n\tv_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(w), "v"(a));
}

template <int TN, int D, bool F16>
__global__ __launch_bounds__(256) void gemm_frame4_kernel(const bf16_t* __restrict__ A, int lda, long strideA, const bf16_t* __restrict__ W,
                                                           int ldw, long strideW, const float* __restrict__ bias, void* __restrict__ Cv, int ldc,
                                                           long strideC, int M, int N, int K, int epi, const int* ctl) {
  DEER_RETURN_IF_EXITED(ctl);
  constexpr int NW = 4, BN = 32 * TN, CH = 17 + BN / 16, STAGE = CH * 1024, XT = TN / 2;
  constexpr int CPW = (CH + NW - 1) / NW, N_HI = CH - (CPW - 1) * NW;
  static_assert(TN == 8 || TN == 6 || TN == 4, "wave tile");
  static_assert(D >= 3 && D * STAGE <= 160 * 1024 && (D - 2) * CPW <= 63 && CPW <= 9, "LDS / vmcnt field / DMA slots of a K-step");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int c = lane & 15, g = lane >> 4;
  const int tiles_n = N / BN, rows_n = gridDim.x / tiles_n;
  int rt, ct;
  {                                                          // XCD blocks, as in gemm_frame_kernel
    int gr = 0, gc = 0;
    long best = 1L << 60;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int r_ = 1 << e, c_ = 8 >> e;
      if (rows_n % r_ == 0 && tiles_n % c_ == 0) {
        const long cost = (long)(rows_n / r_) * 257 + (long)(tiles_n / c_) * BN;
        if (cost < best) { best = cost; gr = r_; gc = c_; }
      }
    }
    const int bid = blockIdx.x, nb = gridDim.x;
    if (gr != 0) {
      const int xcd = bid & 7, idx = bid >> 3, bc = tiles_n / gc, br = rows_n / gr;
      rt = (xcd / gc) * br + idx / bc;
      ct = (xcd % gc) * bc + idx % bc;
    } else {
      const int xq = nb >> 3, xr = nb & 7, xcd = bid & 7;
      const int tile = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
      rt = tile / tiles_n;
      ct = tile % tiles_n;
    }
  }
  const int m0 = rt * 257, n0 = ct * BN;
  const int rows_valid = min(257, M - m0);
  A += (long)blockIdx.z * strideA;
  W += (long)blockIdx.z * strideW;

  const int lr = lane >> 2;
  const int ls = ((lane & 3) ^ ((0x1320 >> (((lr >> 2) & 3) * 4)) & 3)) * 8;
  const bf16_t* base[CPW];
  unsigned vo[CPW];
#pragma unroll
  for (int i = 0; i < CPW; ++i) {
    const int q = min(wave + i * NW, CH - 1);
    const bool is_a = q < 17;
    base[i] = is_a ? A : W;
    vo[i] = is_a ? (unsigned)(((long)min(m0 + q * 16 + lr, M - 1) * lda + ls) * 2) : (unsigned)(((long)min(n0 + (q - 17) * 16 + lr, N - 1) * ldw + ls) * 2);
  }
  const int nk = K >> 5;                                      // even (launcher)
  f32x4 acc[TN][8], accx[XT];
#pragma unroll
  for (int i = 0; i < TN; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int x = 0; x < XT; ++x) accx[x] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int fr_sw = (g ^ ((0x1320 >> (((c >> 2) & 3) * 4)) & 3)) << 4;
  const int a_off = (wm * 128 + c) * 64 + fr_sw;
  const int x_off = (256 + c) * 64 + fr_sw;
  const int w_off = 17 * 1024 + (wn * TN * 16 + c) * 64 + fr_sw;

  struct Frags { bf16x8 a[8], w[TN], x; };
  auto load_frags_a = [&](Frags& F, int stage) {              // first batch: the extra row tile, A fragments 1..7
    const unsigned char* st = smem + (stage % D) * STAGE;
    F.x = *reinterpret_cast<const bf16x8*>(st + x_off);
#pragma unroll
    for (int j = 1; j < 8; ++j) F.a[j] = *reinterpret_cast<const bf16x8*>(st + a_off + j * 1024);
  };
  auto load_frags_b = [&](Frags& F, int stage) {              // second batch: what the head of the next K-step needs
    const unsigned char* st = smem + (stage % D) * STAGE;
#pragma unroll
    for (int i = 0; i < TN; ++i) F.w[i] = *reinterpret_cast<const bf16x8*>(st + w_off + i * 1024);
    F.a[0] = *reinterpret_cast<const bf16x8*>(st + a_off);
  };
  auto mma_j = [&](const Frags& F, int j) {
#pragma unroll
    for (int i = 0; i < TN; ++i) f4_mma_a<F16>(acc[i][j], F.w[i], F.a[j]);
  };
  auto mma_x = [&](const Frags& F) {
#pragma unroll
    for (int x = 0; x < XT; ++x) {
      const bf16x8 wx = wm ? F.w[XT + x] : F.w[x];
      f4_mma_v<F16>(accx[x], wx, F.x);
    }
  };
  auto run = [&](auto cpw_tag) {
    constexpr int CPWL = decltype(cpw_tag)::value;
    auto issue_one = [&](int t, int i) {
      if (i < CPWL) {
        const int k0 = min(t, nk - 1) << 5;
        f8_dma16(base[i], vo[i], k0 * 2, smem + (t % D) * STAGE + (wave + i * NW) * 1024);
      }
    };
    auto half = [&](Frags& Fc, Frags& Fn, int kt) {
      p8_wait_vmcnt<(D - 3) * CPWL>();                        // this wave's share of stage kt+1 has landed
      __builtin_amdgcn_s_barrier();                           // ... everybody's; and every wave is done with the fragments of stage kt-1
      const int t = kt + D - 1;                               // into the slot of stage kt-1
      mma_j(Fc, 0);
      asm volatile("" ::: "memory");                          // the reads below stay below the head
      issue_one(t, 0); issue_one(t, 1);
      load_frags_a(Fn, kt + 1);
      mma_j(Fc, 1); issue_one(t, 2);
      mma_j(Fc, 2); issue_one(t, 3);
      mma_j(Fc, 3); issue_one(t, 4);
      asm volatile("" ::: "memory");
      load_frags_b(Fn, kt + 1);
      mma_j(Fc, 4); issue_one(t, 5);
      mma_j(Fc, 5); issue_one(t, 6);
      mma_j(Fc, 6); issue_one(t, 7);
      mma_j(Fc, 7); issue_one(t, 8);
      mma_x(Fc);
    };
    Frags F0, F1;
#pragma unroll
    for (int t = 0; t < D - 1; ++t)
#pragma unroll
      for (int i = 0; i < CPWL; ++i) issue_one(t, i);
    p8_wait_vmcnt<(D - 2) * CPWL>();
    __builtin_amdgcn_s_barrier();
    load_frags_a(F0, 0);
    load_frags_b(F0, 0);
    for (int kt = 0; kt < nk; kt += 2) {
      half(F0, F1, kt);
      half(F1, F0, kt + 1);
    }
    p8_wait_vmcnt<0>();
    asm volatile("s_nop 15" ::: "memory");
  };
  if (wave < N_HI) run(std::integral_constant<int, CPW>{});
  else run(std::integral_constant<int, CPW - 1>{});

  // epilogue.  A lane holds 8 TN + TN / 2 output tiles (4 x the 16-wave kernel's): bias / activation / pack / staging as straight-line code
  // over them on ONE wave per SIMD is latency-bound and gave back more than the loop gains (first form of this kernel, c_fc in the step
  // 53.0 us against 46.6).  Here the RAW f32 accumulators of one wave row (128 frame rows; with the second row also the frame's 257th
  // row) go to LDS, and all four waves run a rolled loop over the f32 half tile: bias, activation, 16-bit pack (or f32), whole-line
  // stores.  Same arithmetic per element as the other frame tiles: bit-identical results.
  constexpr int FP = BN * 4 + 16;                             // f32 row pitch: ds_write_b128 of a 16 x 16 tile conflict-free (as FPITCH of gemm_frame_kernel)
  static_assert(129 * FP <= 160 * 1024, "f32 half tile");
  const bool f32_out = epi == P8_EPI_F32;
  BIGM_SYNC();                                                 // every wave has read its last fragments: the ring becomes the C tile
#pragma unroll 1
  for (int h = 0; h < 2; ++h) {
    if (wm == h) {
#pragma unroll
      for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j)
          *reinterpret_cast<f32x4*>(smem + (j * 16 + c) * FP + ((wn * TN + i) * 16 + g * 4) * 4) = acc[i][j];
    }
    if (h == 1) {                                             // row 256 of the frame = row 128 of the second half tile (lanes c = 0)
#pragma unroll
      for (int x = 0; x < XT; ++x)
        if (c == 0) *reinterpret_cast<f32x4*>(smem + 128 * FP + ((wn * TN + wm * XT + x) * 16 + g * 4) * 4) = accx[x];
    }
    BIGM_SYNC();
    const int rows = min(h == 0 ? 128 : 129, rows_valid - h * 128);
    const long crow = (long)blockIdx.z * strideC + (long)(m0 + h * 128) * ldc + n0;
    if (f32_out) {
      constexpr int PPR = BN / 4;
      float* Cf = reinterpret_cast<float*>(Cv) + crow;
      for (int p = tid; p < rows * PPR; p += 256) {
        const int r = p / PPR, cp = p - r * PPR;
        float4 v = *reinterpret_cast<const float4*>(smem + r * FP + cp * 16);
        if (bias != nullptr) {
          const float4 bv = *reinterpret_cast<const float4*>(bias + n0 + cp * 4);
          v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
        }
        *reinterpret_cast<float4*>(Cf + (long)r * ldc + cp * 4) = v;
      }
    } else {
      constexpr int PPR = BN / 8;                             // 16-byte output pieces (8 columns) per row
      bf16_t* Cb = reinterpret_cast<bf16_t*>(Cv) + crow;
      const bool qg = epi == P8_EPI_QGELU_BF16;
#pragma unroll 2
      for (int p = tid; p < rows * PPR; p += 256) {
        const int r = p / PPR, cp = p - r * PPR;
        float4 a = *reinterpret_cast<const float4*>(smem + r * FP + cp * 32);
        float4 b = *reinterpret_cast<const float4*>(smem + r * FP + cp * 32 + 16);
        if (bias != nullptr) {
          const float4 ba = *reinterpret_cast<const float4*>(bias + n0 + cp * 8);
          const float4 bb = *reinterpret_cast<const float4*>(bias + n0 + cp * 8 + 4);
          a.x += ba.x; a.y += ba.y; a.z += ba.z; a.w += ba.w;
          b.x += bb.x; b.y += bb.y; b.z += bb.z; b.w += bb.w;
        }
        if (qg) {
          a.x = quick_gelu_bf(a.x); a.y = quick_gelu_bf(a.y); a.z = quick_gelu_bf(a.z); a.w = quick_gelu_bf(a.w);
          b.x = quick_gelu_bf(b.x); b.y = quick_gelu_bf(b.y); b.z = quick_gelu_bf(b.z); b.w = quick_gelu_bf(b.w);
        }
        *reinterpret_cast<uint4*>(Cb + (long)r * ldc + cp * 8) =
            uint4{pack2_epi<F16>(a.x, a.y, epi), pack2_epi<F16>(a.z, a.w, epi), pack2_epi<F16>(b.x, b.y, epi), pack2_epi<F16>(b.z, b.w, epi)};
      }
    }
    if (h == 0) BIGM_SYNC();                                  // the first half tile has been read: the second one may overwrite it
  }
}

template <bool F16, int TN, int D>
static int launch_frame4(const bf16_t* A, int lda, long strideA, const bf16_t* W, int ldw, long strideW, const float* bias, void* C, int ldc,
                         long strideC, int M, int N, int K, int batch, int epi, const float* gate, const int* ctl, hipStream_t st) {
  constexpr int BN = 32 * TN;
  const bool to_bf16 = epi == P8_EPI_BF16 || epi == P8_EPI_QGELU_BF16 || epi == P8_EPI_BF16OUT;
  const bool to_f32 = epi == P8_EPI_F32;
  if ((N % BN) || (K & 63) || M <= 0 || (M % 257) || batch <= 0 || !(to_bf16 || to_f32)) return DEER_ERR_SHAPE;
  // MUBUF byte offsets are 32 bits: every operand (one batch slice) has to end below 4 GiB
  if ((long)M * lda * 2 >= (1L << 32) || (long)N * ldw * 2 >= (1L << 32)) return DEER_ERR_SHAPE;
  constexpr int ring_bytes = D * (17 + BN / 16) * 1024, c_bytes = 129 * (BN * 4 + 16);     // the ring, then an f32 half tile
  constexpr int smem_bytes = ring_bytes > c_bytes ? ring_bytes : c_bytes;
  static std::atomic<bool> attr_set{false};
  auto kern = &gemm_frame4_kernel<TN, D, F16>;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes) != hipSuccess)
      return DEER_ERR_LAUNCH;
    attr_set = true;
  }
  const int tiles = (M / 257) * (N / BN);
  hipLaunchKernelGGL(kern, dim3(tiles, 1, batch), dim3(256), smem_bytes, st, A, lda, strideA, W, ldw, strideW, bias, C, ldc, strideC, M, N, K, epi,
                     ctl);
  DEER_LAUNCH_CHECK();
  return DEER_OK;
}

template <bool F16, int BM, int BN, int D, bool STAG = false>
static int launch_ring32(const bf16_t* A, int lda, long strideA, const bf16_t* W, int ldw, long strideW, const float* bias, void* C,
                         int ldc, long strideC, int M, int N, int K, int batch, int epi, const float* gate, const int* ctl,
                         hipStream_t st) {
  if ((N & 15) || (K & 31) || M <= 0 || batch <= 0) return DEER_ERR_SHAPE;
  constexpr int smem_bytes = D * (BM + BN) * 64;
  static std::atomic<bool> attr_set{false};
  auto kern = &gemm_ring32_kernel<BM, BN, D, STAG, F16>;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem_bytes) != hipSuccess)
      return DEER_ERR_LAUNCH;
    attr_set = true;
  }
  const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN);
  hipLaunchKernelGGL(kern, dim3(tiles, 1, batch), dim3(1024), smem_bytes, st, A, lda, strideA, W, ldw, strideW, bias, C, ldc, strideC, M, N,
                     K, epi, gate, ctl);
  DEER_LAUNCH_CHECK();
  return DEER_OK;
}

// variant: 0 = 256x256 / 5 stages (160 KB), 1 = 256x256 / 4 stages, 2 = 128x128 / 8 stages (128 KB), 3 = 256x256 / 2 stages,
//          4 = 256x256 / 3 stages, 5 = 128x128 / 5 stages (80 KB: two workgroups per CU), 6 / 7 = 257(272)x256 / 3 / 4 stages
template <bool F16>
int deer_launch_gemm_ring32(int variant, const bf16_t* A, int lda, long strideA, const bf16_t* W, int ldw, long strideW, const float* bias,
                            void* C, int ldc, long strideC, int M, int N, int K, int batch, int epi, const float* gate, const int* ctl,
                            hipStream_t st) {
#define P8_ARGS A, lda, strideA, W, ldw, strideW, bias, C, ldc, strideC, M, N, K, batch, epi, gate, ctl, st
  switch (variant) {
    case 0: return launch_ring32<F16, 256, 256, 5>(P8_ARGS);
    case 1: return launch_ring32<F16, 256, 256, 4>(P8_ARGS);
    case 2: return launch_ring32<F16, 128, 128, 8>(P8_ARGS);
    case 3: return launch_ring32<F16, 256, 256, 2>(P8_ARGS);
    case 4: return launch_ring32<F16, 256, 256, 3>(P8_ARGS);
    case 5: return launch_ring32<F16, 128, 128, 5>(P8_ARGS);
    case 8: return launch_ring32<F16, 256, 256, 5, true>(P8_ARGS);   // staggered wave groups, two barriers per K-step
    case 9: return launch_ring32<F16, 128, 128, 5, true>(P8_ARGS);
    case 10: return launch_ring32<F16, 256, 256, 4, true>(P8_ARGS);
    case 11: return launch_ring32<F16, 128, 128, 8, true>(P8_ARGS);
    case 6: return launch_ring272<F16, 3>(P8_ARGS);              // one camera frame (257 rows) per row tile, 99 KB ring
    case 7: return launch_ring272<F16, 4>(P8_ARGS);              // the same, 132 KB ring
    case 12: return launch_frame<F16, 4, 4, 4, 4>(P8_ARGS);         // frame tiles, balanced 17th row tile: 257 x 256, 16 waves, 132 KB
    case 13: return launch_frame<F16, 4, 4, 3, 4>(P8_ARGS);         // 257 x 192, 16 waves, 116 KB
    case 14: return launch_frame<F16, 4, 4, 3, 5>(P8_ARGS);         // 257 x 192, 145 KB ring
    case 15: return launch_frame<F16, 4, 2, 4, 3>(P8_ARGS);         // 257 x 128, 8 waves, 75 KB: two workgroups per CU
    case 16: return launch_frame<F16, 4, 4, 2, 4>(P8_ARGS);         // 257 x 128, 16 waves, 100 KB
    case 17: return launch_frame<F16, 4, 4, 1, 4>(P8_ARGS);         // 257 x 64, 16 waves, 84 KB
    case 18: return launch_frame<F16, 4, 2, 2, 3>(P8_ARGS);         // 257 x 64, 8 waves, 63 KB: two workgroups per CU
    case 19: return launch_frame<F16, 4, 4, 4, 3>(P8_ARGS);         // 257 x 256, 99 KB
    case 20: return launch_frame<F16, 4, 2, 4, 4>(P8_ARGS);         // 257 x 128, 8 waves, 100 KB
    // HALF frames (measured, not auto-selected: out_proj at 16 frames 18.0 us against 17.3 for the 128x128 ring; c_proj without the K split
    // 50-55 us against 42.8 for the 257 x 128 tiles of the two K halves):
    case 21: return launch_frame<F16, 2, 4, 2, 5>(P8_ARGS);      // 129 | 128 x 128, 16 waves (32 x 32 wave tiles), 85 KB
    case 22: return launch_frame<F16, 2, 2, 4, 4>(P8_ARGS);      // 129 | 128 x 128, 8 waves (32 x 64 wave tiles), 68 KB
    case 23: return launch_frame8<F16, 8, 4>(P8_ARGS);           // frame8: 257 x 256, 8 waves (64 x 128 wave tiles), 132 KB
    case 24: return launch_frame8<F16, 6, 4>(P8_ARGS);           // frame8: 257 x 192, 8 waves (64 x 96 wave tiles), 116 KB
    case 25: return launch_frame4<F16, 8, 4>(P8_ARGS);           // frame4: 257 x 256, 4 waves (128 x 128 wave tiles), 132 KB ring / 144 KB C tile
    case 26: return launch_frame4<F16, 6, 4>(P8_ARGS);           // frame4: 257 x 192, 4 waves (128 x 96 wave tiles), 116 KB
    case 27: return launch_frame4<F16, 4, 4>(P8_ARGS);           // frame4: 257 x 128, 4 waves (128 x 64 wave tiles), 100 KB ring / 144 KB f32 C tile
    case 28: return launch_frame4<F16, 4, 6>(P8_ARGS);           // the same on a 150 KB ring
    default: return DEER_ERR_SHAPE;
  }
#undef P8_ARGS
}

template int deer_launch_gemm_ring32<false>(int, const bf16_t*, int, long, const bf16_t*, int, long, const float*, void*, int, long, int, int, int, int, int,
                                            const float*, const int*, hipStream_t);
template int deer_launch_gemm_ring32<true>(int, const bf16_t*, int, long, const bf16_t*, int, long, const float*, void*, int, long, int, int, int, int, int,
                                           const float*, const int*, hipStream_t);
