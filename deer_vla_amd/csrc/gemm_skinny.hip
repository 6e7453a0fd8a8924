// Skinny (M <= 64 rows) bf16 MFMA GEMM for the weight-bandwidth-bound part of the step: the MPT blocks
// and gated x-attn layers at T ~ 14 text tokens (174 MB of weights per layer for 2.7 GFLOP).
//
//   part[ks][m][n] = sum_{k in K-slice ks} A[m,k] * W[n,k]          (f32 partial slabs, split-K)
//
// gfx950 design notes
//  * HBM-bound: the only thing that matters is streaming W once at full bandwidth.  W is PRE-PACKED at
//    load time into MFMA-fragment order  Wp[N/16][K/32][64 lanes][8 bf16]  (deer_pack_weight_mfma16) so
//    every wave-level load is one fully coalesced, contiguous 1 KiB read (no 64-byte row striding),
//    issued non-temporal (each byte is used exactly once per step).
//  * One wave = one 16-column tile over the block's K-slice; 8 fragment loads are kept in flight per
//    wave (deep unroll, late wait) and the weights go straight to VGPRs - an LDS round trip would be pure
//    overhead for an operand that is not shared between waves.
//  * The tiny activation slice A[<=32][KS] is staged once per block in LDS (bf16, padded rows) and read
//    back as MFMA fragments with ds_read_b128; operands are swapped (W fragment = MFMA "A") so each lane
//    ends up with 4 consecutive output columns -> 16-byte stores.
//  * Split-K partials are written as slabs and summed by the CONSUMER kernel's prologue (residual+LN,
//    attention, or the next GEMM's A-staging which also applies the exact GELU).  No atomics, no
//    in-kernel cross-workgroup hand-off: deterministic, and a kernel boundary (~1.5 us) is cheaper than
//    a grid barrier on this chip (MI355X_MICROARCH price list).
#include "common.h"
#include <algorithm>
#include <cstdlib>

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

enum { A_BF16 = 0, A_SLABS_GELU = 1, A_SLABS = 2, A_F32 = 3 };

// SPLIT: the f32 activation is represented as bf16 hi + bf16 lo (hi = bf16(a), lo = bf16(a - hi)) and each weight
// fragment feeds two MFMAs.  The GEMM is HBM-bound (MFMA pipe <5% busy at 16 rows), so the second MFMA is free and the
// activation side of the product keeps ~16 mantissa bits: the LLM trunk then differs from an fp32 reference only
// through the bf16 weights it shares with it -> robust exit decisions (SURVEY §7 "exit-index exactness").
//
// Work decomposition: workgroup (x, y) owns NW 16-column tiles (one per wave) over the K RANGE [y*KR, (y+1)*KR) and walks it
// in CHUNKS of KC = 32*WU columns: stage the activation chunk [MPAD rows x KC] in LDS (hi + lo), barrier, WU weight fragments
// x MT row tiles of MFMAs, barrier.  The accumulators live across chunks, so the split-K factor (= number of f32 slabs the
// consumer has to sum) is decoupled from the LDS slice: with 112 rows (8 environments) a 128-column chunk fills the LDS,
// but the K range per workgroup stays 512-2048 columns (<= 16 slabs instead of 64).
// Weight fragments are double-buffered in registers (wA / wB): the loads of chunk c+1 are issued BEFORE the MFMAs of chunk
// c and fly during its barriers and the staging of chunk c+1; the first chunk's loads are issued before anything else, so the
// activation staging (L2 latency) hides under the first HBM round trip instead of preceding it.
template <int MT, bool SPLIT, int NW, int WU, bool F16 = false>
__global__ __launch_bounds__(64 * NW) void gemm_skinny_kernel(const void* __restrict__ Av, int lda,
                                                          const float* __restrict__ Aslab, int s_in, long slab_stride_in,
                                                          int a_mode, const bf16_t* __restrict__ Wp,
                                                          float* __restrict__ part, int M, int N, int K, int KR,
                                                          const int* ctl, int dbg) {
  DEER_RETURN_IF_EXITED(ctl);
  constexpr int MPAD = MT * 16;
  constexpr int KC = 32 * WU;
  constexpr int pitch = KC + 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* As = reinterpret_cast<bf16_t*>(smem_raw);      // hi: [MPAD][KC + 8]
  bf16_t* Al = As + MPAD * pitch;                         // lo: [MPAD][KC + 8] (SPLIT only)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int ks = blockIdx.y, k_begin = ks * KR;
  const int k_end = min(K, k_begin + KR);                 // multiple of 32
  const int nchunks = (k_end - k_begin + KC - 1) / KC;

  const int ktiles = K >> 5;
  // waves beyond N (last workgroup of a ragged N) stream tile 0 and never store: keeps barriers and loads uniform
  const int tile_raw = blockIdx.x * NW + wave;
  const bool tile_ok = tile_raw * 16 < N;
  const int tile = tile_ok ? tile_raw : 0;
  const u32x4* wp = reinterpret_cast<const u32x4*>(Wp) + (long)tile * ktiles * 64 + lane;
  const int kt_last = (k_end >> 5) - 1;

  u32x4 wA[WU], wB[WU];
  auto issue = [&](u32x4 (&w)[WU], int chunk) {           // fragments of chunk (clamped: loads stay branch-free)
    const int kt0 = (k_begin >> 5) + chunk * WU;
#pragma unroll
    for (int u = 0; u < WU; ++u) w[u] = __builtin_nontemporal_load(wp + (long)min(kt0 + u, kt_last) * 64);
  };
  auto stage = [&](int chunk) {                            // activation chunk -> LDS (rows >= M and columns >= k_end are zero)
    const int k0 = k_begin + chunk * KC;
    const int klen = min(KC, k_end - k0);
    if (a_mode == A_BF16) {
      const bf16_t* A = reinterpret_cast<const bf16_t*>(Av);
      constexpr int segs = KC >> 3;
      for (int idx = tid; idx < MPAD * segs; idx += 64 * NW) {
        const int row = idx / segs, seg = idx - row * segs;
        uint4 v = uint4{0, 0, 0, 0};
        if (row < M && seg * 8 < klen) v = *reinterpret_cast<const uint4*>(A + (long)row * lda + k0 + seg * 8);
        *reinterpret_cast<uint4*>(As + row * pitch + seg * 8) = v;
        if (SPLIT) *reinterpret_cast<uint4*>(Al + row * pitch + seg * 8) = uint4{0, 0, 0, 0};
      }
    } else {
      constexpr int segs = KC >> 2;
      for (int idx = tid; idx < MPAD * segs; idx += 64 * NW) {
        const int row = idx / segs, seg = idx - row * segs;
        float4 s = float4{0.f, 0.f, 0.f, 0.f};
        if (row < M && seg * 4 < klen) {
          if (a_mode == A_F32) {
            s = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(Av) + (long)row * lda + k0 + seg * 4);
          } else {
            s = slab_sum4(Aslab + (long)row * K + k0 + seg * 4, s_in, slab_stride_in);
            if (a_mode == A_SLABS_GELU) { s.x = gelu_erf(s.x); s.y = gelu_erf(s.y); s.z = gelu_erf(s.z); s.w = gelu_erf(s.w); }
          }
        }
        uint32_t h01, h23, l01, l23;
        split2<F16>(s.x, s.y, h01, l01);
        split2<F16>(s.z, s.w, h23, l23);
        *reinterpret_cast<uint2*>(As + row * pitch + seg * 4) = uint2{h01, h23};
        if (SPLIT) *reinterpret_cast<uint2*>(Al + row * pitch + seg * 4) = uint2{l01, l23};
      }
    }
  };

  f32x4 acc[MT];
#pragma unroll
  for (int j = 0; j < MT; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bf16_t* as = As + c * pitch + g * 8;
  const bf16_t* al = Al + c * pitch + g * 8;
  auto mma = [&](const u32x4 (&w)[WU], int chunk) {
    const int nkt = min(WU, ((k_end - k_begin) >> 5) - chunk * WU);      // fragments of this chunk that are inside the K range
#pragma unroll
    for (int u = 0; u < WU; ++u) {
      if (u < nkt) {
        const bf16x8 wf = __builtin_bit_cast(bf16x8, w[u]);
#pragma unroll
        for (int j = 0; j < MT; ++j) {
          const bf16x8 af = *reinterpret_cast<const bf16x8*>(as + j * 16 * pitch + u * 32);
          acc[j] = mfma16<F16>(wf, af, acc[j]);
          if (SPLIT) {
            const bf16x8 lf = *reinterpret_cast<const bf16x8*>(al + j * 16 * pitch + u * 32);
            acc[j] = mfma16<F16>(wf, lf, acc[j]);
          }
        }
      }
    }
  };

  issue(wA, 0);                                            // weights first: they do not depend on the activation
  for (int ch = 0; ch < nchunks; ch += 2) {
    if (!(dbg & 1)) stage(ch);
    __syncthreads();
    if (ch + 1 < nchunks) issue(wB, ch + 1);
    __builtin_amdgcn_sched_barrier(0);                     // keep the next chunk's loads ahead of this chunk's MFMAs
    if (!(dbg & 2)) mma(wA, ch); else { for (int u = 0; u < WU; ++u) acc[0][0] += __uint_as_float(wA[u][0]); }
    __syncthreads();
    if (ch + 1 >= nchunks) break;
    if (!(dbg & 1)) stage(ch + 1);
    __syncthreads();
    if (ch + 2 < nchunks) issue(wA, ch + 2);
    __builtin_amdgcn_sched_barrier(0);
    if (!(dbg & 2)) mma(wB, ch + 1); else { for (int u = 0; u < WU; ++u) acc[0][0] += __uint_as_float(wB[u][0]); }
    __syncthreads();
  }
  if (!tile_ok) return;
  // lane holds part[m = j*16 + c][n = tile*16 + g*4 .. +3]
  float* dst = part + ((long)ks * MPAD) * N + tile * 16 + g * 4;
#pragma unroll
  for (int j = 0; j < MT; ++j)
    *reinterpret_cast<float4*>(dst + (long)(j * 16 + c) * N) = float4{acc[j][0], acc[j][1], acc[j][2], acc[j][3]};
}

// ------------------------------------------------------------------------------------------------------------------------
// Env-batch variant (49..128 rows: 4-8 environments x 14 tokens): at these row counts the GEMM is no longer a pure weight
// stream - per 1 KiB weight fragment a wave issues 2*MT MFMAs and reads 2*MT activation fragments, so MFMA time (3.0 us for
// 8192x2048), LDS-read time (3.0 us) and HBM time (5.6 us) are the same order, and the kernel above loses to its own
// serialisation (measured r03: 5 us per 128-column chunk = load -> cvt -> ds_write -> barrier -> ds_read -> MFMA with nothing
// overlapped; 21.9 us for 33.5 MB) and to its slabs (S = 8 f32 slabs of 128 x 8192 = the weight bytes again, written in a burst
// at the end, then re-read - summed, GELU'ed and split - by EVERY column group of the consumer).
// Here
//  * the activation arrives PRE-SPLIT as two bf16 planes hi = bf16(a), lo = bf16(a - hi) (written by the producing kernel:
//    deer_resadd_ln_split / deer_mpt_attn_small_hl / deer_slab_gelu_split), so staging is a pure copy and runs on the LDS-DMA
//    path (global_load_lds_dwordx4) into a ring of D stages of 64 K-columns, exactly like csrc/gemm_tiled.hip: counted
//    s_waitcnt vmcnt + ONE raw s_barrier per stage, XOR-swizzled 128-byte rows (conflict-free ds_read_b128 fragment reads);
//  * the packed weight fragments ride the same ring (one 1 KiB DMA per fragment: all loads of a wave are LDS-DMA, so the
//    in-order vmcnt accounting stays exact; the fragment is read back lane-linear, conflict-free);
//  * a workgroup is 8 waves = 128 columns and walks a K range of K/S columns: S = 256 * 128 / N slabs instead of 256 * 256 / N
//    (4 instead of 8 for the 8192-wide up-projections, 16 instead of 32 for the down-projections).
// Algorithmic bytes per launch = 2*N*K (weights once); L2->LDS activation traffic = (N/128) * 128 rows * K * 4 B.
// Measured r03 at 112 rows, 8192x2048 (profiles/r03_b_skinny_hl_*.txt; ablations by temporary kernel flags, since removed): 17.7 us
// against 22.6; epilogue (S = 4 slabs, 14.7 MB of stores) 3.5 us; the K loop 1.5 us per 48 KB stage whatever the ring depth
// (2..5 stages: rate-bound at ~35 GB/s of LDS-DMA fill per CU, not latency-bound), with either operand's DMAs removed still
// 1.1 us per stage, with the MFMAs / fragment reads removed 1.2 us; K-blocked activation planes or kt-major packed weights
// (channel spreading) change nothing.  S = 2 (fewer slabs, 16 stages) and S = 8 (more workgroups than CUs) both lose.
static inline int skinny_mt(int M);
typedef __attribute__((address_space(1))) const void sk_gptr_t;
typedef __attribute__((address_space(3))) void sk_lptr_t;

template <int N_>
__device__ __forceinline__ void sk_wait_vmcnt() {
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_) : "memory");
}

template <int MT, int D, bool F16 = false>
__device__ __forceinline__ void gemm_skinny_hl_body(const bf16_t* __restrict__ Ahi, const bf16_t* __restrict__ Alo, int lda,
                                                    const bf16_t* __restrict__ Wp, float* __restrict__ part, int M, int N, int K, int KR,
                                                    int part_rows, int bx, int by) {
  constexpr int MTL = (MT + 1) & ~1;                  // row tiles LOADED per plane (even: every wave issues the same number of DMAs)
  constexpr int A_CH = 4 * MTL;                       // 1 KiB chunks (8 rows x 128 B) of [hi plane ; lo plane] per stage
  constexpr int CH = A_CH + 16;                       // + 8 column tiles x 2 K-tiles of packed weight fragments
  constexpr int CPW = CH / 8;                         // chunks per wave per stage
  static_assert(CH % 8 == 0, "static vmcnt");
  constexpr int A_BYTES = A_CH * 1024;
  constexpr int STAGE = CH * 1024;
  static_assert((D - 2) * CPW <= 63, "vmcnt field");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int ks = by, k_begin = ks * KR;
  const int nk = KR >> 6;                             // stages of 64 K-columns
  const int ktiles = K >> 5;
  const int tile_raw = bx * 8 + wave;
  const bool tile_ok = tile_raw * 16 < N;

  // per-wave DMA sources: chunk q = wave + i*8
  const bf16_t* src[CPW];
  int kstep[CPW];                                     // elements to advance per stage (64 for A rows, 2 fragments for W)
  const int lr = lane >> 3, ls = ((lane & 7) ^ lr) * 8;
#pragma unroll
  for (int i = 0; i < CPW; ++i) {
    const int q = wave + i * 8;                       // wave-uniform
    if (q < A_CH) {
      const int plane = q / (2 * MTL), r8 = q - plane * 2 * MTL;
      const int row = min(r8 * 8 + lr, M - 1);        // rows >= M repeat the last row (their outputs are zeroed in the epilogue)
      src[i] = (plane ? Alo : Ahi) + (long)row * lda + k_begin + ls;
      kstep[i] = 64;
    } else {
      const int w = q - A_CH, ct = w >> 1, kt = w & 1;
      const int t16 = bx * 8 + ct;
      const int tile = (t16 * 16 < N) ? t16 : 0;      // ragged N: stream tile 0, never stored
      src[i] = Wp + (((long)tile * ktiles + (k_begin >> 5) + kt) * 64 + lane) * 8;
      kstep[i] = 2 * 64 * 8;
    }
  }
  auto issue = [&](int t) {
    const int tt = min(t, nk - 1);
    unsigned char* st = smem_raw + (t % D) * STAGE;
#pragma unroll
    for (int i = 0; i < CPW; ++i)
      __builtin_amdgcn_global_load_lds((sk_gptr_t*)(src[i] + (long)tt * kstep[i]), (sk_lptr_t*)(st + (wave + i * 8) * 1024), 16, 0, 0);
  };

  f32x4 acc[MT];
#pragma unroll
  for (int j = 0; j < MT; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int a_row = c * 128;
  const int sw0 = ((0 * 4 + g) ^ (c & 7)) << 4, sw1 = ((1 * 4 + g) ^ (c & 7)) << 4;
  const int w_off = A_BYTES + wave * 2048 + lane * 16;

#pragma unroll
  for (int t = 0; t < D - 1; ++t) issue(t);
  for (int kt = 0; kt < nk; ++kt) {
    sk_wait_vmcnt<(D - 2) * CPW>();                   // this wave's part of stage kt has landed
    __builtin_amdgcn_s_barrier();                     // ... everybody's; and everybody finished reading stage kt-1 (refilled now)
    issue(kt + D - 1);
    const unsigned char* st = smem_raw + (kt % D) * STAGE;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      const int sw = kk ? sw1 : sw0;
      const bf16x8 wf = *reinterpret_cast<const bf16x8*>(st + w_off + kk * 1024);
#pragma unroll
      for (int j = 0; j < MT; ++j) {
        const bf16x8 ah = *reinterpret_cast<const bf16x8*>(st + a_row + j * 2048 + sw);
        const bf16x8 al = *reinterpret_cast<const bf16x8*>(st + a_row + (MTL * 16 + j * 16) * 128 + sw);
        acc[j] = mfma16<F16>(wf, ah, acc[j]);
        acc[j] = mfma16<F16>(wf, al, acc[j]);
      }
    }
  }
  sk_wait_vmcnt<0>();                                 // clamped tail loads must not land in LDS after the workgroup is gone
  if (!tile_ok) return;
  float* dst = part + ((long)ks * part_rows) * N + tile_raw * 16 + g * 4;      // slab ks: rows part_rows * ks .. (this launch's rows first)
#pragma unroll
  for (int j = 0; j < MT; ++j) {
    const int row = j * 16 + c;
    const float4 v = row < M ? float4{acc[j][0], acc[j][1], acc[j][2], acc[j][3]} : float4{0.f, 0.f, 0.f, 0.f};
    *reinterpret_cast<float4*>(dst + (long)row * N) = v;
  }
}

template <int MT, int D, bool F16 = false>
__global__ __launch_bounds__(512) void gemm_skinny_hl_kernel(const bf16_t* __restrict__ Ahi, const bf16_t* __restrict__ Alo, int lda,
                                                            const bf16_t* __restrict__ Wp, float* __restrict__ part, int M, int N,
                                                            int K, int KR, const int* ctl, int part_rows) {
  DEER_RETURN_IF_EXITED(ctl);
  gemm_skinny_hl_body<MT, D, F16>(Ahi, Alo, lda, Wp, part, M, N, K, KR, part_rows, blockIdx.x, blockIdx.y);
}

#ifndef DEER_BODIES_ONLY
// Env batch with COMPACTION of exited environments (common.h: CMAP_*): the number of valid rows is read from the device - the active
// slots x rows_per_env, minus the rows of earlier row blocks - and the workgroup runs the body specialised for that many MFMA row tiles
// (its own ring depth, DMA counts and waits: exactly the kernel a launch with that M would have run).  A launch costs what its ACTIVE rows
// cost: 15.4 us at 112 rows, 6.9 us at 14 (r03).  Rows beyond the active ones are not written.
template <bool F16>
__global__ __launch_bounds__(512) void gemm_skinny_hl_dyn_kernel(const bf16_t* __restrict__ Ahi, const bf16_t* __restrict__ Alo, int lda,
                                                                const bf16_t* __restrict__ Wp, float* __restrict__ part, int M_max, int N,
                                                                int K, int KR, const int* ctl, int part_rows, const int* __restrict__ cmap,
                                                                int rows_per_env, int row0) {
  DEER_RETURN_IF_EXITED(ctl);
  const int M = min(M_max, cmap[CMAP_N] * rows_per_env - row0);
  if (M <= 0) return;
  switch ((M + 15) >> 4) {
    case 1: gemm_skinny_hl_body<1, 4, F16>(Ahi, Alo, lda, Wp, part, M, N, K, KR, part_rows, blockIdx.x, blockIdx.y); break;
    case 2: gemm_skinny_hl_body<2, 4, F16>(Ahi, Alo, lda, Wp, part, M, N, K, KR, part_rows, blockIdx.x, blockIdx.y); break;
    case 3: gemm_skinny_hl_body<3, 4, F16>(Ahi, Alo, lda, Wp, part, M, N, K, KR, part_rows, blockIdx.x, blockIdx.y); break;
    case 4: gemm_skinny_hl_body<4, 4, F16>(Ahi, Alo, lda, Wp, part, M, N, K, KR, part_rows, blockIdx.x, blockIdx.y); break;
    case 5: gemm_skinny_hl_body<5, 3, F16>(Ahi, Alo, lda, Wp, part, M, N, K, KR, part_rows, blockIdx.x, blockIdx.y); break;
    case 6: gemm_skinny_hl_body<6, 3, F16>(Ahi, Alo, lda, Wp, part, M, N, K, KR, part_rows, blockIdx.x, blockIdx.y); break;
    case 7: gemm_skinny_hl_body<7, 3, F16>(Ahi, Alo, lda, Wp, part, M, N, K, KR, part_rows, blockIdx.x, blockIdx.y); break;
    default: gemm_skinny_hl_body<8, 3, F16>(Ahi, Alo, lda, Wp, part, M, N, K, KR, part_rows, blockIdx.x, blockIdx.y); break;
  }
}

// split-K of the env-batch kernel: 128-column workgroups, ~256 of them, K range per workgroup >= 256 columns (4 ring stages)
extern "C" int deer_skinny_hl_splitk(int M, int N, int K) {
  (void)M;
  const int groups = (N + 127) / 128;
  int s = 1;
  while (groups * s * 2 <= 288 && (K / (s * 2)) >= 256 && (K % (s * 2 * 64)) == 0) s *= 2;
  return s;
}

template <bool F16>
static int launch_skinny_hl(const void* Ahi, const void* Alo, int lda, const void* Wp, float* part, int M, int N, int K, int splitk, const int* ctl,
                            void* stream, int part_rows);

// every entry point of the trunk exists twice: on bf16 operands (weights packed bf16, activation planes bf16 hi / lo) and - suffix _f16,
// round 6 - on IEEE fp16 operands (weights fp16 = what fp16 autocast feeds a Linear, planes fp16 hi / lo): same arguments, same kernels
// instantiated on v_mfma_f32_16x16x32_f16
extern "C" int deer_gemm_skinny_hl(const void* Ahi, const void* Alo, int lda, const void* Wp, float* part, int M, int N, int K,
                                   int splitk, const int* ctl, void* stream) {
  return launch_skinny_hl<false>(Ahi, Alo, lda, Wp, part, M, N, K, splitk, ctl, stream, 16 * ((M + 15) / 16));
}
extern "C" int deer_gemm_skinny_hl_f16(const void* Ahi, const void* Alo, int lda, const void* Wp, float* part, int M, int N, int K,
                                       int splitk, const int* ctl, void* stream) {
  return launch_skinny_hl<true>(Ahi, Alo, lda, Wp, part, M, N, K, splitk, ctl, stream, 16 * ((M + 15) / 16));
}

// More than 128 rows (8 environments x 32-token instructions: data.py:905-919 pads to the longest of the batch, max_length = 32; or
// more environments per engine): row blocks of <= 128 rows, one launch each, into slabs of `slab_rows` rows (>= 16 * ceil(M / 16)) -
// part[ks][slab_rows][N].  The weights of the later blocks come from the Infinity Cache (33 MB per projection).
template <bool F16>
static int gemm_skinny_hl_rows(const void* Ahi, const void* Alo, int lda, const void* Wp, float* part, int M, int N, int K, int splitk,
                               int slab_rows, const int* ctl, void* stream) {
  if (M <= 0 || M > 512 || slab_rows < 16 * ((M + 15) / 16)) return DEER_ERR_SHAPE;
  for (int r0 = 0; r0 < M; r0 += 128) {
    const int mb = std::min(128, M - r0);
    const int rc = launch_skinny_hl<F16>(reinterpret_cast<const bf16_t*>(Ahi) + (long)r0 * lda, reinterpret_cast<const bf16_t*>(Alo) + (long)r0 * lda, lda, Wp,
                                         part + (long)r0 * N, mb, N, K, splitk, ctl, stream, slab_rows);
    if (rc != DEER_OK) return rc;
  }
  return DEER_OK;
}
extern "C" int deer_gemm_skinny_hl_rows(const void* Ahi, const void* Alo, int lda, const void* Wp, float* part, int M, int N, int K, int splitk,
                                        int slab_rows, const int* ctl, void* stream) {
  return gemm_skinny_hl_rows<false>(Ahi, Alo, lda, Wp, part, M, N, K, splitk, slab_rows, ctl, stream);
}
extern "C" int deer_gemm_skinny_hl_rows_f16(const void* Ahi, const void* Alo, int lda, const void* Wp, float* part, int M, int N, int K, int splitk,
                                            int slab_rows, const int* ctl, void* stream) {
  return gemm_skinny_hl_rows<true>(Ahi, Alo, lda, Wp, part, M, N, K, splitk, slab_rows, ctl, stream);
}

// deer_gemm_skinny_hl_rows for an env batch with compaction: M = the rows of ALL environments (the launch geometry), the valid rows are read
// from `cmap` on the device (active slots x rows_per_env)
template <bool F16>
static int gemm_skinny_hl_active(const void* Ahi, const void* Alo, int lda, const void* Wp, float* part, int M, int N, int K, int splitk,
                                          int slab_rows, const int* ctl, const int* cmap, int rows_per_env, void* stream) {
  if (M <= 0 || M > 512 || slab_rows < 16 * ((M + 15) / 16) || cmap == nullptr || rows_per_env <= 0 || N <= 0 || (N & 15) || K <= 0 || (K & 63) ||
      splitk <= 0 || (K % (splitk * 64)) != 0 || (lda & 7) || Ahi == nullptr || Alo == nullptr || Wp == nullptr || part == nullptr)
    return DEER_ERR_SHAPE;
  constexpr int smem = 3 * (4 * 8 + 16) * 1024;               // the largest body (8 row tiles, 3 stages); the 4-stage bodies need 4 * (4 * 4 + 16) KB
  static_assert(smem >= 4 * (4 * 4 + 16) * 1024 && smem <= 160 * 1024, "LDS");
  static std::atomic<bool> attr_set{false};
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_skinny_hl_dyn_kernel<F16>), hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
      return DEER_ERR_LAUNCH;
    attr_set = true;
  }
  const int KR = K / splitk;
  dim3 grid((N + 127) / 128, splitk);
  for (int r0 = 0; r0 < M; r0 += 128) {
    const int mb = std::min(128, M - r0);
    hipLaunchKernelGGL(gemm_skinny_hl_dyn_kernel<F16>, grid, dim3(512), smem, reinterpret_cast<hipStream_t>(stream),
                       reinterpret_cast<const bf16_t*>(Ahi) + (long)r0 * lda, reinterpret_cast<const bf16_t*>(Alo) + (long)r0 * lda, lda,
                       reinterpret_cast<const bf16_t*>(Wp), part + (long)r0 * N, mb, N, K, KR, ctl, slab_rows, cmap, rows_per_env, r0);
  }
  DEER_LAUNCH_CHECK();
  return DEER_OK;
}

extern "C" int deer_gemm_skinny_hl_active(const void* Ahi, const void* Alo, int lda, const void* Wp, float* part, int M, int N, int K, int splitk,
                                          int slab_rows, const int* ctl, const int* cmap, int rows_per_env, void* stream) {
  return gemm_skinny_hl_active<false>(Ahi, Alo, lda, Wp, part, M, N, K, splitk, slab_rows, ctl, cmap, rows_per_env, stream);
}
extern "C" int deer_gemm_skinny_hl_active_f16(const void* Ahi, const void* Alo, int lda, const void* Wp, float* part, int M, int N, int K, int splitk,
                                              int slab_rows, const int* ctl, const int* cmap, int rows_per_env, void* stream) {
  return gemm_skinny_hl_active<true>(Ahi, Alo, lda, Wp, part, M, N, K, splitk, slab_rows, ctl, cmap, rows_per_env, stream);
}

template <bool F16>
static int launch_skinny_hl(const void* Ahi, const void* Alo, int lda, const void* Wp, float* part, int M, int N, int K, int splitk, const int* ctl,
                            void* stream, int part_rows) {
  if (M <= 0 || M > 128 || N <= 0 || (N & 15) || K <= 0 || (K & 63) || splitk <= 0 || (K % (splitk * 64)) != 0 || (lda & 7))
    return DEER_ERR_SHAPE;
  if (Ahi == nullptr || Alo == nullptr || Wp == nullptr || part == nullptr) return DEER_ERR_SHAPE;
  const int KR = K / splitk;
  const int mt = skinny_mt(M);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  dim3 grid((N + 127) / 128, splitk);
  const bf16_t* ah = reinterpret_cast<const bf16_t*>(Ahi);
  const bf16_t* al = reinterpret_cast<const bf16_t*>(Alo);
  const bf16_t* wp = reinterpret_cast<const bf16_t*>(Wp);
#define DEER_SKHL_CASE(MT_, D_)                                                                                              \
  case MT_: {                                                                                                                \
    constexpr int smem = D_ * (4 * ((MT_ + 1) & ~1) + 16) * 1024;                                                            \
    static_assert(smem <= 160 * 1024, "LDS");                                                                                \
    static std::atomic<bool> attr_set{false};                                                                                            \
    auto kern = &gemm_skinny_hl_kernel<MT_, D_, F16>;                                                                             \
    if (!attr_set) {                                                                                                         \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem) !=     \
          hipSuccess) return DEER_ERR_LAUNCH;                                                                                \
      attr_set = true;                                                                                                       \
    }                                                                                                                        \
    hipLaunchKernelGGL(kern, grid, dim3(512), smem, st, ah, al, lda, wp, part, M, N, K, KR, ctl, part_rows);                 \
  } break
  switch (mt) {
    DEER_SKHL_CASE(1, 4);
    DEER_SKHL_CASE(2, 4);
    DEER_SKHL_CASE(3, 4);
    DEER_SKHL_CASE(4, 4);
    DEER_SKHL_CASE(5, 3);
    DEER_SKHL_CASE(6, 3);
    DEER_SKHL_CASE(7, 3);
    DEER_SKHL_CASE(8, 3);
    default: return DEER_ERR_SHAPE;
  }
#undef DEER_SKHL_CASE
  DEER_LAUNCH_CHECK();
  return DEER_OK;
}

// ---- producer of the env-batch kernel's activation for the down-projections: GELU(sum of the up-projection's slabs) as
// bf16 hi / lo planes [rows][C].  One pass over the slabs (the kernel above re-did this sum + GELU + split in EVERY column group
// of the consumer: 8x for the 2048-wide down-projections).
template <bool F16>
__global__ __launch_bounds__(256) void slab_gelu_split_kernel(const float* __restrict__ slab, int s_in, long stride, int gelu,
                                                              bf16_t* __restrict__ hi, bf16_t* __restrict__ lo, long total4,
                                                              const int* ctl, const int* __restrict__ cmap = nullptr, long per_env4 = 0) {
  DEER_RETURN_IF_EXITED(ctl);
  if (cmap != nullptr) total4 = min(total4, (long)cmap[CMAP_N] * per_env4);      // active slots only (compaction)
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total4; i += (long)gridDim.x * 256) {
    float4 s = slab_sum4(slab + i * 4, s_in, stride);
    if (gelu) { s.x = gelu_erf(s.x); s.y = gelu_erf(s.y); s.z = gelu_erf(s.z); s.w = gelu_erf(s.w); }
    uint32_t h01, h23, l01, l23;
    split2<F16>(s.x, s.y, h01, l01);
    split2<F16>(s.z, s.w, h23, l23);
    *reinterpret_cast<uint2*>(hi + i * 4) = uint2{h01, h23};
    *reinterpret_cast<uint2*>(lo + i * 4) = uint2{l01, l23};
  }
}

template <bool F16>
static int slab_gelu_split(const float* slab, int s_in, long slab_stride, int gelu, void* out_hi, void* out_lo, int rows,
                                    int C, const int* ctl, void* stream) {
  if (slab == nullptr || s_in <= 0 || out_hi == nullptr || out_lo == nullptr || rows <= 0 || C <= 0 || (C & 3)) return DEER_ERR_SHAPE;
  const long total4 = (long)rows * C / 4;
  const int blocks = (int)std::min<long>((total4 + 255) / 256, 2048);
  hipLaunchKernelGGL(slab_gelu_split_kernel<F16>, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), slab, s_in, slab_stride,
                     gelu, reinterpret_cast<bf16_t*>(out_hi), reinterpret_cast<bf16_t*>(out_lo), total4, ctl);
  DEER_LAUNCH_CHECK();
  return DEER_OK;
}

// the same restricted to the active slots of an env batch with compaction (rows_per_env rows per slot)
template <bool F16>
static int slab_gelu_split_active(const float* slab, int s_in, long slab_stride, int gelu, void* out_hi, void* out_lo, int rows, int C,
                                           const int* ctl, const int* cmap, int rows_per_env, void* stream) {
  if (slab == nullptr || s_in <= 0 || out_hi == nullptr || out_lo == nullptr || rows <= 0 || C <= 0 || (C & 3) || cmap == nullptr || rows_per_env <= 0)
    return DEER_ERR_SHAPE;
  const long total4 = (long)rows * C / 4;
  const int blocks = (int)std::min<long>((total4 + 255) / 256, 2048);
  hipLaunchKernelGGL(slab_gelu_split_kernel<F16>, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), slab, s_in, slab_stride,
                     gelu, reinterpret_cast<bf16_t*>(out_hi), reinterpret_cast<bf16_t*>(out_lo), total4, ctl, cmap, (long)rows_per_env * C / 4);
  DEER_LAUNCH_CHECK();
  return DEER_OK;
}

extern "C" int deer_slab_gelu_split(const float* slab, int s_in, long slab_stride, int gelu, void* out_hi, void* out_lo, int rows, int C, const int* ctl,
                                    void* stream) {
  return slab_gelu_split<false>(slab, s_in, slab_stride, gelu, out_hi, out_lo, rows, C, ctl, stream);
}
extern "C" int deer_slab_gelu_split_f16(const float* slab, int s_in, long slab_stride, int gelu, void* out_hi, void* out_lo, int rows, int C, const int* ctl,
                                        void* stream) {
  return slab_gelu_split<true>(slab, s_in, slab_stride, gelu, out_hi, out_lo, rows, C, ctl, stream);
}
extern "C" int deer_slab_gelu_split_active(const float* slab, int s_in, long slab_stride, int gelu, void* out_hi, void* out_lo, int rows, int C,
                                           const int* ctl, const int* cmap, int rows_per_env, void* stream) {
  return slab_gelu_split_active<false>(slab, s_in, slab_stride, gelu, out_hi, out_lo, rows, C, ctl, cmap, rows_per_env, stream);
}
extern "C" int deer_slab_gelu_split_active_f16(const float* slab, int s_in, long slab_stride, int gelu, void* out_hi, void* out_lo, int rows, int C,
                                               const int* ctl, const int* cmap, int rows_per_env, void* stream) {
  return slab_gelu_split_active<true>(slab, s_in, slab_stride, gelu, out_hi, out_lo, rows, C, ctl, cmap, rows_per_env, stream);
}

// ---- weight packing: row-major W[N,K] bf16 -> Wp[N/16][K/32][64][8] (done once at load time) ----
__global__ void pack_weight_kernel(const bf16_t* __restrict__ W, bf16_t* __restrict__ Wp, int N, int K) {
  const long total = (long)(N >> 4) * (K >> 5) * 64;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int lane = (int)(idx & 63);
    const long tk = idx >> 6;
    const int kt = (int)(tk % (K >> 5));
    const int tile = (int)(tk / (K >> 5));
    const int n = tile * 16 + (lane & 15), k = kt * 32 + (lane >> 4) * 8;
    *reinterpret_cast<uint4*>(Wp + idx * 8) = *reinterpret_cast<const uint4*>(W + (long)n * K + k);
  }
}

extern "C" int deer_pack_weight_mfma16(const void* W, void* Wp, int N, int K, void* stream) {
  if (N <= 0 || K <= 0 || (N & 15) || (K & 31)) return DEER_ERR_SHAPE;
  hipLaunchKernelGGL(pack_weight_kernel, dim3(1024), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<const bf16_t*>(W), reinterpret_cast<bf16_t*>(Wp), N, K);
  DEER_LAUNCH_CHECK();
  return DEER_OK;
}

// Row-tile count / workgroup shape for M rows: up to 16 rows 4 waves (64 columns) per workgroup, beyond that 16 waves (256
// columns) so that the staged activation chunk is amortised over more weight bytes.
static inline int skinny_mt(int M) { return (M + 15) >> 4; }
static inline int skinny_wu(int mt) { return mt > 4 ? 4 : 8; }                 // chunk = 32*WU columns: LDS (hi + lo) <= 70 KB

// Suggested split-K (= number of f32 partial slabs the consumer sums).  Measured on MI355X (tools/bench_skinny.py, 14 / 56 / 112
// rows): 4-wave workgroups (<= 16 rows) want ~512 workgroups, 16-wave ones ~192-256, and a K range per workgroup of >= 128
// columns; beyond that more slabs cost more (slab stores + the consumer's reduction) than the extra parallelism returns.
extern "C" int deer_skinny_splitk(int M, int N, int K) {
  const int mt = skinny_mt(M);
  const int cols = (mt == 1) ? 64 : 256;
  const int groups = (N + cols - 1) / cols;
  static const int want_wide = [] { const char* e = getenv("DEER_SKINNY_WANT"); return e ? atoi(e) : 192; }();   // tuning knob
  const int want = (mt == 1) ? 512 : want_wide;
  int s = 1;
  while (groups * s < want && (K / (s * 2)) >= 128 && (K % (s * 2 * 32)) == 0) s *= 2;
  return s;
}

template <bool F16>
static int gemm_skinny(const void* A, int lda, const float* Aslab, int s_in, long slab_stride_in, int a_mode,
                                const void* Wp, float* part, int M, int N, int K, int splitk, const int* ctl,
                                void* stream) {
  if (M <= 0 || M > 128 || N <= 0 || (N & 15) || K <= 0 || (K & 31) || splitk <= 0 || (K % (splitk * 32)) != 0)
    return DEER_ERR_SHAPE;
  if (a_mode < 0 || a_mode > 3) return DEER_ERR_SHAPE;
  if (a_mode == A_BF16 && (A == nullptr || (lda & 7))) return DEER_ERR_SHAPE;
  if (a_mode == A_F32 && (A == nullptr || (lda & 3))) return DEER_ERR_SHAPE;
  if ((a_mode == A_SLABS || a_mode == A_SLABS_GELU) && (Aslab == nullptr || s_in <= 0)) return DEER_ERR_SHAPE;
  static const int dbg = [] { const char* e = getenv("DEER_SKINNY_DBG"); return e ? atoi(e) : 0; }();   // ablation only (tools/)
  const int KR = K / splitk;
  const int mt = skinny_mt(M);
  const bool split = (a_mode != A_BF16);
  const int wu = skinny_wu(mt);
  const int smem = (split ? 2 : 1) * mt * 16 * (32 * wu + 8) * (int)sizeof(bf16_t);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const int nw = (mt == 1) ? 4 : 16;
  dim3 grid((N + 16 * nw - 1) / (16 * nw), splitk);
  const bf16_t* wp = reinterpret_cast<const bf16_t*>(Wp);
#define DEER_SK_LAUNCH(MT_, SP_, NW_, WU_)                                                                                     \
  do {                                                                                                                         \
    static std::atomic<bool> attr_set{false};                                                                                              \
    auto kern = &gemm_skinny_kernel<MT_, SP_, NW_, WU_, F16>;                                                                       \
    if (!attr_set) {                                                                                                           \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 72 * 1024) !=  \
          hipSuccess) return DEER_ERR_LAUNCH;                                                                                  \
      attr_set = true;                                                                                                         \
    }                                                                                                                          \
    hipLaunchKernelGGL(kern, grid, dim3(64 * NW_), smem, st, A, lda, Aslab, s_in, slab_stride_in, a_mode, wp, part, M, N, K,  \
                       KR, ctl, dbg);                                                                                          \
  } while (0)
#define DEER_SK_CASE(MT_, NW_, WU_) \
  case MT_: if (split) DEER_SK_LAUNCH(MT_, true, NW_, WU_); else DEER_SK_LAUNCH(MT_, false, NW_, WU_); break
  switch (mt) {
    DEER_SK_CASE(1, 4, 8);
    DEER_SK_CASE(2, 16, 8);
    DEER_SK_CASE(3, 16, 8);
    DEER_SK_CASE(4, 16, 8);
    DEER_SK_CASE(5, 16, 4);
    DEER_SK_CASE(6, 16, 4);
    DEER_SK_CASE(7, 16, 4);
    DEER_SK_CASE(8, 16, 4);
    default: return DEER_ERR_SHAPE;
  }
#undef DEER_SK_CASE
#undef DEER_SK_LAUNCH
  DEER_LAUNCH_CHECK();
  return DEER_OK;
}
extern "C" int deer_gemm_skinny(const void* A, int lda, const float* Aslab, int s_in, long slab_stride_in, int a_mode, const void* Wp, float* part,
                                int M, int N, int K, int splitk, const int* ctl, void* stream) {
  return gemm_skinny<false>(A, lda, Aslab, s_in, slab_stride_in, a_mode, Wp, part, M, N, K, splitk, ctl, stream);
}
extern "C" int deer_gemm_skinny_f16(const void* A, int lda, const float* Aslab, int s_in, long slab_stride_in, int a_mode, const void* Wp, float* part,
                                    int M, int N, int K, int splitk, const int* ctl, void* stream) {
  return gemm_skinny<true>(A, lda, Aslab, s_in, slab_stride_in, a_mode, Wp, part, M, N, K, splitk, ctl, stream);
}
#endif  // DEER_BODIES_ONLY
