// The LLM trunk at ONE environment (<= 16 text rows, one MFMA row tile): the decoder block / gated x-attn layer of
// open_flamingo/src/flamingo_lm.py:46-83 + helpers.py:260-279 + SURVEY App. B.1 in NINE launches per layer instead of fifteen.
//
// Round 3 measured the one-environment trunk layer at 112 us for 174 MB of weights (19 % of the HBM peak): the six weight-streaming
// GEMMs ran at 4 TB/s, the other 70 us were nine latency-bound row kernels (slab reduce + residual + LayerNorm x 4, GELU split x 2,
// q/k LayerNorm, attention) and fifteen kernel boundaries.  At <= 16 rows those row operations are tiny (the whole residual
// stream is 16 x 2048 floats = 128 KB), so here they ride INSIDE the GEMMs:
//
//  * deer_trunk_ln_gemm    y = LN(x + tanh(gate) * (slab0 + slab1)) W^T for the wide projections (N = 3d / 4d, K = d).  EVERY
//    workgroup recomputes the residual update and the LayerNorm of all rows in registers (each wave owns a K range: the lanes hold
//    the rows in MFMA-fragment order, the row statistics meet through 1 KB of LDS) while its 16 KB of packed weights per wave are in
//    flight (LDS-DMA, non-temporal): the redundant read is <= 384 KB per workgroup from L2 beside a 5.6 us weight stream.
//    The K split is over the 8 WAVES of a workgroup (reduced through LDS in wave order: deterministic), so the results leave FINAL:
//    the GELU (-> bf16 hi / lo planes for the down-projection), or f32 q|k|v plus per-32-column LayerNorm moments for the q/k
//    LayerNorm of the attention kernel.  No split-K slabs, no separate reduce / GELU / LayerNorm launches.  Workgroup 0 also stores
//    the completed residual stream (and hidden_states[i-1]).
//  * deer_trunk_gemm       the d-wide projections (out_proj, to_out, mlp_down, ff.3): one 16-column tile per workgroup, K split over
//    the waves; K = d / inner: full K per workgroup, the epilogue applies x += tanh(gate) * y IN PLACE (each workgroup owns its 16
//    columns); K = 4d: two K halves -> two f32 slabs that the next deer_trunk_ln_gemm folds into its prologue.
//  * deer_trunk_mpt_attn   q/k LayerNorm over d_model (from the moments above) + causal ALiBi attention, one workgroup per head.
//
// Activations enter every MFMA as bf16 hi + lo; all reductions are in a fixed order (bit-identical across
// schedules and sibling engines).
#include "common.h"
#include <algorithm>

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((address_space(1))) const void tr_gptr_t;
typedef __attribute__((address_space(3))) void tr_lptr_t;

#define TR_NW 8                 // waves per workgroup: the K split
#define TR_OPITCH 36            // f32 pitch of the [16][32] output tile partials (16-byte aligned rows, staggered banks)

enum { TR_EPI_F32 = 0, TR_EPI_GELU_PLANES = 1, TR_EPI_F32_STATS = 2 };
enum { TR_OUT_SLAB = 0, TR_OUT_INPLACE = 1 };
enum { TR_A_PLANES = 0, TR_A_F32 = 1 };

constexpr int tr_w_bytes(int ks) { return TR_NW * 2 * ks * 1024 > TR_NW * 16 * TR_OPITCH * 4 ? TR_NW * 2 * ks * 1024 : TR_NW * 16 * TR_OPITCH * 4; }

__device__ __forceinline__ void tr_split8(const float4 x0, const float4 x1, bf16x8& hi, bf16x8& lo) {
  const uint32_t h0 = pack2bf(x0.x, x0.y), h1 = pack2bf(x0.z, x0.w), h2 = pack2bf(x1.x, x1.y), h3 = pack2bf(x1.z, x1.w);
  const uint4 h = uint4{h0, h1, h2, h3};
  const uint4 l = uint4{pack2bf(x0.x - __uint_as_float(h0 << 16), x0.y - __uint_as_float(h0 & 0xffff0000u)),
                        pack2bf(x0.z - __uint_as_float(h1 << 16), x0.w - __uint_as_float(h1 & 0xffff0000u)),
                        pack2bf(x1.x - __uint_as_float(h2 << 16), x1.y - __uint_as_float(h2 & 0xffff0000u)),
                        pack2bf(x1.z - __uint_as_float(h3 << 16), x1.w - __uint_as_float(h3 & 0xffff0000u))};
  hi = __builtin_bit_cast(bf16x8, h);
  lo = __builtin_bit_cast(bf16x8, l);
}

// ---------------------------------------------------------------------------------------------------------------------------
// deer_trunk_ln_gemm: workgroup = 32 output columns (2 MFMA column tiles), 8 waves x KS k-tiles of 32 (d = 256 * KS)
// LDS: [8 waves][2 tiles][KS] x 1 KiB packed weight fragments (LDS-DMA) | gamma, beta (d floats each) | row-moment partials;
//      the weight region is reused for the waves' output partials once the MFMAs are done
// ---------------------------------------------------------------------------------------------------------------------------
template <int KS, int EPI>
__global__ __launch_bounds__(64 * TR_NW) void trunk_ln_gemm_kernel(const float* __restrict__ x, const float* __restrict__ slab, int s_in,
                                                                  long slab_stride, const float* __restrict__ gate, float* __restrict__ x_out,
                                                                  float* __restrict__ x_copy, const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, float eps, const bf16_t* __restrict__ Wp, int N,
                                                                  float* __restrict__ out_f32, bf16_t* __restrict__ out_hi,
                                                                  bf16_t* __restrict__ out_lo, int ldo, float* __restrict__ stats, int T,
                                                                  const int* ctl) {
  DEER_RETURN_IF_EXITED(ctl);
  constexpr int D = KS * 256;
  constexpr int KT = D / 32;
  constexpr int W_BYTES = tr_w_bytes(KS);                      // the weight region is reused for the waves' output partials
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  float* gl = reinterpret_cast<float*>(smem + W_BYTES);
  float* bl = gl + D;
  float* red = bl + D;                                         // [2][TR_NW][16]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int kt0 = wave * KS;                                   // this wave's k-tiles: kt0 .. kt0 + KS - 1

  // ---- everything that depends on nothing is requested first: the packed weights (HBM, non-temporal) straight into LDS ----
  unsigned char* wl = smem + wave * (2 * KS * 1024);
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int tile = blockIdx.x * 2 + t;
#pragma unroll
    for (int s = 0; s < KS; ++s)
      __builtin_amdgcn_global_load_lds((tr_gptr_t*)(Wp + (((long)tile * KT + kt0 + s) * 64 + lane) * 8), (tr_lptr_t*)(wl + (t * KS + s) * 1024), 16, 0,
                                       2 /* nt */);
  }
  // ---- residual stream + pending branch of this lane's row (MFMA B-fragment order: row c, k = 32*kt + 8*g .. + 7) ----
  const int row = min(c, T - 1);
  const bool row_ok = c < T;
  const float* xr = x + (long)row * D + kt0 * 32 + g * 8;
  float4 xv[KS][2];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    xv[s][0] = *reinterpret_cast<const float4*>(xr + s * 32);
    xv[s][1] = *reinterpret_cast<const float4*>(xr + s * 32 + 4);
  }
  if (s_in > 0) {                                              // x += tanh(gate) * (slab0 + slab1 ...): deer_resadd_ln's arithmetic
    const float sc = gate != nullptr ? tanhf(*gate) : 1.f;
    const float* sr = slab + (long)row * D + kt0 * 32 + g * 8;
    float4 a[KS][2];
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      a[s][0] = *reinterpret_cast<const float4*>(sr + s * 32);
      a[s][1] = *reinterpret_cast<const float4*>(sr + s * 32 + 4);
    }
    for (int q = 1; q < s_in; ++q) {
      const float* sq = sr + (long)q * slab_stride;
#pragma unroll
      for (int s = 0; s < KS; ++s) {
        const float4 b0 = *reinterpret_cast<const float4*>(sq + s * 32), b1 = *reinterpret_cast<const float4*>(sq + s * 32 + 4);
        a[s][0].x += b0.x; a[s][0].y += b0.y; a[s][0].z += b0.z; a[s][0].w += b0.w;
        a[s][1].x += b1.x; a[s][1].y += b1.y; a[s][1].z += b1.z; a[s][1].w += b1.w;
      }
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      xv[s][0].x += sc * a[s][0].x; xv[s][0].y += sc * a[s][0].y; xv[s][0].z += sc * a[s][0].z; xv[s][0].w += sc * a[s][0].w;
      xv[s][1].x += sc * a[s][1].x; xv[s][1].y += sc * a[s][1].y; xv[s][1].z += sc * a[s][1].z; xv[s][1].w += sc * a[s][1].w;
    }
  }
  // LayerNorm affine -> LDS (shared by the waves)
  if (tid * 4 < D) {
    *reinterpret_cast<float4*>(gl + tid * 4) = *reinterpret_cast<const float4*>(gamma + tid * 4);
    *reinterpret_cast<float4*>(bl + tid * 4) = beta != nullptr ? *reinterpret_cast<const float4*>(beta + tid * 4) : float4{0.f, 0.f, 0.f, 0.f};
  }
  // workgroup 0 stores the completed residual stream (and hidden_states[i-1])
  if (blockIdx.x == 0 && row_ok && (x_out != nullptr || x_copy != nullptr)) {
    const long off = (long)c * D + kt0 * 32 + g * 8;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      if (x_out != nullptr) {
        *reinterpret_cast<float4*>(x_out + off + s * 32) = xv[s][0];
        *reinterpret_cast<float4*>(x_out + off + s * 32 + 4) = xv[s][1];
      }
      if (x_copy != nullptr) {
        *reinterpret_cast<float4*>(x_copy + off + s * 32) = xv[s][0];
        *reinterpret_cast<float4*>(x_copy + off + s * 32 + 4) = xv[s][1];
      }
    }
  }
  if (!row_ok) {
#pragma unroll
    for (int s = 0; s < KS; ++s) xv[s][0] = xv[s][1] = float4{0.f, 0.f, 0.f, 0.f};
  }
  // ---- row moments: mean, then centred second moment (two passes, like deer_resadd_ln); partials meet in LDS in wave order ----
  float ps = 0.f;
#pragma unroll
  for (int s = 0; s < KS; ++s) ps += (xv[s][0].x + xv[s][0].y + xv[s][0].z + xv[s][0].w) + (xv[s][1].x + xv[s][1].y + xv[s][1].z + xv[s][1].w);
  ps += __shfl_xor(ps, 16, 64);
  ps += __shfl_xor(ps, 32, 64);
  if (g == 0) red[wave * 16 + c] = ps;
  __syncthreads();
  float mean = 0.f;
#pragma unroll
  for (int w = 0; w < TR_NW; ++w) mean += red[w * 16 + c];
  mean *= (1.f / D);
  float pv = 0.f;
#pragma unroll
  for (int s = 0; s < KS; ++s)
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float a0 = xv[s][h].x - mean, a1 = xv[s][h].y - mean, a2 = xv[s][h].z - mean, a3 = xv[s][h].w - mean;
      pv += a0 * a0 + a1 * a1 + a2 * a2 + a3 * a3;
    }
  pv += __shfl_xor(pv, 16, 64);
  pv += __shfl_xor(pv, 32, 64);
  if (g == 0) red[TR_NW * 16 + wave * 16 + c] = pv;
  __syncthreads();
  float var = 0.f;
#pragma unroll
  for (int w = 0; w < TR_NW; ++w) var += red[TR_NW * 16 + w * 16 + c];
  const float rstd = rsqrtf(var * (1.f / D) + eps);

  // ---- this wave's weights have landed (its own DMAs: no barrier needed) ----
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const float* gp = gl + (kt0 + s) * 32 + g * 8;
    const float* bp = bl + (kt0 + s) * 32 + g * 8;
    const float4 g0 = *reinterpret_cast<const float4*>(gp), g1 = *reinterpret_cast<const float4*>(gp + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(bp), b1 = *reinterpret_cast<const float4*>(bp + 4);
    float4 y0, y1;
    y0.x = (xv[s][0].x - mean) * rstd * g0.x + b0.x; y0.y = (xv[s][0].y - mean) * rstd * g0.y + b0.y;
    y0.z = (xv[s][0].z - mean) * rstd * g0.z + b0.z; y0.w = (xv[s][0].w - mean) * rstd * g0.w + b0.w;
    y1.x = (xv[s][1].x - mean) * rstd * g1.x + b1.x; y1.y = (xv[s][1].y - mean) * rstd * g1.y + b1.y;
    y1.z = (xv[s][1].z - mean) * rstd * g1.z + b1.z; y1.w = (xv[s][1].w - mean) * rstd * g1.w + b1.w;
    bf16x8 ah, al;
    tr_split8(y0, y1, ah, al);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const bf16x8 wf = *reinterpret_cast<const bf16x8*>(wl + (t * KS + s) * 1024 + lane * 16);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, ah, acc[t], 0, 0, 0);
      acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, al, acc[t], 0, 0, 0);
    }
  }
  // ---- K partials of the 8 waves -> LDS -> summed in wave order; lane holds out[m = c][n = 16 t + 4 g .. + 3] ----
  __syncthreads();                                              // every wave is done with the weight region
  float* opart = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int t = 0; t < 2; ++t)
    *reinterpret_cast<float4*>(opart + (wave * 16 + c) * TR_OPITCH + t * 16 + g * 4) = float4{acc[t][0], acc[t][1], acc[t][2], acc[t][3]};
  __syncthreads();
  const int m = tid >> 5, n = tid & 31;                          // one output element per thread: 16 rows x 32 columns
  float v = 0.f;
#pragma unroll
  for (int w = 0; w < TR_NW; ++w) v += opart[(w * 16 + m) * TR_OPITCH + n];
  const int col = blockIdx.x * 32 + n;
  if (EPI == TR_EPI_GELU_PLANES) {
    v = gelu_erf(v);
    const float vn = __shfl_down(v, 1, 64);
    if ((n & 1) == 0 && m < T) {
      const uint32_t h = pack2bf(v, vn);
      *reinterpret_cast<uint32_t*>(out_hi + (long)m * ldo + col) = h;
      *reinterpret_cast<uint32_t*>(out_lo + (long)m * ldo + col) = pack2bf(v - __uint_as_float(h << 16), vn - __uint_as_float(h & 0xffff0000u));
    }
  } else {
    if (m < T) out_f32[(long)m * ldo + col] = v;
    if (EPI == TR_EPI_F32_STATS) {                               // moments of this row over the workgroup's 32 columns (half-wave reductions)
      float s1 = v;
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) s1 += __shfl_xor(s1, o, 64);
      const float mu = s1 * (1.f / 32.f);
      float s2 = (v - mu) * (v - mu);
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) s2 += __shfl_xor(s2, o, 64);
      if (n == 0) *reinterpret_cast<float2*>(stats + ((long)blockIdx.x * 16 + m) * 2) = float2{mu, s2};
    }
  }
}

// y = LN(x + tanh(*gate) * sum_s slab[s]) W^T.  x f32 [T][d], slab f32 [s_in][slab_stride] (row-major [.][d]), Wp packed [N/16][d/32][64][8].
// epi: 0 = out_f32 [T][ldo]; 1 = exact GELU -> bf16 hi / lo planes [T][ldo]; 2 = out_f32 + stats [N/32][16][2] (mean, centred sum of
// squares of the row over each 32-column group).  x_out / x_copy (optional): the completed residual stream, stored by workgroup 0.
extern "C" int deer_trunk_ln_gemm(const float* x, const float* slab, int s_in, long slab_stride, const float* gate, float* x_out, float* x_copy,
                                  const float* gamma, const float* beta, float eps, const void* Wp, int N, int d, int epi, float* out_f32,
                                  void* out_hi, void* out_lo, int ldo, float* stats, int T, const int* ctl, void* stream) {
  if (x == nullptr || gamma == nullptr || Wp == nullptr || T <= 0 || T > 16 || N <= 0 || (N & 31) || s_in < 0 || (s_in > 0 && slab == nullptr) ||
      epi < 0 || epi > 2 || x_out == x)
    return DEER_ERR_SHAPE;
  if (epi == TR_EPI_GELU_PLANES ? (out_hi == nullptr || out_lo == nullptr || (ldo & 1)) : out_f32 == nullptr) return DEER_ERR_SHAPE;
  if (epi == TR_EPI_F32_STATS && stats == nullptr) return DEER_ERR_SHAPE;
  if (d != 256 && d != 2048) return DEER_ERR_SHAPE;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const bf16_t* wp = reinterpret_cast<const bf16_t*>(Wp);
  bf16_t* oh = reinterpret_cast<bf16_t*>(out_hi);
  bf16_t* ol = reinterpret_cast<bf16_t*>(out_lo);
#define DEER_TLG(KS_, EPI_)                                                                                                       \
  do {                                                                                                                            \
    constexpr int smem = tr_w_bytes(KS_) + 2 * KS_ * 256 * 4 + 2 * TR_NW * 16 * 4;                                         \
    static std::atomic<bool> attr_set{false};                                                                                     \
    auto kern = &trunk_ln_gemm_kernel<KS_, EPI_>;                                                                                 \
    if (!attr_set) {                                                                                                              \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess) \
        return DEER_ERR_LAUNCH;                                                                                                   \
      attr_set = true;                                                                                                            \
    }                                                                                                                             \
    hipLaunchKernelGGL(kern, dim3(N / 32), dim3(64 * TR_NW), smem, st, x, slab, s_in, slab_stride, gate, x_out, x_copy, gamma, beta, eps, wp, N, \
                       out_f32, oh, ol, ldo, stats, T, ctl);                                                                      \
  } while (0)
#define DEER_TLG_E(KS_)                                          \
  do {                                                           \
    if (epi == 0) DEER_TLG(KS_, 0);                              \
    else if (epi == 1) DEER_TLG(KS_, 1);                         \
    else DEER_TLG(KS_, 2);                                       \
  } while (0)
  if (d == 2048) DEER_TLG_E(8); else DEER_TLG_E(1);
#undef DEER_TLG_E
#undef DEER_TLG
  DEER_LAUNCH_CHECK();
  return DEER_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// deer_trunk_gemm: workgroup (tile, ks) = one 16-column tile over the K range [ks * 256 * KS, (ks + 1) * 256 * KS); 8 waves x KS k-tiles
// ---------------------------------------------------------------------------------------------------------------------------
template <int KS, int AMODE, int OUT>
__global__ __launch_bounds__(64 * TR_NW) void trunk_gemm_kernel(const bf16_t* __restrict__ Ahi, const bf16_t* __restrict__ Alo,
                                                               const float* __restrict__ Af, int lda, const bf16_t* __restrict__ Wp, int N, int K,
                                                               float* __restrict__ out, long slab_stride, const float* __restrict__ gate,
                                                               float* __restrict__ x_copy, int T, const int* ctl) {
  DEER_RETURN_IF_EXITED(ctl);
  __shared__ __attribute__((aligned(16))) float opart[TR_NW * 16 * 20];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int tile = blockIdx.x, ks = blockIdx.y;
  const int kt0 = (ks * TR_NW + wave) * KS;
  const int KT = K >> 5;
  const u32x4* wp = reinterpret_cast<const u32x4*>(Wp) + ((long)tile * KT + kt0) * 64 + lane;
  u32x4 w[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) w[s] = __builtin_nontemporal_load(wp + s * 64);
  const int row = min(c, T - 1);
  bf16x8 ah[KS], al[KS];
  if (AMODE == TR_A_PLANES) {
    const long off = (long)row * lda + kt0 * 32 + g * 8;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      ah[s] = *reinterpret_cast<const bf16x8*>(Ahi + off + s * 32);
      al[s] = *reinterpret_cast<const bf16x8*>(Alo + off + s * 32);
    }
  } else {
    const float* ap = Af + (long)row * lda + kt0 * 32 + g * 8;
#pragma unroll
    for (int s = 0; s < KS; ++s) tr_split8(*reinterpret_cast<const float4*>(ap + s * 32), *reinterpret_cast<const float4*>(ap + s * 32 + 4), ah[s], al[s]);
  }
  f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const bf16x8 wf = __builtin_bit_cast(bf16x8, w[s]);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, ah[s], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, al[s], acc, 0, 0, 0);
  }
  *reinterpret_cast<float4*>(opart + (wave * 16 + c) * 20 + g * 4) = float4{acc[0], acc[1], acc[2], acc[3]};
  __syncthreads();
  if (tid < 256) {
    const int m = tid >> 4, n = tid & 15;
    float v = 0.f;
#pragma unroll
    for (int wv = 0; wv < TR_NW; ++wv) v += opart[(wv * 16 + m) * 20 + n];
    const int col = tile * 16 + n;
    if (OUT == TR_OUT_SLAB) {
      out[(long)ks * slab_stride + (long)m * N + col] = m < T ? v : 0.f;
    } else if (m < T) {
      const float sc = gate != nullptr ? tanhf(*gate) : 1.f;
      const float xn = out[(long)m * N + col] + sc * v;
      out[(long)m * N + col] = xn;
      if (x_copy != nullptr) x_copy[(long)m * N + col] = xn;
    }
  }
}

// y = A W^T for the d-wide projections.  A: bf16 hi / lo planes [T][lda] (a_hi, a_lo) or f32 [T][lda] (a_f32, planes NULL).
// splitk == 1 and inplace: out = x [T][N], x += tanh(*gate or 1) * y (+ optional copy);  otherwise out = f32 slabs [splitk][slab_stride].
extern "C" int deer_trunk_gemm(const void* a_hi, const void* a_lo, const float* a_f32, int lda, const void* Wp, int N, int K, int splitk, int inplace,
                               float* out, long slab_stride, const float* gate, float* x_copy, int T, const int* ctl, void* stream) {
  if (Wp == nullptr || out == nullptr || T <= 0 || T > 16 || N <= 0 || (N & 15) || K <= 0 || splitk <= 0 || (K % (splitk * 256)) != 0 || (lda & 7))
    return DEER_ERR_SHAPE;
  const bool planes = a_hi != nullptr;
  if (planes ? (a_lo == nullptr || a_f32 != nullptr) : a_f32 == nullptr) return DEER_ERR_SHAPE;
  if (inplace && splitk != 1) return DEER_ERR_SHAPE;
  if (!inplace && slab_stride < (long)16 * N) return DEER_ERR_SHAPE;
  const int ks = K / (splitk * 256);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  dim3 grid(N / 16, splitk);
  const bf16_t* ah = reinterpret_cast<const bf16_t*>(a_hi);
  const bf16_t* al = reinterpret_cast<const bf16_t*>(a_lo);
  const bf16_t* wp = reinterpret_cast<const bf16_t*>(Wp);
#define DEER_TG(KS_, AM_, OUT_) \
  hipLaunchKernelGGL((trunk_gemm_kernel<KS_, AM_, OUT_>), grid, dim3(64 * TR_NW), 0, st, ah, al, a_f32, lda, wp, N, K, out, slab_stride, gate, x_copy, T, ctl)
#define DEER_TG_K(AM_, OUT_)                         \
  do {                                               \
    switch (ks) {                                    \
      case 1: DEER_TG(1, AM_, OUT_); break;          \
      case 2: DEER_TG(2, AM_, OUT_); break;          \
      case 8: DEER_TG(8, AM_, OUT_); break;          \
      case 16: DEER_TG(16, AM_, OUT_); break;        \
      default: return DEER_ERR_SHAPE;                \
    }                                                \
  } while (0)
  if (planes) { if (inplace) DEER_TG_K(TR_A_PLANES, TR_OUT_INPLACE); else DEER_TG_K(TR_A_PLANES, TR_OUT_SLAB); }
  else { if (inplace) DEER_TG_K(TR_A_F32, TR_OUT_INPLACE); else DEER_TG_K(TR_A_F32, TR_OUT_SLAB); }
#undef DEER_TG_K
#undef DEER_TG
  DEER_LAUNCH_CHECK();
  return DEER_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// deer_trunk_mpt_attn: MPT attention core on final f32 q|k|v [T][3d]: q/k LayerNorm over the FULL d_model from the
// per-32-column moments of deer_trunk_ln_gemm (combined in group order: equal group sizes, Chan's formula), then per head
// softmax(q k^T / sqrt(hd) + alibi + causal + key-pad) v -> bf16 hi / lo planes [T][ldo].  One workgroup per head.
// ---------------------------------------------------------------------------------------------------------------------------
#define TM_MAXT 16
__global__ __launch_bounds__(256) void trunk_mpt_attn_kernel(const float* __restrict__ qkv, const float* __restrict__ stats, int d_model, int hd,
                                                             const float* __restrict__ q_ln_w, const float* __restrict__ k_ln_w, float eps,
                                                             const unsigned char* __restrict__ key_mask, float alibi_slope_base, int n_heads,
                                                             bf16_t* __restrict__ out_hi, bf16_t* __restrict__ out_lo, int ldo, int T, const int* ctl) {
  DEER_RETURN_IF_EXITED(ctl);
  __shared__ __attribute__((aligned(16))) float qs[TM_MAXT][128 + 4];
  __shared__ __attribute__((aligned(16))) float ks[TM_MAXT][128 + 4];
  __shared__ __attribute__((aligned(16))) float vs[TM_MAXT][128 + 4];
  __shared__ float sim[TM_MAXT][TM_MAXT + 1];
  __shared__ float mom[2][TM_MAXT][2];                          // (mean, rstd) of the q row / k row over d_model
  const int h = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ld = 3 * d_model;
  const bool qk_ln = q_ln_w != nullptr;
  if (qk_ln && tid < 2 * TM_MAXT) {
    const int which = tid >> 4, r = tid & 15;
    const int G = d_model >> 5;
    const float* sp = stats + ((long)which * G * 16 + r) * 2;   // group gq of q|k: index (which * G + gq) * 16 + r
    float ms = 0.f;
    for (int gq = 0; gq < G; ++gq) ms += sp[(long)gq * 32];
    const float mean = ms / (float)G;
    float m2 = 0.f;
    for (int gq = 0; gq < G; ++gq) {
      const float dm = sp[(long)gq * 32] - mean;
      m2 += sp[(long)gq * 32 + 1] + 32.f * dm * dm;
    }
    mom[which][r][0] = mean;
    mom[which][r][1] = rsqrtf(m2 / (float)d_model + eps);
  }
  __syncthreads();
  const int hd4 = hd >> 2;
  for (int idx = tid; idx < T * hd4; idx += 256) {
    const int t = idx / hd4, dd = (idx - t * hd4) * 4;
    const float* p = qkv + (long)t * ld + h * hd + dd;
    float4 q = *reinterpret_cast<const float4*>(p);
    float4 k = *reinterpret_cast<const float4*>(p + d_model);
    if (qk_ln) {
      const float4 gq = *reinterpret_cast<const float4*>(q_ln_w + h * hd + dd), gk = *reinterpret_cast<const float4*>(k_ln_w + h * hd + dd);
      const float mq = mom[0][t][0], rq = mom[0][t][1], mk = mom[1][t][0], rk = mom[1][t][1];
      q.x = (q.x - mq) * rq * gq.x; q.y = (q.y - mq) * rq * gq.y; q.z = (q.z - mq) * rq * gq.z; q.w = (q.w - mq) * rq * gq.w;
      k.x = (k.x - mk) * rk * gk.x; k.y = (k.y - mk) * rk * gk.y; k.z = (k.z - mk) * rk * gk.z; k.w = (k.w - mk) * rk * gk.w;
    }
    *reinterpret_cast<float4*>(&qs[t][dd]) = q;
    *reinterpret_cast<float4*>(&ks[t][dd]) = k;
    *reinterpret_cast<float4*>(&vs[t][dd]) = *reinterpret_cast<const float4*>(p + 2 * d_model);
  }
  __syncthreads();
  const float sc = rsqrtf((float)hd);
  const float slope = exp2f(-alibi_slope_base * (float)(h + 1) / (float)n_heads);
  for (int idx = tid; idx < T * T; idx += 256) {
    const int i = idx / T, j = idx - i * T;
    float a = 0.f;
    if (j <= i)
      for (int dd = 0; dd < hd; dd += 4) {
        const float4 qv = *reinterpret_cast<const float4*>(&qs[i][dd]);
        const float4 kv = *reinterpret_cast<const float4*>(&ks[j][dd]);
        a += qv.x * kv.x + qv.y * kv.y + qv.z * kv.z + qv.w * kv.w;
      }
    a = a * sc - (float)(T - 1 - j) * slope;
    if (j > i || (key_mask != nullptr && key_mask[j] == 0)) a = -INFINITY;
    sim[i][j] = a;
  }
  __syncthreads();
  for (int i = wave; i < T; i += 4) {
    float v = (lane < T) ? sim[i][lane] : -INFINITY;
    const float mx = wave_max(v);
    const float p = (lane < T) ? expf(v - mx) : 0.f;
    const float sum = wave_sum(p);
    if (lane < T) sim[i][lane] = p / sum;
  }
  __syncthreads();
  for (int idx = tid; idx < T * hd; idx += 256) {
    const int t = idx / hd, dd = idx - t * hd;
    float a = 0.f;
    for (int j = 0; j <= t; ++j) a += sim[t][j] * vs[j][dd];
    const bf16_t hi = f2bf(a);
    out_hi[(long)t * ldo + h * hd + dd] = hi;
    out_lo[(long)t * ldo + h * hd + dd] = f2bf(a - bf2f(hi));
  }
}

extern "C" int deer_trunk_mpt_attn(const float* qkv, const float* stats, int d_model, int n_heads, const float* q_ln_w, const float* k_ln_w, float eps,
                                   const unsigned char* key_mask, float alibi_bias_max, void* out_hi, void* out_lo, int ldo, int T, const int* ctl,
                                   void* stream) {
  const int hd = d_model / n_heads;
  if (qkv == nullptr || T <= 0 || T > TM_MAXT || hd > 128 || (hd & 3) || hd * n_heads != d_model || (d_model & 31) || out_hi == nullptr ||
      out_lo == nullptr)
    return DEER_ERR_SHAPE;
  if ((q_ln_w == nullptr) != (k_ln_w == nullptr) || (q_ln_w != nullptr && stats == nullptr)) return DEER_ERR_SHAPE;
  hipLaunchKernelGGL(trunk_mpt_attn_kernel, dim3(n_heads), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), qkv, stats, d_model, hd, q_ln_w, k_ln_w,
                     eps, key_mask, alibi_bias_max, n_heads, reinterpret_cast<bf16_t*>(out_hi), reinterpret_cast<bf16_t*>(out_lo), ldo, T, ctl);
  DEER_LAUNCH_CHECK();
  return DEER_OK;
}
