// The LLM trunk at ONE environment (<= 16 text rows, one MFMA row tile): the wide projections of the decoder block / gated x-attn
// layer (open_flamingo/src/helpers.py:15-22,260-279; MPT block) with their consumers' row operations folded in.
//
// Round 3 measured the one-environment trunk layer at 112 us for 174 MB of weights: the six weight-streaming GEMMs ran at 4 TB/s, the
// other 70 us were nine latency-bound row kernels and fifteen kernel boundaries.  Three of those row kernels exist only because
// the wide projections (ff.1 / mlp_up: N = 4d, Wqkv: N = 3d; K = d) leave their result as split-K slabs: the GELU pass in front of each
// down-projection (deer_slab_gelu_split) and the slab reduction in front of the attention (qkv_reduce_ln_kernel).  With <= 16 rows the
// K split can live INSIDE the workgroup instead:
//
//  * deer_trunk_wide_gemm   workgroup = 32 output columns x the FULL K; the 8 waves split K, their partial tiles meet in LDS and are
//    summed in wave order (deterministic), so the result leaves FINAL: exact GELU -> bf16 hi / lo planes (the operand of the
//    down-projection), or f32 q|k|v plus the moments of every row over the workgroup's 32 columns, from which deer_trunk_mpt_attn
//    rebuilds the q / k LayerNorm over d_model.  The activation (LayerNorm output of deer_resadd_ln_packed) arrives as bf16 hi / lo
//    planes in MFMA-FRAGMENT ORDER [k-tile][lane][8]: one contiguous, fully coalesced 1 KiB read per wave instruction.  (A row-major
//    operand read in fragment shape - 16 rows x 16 B per lane, every lane its own cache line - was measured at ~30 GB/s per CU in
//    this round's first form of the kernel: 128-384 KB of activation per workgroup then cost more than the 128 KB of weights.)
//  * deer_trunk_mpt_attn    q/k LayerNorm over d_model (from the moments above) + causal ALiBi attention, one workgroup per head.
//
// Activations enter every MFMA as bf16 hi + lo; all reductions are in a fixed order (bit-identical across
// schedules and sibling engines).
#include "common.h"
#include <type_traits>
#include <algorithm>

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

#define TR_NW 8                 // waves per workgroup: the K split
#define TR_OPITCH 36            // f32 pitch of the [16][32] output tile partials (16-byte aligned rows, staggered banks)

enum { TR_EPI_F32 = 0, TR_EPI_GELU_PLANES = 1, TR_EPI_F32_STATS = 2 };

// workgroup = 16 CT output columns (CT MFMA column tiles), 8 waves x KS k-tiles of 32 (K = 256 * KS): CT = 2 up to K = 2048, CT = 1 at
// K = 4096 (MPT-7B, round 5: 16 k-tiles per wave - the weight + activation fragments of TWO column tiles would not fit the registers;
// the workgroup streams the same 128 KB of weights either way)
// (bodies are device functions taking the LOGICAL workgroup index: csrc/persistent_layer.hip runs them as phases of one launch)
template <int KS, int EPI, int CT = 2, bool F16 = false>
__device__ __forceinline__ void trunk_wide_gemm_body(const bf16_t* __restrict__ Ahi, const bf16_t* __restrict__ Alo,
                                                     const bf16_t* __restrict__ Wp, float* __restrict__ out_f32,
                                                     bf16_t* __restrict__ out_hi, bf16_t* __restrict__ out_lo, int ldo,
                                                     float* __restrict__ stats, int T, int bx, float* opart /* LDS: TR_NW * 16 * TR_OPITCH floats */) {
  static_assert(CT == 2 || EPI != TR_EPI_F32_STATS, "the q/k LayerNorm moments are per 32 columns");
  constexpr int KT = KS * TR_NW, NC = 16 * CT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int kt0 = wave * KS;                                   // this wave's k-tiles: kt0 .. kt0 + KS - 1
  // every request of the wave is issued before anything is consumed: CT KS weight fragments (HBM, non-temporal) + 2 KS activation
  // fragments (L2) of 1 KiB each
  u32x4 w[CT][KS];
#pragma unroll
  for (int t = 0; t < CT; ++t) {
    const u32x4* wp = reinterpret_cast<const u32x4*>(Wp) + ((long)(bx * CT + t) * KT + kt0) * 64 + lane;
#pragma unroll
    for (int s = 0; s < KS; ++s) w[t][s] = __builtin_nontemporal_load(wp + s * 64);
  }
  bf16x8 ah[KS], al[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    ah[s] = *reinterpret_cast<const bf16x8*>(Ahi + ((long)(kt0 + s) * 64 + lane) * 8);
    al[s] = *reinterpret_cast<const bf16x8*>(Alo + ((long)(kt0 + s) * 64 + lane) * 8);
  }
  f32x4 acc[CT];
#pragma unroll
  for (int t = 0; t < CT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int s = 0; s < KS; ++s)
#pragma unroll
    for (int t = 0; t < CT; ++t) {
      const bf16x8 wf = __builtin_bit_cast(bf16x8, w[t][s]);
      acc[t] = mfma16<F16>(wf, ah[s], acc[t]);
      acc[t] = mfma16<F16>(wf, al[s], acc[t]);
    }
  // K partials of the 8 waves -> LDS -> summed in wave order; lane holds out[m = c][n = 16 t + 4 g .. + 3]
#pragma unroll
  for (int t = 0; t < CT; ++t)
    *reinterpret_cast<float4*>(opart + (wave * 16 + c) * TR_OPITCH + t * 16 + g * 4) = float4{acc[t][0], acc[t][1], acc[t][2], acc[t][3]};
  __syncthreads();
  if (CT == 1 && tid >= 16 * NC) return;                         // 16 rows x 16 columns: the first 256 threads (no barrier follows)
  const int m = CT == 2 ? tid >> 5 : tid >> 4, n = CT == 2 ? tid & 31 : tid & 15;   // one output element per thread: 16 rows x NC columns
  float v = 0.f;
#pragma unroll
  for (int wv = 0; wv < TR_NW; ++wv) v += opart[(wv * 16 + m) * TR_OPITCH + n];
  const int col = bx * NC + n;
  if (EPI == TR_EPI_GELU_PLANES) {
    v = gelu_erf(v);
    const float vn = __shfl_down(v, 1, 64);
    if ((n & 1) == 0 && m < T) {
      uint32_t h, l;
      split2<F16>(v, vn, h, l);
      *reinterpret_cast<uint32_t*>(out_hi + (long)m * ldo + col) = h;
      *reinterpret_cast<uint32_t*>(out_lo + (long)m * ldo + col) = l;
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
      if (n == 0) *reinterpret_cast<float2*>(stats + ((long)bx * 16 + m) * 2) = float2{mu, s2};
    }
  }
}

template <int KS, int EPI, int CT = 2, bool F16 = false>
__global__ __launch_bounds__(64 * TR_NW) void trunk_wide_gemm_kernel(const bf16_t* __restrict__ Ahi, const bf16_t* __restrict__ Alo,
                                                                    const bf16_t* __restrict__ Wp, float* __restrict__ out_f32,
                                                                    bf16_t* __restrict__ out_hi, bf16_t* __restrict__ out_lo, int ldo,
                                                                    float* __restrict__ stats, int T, const int* ctl) {
  DEER_RETURN_IF_EXITED(ctl);
  __shared__ __attribute__((aligned(16))) float opart[TR_NW * 16 * TR_OPITCH];
  trunk_wide_gemm_body<KS, EPI, CT, F16>(Ahi, Alo, Wp, out_f32, out_hi, out_lo, ldo, stats, T, blockIdx.x, opart);
}

#ifndef DEER_BODIES_ONLY
// y = A W^T for the wide bias-free Linears at <= 16 rows.  a_hi / a_lo: the activation as bf16 planes in MFMA-fragment order
// [K/32][64 lanes][8] (lane = 16 * (k % 32 / 8) + row; written by deer_resadd_ln_packed), Wp packed [N/16][K/32][64][8].
// epi: 0 = out_f32 [T][ldo]; 1 = exact GELU -> ROW-MAJOR bf16 hi / lo planes [T][ldo]; 2 = out_f32 + stats [N/32][16][2] (mean, centred
// sum of squares of the row over each 32-column group; not at K = 4096).  K = 256, 2048 or 4096.
template <bool F16>
static int trunk_wide_gemm(const void* a_hi, const void* a_lo, const void* Wp, int N, int K, int epi, float* out_f32, void* out_hi,
                                    void* out_lo, int ldo, float* stats, int T, const int* ctl, void* stream) {
  if (a_hi == nullptr || a_lo == nullptr || Wp == nullptr || T <= 0 || T > 16 || N <= 0 || (N & 31) || epi < 0 || epi > 2) return DEER_ERR_SHAPE;
  if (epi == TR_EPI_GELU_PLANES ? (out_hi == nullptr || out_lo == nullptr || (ldo & 1)) : out_f32 == nullptr) return DEER_ERR_SHAPE;
  if (epi == TR_EPI_F32_STATS && stats == nullptr) return DEER_ERR_SHAPE;
  if (K != 256 && K != 2048 && K != 4096) return DEER_ERR_SHAPE;
  if (K == 4096 && epi == TR_EPI_F32_STATS) return DEER_ERR_SHAPE;   // (MPT-7B has no q/k LayerNorm; the moments are per 32 columns)
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const bf16_t* ah = reinterpret_cast<const bf16_t*>(a_hi);
  const bf16_t* al = reinterpret_cast<const bf16_t*>(a_lo);
  const bf16_t* wp = reinterpret_cast<const bf16_t*>(Wp);
  bf16_t* oh = reinterpret_cast<bf16_t*>(out_hi);
  bf16_t* ol = reinterpret_cast<bf16_t*>(out_lo);
#define DEER_TWG(KS_, EPI_) \
  hipLaunchKernelGGL((trunk_wide_gemm_kernel<KS_, EPI_, 2, F16>), dim3(N / 32), dim3(64 * TR_NW), 0, st, ah, al, wp, out_f32, oh, ol, ldo, stats, T, ctl)
#define DEER_TWG_E(KS_)                       \
  do {                                        \
    if (epi == 0) DEER_TWG(KS_, 0);           \
    else if (epi == 1) DEER_TWG(KS_, 1);      \
    else DEER_TWG(KS_, 2);                    \
  } while (0)
  if (K == 4096) {                                            // 16 columns per workgroup
    if (epi == 0) hipLaunchKernelGGL((trunk_wide_gemm_kernel<16, 0, 1, F16>), dim3(N / 16), dim3(64 * TR_NW), 0, st, ah, al, wp, out_f32, oh, ol, ldo, stats, T, ctl);
    else hipLaunchKernelGGL((trunk_wide_gemm_kernel<16, 1, 1, F16>), dim3(N / 16), dim3(64 * TR_NW), 0, st, ah, al, wp, out_f32, oh, ol, ldo, stats, T, ctl);
  } else if (K == 2048) DEER_TWG_E(8); else DEER_TWG_E(1);
#undef DEER_TWG_E
#undef DEER_TWG
  DEER_LAUNCH_CHECK();
  return DEER_OK;
}
extern "C" int deer_trunk_wide_gemm(const void* a_hi, const void* a_lo, const void* Wp, int N, int K, int epi, float* out_f32, void* out_hi,
                                    void* out_lo, int ldo, float* stats, int T, const int* ctl, void* stream) {
  return trunk_wide_gemm<false>(a_hi, a_lo, Wp, N, K, epi, out_f32, out_hi, out_lo, ldo, stats, T, ctl, stream);
}
extern "C" int deer_trunk_wide_gemm_f16(const void* a_hi, const void* a_lo, const void* Wp, int N, int K, int epi, float* out_f32, void* out_hi,
                                        void* out_lo, int ldo, float* stats, int T, const int* ctl, void* stream) {   // fp16 weights / planes (round 6)
  return trunk_wide_gemm<true>(a_hi, a_lo, Wp, N, K, epi, out_f32, out_hi, out_lo, ldo, stats, T, ctl, stream);
}
#endif  // DEER_BODIES_ONLY

// ---------------------------------------------------------------------------------------------------------------------------
// deer_trunk_mpt_attn: MPT attention core on final f32 q|k|v [T][3d]: q/k LayerNorm over the FULL d_model from the
// per-32-column moments of deer_trunk_wide_gemm (combined in group order: equal group sizes, Chan's formula), then per head
// softmax(q k^T / sqrt(hd) + alibi + causal + key-pad) v -> bf16 hi / lo planes [T][ldo].  One workgroup per head.
// ---------------------------------------------------------------------------------------------------------------------------
#define TM_MAXT 16
#define TM_LDS_FLOATS (3 * TM_MAXT * 132 + TM_MAXT * (TM_MAXT + 1) + 2 * TM_MAXT * 2)
#if defined(DEER_KTRACE) && !defined(DEER_BODIES_ONLY)
KT_DEFINE(mpt_attn)
#define MKT(slot) KT(mpt_attn, h == 0, slot)
#else
#define MKT(slot) do { } while (0)
#endif
// Round 5 (tools/ktrace_trunk.py: the five-barrier form spent 1.9 us on the moments, 1.9 on the q k v loads, 1.3 on the scores, 2.3
// on the softmax and 2.8 on P V of a 10.2 us launch): the q k v pieces are REQUESTED before the moments are combined (one latency
// instead of two), the moments are combined in one pass over registers, and a query row is ONE WAVE from scores to output - lane =
// (key j = lane / 4, quarter of the head dimension): partial dot products meet by two xor-shuffles, the softmax runs over the 16
// key groups by four more, P V takes the probabilities from lane registers (v_readlane) - two barriers instead of five.
template <int NT, bool F16 = false>     // threads of the workgroup (256; 512 as a phase of the persistent layer); F16: fp16 hi / lo output planes
__device__ __forceinline__ void trunk_mpt_attn_body(const float* __restrict__ qkv, const float* __restrict__ stats, int d_model, int hd,
                                                    const float* __restrict__ q_ln_w, const float* __restrict__ k_ln_w, float eps,
                                                    const unsigned char* __restrict__ key_mask, float alibi_slope_base, int n_heads,
                                                    bf16_t* __restrict__ out_hi, bf16_t* __restrict__ out_lo, int ldo, int T, int h,
                                                    float* lds /* TM_LDS_FLOATS floats, 16-byte aligned */) {
  float (*qs)[128 + 4] = reinterpret_cast<float (*)[128 + 4]>(lds);
  float (*ks)[128 + 4] = reinterpret_cast<float (*)[128 + 4]>(lds + TM_MAXT * 132);
  float (*vs)[128 + 4] = reinterpret_cast<float (*)[128 + 4]>(lds + 2 * TM_MAXT * 132);
  float (*mom)[TM_MAXT][2] = reinterpret_cast<float (*)[TM_MAXT][2]>(lds + 3 * TM_MAXT * 132 + TM_MAXT * (TM_MAXT + 1));   // (mean, rstd) of the q / k row over d_model
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ld = 3 * d_model;
  const bool qk_ln = q_ln_w != nullptr;
  MKT(8);
  // ---- requests first: the first 8 moment groups of this thread (returned first: vmcnt counts in order), then this thread's float4
  //      pieces of q | k | v of head h (T * hd / 4 <= 512 pieces) ----
  const int st_which = (tid >> 7) & 1, st_r = (tid >> 3) & 15, st_part = tid & 7;
  const int G = d_model >> 5, per = G >> 3;                     // launcher: d_model % 256 == 0
  const float* sp = stats + (((long)st_which * G + st_part * per) * 16 + st_r) * 2;
  float2 v[8];
  if (qk_ln && tid < 256) {
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = (u < per) ? *reinterpret_cast<const float2*>(sp + (long)u * 32) : float2{0.f, 0.f};
  }
  const int hd4 = hd >> 2;
  constexpr int PIECES = (TM_MAXT * 32 + NT - 1) / NT;
  float4 qr[PIECES], kr[PIECES], vr[PIECES], gq[PIECES], gk[PIECES];
#pragma unroll
  for (int u = 0; u < PIECES; ++u) {
    const int idx = tid + u * NT;
    if (idx < T * hd4) {
      const int t = idx / hd4, dd = (idx - t * hd4) * 4;
      const float* p = qkv + (long)t * ld + h * hd + dd;
      qr[u] = *reinterpret_cast<const float4*>(p);
      kr[u] = *reinterpret_cast<const float4*>(p + d_model);
      vr[u] = *reinterpret_cast<const float4*>(p + 2 * d_model);
      if (qk_ln) {
        gq[u] = *reinterpret_cast<const float4*>(q_ln_w + h * hd + dd);
        gk[u] = *reinterpret_cast<const float4*>(k_ln_w + h * hd + dd);
      }
    }
  }
  if (qk_ln && tid < 256) {
    // 2 x 16 rows x G groups of 32 columns: thread (which, r, part) combines G/8 groups, the 8 parts of a row meet by xor-shuffles
    // (equal group sizes -> mean = average of the group means; M2 = sum of the group M2 + 32 * sum (group mean - mean)^2)
    const int which = st_which, r = st_r, part = st_part;
    float ms = 0.f, m2 = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) { ms += v[u].x; m2 += v[u].y; }
    for (int j0 = 8; j0 < per; j0 += 8) {                       // d_model > 2048
      float2 w[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) w[u] = (j0 + u < per) ? *reinterpret_cast<const float2*>(sp + (long)(j0 + u) * 32) : float2{0.f, 0.f};
#pragma unroll
      for (int u = 0; u < 8; ++u) { ms += w[u].x; m2 += w[u].y; }
    }
    ms += __shfl_xor(ms, 1, 64); ms += __shfl_xor(ms, 2, 64); ms += __shfl_xor(ms, 4, 64);
    const float mean = ms / (float)G;
    float dv = 0.f;
    if (per <= 8) {                                             // the group means are still in registers (d_model <= 2048)
#pragma unroll
      for (int u = 0; u < 8; ++u) dv += (u < per) ? (v[u].x - mean) * (v[u].x - mean) : 0.f;
    } else {
      for (int j0 = 0; j0 < per; j0 += 8) {
        float mg[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) mg[u] = (j0 + u < per) ? sp[(long)(j0 + u) * 32] : mean;
#pragma unroll
        for (int u = 0; u < 8; ++u) dv += (mg[u] - mean) * (mg[u] - mean);
      }
    }
    m2 += 32.f * dv;
    m2 += __shfl_xor(m2, 1, 64); m2 += __shfl_xor(m2, 2, 64); m2 += __shfl_xor(m2, 4, 64);
    if (part == 0) {
      mom[which][r][0] = mean;
      mom[which][r][1] = rsqrtf(m2 / (float)d_model + eps);
    }
  }
  __syncthreads();
  MKT(9);
#pragma unroll
  for (int u = 0; u < PIECES; ++u) {
    const int idx = tid + u * NT;
    if (idx < T * hd4) {
      const int t = idx / hd4, dd = (idx - t * hd4) * 4;
      float4 q = qr[u], k = kr[u];
      if (qk_ln) {
        const float mq = mom[0][t][0], rq = mom[0][t][1], mk = mom[1][t][0], rk = mom[1][t][1];
        q.x = (q.x - mq) * rq * gq[u].x; q.y = (q.y - mq) * rq * gq[u].y; q.z = (q.z - mq) * rq * gq[u].z; q.w = (q.w - mq) * rq * gq[u].w;
        k.x = (k.x - mk) * rk * gk[u].x; k.y = (k.y - mk) * rk * gk[u].y; k.z = (k.z - mk) * rk * gk[u].z; k.w = (k.w - mk) * rk * gk[u].w;
      }
      *reinterpret_cast<float4*>(&qs[t][dd]) = q;
      *reinterpret_cast<float4*>(&ks[t][dd]) = k;
      *reinterpret_cast<float4*>(&vs[t][dd]) = vr[u];
    }
  }
  __syncthreads();
  MKT(10);
  const float sc = rsqrtf((float)hd);
  const float slope = exp2f(-alibi_slope_base * (float)(h + 1) / (float)n_heads);
  const int j = lane >> 2, part = lane & 3;                    // key of this lane, quarter of the head dimension
  auto rows = [&](auto hd_tag) {
    constexpr int HDC = decltype(hd_tag)::value;                // 128: unrolled, unconditional LDS reads (stale rows are selected away); 0: any hd
    const int qd = HDC ? HDC / 4 : hd >> 2;
    const int d0 = part * qd;
    for (int i = wave; i < T; i += NT / 64) {
      float a = 0.f;
      if (HDC) {
        float4 qv[HDC ? HDC / 16 : 1], kv[HDC ? HDC / 16 : 1];
#pragma unroll
        for (int u = 0; u < HDC / 16; ++u) {
          qv[u] = *reinterpret_cast<const float4*>(&qs[i][d0 + 4 * u]);
          kv[u] = *reinterpret_cast<const float4*>(&ks[j][d0 + 4 * u]);
        }
#pragma unroll
        for (int u = 0; u < HDC / 16; ++u) a += qv[u].x * kv[u].x + qv[u].y * kv[u].y + qv[u].z * kv[u].z + qv[u].w * kv[u].w;
      } else if (j <= i) {
        if ((qd & 3) == 0) {
          for (int dd = 0; dd < qd; dd += 4) {
            const float4 qv = *reinterpret_cast<const float4*>(&qs[i][d0 + dd]);
            const float4 kv = *reinterpret_cast<const float4*>(&ks[j][d0 + dd]);
            a += qv.x * kv.x + qv.y * kv.y + qv.z * kv.z + qv.w * kv.w;
          }
        } else {
          for (int dd = 0; dd < qd; ++dd) a += qs[i][d0 + dd] * ks[j][d0 + dd];
        }
      }
      if (j > i) a = 0.f;                                         // (stale LDS rows of the unrolled form)
      a += __shfl_xor(a, 1, 64);
      a += __shfl_xor(a, 2, 64);
      a = a * sc - (float)(T - 1 - j) * slope;
      if (j > i || (key_mask != nullptr && key_mask[min(j, T - 1)] == 0)) a = -INFINITY;
      float mx = a;
      mx = fmaxf(mx, __shfl_xor(mx, 4, 64)); mx = fmaxf(mx, __shfl_xor(mx, 8, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 16, 64)); mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      float pr = (j <= i) ? expf(a - mx) : 0.f;
      float sum = pr;
      sum += __shfl_xor(sum, 4, 64); sum += __shfl_xor(sum, 8, 64); sum += __shfl_xor(sum, 16, 64); sum += __shfl_xor(sum, 32, 64);
      pr = pr / sum;
      // P V: lane owns output columns 2 * lane, 2 * lane + 1 of the head; p_j from lane 4 j
      const int dd = 2 * lane;
      const bool col_ok = HDC ? true : dd < hd;
      float o0 = 0.f, o1 = 0.f;
      float2 vv[TM_MAXT];
#pragma unroll
      for (int jj = 0; jj < TM_MAXT; ++jj)
        vv[jj] = (HDC || (jj <= i && col_ok)) ? *reinterpret_cast<const float2*>(&vs[jj][col_ok ? dd : 0]) : float2{0.f, 0.f};
#pragma unroll
      for (int jj = 0; jj < TM_MAXT; ++jj) {
        const float pj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, pr), jj * 4));
        o0 += (jj <= i) ? pj * vv[jj].x : 0.f;
        o1 += (jj <= i) ? pj * vv[jj].y : 0.f;
      }
      if (col_ok) {
        uint32_t hi2, lo2;
        split2<F16>(o0, o1, hi2, lo2);
        *reinterpret_cast<uint32_t*>(out_hi + (long)i * ldo + h * hd + dd) = hi2;
        *reinterpret_cast<uint32_t*>(out_lo + (long)i * ldo + h * hd + dd) = lo2;
      }
    }
  };
  if (hd == 128) rows(std::integral_constant<int, 128>{});
  else rows(std::integral_constant<int, 0>{});
  MKT(13);
}

#ifndef DEER_BODIES_ONLY
template <bool F16>
__global__ __launch_bounds__(1024) void trunk_mpt_attn_kernel(const float* __restrict__ qkv, const float* __restrict__ stats, int d_model, int hd,
                                                             const float* __restrict__ q_ln_w, const float* __restrict__ k_ln_w, float eps,
                                                             const unsigned char* __restrict__ key_mask, float alibi_slope_base, int n_heads,
                                                             bf16_t* __restrict__ out_hi, bf16_t* __restrict__ out_lo, int ldo, int T, const int* ctl) {
  DEER_RETURN_IF_EXITED(ctl);
  __shared__ __attribute__((aligned(16))) float lds[TM_LDS_FLOATS];
  trunk_mpt_attn_body<1024, F16>(qkv, stats, d_model, hd, q_ln_w, k_ln_w, eps, key_mask, alibi_slope_base, n_heads, out_hi, out_lo, ldo, T, blockIdx.x, lds);
}

template <bool F16>
static int trunk_mpt_attn(const float* qkv, const float* stats, int d_model, int n_heads, const float* q_ln_w, const float* k_ln_w, float eps,
                                   const unsigned char* key_mask, float alibi_bias_max, void* out_hi, void* out_lo, int ldo, int T, const int* ctl,
                                   void* stream) {
  const int hd = d_model / n_heads;
  if (qkv == nullptr || T <= 0 || T > TM_MAXT || hd > 128 || (hd & 3) || hd * n_heads != d_model || (d_model & 255) || out_hi == nullptr ||
      out_lo == nullptr)
    return DEER_ERR_SHAPE;
  if ((q_ln_w == nullptr) != (k_ln_w == nullptr) || (q_ln_w != nullptr && stats == nullptr)) return DEER_ERR_SHAPE;
  hipLaunchKernelGGL(trunk_mpt_attn_kernel<F16>, dim3(n_heads), dim3(1024), 0, reinterpret_cast<hipStream_t>(stream), qkv, stats, d_model, hd, q_ln_w, k_ln_w,
                     eps, key_mask, alibi_bias_max, n_heads, reinterpret_cast<bf16_t*>(out_hi), reinterpret_cast<bf16_t*>(out_lo), ldo, T, ctl);
  DEER_LAUNCH_CHECK();
  return DEER_OK;
}
extern "C" int deer_trunk_mpt_attn(const float* qkv, const float* stats, int d_model, int n_heads, const float* q_ln_w, const float* k_ln_w, float eps,
                                   const unsigned char* key_mask, float alibi_bias_max, void* out_hi, void* out_lo, int ldo, int T, const int* ctl,
                                   void* stream) {
  return trunk_mpt_attn<false>(qkv, stats, d_model, n_heads, q_ln_w, k_ln_w, eps, key_mask, alibi_bias_max, out_hi, out_lo, ldo, T, ctl, stream);
}
extern "C" int deer_trunk_mpt_attn_f16(const float* qkv, const float* stats, int d_model, int n_heads, const float* q_ln_w, const float* k_ln_w, float eps,
                                       const unsigned char* key_mask, float alibi_bias_max, void* out_hi, void* out_lo, int ldo, int T, const int* ctl,
                                       void* stream) {   // fp16 hi / lo output planes (round 6)
  return trunk_mpt_attn<true>(qkv, stats, d_model, n_heads, q_ln_w, k_ln_w, eps, key_mask, alibi_bias_max, out_hi, out_lo, ldo, T, ctl, stream);
}
#endif  // DEER_BODIES_ONLY
