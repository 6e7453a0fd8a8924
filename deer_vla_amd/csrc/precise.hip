// fp32 arithmetic for the vision tower and the gated cross-attention (deer_config.precision = 1).
//
// The default path rounds activations to bf16 wherever they enter an MFMA in the ViT, the Perceiver, the media K/V projection
// and the x-attn core (1.2-1.6e-3 on the action against the fp32 oracle, half of it from the vision tower, half
// from the bf16 media tokens / K/V / x-attn operands).  This file is the second arithmetic north_star's "1e-3 fp32" clause asks
// for: every activation stays fp32 from the image to the action -
//   * deer_gemm_f32_nt   C[M,N] f32 (op)= A[M,K] f32 * W[N,K]^T + bias with the exact-f32 MFMA (v_mfma_f32_16x16x4_f32, 1/16 of
//                        the bf16 rate); in this arithmetic the arena keeps the weights of these GEMMs in f32;
//   * deer_attn_f32      softmax(q k^T * scale) v, head_dim 64, keys in one or two segments (Perceiver: [media ; latents]),
//                        all fp32 VALU, K then V of a head staged in LDS;
//   * deer_xattn_f32     the gated x-attn core (media-time mask, helpers.py:192-232) with fp32 K/V.
// The LLM trunk multiplies fp32 activations as bf16 hi + lo pairs with hi + lo weight planes (model.hip::skinny), the embedding table and
// the head's weights are f32 in this mode, the MPT attention, the LSTM state and the exit gate are fp32 anyway.  Throughput is not the point of this mode (ViT-L at ~1/10 of the bf16 path); parity is.
#include "common.h"

enum { PE_F32 = 0, PE_QGELU = 1, PE_GELU = 2, PE_RESADD = 3 };

#define PG_BK 32
#define PG_PITCH 36          // floats per LDS row: 32 + 4 (keeps float4 alignment; (row*36 + g) mod 64 is conflict-free for 16 rows x 4 k)

__global__ __launch_bounds__(256) void gemm_f32_kernel(const float* __restrict__ A, int lda, const float* __restrict__ W, int ldw,
                                                       const float* __restrict__ bias, float* __restrict__ C, int ldc, int M, int N,
                                                       int K, int epi) {
  __shared__ __attribute__((aligned(16))) float As[2][64][PG_PITCH];
  __shared__ __attribute__((aligned(16))) float Ws[2][64][PG_PITCH];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int c = lane & 15, g = lane >> 4;
  const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;

  f32x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  float4 ra[2], rw[2];
  auto gload = [&](int k0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int id = tid + i * 256, row = id >> 3, seg = id & 7;
      const int k = k0 + seg * 4;
      ra[i] = (m0 + row < M && k < K) ? *reinterpret_cast<const float4*>(A + (long)(m0 + row) * lda + k) : float4{0.f, 0.f, 0.f, 0.f};
      rw[i] = (n0 + row < N && k < K) ? *reinterpret_cast<const float4*>(W + (long)(n0 + row) * ldw + k) : float4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto swrite = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int id = tid + i * 256, row = id >> 3, seg = id & 7;
      *reinterpret_cast<float4*>(&As[buf][row][seg * 4]) = ra[i];
      *reinterpret_cast<float4*>(&Ws[buf][row][seg * 4]) = rw[i];
    }
  };

  const int nk = (K + PG_BK - 1) / PG_BK;
  gload(0);
  swrite(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) gload((kt + 1) * PG_BK);
#pragma unroll
    for (int kk = 0; kk < PG_BK / 4; ++kk) {
      float af[2], wf[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) af[j] = As[buf][wm * 32 + j * 16 + c][kk * 4 + g];
#pragma unroll
      for (int i = 0; i < 2; ++i) wf[i] = Ws[buf][wn * 32 + i * 16 + c][kk * 4 + g];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i], af[j], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) swrite(buf ^ 1);
    __syncthreads();
  }

  // lane holds C[m = .. + c][n = .. + g*4 .. +3] (W fragment as the MFMA "A" operand, like the bf16 kernels)
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int n = n0 + wn * 32 + i * 16 + g * 4;
    if (n >= N) continue;
    float4 bv = float4{0.f, 0.f, 0.f, 0.f};
    if (bias != nullptr) bv = *reinterpret_cast<const float4*>(bias + n);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int m = m0 + wm * 32 + j * 16 + c;
      if (m >= M) continue;
      float v0 = acc[i][j][0] + bv.x, v1 = acc[i][j][1] + bv.y, v2 = acc[i][j][2] + bv.z, v3 = acc[i][j][3] + bv.w;
      float4* p = reinterpret_cast<float4*>(C + (long)m * ldc + n);
      if (epi == PE_RESADD) {
        float4 r = *p;
        r.x += v0; r.y += v1; r.z += v2; r.w += v3;
        *p = r;
      } else {
        if (epi == PE_QGELU) { v0 = quick_gelu(v0); v1 = quick_gelu(v1); v2 = quick_gelu(v2); v3 = quick_gelu(v3); }
        else if (epi == PE_GELU) { v0 = gelu_erf(v0); v1 = gelu_erf(v1); v2 = gelu_erf(v2); v3 = gelu_erf(v3); }
        *p = float4{v0, v1, v2, v3};
      }
    }
  }
}

// A f32 [M,K] (lda), W f32 [N,K] (ldw, nn.Linear layout), bias f32[N] or NULL, C f32 [M,N] (ldc).
// epi: 0 store, 1 QuickGELU, 2 exact GELU, 3 C += (A W^T + bias).  K % 4 == 0, N % 4 == 0.
extern "C" int deer_gemm_f32_nt(const float* A, int lda, const float* W, int ldw, const float* bias, float* C, int ldc, int M, int N, int K,
                                int epi, void* stream) {
  if (A == nullptr || W == nullptr || C == nullptr || M <= 0 || N <= 0 || K <= 0 || (K & 3) || (N & 3) || (lda & 3) || (ldw & 3) || (ldc & 3) ||
      epi < 0 || epi > 3)
    return DEER_ERR_SHAPE;
  dim3 grid((N + 63) / 64, (M + 63) / 64);
  hipLaunchKernelGGL(gemm_f32_kernel, grid, dim3(256), 0, reinterpret_cast<hipStream_t>(stream), A, lda, W, ldw,
                     bias, C, ldc, M, N, K, epi);
  DEER_LAUNCH_CHECK();
  return DEER_OK;
}

// ---- fp32 attention, head_dim 64 ----------------------------------------------------------------------------------------
// One workgroup = PA_QB queries of one (head, batch).  Phase 1: K of the head in LDS (pitch 65: lane j reads row j, conflict-
// free), scores lane-parallel over keys, softmax per query inside its wave, P to LDS.  Phase 2: V takes K's place, lane = d.
#define PA_QB 32
#define PA_MAXKV 352
__global__ __launch_bounds__(256) void attn_f32_kernel(const float* __restrict__ Q, const float* __restrict__ K1, const float* __restrict__ V1,
                                                       const float* __restrict__ K2, const float* __restrict__ V2, float* __restrict__ O,
                                                       int q_len, int kv1, int kv2, int ldq, int ld1, int ld2, int ldo, long q_bstride,
                                                       long bstride1, long bstride2, long o_bstride, float scale) {
  extern __shared__ __attribute__((aligned(16))) float pa_lds[];
  const int kv = kv1 + kv2;
  float* kvs = pa_lds;                                   // [kv][65] (K), then [kv][64] (V)
  float* qs = kvs + (long)kv * 65;                       // [PA_QB][64]
  float* ps = qs + PA_QB * 64;                           // [PA_QB][kv]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.y, b = blockIdx.z, q0 = blockIdx.x * PA_QB;
  const int nq = min(PA_QB, q_len - q0);
  Q += b * q_bstride + (long)q0 * ldq + h * 64;
  K1 += b * bstride1 + h * 64; V1 += b * bstride1 + h * 64;
  if (kv2 > 0) { K2 += b * bstride2 + h * 64; V2 += b * bstride2 + h * 64; }
  for (int idx = tid; idx < kv * 16; idx += 256) {       // K rows, float4 pieces
    const int j = idx >> 4, d4 = (idx & 15) * 4;
    const float4 v = *reinterpret_cast<const float4*>(j < kv1 ? K1 + (long)j * ld1 + d4 : K2 + (long)(j - kv1) * ld2 + d4);
    float* dst = kvs + j * 65 + d4;
    dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
  }
  for (int idx = tid; idx < nq * 16; idx += 256) {
    const int t = idx >> 4, d4 = (idx & 15) * 4;
    const float4 v = *reinterpret_cast<const float4*>(Q + (long)t * ldq + d4);
    *reinterpret_cast<float4*>(qs + t * 64 + d4) = float4{v.x * scale, v.y * scale, v.z * scale, v.w * scale};
  }
  __syncthreads();
  const int ngrp = (kv + 63) >> 6;                        // <= 6 key groups of 64
  for (int t = wave; t < nq; t += 4) {
    float s[6];
    float mx = -INFINITY;
#pragma unroll
    for (int gi = 0; gi < 6; ++gi) {
      s[gi] = -INFINITY;
      const int j = gi * 64 + lane;
      if (gi < ngrp && j < kv) {
        const float* kr = kvs + j * 65;
        const float* qr = qs + t * 64;
        float a = 0.f;
#pragma unroll 16
        for (int d = 0; d < 64; ++d) a += qr[d] * kr[d];
        s[gi] = a;
        mx = fmaxf(mx, a);
      }
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int gi = 0; gi < 6; ++gi) {
      const int j = gi * 64 + lane;
      if (gi < ngrp && j < kv) {
        s[gi] = expf(s[gi] - mx);
        sum += s[gi];
      }
    }
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
#pragma unroll
    for (int gi = 0; gi < 6; ++gi) {
      const int j = gi * 64 + lane;
      if (gi < ngrp && j < kv) ps[t * kv + j] = s[gi] * inv;
    }
  }
  __syncthreads();
  for (int idx = tid; idx < kv * 16; idx += 256) {       // V rows over the K image
    const int j = idx >> 4, d4 = (idx & 15) * 4;
    const float4 v = *reinterpret_cast<const float4*>(j < kv1 ? V1 + (long)j * ld1 + d4 : V2 + (long)(j - kv1) * ld2 + d4);
    *reinterpret_cast<float4*>(kvs + j * 64 + d4) = v;
  }
  __syncthreads();
  for (int t = wave; t < nq; t += 4) {
    const float* pr = ps + t * kv;
    float a = 0.f;
    for (int j = 0; j < kv; ++j) a += pr[j] * kvs[j * 64 + lane];
    O[b * o_bstride + (long)(q0 + t) * ldo + h * 64 + lane] = a;
  }
}

// The same for key sequences beyond PA_MAXKV (pre fusion at full size, flamingo_mpt.py:585-607: 2 x 256 patch tokens + 64 latents = 576
// keys): K and V pass through LDS in chunks of PAL_CH keys, the raw scores of all keys wait in `ps`, the softmax runs once over the row.
#define PAL_CH 256
#define PAL_MAXKV 1024
__global__ __launch_bounds__(256) void attn_f32_long_kernel(const float* __restrict__ Q, const float* __restrict__ K1, const float* __restrict__ V1,
                                                            const float* __restrict__ K2, const float* __restrict__ V2, float* __restrict__ O,
                                                            int q_len, int kv1, int kv2, int ldq, int ld1, int ld2, int ldo, long q_bstride,
                                                            long bstride1, long bstride2, long o_bstride, float scale) {
  extern __shared__ __attribute__((aligned(16))) float pa_lds[];
  const int kv = kv1 + kv2;
  float* kvs = pa_lds;                                   // [PAL_CH][65] (a K chunk), then [PAL_CH][64] (a V chunk)
  float* qs = kvs + PAL_CH * 65;                         // [PA_QB][64]
  float* ps = qs + PA_QB * 64;                           // [PA_QB][kv]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int h = blockIdx.y, b = blockIdx.z, q0 = blockIdx.x * PA_QB;
  const int nq = min(PA_QB, q_len - q0);
  Q += b * q_bstride + (long)q0 * ldq + h * 64;
  K1 += b * bstride1 + h * 64; V1 += b * bstride1 + h * 64;
  if (kv2 > 0) { K2 += b * bstride2 + h * 64; V2 += b * bstride2 + h * 64; }
  for (int idx = tid; idx < nq * 16; idx += 256) {
    const int t = idx >> 4, d4 = (idx & 15) * 4;
    const float4 v = *reinterpret_cast<const float4*>(Q + (long)t * ldq + d4);
    *reinterpret_cast<float4*>(qs + t * 64 + d4) = float4{v.x * scale, v.y * scale, v.z * scale, v.w * scale};
  }
  for (int c0 = 0; c0 < kv; c0 += PAL_CH) {              // raw scores, chunk by chunk
    const int cn = min(PAL_CH, kv - c0);
    __syncthreads();
    for (int idx = tid; idx < cn * 16; idx += 256) {
      const int j = c0 + (idx >> 4), d4 = (idx & 15) * 4;
      const float4 v = *reinterpret_cast<const float4*>(j < kv1 ? K1 + (long)j * ld1 + d4 : K2 + (long)(j - kv1) * ld2 + d4);
      float* dst = kvs + (j - c0) * 65 + d4;
      dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
    }
    __syncthreads();
    for (int t = wave; t < nq; t += 4) {
      const float* qr = qs + t * 64;
      for (int jj = lane; jj < cn; jj += 64) {
        const float* kr = kvs + jj * 65;
        float a = 0.f;
#pragma unroll 16
        for (int d = 0; d < 64; ++d) a += qr[d] * kr[d];
        ps[t * kv + c0 + jj] = a;
      }
    }
  }
  __syncthreads();
  for (int t = wave; t < nq; t += 4) {                   // softmax of a query row inside its wave
    float* pr = ps + t * kv;
    float mx = -INFINITY;
    for (int j = lane; j < kv; j += 64) mx = fmaxf(mx, pr[j]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < kv; j += 64) {
      const float e = expf(pr[j] - mx);
      pr[j] = e;
      sum += e;
    }
    sum = wave_sum(sum);
    const float inv = 1.f / sum;
    for (int j = lane; j < kv; j += 64) pr[j] *= inv;
  }
  float acc[PA_QB / 4];
#pragma unroll
  for (int i = 0; i < PA_QB / 4; ++i) acc[i] = 0.f;
  for (int c0 = 0; c0 < kv; c0 += PAL_CH) {              // O = P V, chunk by chunk; lane = d
    const int cn = min(PAL_CH, kv - c0);
    __syncthreads();
    for (int idx = tid; idx < cn * 16; idx += 256) {
      const int j = c0 + (idx >> 4), d4 = (idx & 15) * 4;
      const float4 v = *reinterpret_cast<const float4*>(j < kv1 ? V1 + (long)j * ld1 + d4 : V2 + (long)(j - kv1) * ld2 + d4);
      *reinterpret_cast<float4*>(kvs + (j - c0) * 64 + d4) = v;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < PA_QB / 4; ++i) {
      const int t = wave + 4 * i;
      if (t < nq) {
        const float* pr = ps + t * kv + c0;
        float a = acc[i];
        for (int j = 0; j < cn; ++j) a += pr[j] * kvs[j * 64 + lane];
        acc[i] = a;
      }
    }
  }
#pragma unroll
  for (int i = 0; i < PA_QB / 4; ++i) {
    const int t = wave + 4 * i;
    if (t < nq) O[b * o_bstride + (long)(q0 + t) * ldo + h * 64 + lane] = acc[i];
  }
}

// q f32 [batch][q_len][ldq] (head h at column h*64), keys/values f32 in one or two segments (segment s: [batch][kv_s][ld_s], K and V
// pointers already at their column offset), out f32 [batch][q_len][ldo].  open_clip MHA / PerceiverAttention (helpers.py:47-73).
extern "C" int deer_attn_f32(const float* Q, const float* K1, const float* V1, const float* K2, const float* V2, float* O, int batch, int heads,
                             int q_len, int kv1, int kv2, int ldq, int ld1, int ld2, int ldo, long q_bstride, long bstride1, long bstride2,
                             long o_bstride, float scale, void* stream) {
  if (Q == nullptr || K1 == nullptr || V1 == nullptr || O == nullptr || batch <= 0 || heads <= 0 || q_len <= 0 || kv1 <= 0 || kv2 < 0 ||
      kv1 + kv2 > PAL_MAXKV || (kv2 > 0 && (K2 == nullptr || V2 == nullptr)) || (ldq & 3) || (ld1 & 3) || (ld2 & 3))
    return DEER_ERR_SHAPE;
  const int kv = kv1 + kv2;
  if (kv > PA_MAXKV) {                                    // long key sequences: chunked K / V
    const int smem_l = (PAL_CH * 65 + PA_QB * 64 + PA_QB * kv) * (int)sizeof(float);
    static std::atomic<bool> attr_l{false};
    if (!attr_l) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_f32_long_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 156 * 1024) != hipSuccess)
        return DEER_ERR_LAUNCH;
      attr_l = true;
    }
    if (smem_l > 156 * 1024) return DEER_ERR_SHAPE;
    hipLaunchKernelGGL(attn_f32_long_kernel, dim3((q_len + PA_QB - 1) / PA_QB, heads, batch), dim3(256), smem_l, reinterpret_cast<hipStream_t>(stream), Q,
                       K1, V1, K2, V2, O, q_len, kv1, kv2, ldq, ld1, ld2, ldo, q_bstride, bstride1, bstride2, o_bstride, scale);
    DEER_LAUNCH_CHECK();
    return DEER_OK;
  }
  const int smem = (kv * 65 + PA_QB * 64 + PA_QB * kv) * (int)sizeof(float);
  static std::atomic<bool> attr_set{false};
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_f32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess)
      return DEER_ERR_LAUNCH;
    attr_set = true;
  }
  if (smem > 150 * 1024) return DEER_ERR_SHAPE;
  hipLaunchKernelGGL(attn_f32_kernel, dim3((q_len + PA_QB - 1) / PA_QB, heads, batch), dim3(256), smem, reinterpret_cast<hipStream_t>(stream), Q,
                     K1, V1, K2, V2, O, q_len, kv1, kv2, ldq, ld1, ld2, ldo, q_bstride, bstride1, bstride2, o_bstride, scale);
  DEER_LAUNCH_CHECK();
  return DEER_OK;
}

// ---- gated x-attn core with fp32 K/V (the fp32 twin deer_xattn_small keeps bf16 K/V) ---------------------------------------------
#define PX_MAXT 32
#define PX_MAXKV 128
__global__ __launch_bounds__(256) void xattn_f32_kernel(const float* __restrict__ qslab, int s_in, long slab_stride, int ldqs,
                                                        const float* __restrict__ kv, int ldkv, int inner, const int* __restrict__ text_time,
                                                        int n_per_media, float* __restrict__ out, int ldo, int T, int n_kv, float scale,
                                                        const int* ctl) {
  DEER_RETURN_IF_EXITED(ctl);
  extern __shared__ __attribute__((aligned(16))) float px_lds[];
  float* qs = px_lds;                                    // [T][65]
  float* sim = qs + PX_MAXT * 65;                        // [T][n_kv + 1]
  float* ks = sim + PX_MAXT * (PX_MAXKV + 1);            // [n_kv][65]
  float* vs = ks + PX_MAXKV * 65;                        // [n_kv][64]
  const int h = blockIdx.x, tid = threadIdx.x;
  qslab += (long)blockIdx.y * T * ldqs;
  kv += (long)blockIdx.y * n_kv * ldkv;
  text_time += blockIdx.y * T;
  out += (long)blockIdx.y * T * ldo;
  for (int idx = tid; idx < T * 16; idx += 256) {
    const int t = idx >> 4, d4 = (idx & 15) * 4;
    const float4 a = slab_sum4(qslab + (long)t * ldqs + h * 64 + d4, s_in, slab_stride);
    float* dst = qs + t * 65 + d4;
    dst[0] = a.x * scale; dst[1] = a.y * scale; dst[2] = a.z * scale; dst[3] = a.w * scale;
  }
  for (int idx = tid; idx < n_kv * 32; idx += 256) {
    const int j = idx >> 5, part = idx & 31, isv = part >> 4, d4 = (part & 15) * 4;
    const float4 v = *reinterpret_cast<const float4*>(kv + (long)j * ldkv + (isv ? inner : 0) + h * 64 + d4);
    if (isv) *reinterpret_cast<float4*>(vs + j * 64 + d4) = v;
    else { float* dst = ks + j * 65 + d4; dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w; }
  }
  __syncthreads();
  for (int idx = tid; idx < T * n_kv; idx += 256) {
    const int t = idx / n_kv, j = idx - t * n_kv;
    float a = 0.f;
#pragma unroll 16
    for (int d = 0; d < 64; ++d) a += qs[t * 65 + d] * ks[j * 65 + d];
    if (text_time[t] != j / n_per_media + 1) a = -3.4028234663852886e38f;      // -finfo.max (helpers.py:218)
    sim[t * (PX_MAXKV + 1) + j] = a;
  }
  __syncthreads();
  const int lane = tid & 63, wave = tid >> 6;
  for (int t = wave; t < T; t += 4) {
    float* sr = sim + t * (PX_MAXKV + 1);
    float mx = -INFINITY;
    for (int j = lane; j < n_kv; j += 64) mx = fmaxf(mx, sr[j]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < n_kv; j += 64) {
      const float p = expf(sr[j] - mx);
      sr[j] = p;
      sum += p;
    }
    sum = wave_sum(sum);
    const float inv = (text_time[t] == 0) ? 0.f : 1.f / sum;                    // helpers.py:223-229
    for (int j = lane; j < n_kv; j += 64) sr[j] *= inv;
  }
  __syncthreads();
  for (int idx = tid; idx < T * 64; idx += 256) {
    const int t = idx >> 6, d = idx & 63;
    const float* sr = sim + t * (PX_MAXKV + 1);
    float a = 0.f;
    for (int j = 0; j < n_kv; ++j) a += sr[j] * vs[j * 64 + d];
    out[(long)t * ldo + h * 64 + d] = a;
  }
}

// q from split-K slabs (x scale), kv f32 [batch][n_kv][ldkv] (k at column h*64, v at column inner + h*64), out f32 [batch][T][ldo]
extern "C" int deer_xattn_f32(const float* qslab, int s_in, long slab_stride, int ldqs, const float* kv, int ldkv, int inner,
                              const int* text_time, int n_per_media, float* out, int ldo, int T, int n_kv, int heads, int batch, float scale,
                              const int* ctl, void* stream) {
  if (T <= 0 || T > PX_MAXT || n_kv <= 0 || n_kv > PX_MAXKV || s_in <= 0 || n_per_media <= 0 || batch <= 0 || (ldkv & 3) || (ldqs & 3))
    return DEER_ERR_SHAPE;
  const int smem = (PX_MAXT * 65 + PX_MAXT * (PX_MAXKV + 1) + PX_MAXKV * 65 + PX_MAXKV * 64) * (int)sizeof(float);
  static std::atomic<bool> attr_set{false};
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&xattn_f32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess)
      return DEER_ERR_LAUNCH;
    attr_set = true;
  }
  hipLaunchKernelGGL(xattn_f32_kernel, dim3(heads, batch), dim3(256), smem, reinterpret_cast<hipStream_t>(stream), qslab, s_in, slab_stride, ldqs,
                     kv, ldkv, inner, text_time, n_per_media, out, ldo, T, n_kv, scale, ctl);
  DEER_LAUNCH_CHECK();
  return DEER_OK;
}

// ---- ViT patch embedding, step 1 in f32: im2col of p x p / p patches -> f32 [N*P, Kpad] (k = ch*p*p + py*p + px, zero padded) -----
__global__ void im2col_f32_kernel(const float* __restrict__ img, int N, int S, int p, int gw, float* __restrict__ out, int Kpad) {
  const long total = (long)N * gw * gw * Kpad;
  const int kk = 3 * p * p;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int k = (int)(idx % Kpad);
    const long row = idx / Kpad;
    float v = 0.f;
    if (k < kk) {
      const int P = gw * gw;
      const int n = (int)(row / P), pi = (int)(row - (long)n * P);
      const int gy = pi / gw, gx = pi - gy * gw;
      const int ch = k / (p * p), rem = k - ch * p * p, py = rem / p, px = rem - py * p;
      v = img[(((long)n * 3 + ch) * S + (gy * p + py)) * S + gx * p + px];
    }
    out[idx] = v;
  }
}

extern "C" int deer_vit_im2col_f32(const float* img, int N, int S, int patch, float* out, int Kpad, void* stream) {
  if (img == nullptr || out == nullptr || N <= 0 || S <= 0 || patch <= 0 || S % patch != 0 || Kpad < 3 * patch * patch) return DEER_ERR_SHAPE;
  const int gw = S / patch;
  const long total = (long)N * gw * gw * Kpad;
  const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(im2col_f32_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), img, N, S, patch, gw, out, Kpad);
  DEER_LAUNCH_CHECK();
  return DEER_OK;
}

// ---- token embedding from an f32 table (deer_embed_tokens keeps bf16) + media bookkeeping ---------------------------------
__global__ __launch_bounds__(256) void embed_tokens_f32_kernel(const long long* __restrict__ ids, const float* __restrict__ wte, float* __restrict__ x,
                                                               int* __restrict__ text_time, int T, int d, int vocab, int media_id) {
  const int row = blockIdx.x, t = row % T, e0 = row - t;   // rows are [env][T]; the media count restarts per environment
  long long id = ids[row];
  if (id < 0) id = 0;
  if (id >= vocab) id = vocab - 1;
  for (int i = threadIdx.x; i < d; i += 256) x[(long)row * d + i] = wte[id * d + i];
  if (threadIdx.x == 0) {
    int c = 0;
    for (int j = 0; j <= t; ++j) c += (ids[e0 + j] == media_id) ? 1 : 0;
    text_time[row] = c;
  }
}

extern "C" int deer_embed_tokens_f32(const long long* ids, const float* wte, float* x, int* text_time, int T, int batch, int d, int vocab,
                                     int media_id, void* stream) {
  if (T <= 0 || batch <= 0 || d <= 0 || vocab <= 0) return DEER_ERR_SHAPE;
  hipLaunchKernelGGL(embed_tokens_f32_kernel, dim3(T * batch), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), ids, wte, x, text_time, T, d,
                     vocab, media_id);
  DEER_LAUNCH_CHECK();
  return DEER_OK;
}
