// Row-wise kernels of the DeeR-VLA step: LayerNorm, fused "split-K reduce + gated residual + LayerNorm",
// ViT patch im2col and cls/pos/ln_pre embedding, token embedding, media-time bookkeeping.
// All statistics are fp32 two-pass (mean, then biased variance), eps 1e-5, like torch.nn.LayerNorm.
#include "common.h"
#include <cstdlib>

// ---- LayerNorm over rows: f32 in -> bf16 out (GEMM A operand) and/or f32 out -------------------------
// Row r of batch b is read at  x + b*in_bstride + r*in_rstride  and written at  out + b*out_bstride + r*out_rstride
// (lets the Perceiver write LN_media(x) and LN_latents(latents) into one [x; latents] buffer, helpers.py:51).
// one workgroup per row; the row lives in registers (float4 per thread), two block reductions.
template <bool F16>
__global__ __launch_bounds__(256) void ln_rows_kernel(const float* __restrict__ x, long in_rstride, long in_bstride,
                                                      int rows_per_batch, int total_rows, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, bf16_t* __restrict__ out_bf,
                                                      float* __restrict__ out_f32, long out_rstride, long out_bstride,
                                                      int C, float eps, int n_sets, long param_stride, long out_set_stride) {
  __shared__ float red[16];
  const int row = blockIdx.x;
  const int b = row / rows_per_batch, r = row - b * rows_per_batch;
  const float* xr = x + b * in_bstride + r * in_rstride;
  const int n4 = C >> 2;
  float4 v[4];                                             // C <= 4096
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i4 = threadIdx.x + j * 256;
    v[j] = float4{0.f, 0.f, 0.f, 0.f};
    if (i4 < n4) {
      v[j] = *reinterpret_cast<const float4*>(xr + (long)i4 * 4);
      s += v[j].x + v[j].y + v[j].z + v[j].w;
    }
  }
  const float mean = block_sum(s, red) / C;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j)
    if (threadIdx.x + j * 256 < n4) {
      const float a = v[j].x - mean, bb = v[j].y - mean, c = v[j].z - mean, d = v[j].w - mean;
      q += a * a + bb * bb + c * c + d * d;
    }
  const float rstd = rsqrtf(block_sum(q, red) / C + eps);
  // n_sets > 1: the same normalised row under several (gamma, beta) pairs - statistics once (Perceiver norm_media of all layers)
  for (int ps = 0; ps < n_sets; ++ps) {
    const long o = ps * out_set_stride + b * out_bstride + r * out_rstride;
    const float* gm = gamma + ps * param_stride;
    const float* bt_ = beta != nullptr ? beta + ps * param_stride : nullptr;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i4 = threadIdx.x + j * 256;
      if (i4 < n4) {
        const float4 g = *reinterpret_cast<const float4*>(gm + (long)i4 * 4);
        float4 y;
        y.x = (v[j].x - mean) * rstd * g.x; y.y = (v[j].y - mean) * rstd * g.y;
        y.z = (v[j].z - mean) * rstd * g.z; y.w = (v[j].w - mean) * rstd * g.w;
        if (bt_ != nullptr) {
          const float4 bt = *reinterpret_cast<const float4*>(bt_ + (long)i4 * 4);
          y.x += bt.x; y.y += bt.y; y.z += bt.z; y.w += bt.w;
        }
        if (out_bf != nullptr) *reinterpret_cast<uint2*>(out_bf + o + (long)i4 * 4) = uint2{pack2x<F16>(y.x, y.y), pack2x<F16>(y.z, y.w)};
        if (out_f32 != nullptr) *reinterpret_cast<float4*>(out_f32 + o + (long)i4 * 4) = y;
      }
    }
  }
}

extern "C" int deer_layernorm_rows(const float* x, long in_rstride, long in_bstride, int rows_per_batch, int batch,
                                   const float* gamma, const float* beta, void* out_bf16, float* out_f32,
                                   long out_rstride, long out_bstride, int C, float eps, void* stream) {
  if (rows_per_batch <= 0 || batch <= 0 || C <= 0 || (C & 3) || C > 4096 || gamma == nullptr ||
      (out_bf16 == nullptr && out_f32 == nullptr) || (in_rstride & 3) || (in_bstride & 3) || (out_rstride & 3) || (out_bstride & 3))
    return DEER_ERR_SHAPE;
  const int total = rows_per_batch * batch;
  hipLaunchKernelGGL(ln_rows_kernel<false>, dim3(total), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     x, in_rstride, in_bstride, rows_per_batch, total, gamma, beta, reinterpret_cast<bf16_t*>(out_bf16),
                     out_f32, out_rstride, out_bstride, C, eps, 1, 0L, 0L);
  DEER_LAUNCH_CHECK();
  return DEER_OK;
}

// One pass of statistics, n_sets affine outputs: out[s] = LN(x) * gamma[s] + beta[s] (gamma/beta rows param_stride apart,
// outputs out_set_stride apart).  The Perceiver applies a different norm_media to the SAME media tokens in each of its
// layers (helpers.py:47), so all of them are produced up front.
extern "C" int deer_layernorm_rows_multi(const float* x, long in_rstride, long in_bstride, int rows_per_batch, int batch,
                                         const float* gamma, const float* beta, int n_sets, long param_stride, void* out_bf16,
                                         long out_set_stride, long out_rstride, long out_bstride, int C, float eps,
                                         void* stream) {
  if (rows_per_batch <= 0 || batch <= 0 || C <= 0 || (C & 3) || C > 4096 || gamma == nullptr || out_bf16 == nullptr ||
      n_sets <= 0 || (param_stride & 3) || (out_set_stride & 3) || (in_rstride & 3) || (in_bstride & 3) || (out_rstride & 3) ||
      (out_bstride & 3))
    return DEER_ERR_SHAPE;
  const int total = rows_per_batch * batch;
  hipLaunchKernelGGL(ln_rows_kernel<false>, dim3(total), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     x, in_rstride, in_bstride, rows_per_batch, total, gamma, beta, reinterpret_cast<bf16_t*>(out_bf16),
                     (float*)nullptr, out_rstride, out_bstride, C, eps, n_sets, param_stride, out_set_stride);
  DEER_LAUNCH_CHECK();
  return DEER_OK;
}

// ---- the same two entry points with the 16-bit output in fp16 (round 6: the vision tower's fp16 arithmetic; the argument is still called
// out_bf16) ----
extern "C" int deer_layernorm_rows_f16(const float* x, long in_rstride, long in_bstride, int rows_per_batch, int batch,
                                   const float* gamma, const float* beta, void* out_bf16, float* out_f32,
                                   long out_rstride, long out_bstride, int C, float eps, void* stream) {
  if (rows_per_batch <= 0 || batch <= 0 || C <= 0 || (C & 3) || C > 4096 || gamma == nullptr ||
      (out_bf16 == nullptr && out_f32 == nullptr) || (in_rstride & 3) || (in_bstride & 3) || (out_rstride & 3) || (out_bstride & 3))
    return DEER_ERR_SHAPE;
  const int total = rows_per_batch * batch;
  hipLaunchKernelGGL(ln_rows_kernel<true>, dim3(total), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     x, in_rstride, in_bstride, rows_per_batch, total, gamma, beta, reinterpret_cast<bf16_t*>(out_bf16),
                     out_f32, out_rstride, out_bstride, C, eps, 1, 0L, 0L);
  DEER_LAUNCH_CHECK();
  return DEER_OK;
}

// One pass of statistics, n_sets affine outputs: out[s] = LN(x) * gamma[s] + beta[s] (gamma/beta rows param_stride apart,
// outputs out_set_stride apart).  The Perceiver applies a different norm_media to the SAME media tokens in each of its
// layers (helpers.py:47), so all of them are produced up front.
extern "C" int deer_layernorm_rows_multi_f16(const float* x, long in_rstride, long in_bstride, int rows_per_batch, int batch,
                                         const float* gamma, const float* beta, int n_sets, long param_stride, void* out_bf16,
                                         long out_set_stride, long out_rstride, long out_bstride, int C, float eps,
                                         void* stream) {
  if (rows_per_batch <= 0 || batch <= 0 || C <= 0 || (C & 3) || C > 4096 || gamma == nullptr || out_bf16 == nullptr ||
      n_sets <= 0 || (param_stride & 3) || (out_set_stride & 3) || (in_rstride & 3) || (in_bstride & 3) || (out_rstride & 3) ||
      (out_bstride & 3))
    return DEER_ERR_SHAPE;
  const int total = rows_per_batch * batch;
  hipLaunchKernelGGL(ln_rows_kernel<true>, dim3(total), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     x, in_rstride, in_bstride, rows_per_batch, total, gamma, beta, reinterpret_cast<bf16_t*>(out_bf16),
                     (float*)nullptr, out_rstride, out_bstride, C, eps, n_sets, param_stride, out_set_stride);
  DEER_LAUNCH_CHECK();
  return DEER_OK;
}

// ---- LLM row op: x += scale * sum_s slab[s] ; [copy x] ; [LN(x) -> bf16] ------------------------------
// The consumer-side half of the skinny GEMM's split-K (launch-boundary reduce) fused with the gated
// residual update (helpers.py:267-279: x + tanh(gate) * y; MPT block: x + y) and the following LayerNorm.
#include "resadd_body.h"

template <int NT, bool F16 = false>
__global__ __launch_bounds__(NT) void resadd_ln_kernel(float* __restrict__ x, const float* __restrict__ slab, int s_in,
                                                       long slab_stride, const float* __restrict__ gate,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       bf16_t* __restrict__ out_bf, float* __restrict__ out_f32,
                                                       float* __restrict__ x_copy, int d, float eps, const int* ctl,
                                                       const float* __restrict__ bias, bf16_t* __restrict__ out_lo, int packed = 0,
                                                       deer_rowmap rm = deer_rowmap{nullptr, 0, nullptr, nullptr, nullptr, 0, 0}) {
  DEER_RETURN_IF_EXITED(ctl);
  resadd_ln_body<NT, F16>(x, slab, s_in, slab_stride, gate, gamma, beta, out_bf, out_f32, x_copy, d, eps, ctl, bias, out_lo, packed, rm, blockIdx.x);
}

// ---- the same row op for the env-batch vision tower (thousands of 1024-wide rows): R rows per workgroup ----------------------------
// One row per workgroup leaves a 4112-row launch at 3.5-4.1 TB/s: a workgroup's life is load -> two block reductions -> store, and only
// its first third has bytes in flight.  With R rows per workgroup every thread requests its float4 of all R rows (x and every slab) before
// the first use, the R rows' statistics go through ONE pair of block reductions (R values per wave, leading barrier before the scratch
// is rewritten), and gamma / beta are read once.  Per row the arithmetic and its order are those of resadd_ln_body: results are
// BIT-identical to the one-row kernel (tests/test_hip_ops.py).  d == 1024 exactly (a thread owns one float4 column).
template <int R>
__device__ __forceinline__ void block_sum_rows(float (&v)[R], float* red) {   // red: R x 16 floats
#pragma unroll
  for (int r = 0; r < R; ++r) v[r] = wave_sum(v[r]);
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();                                         // the previous round's readers are done with `red`
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int r = 0; r < R; ++r) red[r * 16 + w] = v[r];
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < R; ++r) {
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[r * 16 + i];
    v[r] = t;
  }
}

template <int R, bool F16>
__global__ __launch_bounds__(256) void resadd_ln_multirow_kernel(float* __restrict__ x, const float* __restrict__ slab, int s_in, long slab_stride,
                                                                 const float* __restrict__ gate, const float* __restrict__ bias,
                                                                 const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                 bf16_t* __restrict__ out_bf, float eps, int rows) {
  constexpr int D = 1024;
  __shared__ float red[R * 16];
  const int c = threadIdx.x * 4;
  const int r0 = blockIdx.x * R;
  long off[R];
#pragma unroll
  for (int r = 0; r < R; ++r) off[r] = (long)min(r0 + r, rows - 1) * D + c;   // a ragged last workgroup repeats the last row and does not store it
  float4 a[R], v[R];
#pragma unroll
  for (int r = 0; r < R; ++r) a[r] = slab != nullptr ? slab_sum4(slab + off[r], s_in, slab_stride) : float4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int r = 0; r < R; ++r) v[r] = *reinterpret_cast<const float4*>(x + off[r]);
  const float sc = (slab != nullptr && gate != nullptr) ? tanhf(*gate) : 1.f;
  float4 bv = float4{0.f, 0.f, 0.f, 0.f};
  if (slab != nullptr && bias != nullptr) bv = *reinterpret_cast<const float4*>(bias + c);
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (slab != nullptr) {
      if (bias != nullptr) { a[r].x += bv.x; a[r].y += bv.y; a[r].z += bv.z; a[r].w += bv.w; }
      v[r].x += sc * a[r].x; v[r].y += sc * a[r].y; v[r].z += sc * a[r].z; v[r].w += sc * a[r].w;
      if (r0 + r < rows) *reinterpret_cast<float4*>(x + off[r]) = v[r];
    }
  }
  if (gamma == nullptr) return;
  const float4 g = *reinterpret_cast<const float4*>(gamma + c);          // requested before the reductions (see resadd_ln_body)
  float4 bb = float4{0.f, 0.f, 0.f, 0.f};
  if (beta != nullptr) bb = *reinterpret_cast<const float4*>(beta + c);
  float s[R];
#pragma unroll
  for (int r = 0; r < R; ++r) { s[r] = 0.f; s[r] += v[r].x + v[r].y + v[r].z + v[r].w; }
  block_sum_rows<R>(s, red);
  float mean[R], q[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    mean[r] = s[r] / D;
    q[r] = 0.f;
    q[r] += ln_sq4(v[r], mean[r]);
  }
  block_sum_rows<R>(q, red);
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (r0 + r >= rows) break;
    const float rstd = rsqrtf(q[r] / D + eps);
    float4 y = ln_norm4(v[r], mean[r], rstd, g);
    if (beta != nullptr) y = ln_add4(y, bb);
    *reinterpret_cast<uint2*>(out_bf + off[r]) = uint2{pack2x<F16>(y.x, y.y), pack2x<F16>(y.z, y.w)};
  }
}

// rows_per_wg: 2 or 4.  d must be 1024, out bf16 (fp16 when F16); no control block (the vision tower runs before the first exit check).
template <bool F16>
static int resadd_ln_multirow(float* x, const float* slab, int s_in, long slab_stride, const float* gate, const float* bias,
                              const float* gamma, const float* beta, void* out_bf16, int T, int d, float eps, int rows_per_wg, void* stream) {
  if (T <= 0 || d != 1024 || x == nullptr || (slab != nullptr && s_in <= 0) || (gamma != nullptr && out_bf16 == nullptr) ||
      (rows_per_wg != 2 && rows_per_wg != 4))
    return DEER_ERR_SHAPE;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (rows_per_wg == 2)
    hipLaunchKernelGGL((resadd_ln_multirow_kernel<2, F16>), dim3((T + 1) / 2), dim3(256), 0, st, x, slab, s_in, slab_stride, gate, bias, gamma, beta,
                       reinterpret_cast<bf16_t*>(out_bf16), eps, T);
  else
    hipLaunchKernelGGL((resadd_ln_multirow_kernel<4, F16>), dim3((T + 3) / 4), dim3(256), 0, st, x, slab, s_in, slab_stride, gate, bias, gamma, beta,
                       reinterpret_cast<bf16_t*>(out_bf16), eps, T);
  DEER_LAUNCH_CHECK();
  return DEER_OK;
}

extern "C" int deer_resadd_ln_multirow(float* x, const float* slab, int s_in, long slab_stride, const float* gate, const float* bias,
                                       const float* gamma, const float* beta, void* out_bf16, int T, int d, float eps, int rows_per_wg,
                                       void* stream) {
  return resadd_ln_multirow<false>(x, slab, s_in, slab_stride, gate, bias, gamma, beta, out_bf16, T, d, eps, rows_per_wg, stream);
}

template <bool F16>
static int resadd_ln(float* x, const float* slab, int s_in, long slab_stride, const float* gate, const float* bias, const float* gamma,
                     const float* beta, void* out_bf16, float* out_f32, float* x_copy, int T, int d, float eps, const int* ctl, void* stream) {
  if (T <= 0 || d <= 0 || (d & 3) || d > 4096 || (slab != nullptr && s_in <= 0) ||
      (gamma != nullptr && out_bf16 == nullptr && out_f32 == nullptr))
    return DEER_ERR_SHAPE;
  // env-batch vision rows: several rows per workgroup (bit-identical per row; DEER_RESADD_ROWS=1 keeps the one-row kernel, 2 / 4 force R)
  static const int rows_knob = [] { const char* e = getenv("DEER_RESADD_ROWS"); return e ? atoi(e) : 0; }();
  if (d == 1024 && T >= 2048 && out_f32 == nullptr && x_copy == nullptr && ctl == nullptr && rows_knob != 1 && (slab != nullptr || gamma != nullptr))
    return resadd_ln_multirow<F16>(x, slab, s_in, slab_stride, gate, bias, gamma, beta, out_bf16, T, d, eps, rows_knob == 4 ? 4 : 2, stream);
  hipLaunchKernelGGL((resadd_ln_kernel<256, F16>), dim3(T), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, slab, s_in,
                     slab_stride, gate, gamma, beta, reinterpret_cast<bf16_t*>(out_bf16), out_f32, x_copy, d, eps, ctl, bias,
                     static_cast<bf16_t*>(nullptr), 0, deer_rowmap{nullptr, 0, nullptr, nullptr, nullptr, 0, 0});
  DEER_LAUNCH_CHECK();
  return DEER_OK;
}

extern "C" int deer_resadd_ln(float* x, const float* slab, int s_in, long slab_stride, const float* gate,
                              const float* bias, const float* gamma, const float* beta, void* out_bf16, float* out_f32, float* x_copy, int T,
                              int d, float eps, const int* ctl, void* stream) {
  return resadd_ln<false>(x, slab, s_in, slab_stride, gate, bias, gamma, beta, out_bf16, out_f32, x_copy, T, d, eps, ctl, stream);
}

// the same with the 16-bit LayerNorm output in fp16 (round 6: the vision tower's fp16 arithmetic)
extern "C" int deer_resadd_ln_f16(float* x, const float* slab, int s_in, long slab_stride, const float* gate,
                                  const float* bias, const float* gamma, const float* beta, void* out_f16, float* out_f32, float* x_copy, int T,
                                  int d, float eps, const int* ctl, void* stream) {
  return resadd_ln<true>(x, slab, s_in, slab_stride, gate, bias, gamma, beta, out_f16, out_f32, x_copy, T, d, eps, ctl, stream);
}

// deer_resadd_ln with the LayerNorm output ALSO as two bf16 planes hi = bf16(y), lo = bf16(y - hi) (the pre-split activation operand
// of deer_gemm_skinny_hl); out_f32 optional (deer_xattn_fused reads it).
template <bool F16>
static int resadd_ln_split(float* x, const float* slab, int s_in, long slab_stride, const float* gate, const float* bias,
                                    const float* gamma, const float* beta, void* out_hi, void* out_lo, float* out_f32, float* x_copy,
                                    int T, int d, float eps, const int* ctl, void* stream) {
  if (T <= 0 || d <= 0 || (d & 3) || d > 4096 || (slab != nullptr && s_in <= 0) || gamma == nullptr || out_hi == nullptr || out_lo == nullptr)
    return DEER_ERR_SHAPE;
  hipLaunchKernelGGL((resadd_ln_kernel<256, F16>), dim3(T), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, slab, s_in, slab_stride, gate,
                     gamma, beta, reinterpret_cast<bf16_t*>(out_hi), out_f32, x_copy, d, eps, ctl, bias, reinterpret_cast<bf16_t*>(out_lo));
  DEER_LAUNCH_CHECK();
  return DEER_OK;
}

// Env batches with compaction of exited environments: deer_resadd_ln / deer_resadd_ln_split (out_lo != NULL) restricted to the active slots
// of `cmap` (rows_per_env rows per slot); x_in != NULL: the gathering first row operation of a compaction layer (see deer_rowmap above)
template <bool F16>
static int resadd_ln_rows(float* x, const float* slab, int s_in, long slab_stride, const float* gate, const float* gamma, const float* beta,
                                   void* out_bf16, void* out_lo, float* out_f32, float* x_copy, int T_rows, int d, float eps, const int* ctl,
                                   const int* cmap, int rows_per_env, const float* x_in, const int* cmap_old, int B, int drop_upto, void* stream) {
  if (T_rows <= 0 || d <= 0 || (d & 3) || d > 4096 || (slab != nullptr && s_in <= 0) || (gamma != nullptr && out_bf16 == nullptr && out_f32 == nullptr) ||
      cmap == nullptr || rows_per_env <= 0 || ctl == nullptr || B <= 0 || B > DEER_MAX_ENVS || (x_in != nullptr && (cmap_old == nullptr || x_in == x)))
    return DEER_ERR_SHAPE;
  deer_rowmap rm{cmap, rows_per_env, x_in, cmap_old, ctl, B, drop_upto};
  hipLaunchKernelGGL((resadd_ln_kernel<256, F16>), dim3(T_rows), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, slab, s_in, slab_stride, gate,
                     gamma, beta, reinterpret_cast<bf16_t*>(out_bf16), out_f32, x_copy, d, eps, ctl, static_cast<const float*>(nullptr),
                     reinterpret_cast<bf16_t*>(out_lo), 0, rm);
  DEER_LAUNCH_CHECK();
  return DEER_OK;
}

// the same with the two planes in MFMA-fragment order [d/32][64][8] (T <= 16 rows: lane = 16 * (k % 32 / 8) + row); d % 32 == 0
template <bool F16>
static int resadd_ln_packed(float* x, const float* slab, int s_in, long slab_stride, const float* gate, const float* bias,
                                     const float* gamma, const float* beta, void* out_hi, void* out_lo, float* out_f32, float* x_copy,
                                     int T, int d, float eps, const int* ctl, void* stream) {
  if (T <= 0 || T > 16 || d <= 0 || (d & 31) || d > 4096 || (slab != nullptr && s_in <= 0) || gamma == nullptr || out_hi == nullptr || out_lo == nullptr)
    return DEER_ERR_SHAPE;
  hipLaunchKernelGGL((resadd_ln_kernel<512, F16>), dim3(T), dim3(512), 0, reinterpret_cast<hipStream_t>(stream), x, slab, s_in, slab_stride, gate,
                     gamma, beta, reinterpret_cast<bf16_t*>(out_hi), out_f32, x_copy, d, eps, ctl, bias, reinterpret_cast<bf16_t*>(out_lo), 1);
  DEER_LAUNCH_CHECK();
  return DEER_OK;
}

// C entry points of the three forms above: bf16 planes, and - suffix _f16, round 6 - fp16 planes (operands of the *_f16 trunk GEMMs)
extern "C" int deer_resadd_ln_split(float* x, const float* slab, int s_in, long slab_stride, const float* gate, const float* bias, const float* gamma,
                                    const float* beta, void* out_hi, void* out_lo, float* out_f32, float* x_copy, int T, int d, float eps, const int* ctl,
                                    void* stream) {
  return resadd_ln_split<false>(x, slab, s_in, slab_stride, gate, bias, gamma, beta, out_hi, out_lo, out_f32, x_copy, T, d, eps, ctl, stream);
}
extern "C" int deer_resadd_ln_split_f16(float* x, const float* slab, int s_in, long slab_stride, const float* gate, const float* bias, const float* gamma,
                                        const float* beta, void* out_hi, void* out_lo, float* out_f32, float* x_copy, int T, int d, float eps,
                                        const int* ctl, void* stream) {
  return resadd_ln_split<true>(x, slab, s_in, slab_stride, gate, bias, gamma, beta, out_hi, out_lo, out_f32, x_copy, T, d, eps, ctl, stream);
}
extern "C" int deer_resadd_ln_rows(float* x, const float* slab, int s_in, long slab_stride, const float* gate, const float* gamma, const float* beta,
                                   void* out_bf16, void* out_lo, float* out_f32, float* x_copy, int T_rows, int d, float eps, const int* ctl,
                                   const int* cmap, int rows_per_env, const float* x_in, const int* cmap_old, int B, int drop_upto, void* stream) {
  return resadd_ln_rows<false>(x, slab, s_in, slab_stride, gate, gamma, beta, out_bf16, out_lo, out_f32, x_copy, T_rows, d, eps, ctl, cmap, rows_per_env, x_in,
                               cmap_old, B, drop_upto, stream);
}
extern "C" int deer_resadd_ln_rows_f16(float* x, const float* slab, int s_in, long slab_stride, const float* gate, const float* gamma, const float* beta,
                                       void* out_f16, void* out_lo, float* out_f32, float* x_copy, int T_rows, int d, float eps, const int* ctl,
                                       const int* cmap, int rows_per_env, const float* x_in, const int* cmap_old, int B, int drop_upto, void* stream) {
  return resadd_ln_rows<true>(x, slab, s_in, slab_stride, gate, gamma, beta, out_f16, out_lo, out_f32, x_copy, T_rows, d, eps, ctl, cmap, rows_per_env, x_in,
                              cmap_old, B, drop_upto, stream);
}
extern "C" int deer_resadd_ln_packed(float* x, const float* slab, int s_in, long slab_stride, const float* gate, const float* bias, const float* gamma,
                                     const float* beta, void* out_hi, void* out_lo, float* out_f32, float* x_copy, int T, int d, float eps, const int* ctl,
                                     void* stream) {
  return resadd_ln_packed<false>(x, slab, s_in, slab_stride, gate, bias, gamma, beta, out_hi, out_lo, out_f32, x_copy, T, d, eps, ctl, stream);
}
extern "C" int deer_resadd_ln_packed_f16(float* x, const float* slab, int s_in, long slab_stride, const float* gate, const float* bias, const float* gamma,
                                         const float* beta, void* out_hi, void* out_lo, float* out_f32, float* x_copy, int T, int d, float eps,
                                         const int* ctl, void* stream) {
  return resadd_ln_packed<true>(x, slab, s_in, slab_stride, gate, bias, gamma, beta, out_hi, out_lo, out_f32, x_copy, T, d, eps, ctl, stream);
}

// ---- ViT patch embedding, step 1: im2col of 14x14/14 patches -> 16-bit [N*P, Kpad] -----------------------
// k = ch*p*p + py*p + px matches conv1.weight[W,3,p,p].reshape(W, 3*p*p); columns >= 3*p*p are zero.
// img_kind: 0 = f32 frames, 1 = bf16, 2 = fp16; OUT_F16: patches in fp16 (else bf16).  Same-format frames are copied bit for bit.
template <bool OUT_F16>
__global__ void im2col_kernel(const void* __restrict__ img, int img_kind, int N, int S, int p, int gw,
                              bf16_t* __restrict__ out, int Kpad) {
  const long total = (long)N * gw * gw * Kpad;
  const int kk = 3 * p * p;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int k = (int)(idx % Kpad);
    const long row = idx / Kpad;
    bf16_t v = 0;
    if (k < kk) {
      const int P = gw * gw;
      const int n = (int)(row / P), pi = (int)(row - (long)n * P);
      const int gy = pi / gw, gx = pi - gy * gw;
      const int ch = k / (p * p), rem = k - ch * p * p, py = rem / p, px = rem - py * p;
      const long src = (((long)n * 3 + ch) * S + (gy * p + py)) * S + gx * p + px;
      if (img_kind == (OUT_F16 ? 2 : 1)) v = reinterpret_cast<const bf16_t*>(img)[src];
      else {
        const float f = img_kind == 0 ? reinterpret_cast<const float*>(img)[src]
                                      : (img_kind == 1 ? bf2f(reinterpret_cast<const bf16_t*>(img)[src]) : h2f(reinterpret_cast<const bf16_t*>(img)[src]));
        v = f2x<OUT_F16>(f);
      }
    }
    out[idx] = v;
  }
}

template <bool OUT_F16>
static int vit_im2col(const void* img, int img_kind, int N, int S, int patch, void* out, int Kpad, void* stream) {
  if (N <= 0 || S <= 0 || patch <= 0 || S % patch != 0 || Kpad < 3 * patch * patch || img_kind < 0 || img_kind > 2) return DEER_ERR_SHAPE;
  const int gw = S / patch;
  const long total = (long)N * gw * gw * Kpad;
  const int blocks = (int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256);
  hipLaunchKernelGGL(im2col_kernel<OUT_F16>, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), img, img_kind,
                     N, S, patch, gw, reinterpret_cast<bf16_t*>(out), Kpad);
  DEER_LAUNCH_CHECK();
  return DEER_OK;
}

extern "C" int deer_vit_im2col(const void* img, int img_is_bf16, int N, int S, int patch, void* out, int Kpad,
                               void* stream) {
  return vit_im2col<false>(img, img_is_bf16 ? 1 : 0, N, S, patch, out, Kpad, stream);
}

// fp16 patches (round 6); img_kind: 0 = f32 frames, 1 = bf16, 2 = fp16
extern "C" int deer_vit_im2col_f16(const void* img, int img_kind, int N, int S, int patch, void* out, int Kpad, void* stream) {
  return vit_im2col<true>(img, img_kind, N, S, patch, out, Kpad, stream);
}

// ---- ViT patch embedding, step 2: [cls ; patches] + positional embedding, then ln_pre ----
__global__ __launch_bounds__(256) void vit_embed_lnpre_kernel(const float* __restrict__ patch, const float* __restrict__ cls,
                                                              const float* __restrict__ pos, const float* __restrict__ g,
                                                              const float* __restrict__ bta, float* __restrict__ x, int P,
                                                              int W, float eps) {
  __shared__ float red[16];
  const int row = blockIdx.x, n = row / (P + 1), t = row - n * (P + 1);
  const float* src = (t == 0) ? cls : patch + ((long)n * P + (t - 1)) * W;
  const float* pp = pos + (long)t * W;
  float s = 0.f;
  for (int i = threadIdx.x; i < W; i += 256) s += src[i] + pp[i];
  const float mean = block_sum(s, red) / W;
  float v = 0.f;
  for (int i = threadIdx.x; i < W; i += 256) {
    const float d = src[i] + pp[i] - mean;
    v += d * d;
  }
  const float rstd = rsqrtf(block_sum(v, red) / W + eps);
  for (int i = threadIdx.x; i < W; i += 256) x[(long)row * W + i] = (src[i] + pp[i] - mean) * rstd * g[i] + bta[i];
}

extern "C" int deer_vit_embed_lnpre(const float* patch, const float* cls, const float* pos, const float* ln_w,
                                    const float* ln_b, float* x, int N, int P, int W, float eps, void* stream) {
  if (N <= 0 || P <= 0 || W <= 0) return DEER_ERR_SHAPE;
  hipLaunchKernelGGL(vit_embed_lnpre_kernel, dim3(N * (P + 1)), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), patch,
                     cls, pos, ln_w, ln_b, x, P, W, eps);
  DEER_LAUNCH_CHECK();
  return DEER_OK;
}

// ---- token embedding (mosaic_gpt_3b.py:341) + media bookkeeping (flamingo_lm.py:211, helpers.py:208) ----
// x[t] = wte[ids[t]] (bf16 table -> f32 residual stream); text_time[t] = cumsum(ids == media_token_id).
template <bool F16>
__global__ __launch_bounds__(256) void embed_tokens_kernel(const long long* __restrict__ ids, const bf16_t* __restrict__ wte,
                                                           float* __restrict__ x, int* __restrict__ text_time, int T, int d,
                                                           int vocab, int media_id) {
  const int row = blockIdx.x, t = row % T, e0 = row - t;   // rows are [env][T]; the media count restarts per environment
  long long id = ids[row];
  if (id < 0) id = 0;
  if (id >= vocab) id = vocab - 1;
  for (int i = threadIdx.x; i < d; i += 256) x[(long)row * d + i] = x2f<F16>(wte[id * d + i]);
  if (threadIdx.x == 0) {
    int c = 0;
    for (int j = 0; j <= t; ++j) c += (ids[e0 + j] == media_id) ? 1 : 0;
    text_time[row] = c;
  }
}

template <bool F16>
static int embed_tokens(const long long* ids, const void* wte, float* x, int* text_time, int T, int batch, int d, int vocab, int media_id, void* stream) {
  if (T <= 0 || batch <= 0 || d <= 0 || vocab <= 0) return DEER_ERR_SHAPE;
  hipLaunchKernelGGL(embed_tokens_kernel<F16>, dim3(T * batch), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), ids,
                     reinterpret_cast<const bf16_t*>(wte), x, text_time, T, d, vocab, media_id);
  DEER_LAUNCH_CHECK();
  return DEER_OK;
}
extern "C" int deer_embed_tokens(const long long* ids, const void* wte, float* x, int* text_time, int T, int batch, int d,
                                 int vocab, int media_id, void* stream) {
  return embed_tokens<false>(ids, wte, x, text_time, T, batch, d, vocab, media_id, stream);
}
extern "C" int deer_embed_tokens_f16(const long long* ids, const void* wte, float* x, int* text_time, int T, int batch, int d,
                                     int vocab, int media_id, void* stream) {   // fp16 embedding table (round 6)
  return embed_tokens<true>(ids, wte, x, text_time, T, batch, d, vocab, media_id, stream);
}

// ---- broadcast a [rows, C] f32 parameter to `batch` copies (Perceiver latents, helpers.py:128) -------------
__global__ void broadcast_rows_kernel(const float* __restrict__ src, float* __restrict__ dst, long n, int batch) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n * batch; i += (long)gridDim.x * blockDim.x)
    dst[i] = src[i % n];
}

extern "C" int deer_broadcast_rows(const float* src, float* dst, long n, int batch, void* stream) {
  if (n <= 0 || batch <= 0) return DEER_ERR_SHAPE;
  const long total = n * batch;
  const int blocks = (int)((total + 255) / 256 > 2048 ? 2048 : (total + 255) / 256);
  hipLaunchKernelGGL(broadcast_rows_kernel, dim3(blocks), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), src, dst, n,
                     batch);
  DEER_LAUNCH_CHECK();
  return DEER_OK;
}
