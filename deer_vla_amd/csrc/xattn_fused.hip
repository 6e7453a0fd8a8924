// The attention half of GatedCrossAttentionBlock (open_flamingo/src/helpers.py:184-233, called at :267) as ONE launch:
//     q = LN(x) Wq^T * 64^-0.5 ;  sim = q k^T, keys masked by media time ;  P = softmax ;  o = P v ;  y = o Wo^T
// Before: to_q skinny GEMM (2 MB of weights, launch-floor bound) -> attention (q summed from 16 split-K slabs) -> to_out skinny GEMM
// (2 MB) = three dependent launches of ~4 + 11 + 4 us.  Here workgroup (h, cs, env) streams Wq of head h (64 x d), computes q_h and
// the attention of head h, then multiplies o_h (T x 64) with the 64 input columns of head h of Wo for its share `cs` of the output
// columns, and writes y as ONE f32 slab PER HEAD: the consumer (deer_resadd_ln) sums the `heads` slabs, applies tanh(attn_gate) and the
// residual - the same split-K slab contract as every other projection of the trunk (deterministic, no atomics).
//  * the NS workgroups that share a head recompute q_h and the attention (256 KB of Wq from L2 after the first touch): that buys
//    NS x more workgroups streaming (8 heads alone would pull 4 MB through 8 CUs)
//  * weights stay in the skinny GEMM's packed MFMA-fragment layout (one contiguous 1 KiB read per wave instruction, non-temporal);
//    the f32 activation enters the q projection as bf16 hi + lo, o enters the output projection as bf16 hi + lo (as in gemm_skinny.hip)
//  * the Wo fragments of a wave (8 KiB) and the K/V tile are requested before the q projection starts
#include "common.h"
#include <cstdlib>

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(4))) short xf_s16x4_t;
typedef __attribute__((address_space(3))) xf_s16x4_t xf_lds_s16x4_t;

#define XF_NW 8            // waves per workgroup
#define XF_HD 64
#define XF_KP 72           // K rows: 64 + 8 bf16
#define XF_MAXKV 128
#if defined(DEER_KTRACE) && !defined(DEER_BODIES_ONLY)
KT_DEFINE(xattn)
#define XKT(slot) KT(xattn, bx == 0 && by == 0, slot)
#else
#define XKT(slot) do { } while (0)
#endif

// PACKED (one environment, <= 16 rows): the activation arrives as bf16 hi / lo planes in MFMA-fragment order (xn = hi plane, xlo = lo
// plane; deer_resadd_ln_packed) - one coalesced 1 KiB read per k-tile and plane instead of 16 rows x 16 B per lane
template <int MT, bool PACKED, bool F16 = false>   // F16: weights, K / V and every 16-bit intermediate in fp16
__device__ __forceinline__ void xattn_fused_body(const float* __restrict__ xn, int d, const bf16_t* __restrict__ Wq_p,
                                                                const bf16_t* __restrict__ kv, int ldkv, int inner,
                                                                const int* __restrict__ text_time, int n_per_media, int n_kv,
                                                                const bf16_t* __restrict__ Wo_p, float* __restrict__ out, long slab_stride,
                                                                int T, int heads, int NS, float scale,
                                                                const bf16_t* __restrict__ xlo, const int* __restrict__ cmap, int bx, int by) {
  // compaction: `by` is a SLOT of the row map - activation / output rows by slot, media K/V and text_time by its environment
  if (cmap != nullptr && by >= cmap[CMAP_N]) return;
  const int env = cmap != nullptr ? cmap[CMAP_SLOT_ENV + by] : by;
  static_assert(!PACKED || MT == 1, "packed planes hold one 16-row tile");
  constexpr int MPAD = MT * 16;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* red = reinterpret_cast<float*>(smem_raw);                                   // [NW][MPAD][64] f32 partial q
  bf16_t* qs = reinterpret_cast<bf16_t*>(red + XF_NW * MPAD * XF_HD);                // [MPAD][72] bf16 q (scaled)
  bf16_t* Ks = qs + MPAD * XF_KP;                                                    // [128][72]
  bf16_t* Vs = Ks + XF_MAXKV * XF_KP;                                                // [128][72] ROW-major (round 5): read transposed
  bf16_t* oh = Vs + XF_MAXKV * XF_KP;                                                // [MPAD][72] o hi
  bf16_t* ol = oh + MPAD * XF_KP;                                                    // [MPAD][72] o lo

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  // NS < 0 (experiment, DEER_XF_MAP=1): the |NS| column shares of a head sit on ONE XCD (workgroup bx runs on XCD bx % 8: h = bx % heads),
  // so that its L2 fetches the head's Wq once for all shares
  const bool xmap = NS < 0;
  if (xmap) NS = -NS;
  const int h = xmap ? bx % heads : bx / NS, cs = xmap ? bx / heads : bx - h * NS, b = by;
  XKT(0);
  const float* xb = xn + (long)b * T * d;
  const bf16_t* kvb = kv + (long)env * n_kv * ldkv + h * XF_HD;
  const int ktiles = d >> 5;

  // ---- requests that do not depend on anything computed here: this wave's Wo fragments, the K/V tile of (env, head) ----
  const int xk = inner >> 5;                                   // k-fragments per output-column tile of Wo (K = inner)
  const int tiles_per_wg = (d >> 4) / NS, tiles_per_wave = tiles_per_wg / XF_NW;      // launcher guarantees divisibility, <= 4
  u32x4 wo[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int nt = cs * tiles_per_wg + wave * tiles_per_wave + min(i, tiles_per_wave - 1);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
      wo[i][kk] = __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(Wo_p) + ((long)nt * xk + 2 * h + kk) * 64 + lane);
  }
  uint4 kreg[2], vreg[2];                                      // 128 keys x 8 segments of 16 B = 1024 pieces / 512 threads
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = tid + i * 512, row = idx >> 3, seg = idx & 7;
    kreg[i] = vreg[i] = uint4{0, 0, 0, 0};
    if (row < n_kv) {
      kreg[i] = *reinterpret_cast<const uint4*>(kvb + (long)row * ldkv + seg * 8);
      vreg[i] = *reinterpret_cast<const uint4*>(kvb + (long)row * ldkv + inner + seg * 8);
    }
  }

  XKT(1);
  // ---- q_h = xn Wq_h^T: the 8 waves split K (fragment kt = wave, wave + 8, ...), partial sums meet in LDS ----
  f32x4 acc[4][MT];
#pragma unroll
  for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[t4][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const u32x4* wq = reinterpret_cast<const u32x4*>(Wq_p) + ((long)(4 * h) * ktiles) * 64 + lane;
  // k-tiles per round: the packed form (one environment: the launch is a latency chain) keeps 8 k-tiles = 32 weight fragments + 16
  // activation fragments of the wave in flight - d = 2048 in ONE round trip instead of two
  constexpr int XU = PACKED ? 8 : 4;
  for (int kt0 = wave; kt0 < ktiles; kt0 += XU * XF_NW) {
    u32x4 w[XU][4];
    float4 a[PACKED ? 1 : 4][MT][2];
    bf16x8 ph[XU], pl[XU];
#pragma unroll
    for (int u = 0; u < XU; ++u) {                              // the weight fragments + the activation pieces in flight
      const int kt = min(kt0 + u * XF_NW, ktiles - 1);
#pragma unroll
      for (int t4 = 0; t4 < 4; ++t4) w[u][t4] = xmap ? wq[((long)t4 * ktiles + kt) * 64] : __builtin_nontemporal_load(wq + ((long)t4 * ktiles + kt) * 64);
      if (PACKED) {
        ph[u] = *reinterpret_cast<const bf16x8*>(reinterpret_cast<const bf16_t*>(xn) + ((long)kt * 64 + lane) * 8);
        pl[u] = *reinterpret_cast<const bf16x8*>(xlo + ((long)kt * 64 + lane) * 8);
        continue;
      }
#pragma unroll
      for (int j = 0; j < MT; ++j) {
        const int row = j * 16 + c;
        const float* p = xb + (long)min(row, T - 1) * d + kt * 32 + g * 8;
        a[PACKED ? 0 : u][j][0] = *reinterpret_cast<const float4*>(p);
        a[PACKED ? 0 : u][j][1] = *reinterpret_cast<const float4*>(p + 4);
        if (row >= T) a[PACKED ? 0 : u][j][0] = a[PACKED ? 0 : u][j][1] = float4{0.f, 0.f, 0.f, 0.f};
      }
    }
#pragma unroll
    for (int u = 0; u < XU; ++u) {
      if (kt0 + u * XF_NW < ktiles) {
#pragma unroll
        for (int j = 0; j < MT; ++j) {
          if (PACKED) {
#pragma unroll
            for (int t4 = 0; t4 < 4; ++t4) {
              const bf16x8 wf = __builtin_bit_cast(bf16x8, w[u][t4]);
              acc[t4][j] = mfma16<F16>(wf, ph[u], acc[t4][j]);
              acc[t4][j] = mfma16<F16>(wf, pl[u], acc[t4][j]);
            }
            continue;
          }
          const float4 x0 = a[PACKED ? 0 : u][j][0], x1 = a[PACKED ? 0 : u][j][1];
          uint4 hi, lo;
          split2<F16>(x0.x, x0.y, hi.x, lo.x);
          split2<F16>(x0.z, x0.w, hi.y, lo.y);
          split2<F16>(x1.x, x1.y, hi.z, lo.z);
          split2<F16>(x1.z, x1.w, hi.w, lo.w);
          const bf16x8 ah = __builtin_bit_cast(bf16x8, hi), al = __builtin_bit_cast(bf16x8, lo);
#pragma unroll
          for (int t4 = 0; t4 < 4; ++t4) {
            const bf16x8 wf = __builtin_bit_cast(bf16x8, w[u][t4]);
            acc[t4][j] = mfma16<F16>(wf, ah, acc[t4][j]);
            acc[t4][j] = mfma16<F16>(wf, al, acc[t4][j]);
          }
        }
      }
    }
  }
  XKT(2);
  // lane holds q_partial[m = j*16 + c][n = t4*16 + g*4 .. +3]
#pragma unroll
  for (int t4 = 0; t4 < 4; ++t4)
#pragma unroll
    for (int j = 0; j < MT; ++j)
      *reinterpret_cast<float4*>(red + ((wave * MPAD + j * 16 + c) * XF_HD) + t4 * 16 + g * 4) =
          float4{acc[t4][j][0], acc[t4][j][1], acc[t4][j][2], acc[t4][j][3]};
  // K (row-major) and V (transposed) of this head into LDS
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int idx = tid + i * 512, row = idx >> 3, seg = idx & 7;
    if (row < XF_MAXKV) {
      *reinterpret_cast<uint4*>(Ks + row * XF_KP + seg * 8) = kreg[i];
      // V stays row-major (one conflict-free 16-byte store; the transposing 2-byte stores of rounds 2-4 were eight-way bank-conflicted):
      // the P V operand is fetched with ds_read_b64_tr_b16 like csrc/attention.hip::attn_vit_kernel
      *reinterpret_cast<uint4*>(Vs + row * XF_KP + seg * 8) = vreg[i];
    }
  }
  __syncthreads();
  XKT(3);
  for (int idx = tid; idx < MPAD * XF_HD; idx += 64 * XF_NW) {          // q = sum over the 8 K-slices (fixed order), scaled, -> bf16
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < XF_NW; ++w) s += red[w * MPAD * XF_HD + idx];
    qs[(idx >> 6) * XF_KP + (idx & 63)] = f2x<F16>(s * scale);
  }
  __syncthreads();
  XKT(4);

  // ---- attention of head h: wave j < MT owns query rows j*16 .. j*16+15 (swapped QK^T as in attention.hip) ----
  if (wave < MT) {
    const int q0 = wave * 16;
    bf16x8 qf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qs + (q0 + c) * XF_KP + ks * 32 + g * 8);
    const int tt_q = (q0 + c < T) ? text_time[env * T + q0 + c] : 1;
    const int klo = (tt_q - 1) * n_per_media, khi = tt_q * n_per_media;
    constexpr int NT = XF_MAXKV / 16;
    f32x4 s[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks + (t * 16 + c) * XF_KP + ks * 32 + g * 8);
        s[t] = mfma16<F16>(kf, qf[ks], s[t]);
      }
    }
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = t * 16 + g * 4 + r;
        float v = (key < n_kv) ? s[t][r] : -INFINITY;
        if (key < n_kv && (key < klo || key >= khi)) v = -3.4028234663852886e38f;     // helpers.py:218
        s[t][r] = v;
        mx = fmaxf(mx, v);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __expf(s[t][r] - mx);
        s[t][r] = p;
        sum += p;
      }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = (tt_q == 0) ? 0.f : 1.f / sum;                                   // helpers.py:223-229
    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    xf_lds_s16x4_t* vt = (xf_lds_s16x4_t*)(Vs + (g * 4 + (c >> 2)) * XF_KP + (c & 3) * 4);   // transpose-read base of this lane (generic -> LDS: C-style cast)
#pragma unroll
    for (int ch = 0; ch < NT / 2; ++ch) {
      uint4 pw;
      pw.x = pack2x<F16>(s[2 * ch][0], s[2 * ch][1]);
      pw.y = pack2x<F16>(s[2 * ch][2], s[2 * ch][3]);
      pw.z = pack2x<F16>(s[2 * ch + 1][0], s[2 * ch + 1][1]);
      pw.w = pack2x<F16>(s[2 * ch + 1][2], s[2 * ch + 1][3]);
      const bf16x8 pf = __builtin_bit_cast(bf16x8, pw);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        // k-slot (g, j < 4) <-> key 32 ch + 4 g + j, (g, j >= 4) <-> key 32 ch + 16 + 4 g + (j - 4): a 16-lane group hands in the 16
        // eight-byte pieces of a [4 keys][16 d] block, lane c receives column c (tools/tr_probe.hip)
        const xf_s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(vt + ((ch * 32) * XF_KP + dt * 16) / 4);
        const xf_s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(vt + ((ch * 32 + 16) * XF_KP + dt * 16) / 4);
        const uint2 lo2 = __builtin_bit_cast(uint2, lo), hi2 = __builtin_bit_cast(uint2, hi);
        const uint4 vw = uint4{lo2.x, lo2.y, hi2.x, hi2.y};
        o[dt] = mfma16<F16>(__builtin_bit_cast(bf16x8, vw), pf, o[dt]);
      }
    }
    // lane holds O[q = q0 + c][dd = dt*16 + g*4 .. +3]  ->  bf16 hi + lo rows for the output projection
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const float v0 = o[dt][0] * inv, v1 = o[dt][1] * inv, v2 = o[dt][2] * inv, v3 = o[dt][3] * inv;
      uint32_t h01, h23, l01, l23;
      split2<F16>(v0, v1, h01, l01);
      split2<F16>(v2, v3, h23, l23);
      *reinterpret_cast<uint2*>(oh + (q0 + c) * XF_KP + dt * 16 + g * 4) = uint2{h01, h23};
      *reinterpret_cast<uint2*>(ol + (q0 + c) * XF_KP + dt * 16 + g * 4) = uint2{l01, l23};
    }
  }
  XKT(5);
  __syncthreads();
  XKT(6);

  // ---- y[:, cols of this workgroup] = o_h Wo[cols, h*64 .. h*64+63]^T : one f32 slab per head ----
  float* dst = out + (long)h * slab_stride + (long)b * T * d;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (i < tiles_per_wave) {
      const int nt = cs * tiles_per_wg + wave * tiles_per_wave + i;
#pragma unroll
      for (int j = 0; j < MT; ++j) {
        f32x4 y = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          const bf16x8 wf = __builtin_bit_cast(bf16x8, wo[i][kk]);
          const bf16x8 ah = *reinterpret_cast<const bf16x8*>(oh + (j * 16 + c) * XF_KP + kk * 32 + g * 8);
          const bf16x8 al = *reinterpret_cast<const bf16x8*>(ol + (j * 16 + c) * XF_KP + kk * 32 + g * 8);
          y = mfma16<F16>(wf, ah, y);
          y = mfma16<F16>(wf, al, y);
        }
        const int m = j * 16 + c;
        if (m < T) *reinterpret_cast<float4*>(dst + (long)m * d + nt * 16 + g * 4) = float4{y[0], y[1], y[2], y[3]};
      }
    }
  }
  XKT(7);
}

template <int MT, bool PACKED = false, bool F16 = false>
__global__ __launch_bounds__(64 * XF_NW) void xattn_fused_kernel(const float* __restrict__ xn, int d, const bf16_t* __restrict__ Wq_p,
                                                                const bf16_t* __restrict__ kv, int ldkv, int inner,
                                                                const int* __restrict__ text_time, int n_per_media, int n_kv,
                                                                const bf16_t* __restrict__ Wo_p, float* __restrict__ out, long slab_stride,
                                                                int T, int heads, int NS, float scale, const int* ctl,
                                                                const bf16_t* __restrict__ xlo = nullptr, const int* __restrict__ cmap = nullptr) {
  DEER_RETURN_IF_EXITED(ctl);
  xattn_fused_body<MT, PACKED, F16>(xn, d, Wq_p, kv, ldkv, inner, text_time, n_per_media, n_kv, Wo_p, out, slab_stride, T, heads, NS, scale, xlo, cmap,
                               blockIdx.x, blockIdx.y);
}

#ifndef DEER_BODIES_ONLY
// xn: f32 [batch*T, d] = LN(x) of the block's attention branch; Wq_p / Wo_p: to_q [inner, d] and to_out [d, inner] in the packed
// layout of deer_pack_weight_mfma16; kv: bf16 [batch*n_kv, ldkv] (k of head h at column h*64, v at inner + h*64); out: f32
// [heads][slab_stride] with slab_stride >= batch*T*d - slab h holds head h's contribution to all rows.  T <= 32, n_kv <= 128.
template <bool F16>
static int launch_xattn_fused(const float* xn, const void* x_lo_packed, int d, const void* Wq_p, const void* kv, int ldkv, int inner, const int* text_time,
                              int n_per_media, int n_kv, const void* Wo_p, float* out, long slab_stride, int T, int heads, int batch, float scale,
                              const int* ctl, void* stream, const int* cmap = nullptr);

extern "C" int deer_xattn_fused(const float* xn, int d, const void* Wq_p, const void* kv, int ldkv, int inner, const int* text_time,
                                int n_per_media, int n_kv, const void* Wo_p, float* out, long slab_stride, int T, int heads, int batch,
                                float scale, const int* ctl, void* stream) {
  return launch_xattn_fused<false>(xn, nullptr, d, Wq_p, kv, ldkv, inner, text_time, n_per_media, n_kv, Wo_p, out, slab_stride, T, heads, batch, scale, ctl, stream);
}

// deer_xattn_fused for an env batch with compaction: grid rows are SLOTS of `cmap`
extern "C" int deer_xattn_fused_active(const float* xn, int d, const void* Wq_p, const void* kv, int ldkv, int inner, const int* text_time,
                                       int n_per_media, int n_kv, const void* Wo_p, float* out, long slab_stride, int T, int heads, int batch,
                                       float scale, const int* ctl, const int* cmap, void* stream) {
  if (cmap == nullptr) return DEER_ERR_SHAPE;
  return launch_xattn_fused<false>(xn, nullptr, d, Wq_p, kv, ldkv, inner, text_time, n_per_media, n_kv, Wo_p, out, slab_stride, T, heads, batch, scale, ctl, stream, cmap);
}

// the same for ONE environment of <= 16 rows with LN(x) as bf16 hi / lo planes in MFMA-fragment order (deer_resadd_ln_packed)
extern "C" int deer_xattn_fused_packed(const void* x_hi, const void* x_lo, int d, const void* Wq_p, const void* kv, int ldkv, int inner,
                                       const int* text_time, int n_per_media, int n_kv, const void* Wo_p, float* out, long slab_stride, int T,
                                       int heads, float scale, const int* ctl, void* stream) {
  if (x_lo == nullptr || T > 16) return DEER_ERR_SHAPE;
  return launch_xattn_fused<false>(reinterpret_cast<const float*>(x_hi), x_lo, d, Wq_p, kv, ldkv, inner, text_time, n_per_media, n_kv, Wo_p, out, slab_stride, T,
                            heads, 1, scale, ctl, stream);
}

// ---- the same three entry points on fp16 operands: Wq / Wo packed fp16, K / V fp16, hi / lo planes fp16 (round 6) ----
extern "C" int deer_xattn_fused_f16(const float* xn, int d, const void* Wq_p, const void* kv, int ldkv, int inner, const int* text_time,
                                int n_per_media, int n_kv, const void* Wo_p, float* out, long slab_stride, int T, int heads, int batch,
                                float scale, const int* ctl, void* stream) {
  return launch_xattn_fused<true>(xn, nullptr, d, Wq_p, kv, ldkv, inner, text_time, n_per_media, n_kv, Wo_p, out, slab_stride, T, heads, batch, scale, ctl, stream);
}

// deer_xattn_fused for an env batch with compaction: grid rows are SLOTS of `cmap`
extern "C" int deer_xattn_fused_active_f16(const float* xn, int d, const void* Wq_p, const void* kv, int ldkv, int inner, const int* text_time,
                                       int n_per_media, int n_kv, const void* Wo_p, float* out, long slab_stride, int T, int heads, int batch,
                                       float scale, const int* ctl, const int* cmap, void* stream) {
  if (cmap == nullptr) return DEER_ERR_SHAPE;
  return launch_xattn_fused<true>(xn, nullptr, d, Wq_p, kv, ldkv, inner, text_time, n_per_media, n_kv, Wo_p, out, slab_stride, T, heads, batch, scale, ctl, stream, cmap);
}

// the same for ONE environment of <= 16 rows with LN(x) as bf16 hi / lo planes in MFMA-fragment order (deer_resadd_ln_packed)
extern "C" int deer_xattn_fused_packed_f16(const void* x_hi, const void* x_lo, int d, const void* Wq_p, const void* kv, int ldkv, int inner,
                                       const int* text_time, int n_per_media, int n_kv, const void* Wo_p, float* out, long slab_stride, int T,
                                       int heads, float scale, const int* ctl, void* stream) {
  if (x_lo == nullptr || T > 16) return DEER_ERR_SHAPE;
  return launch_xattn_fused<true>(reinterpret_cast<const float*>(x_hi), x_lo, d, Wq_p, kv, ldkv, inner, text_time, n_per_media, n_kv, Wo_p, out, slab_stride, T,
                            heads, 1, scale, ctl, stream);
}

template <bool F16>
static int launch_xattn_fused(const float* xn, const void* x_lo_packed, int d, const void* Wq_p, const void* kv, int ldkv, int inner, const int* text_time,
                              int n_per_media, int n_kv, const void* Wo_p, float* out, long slab_stride, int T, int heads, int batch, float scale,
                              const int* ctl, void* stream, const int* cmap) {
  if (xn == nullptr || Wq_p == nullptr || kv == nullptr || Wo_p == nullptr || out == nullptr || text_time == nullptr) return DEER_ERR_SHAPE;
  if (T <= 0 || T > 32 || n_kv <= 0 || n_kv > XF_MAXKV || heads <= 0 || inner != heads * XF_HD || (d & 127) || (ldkv & 7) || batch <= 0 ||
      n_per_media <= 0 || slab_stride < (long)batch * T * d)
    return DEER_ERR_SHAPE;
  // column split of the output projection: workgroups = heads * NS per environment; every wave gets 1..4 output tiles
  const int tiles = d >> 4;                                    // 16-column output tiles of Wo
  static const int ns_mul = [] { const char* e = getenv("DEER_XF_NS"); return e ? atoi(e) : 1; }();   // tuning knob: x more column splits
  int NS = (tiles + 4 * XF_NW - 1) / (4 * XF_NW);              // <= 4 tiles per wave
  if (ns_mul > 1 && tiles % (NS * ns_mul * XF_NW) == 0) NS *= ns_mul;
  if (tiles % (NS * XF_NW) != 0) return DEER_ERR_SHAPE;
  static const bool xf_map = [] { const char* e = getenv("DEER_XF_MAP"); return e != nullptr && e[0] == '1'; }();
  const int ns_arg = (xf_map && heads == 8) ? -NS : NS;
  const int mt = (T + 15) >> 4;
  const int mpad = mt * 16;
  const int smem = XF_NW * mpad * XF_HD * 4 + (mpad * XF_KP + 2 * XF_MAXKV * XF_KP + 2 * mpad * XF_KP) * 2;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  dim3 grid(heads * NS, batch);
#define DEER_XF_LAUNCH(MT_, PK_)                                                                                                \
  do {                                                                                                                          \
    static std::atomic<bool> attr_set{false};                                                                                               \
    auto kern = &xattn_fused_kernel<MT_, PK_, F16>;                                                                                  \
    if (!attr_set) {                                                                                                            \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024) !=  \
          hipSuccess) return DEER_ERR_LAUNCH;                                                                                   \
      attr_set = true;                                                                                                          \
    }                                                                                                                           \
    hipLaunchKernelGGL(kern, grid, dim3(64 * XF_NW), smem, st, xn, d, reinterpret_cast<const bf16_t*>(Wq_p),                   \
                       reinterpret_cast<const bf16_t*>(kv), ldkv, inner, text_time, n_per_media, n_kv,                          \
                       reinterpret_cast<const bf16_t*>(Wo_p), out, slab_stride, T, heads, ns_arg, scale, ctl,                    \
                       reinterpret_cast<const bf16_t*>(x_lo_packed), cmap);                                                     \
  } while (0)
  if (x_lo_packed != nullptr) DEER_XF_LAUNCH(1, true);
  else if (mt == 1) DEER_XF_LAUNCH(1, false);
  else DEER_XF_LAUNCH(2, false);
#undef DEER_XF_LAUNCH
  DEER_LAUNCH_CHECK();
  return DEER_OK;
}
#endif  // DEER_BODIES_ONLY
