// The LLM row operation x += scale * sum_s slab[s] ; [copy x] ; [LayerNorm(x) -> bf16 planes / f32] as a device function of the ROW index
// (csrc/norm_embed.hip launches it one workgroup per row; csrc/persistent_layer.hip runs it as a phase of one launch).
#pragma once
#include "common.h"

// Env batches with compaction (common.h: CMAP_*): `rows_per_env` > 0 and `cmap` = the row map of this layer -> rows beyond the active
// slots return at once.  GATHER (x_in != NULL): this launch is the first row operation of a compaction layer - workgroup r' owns
// DESTINATION row r' of the new packing: the surviving slots of the old map `cmap_old` (environments whose EXIT_FLAG is still 0) are
// counted in slot order, row r' reads x_in / the slabs at its SOURCE row, and x (a different buffer than x_in) receives the packed rows;
// workgroup 0 publishes the new map into `cmap` (which the later kernels of the layer read).
// Which environments leave is decided ONLY from verdicts the launch stream has ordered in front of this kernel: an
// environment is dropped iff it exited at a layer <= `drop_upto` (= this layer - 2: the step driver makes the trunk wait for that check).
// The check of layer - 1 may still be running beside this kernel and raise EXIT_FLAG while the workgroups read it (consecutive exit
// layers): head_final publishes EXIT_LAYER (= layer - 1 > drop_upto) in front of EXIT_FLAG (release / acquire), so every workgroup sees either
// "not exited" or "exited too late to drop" and all of them build the same packing; that environment leaves one compaction later.
struct deer_rowmap {
  const int* cmap;          // map of this layer (gather mode: WRITTEN by workgroup 0)
  int rows_per_env;         // T (0 = no map)
  const float* x_in;        // gather source (NULL = in place)
  const int* cmap_old;      // map the source rows are packed by
  const int* ctl0;          // control blocks (EXIT_FLAG / EXIT_LAYER per environment)
  int B;
  int drop_upto;            // gather mode: deepest exit layer whose environments leave the packing
};

// bit e: environment e leaves the packing.  The exit check of layer - 1 may publish its verdict while this runs: flags first (relaxed,
// agent scope), ONE acquire fence, then the exit layers of the flagged environments - pairs with head.hip::publish_exit_flag, so a raised
// flag always comes with its own EXIT_LAYER.  B <= 16.
__device__ __forceinline__ unsigned rowmap_dropped_mask(const deer_rowmap& rm) {
  int* c = const_cast<int*>(rm.ctl0);
  unsigned flagged = 0;
  for (int e = 0; e < rm.B; ++e)
    if (__hip_atomic_load(c + e * CTL_WORDS + CTL_EXIT_FLAG, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) flagged |= 1u << e;
  if (flagged == 0) return 0;
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  unsigned m = 0;
  for (int e = 0; e < rm.B; ++e) {
    if (!(flagged >> e & 1)) continue;
    const int el = __hip_atomic_load(c + e * CTL_WORDS + CTL_EXIT_LAYER, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (el >= 0 && el <= rm.drop_upto) m |= 1u << e;
  }
  return m;
}

// LayerNorm arithmetic shared by every row kernel (one row per workgroup here, R rows per workgroup in norm_embed.hip): every product and
// sum is rounded on its own (no fused multiply-add, whatever the caller's expression looks like after inlining), so that kernels with
// different shapes give the same bits per row.
__device__ __forceinline__ float ln_sq4(const float4 v, float mean) {
#pragma clang fp contract(off)
  const float a = v.x - mean, b = v.y - mean, c = v.z - mean, e = v.w - mean;
  return a * a + b * b + c * c + e * e;
}
__device__ __forceinline__ float4 ln_norm4(const float4 v, float mean, float rstd, const float4 g) {
#pragma clang fp contract(off)
  float4 y;
  y.x = (v.x - mean) * rstd * g.x; y.y = (v.y - mean) * rstd * g.y;
  y.z = (v.z - mean) * rstd * g.z; y.w = (v.w - mean) * rstd * g.w;
  return y;
}
__device__ __forceinline__ float4 ln_add4(const float4 y, const float4 b) {
#pragma clang fp contract(off)
  return float4{y.x + b.x, y.y + b.y, y.z + b.z, y.w + b.w};
}

// NT threads per row: 256, or 512 for the one-environment trunk (<= 16 rows: the launch is a latency chain - with 512 threads a thread
// owns ONE float4 column of a 2048-wide row and all of its slab loads are in flight together, one L2 round trip instead of four)
template <int NT, bool F16 = false>   // F16: the 16-bit outputs (out_bf, and the hi / lo planes out_bf + out_lo) are fp16
__device__ __forceinline__ void resadd_ln_body(float* __restrict__ x, const float* __restrict__ slab, int s_in,
                                                        long slab_stride, const float* __restrict__ gate,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        bf16_t* __restrict__ out_bf, float* __restrict__ out_f32,
                                                        float* __restrict__ x_copy, int d, float eps, const int* ctl,
                                                        const float* __restrict__ bias, bf16_t* __restrict__ out_lo, int packed,
                                                        deer_rowmap rm, int r) {
  __shared__ float red[16];
  __shared__ int s_src;
  int rs = r;                                              // source row (== r unless gathering)
  if (rm.rows_per_env > 0) {
    const int T = rm.rows_per_env;
    if (rm.x_in != nullptr) {
      if (threadIdx.x == 0) {
        const int slot = r / T, t = r - slot * T;
        const int n_old = rm.cmap_old[CMAP_N];
        const unsigned gone = rowmap_dropped_mask(rm);
        int kept = 0, src = -1;
        for (int s = 0; s < n_old; ++s) {
          const int e = rm.cmap_old[CMAP_SLOT_ENV + s];
          if (gone >> e & 1) continue;
          if (kept == slot) src = s * T + t;
          if (r == 0) {                                    // workgroup 0 publishes the new map
            int* cm = const_cast<int*>(rm.cmap);
            cm[CMAP_SLOT_ENV + kept] = e;
            cm[CMAP_ENV_SLOT + e] = kept;
          }
          ++kept;
        }
        if (r == 0) {
          int* cm = const_cast<int*>(rm.cmap);
          cm[CMAP_N] = kept;
          for (int s = 0; s < n_old; ++s) {
            const int e = rm.cmap_old[CMAP_SLOT_ENV + s];
            if (gone >> e & 1) cm[CMAP_ENV_SLOT + e] = -1;
          }
          for (int e = 0; e < rm.B; ++e)
            if (rm.cmap_old[CMAP_ENV_SLOT + e] < 0) cm[CMAP_ENV_SLOT + e] = -1;
        }
        s_src = src;
      }
      __syncthreads();
      rs = s_src;
      if (rs < 0) return;                                  // beyond the surviving slots
    } else if (rm.cmap != nullptr && r >= rm.cmap[CMAP_N] * T) {
      return;
    }
  }
  const int n4 = d >> 2;
  float* xr = x + (long)r * d;
  const float* xs = (rm.x_in != nullptr ? rm.x_in : x) + (long)rs * d;
  constexpr int NV = 1024 / NT;                           // float4 per thread: d <= 4096
  float4 v[NV];                                           // the row stays in registers
  const float sc = (slab != nullptr && gate != nullptr) ? tanhf(*gate) : 1.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int i4 = threadIdx.x + j * NT;
    v[j] = float4{0.f, 0.f, 0.f, 0.f};
    if (i4 < n4) {
      float4 a = float4{0.f, 0.f, 0.f, 0.f};
      if (slab != nullptr) {
        a = slab_sum4(slab + (long)rs * d + (long)i4 * 4, s_in, slab_stride);
        if (bias != nullptr) {
          const float4 t = *reinterpret_cast<const float4*>(bias + (long)i4 * 4);
          a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w;
        }
      }
      float4 xv = *reinterpret_cast<const float4*>(xs + (long)i4 * 4);
      xv.x += sc * a.x; xv.y += sc * a.y; xv.z += sc * a.z; xv.w += sc * a.w;
      v[j] = xv;
      if (slab != nullptr || rm.x_in != nullptr) *reinterpret_cast<float4*>(xr + (long)i4 * 4) = xv;
      if (x_copy != nullptr) *reinterpret_cast<float4*>(x_copy + (long)rs * d + (long)i4 * 4) = xv;
    }
  }
  if (gamma == nullptr) return;
  // gamma / beta are requested BEFORE the two block reductions: their L2 round trip overlaps the reductions instead of following them (the
  // launch is a latency chain - 0.5-0.7 us of it was this second trip)
  float4 gv[NV], bv[NV];
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int i4 = threadIdx.x + j * NT;
    gv[j] = bv[j] = float4{0.f, 0.f, 0.f, 0.f};
    if (i4 < n4) {
      gv[j] = *reinterpret_cast<const float4*>(gamma + (long)i4 * 4);
      if (beta != nullptr) bv[j] = *reinterpret_cast<const float4*>(beta + (long)i4 * 4);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j) s += v[j].x + v[j].y + v[j].z + v[j].w;
  const float mean = block_sum(s, red) / d;
  float var = 0.f;
#pragma unroll
  for (int j = 0; j < NV; ++j)
    if (threadIdx.x + j * NT < n4) var += ln_sq4(v[j], mean);
  const float rstd = rsqrtf(block_sum(var, red) / d + eps);
#pragma unroll
  for (int j = 0; j < NV; ++j) {
    const int i4 = threadIdx.x + j * NT;
    if (i4 < n4) {
      float4 y = ln_norm4(v[j], mean, rstd, gv[j]);
      if (beta != nullptr) y = ln_add4(y, bv[j]);
      if (out_bf != nullptr) {
        uint32_t h01, h23, l01, l23;
        split2<F16>(y.x, y.y, h01, l01);
        split2<F16>(y.z, y.w, h23, l23);
        // packed: MFMA-fragment order [k-tile][lane = 16 * (k % 32 / 8) + row][8] - the <= 16 rows of one environment read back as ONE
        // contiguous 1 KiB per k-tile and plane (deer_trunk_wide_gemm, deer_xattn_fused_packed)
        const int col = i4 * 4;
        const long o = packed ? (((long)(col >> 5) * 64 + ((col & 31) >> 3) * 16 + r) * 8 + (col & 7)) : ((long)r * d + col);
        *reinterpret_cast<uint2*>(out_bf + o) = uint2{h01, h23};
        if (out_lo != nullptr)      // second plane: y = hi + lo to ~16 (bf16) / ~22 (fp16) significand bits (activation operand of deer_gemm_skinny_hl)
          *reinterpret_cast<uint2*>(out_lo + o) = uint2{l01, l23};
      }
      if (out_f32 != nullptr) *reinterpret_cast<float4*>(out_f32 + (long)r * d + (long)i4 * 4) = y;
    }
  }
}

