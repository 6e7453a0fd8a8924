// Attention kernels of the DeeR-VLA step.
//
//  (1) attn_mfma_kernel      - ViT-L/14 self-attention (257x257, 16 heads x 64) and Perceiver cross-attention
//                              (64 latents x 320 keys, 8 heads x 64): MFMA, whole K/V of one head in LDS.
//  (2) xattn_small_kernel    - gated cross-attention inside the LLM (T<=32 text tokens x 128 media tokens,
//                              8 heads x 64): fp32 VALU, one workgroup per head.
//  (3) mpt_attn_small_kernel - MPT causal self-attention with ALiBi, optional q/k LayerNorm over d_model and
//                              key-padding mask (T<=32, head_dim 128): fp32 VALU, one workgroup per head.
//
// (2) and (3) read their q / qkv input as split-K partial slabs of the skinny GEMM and reduce them while
// loading (launch-boundary reduce), and write a bf16 activation that is the A operand of the next GEMM.
#include "common.h"
#include <cstdlib>

// ------------------------------------------------------------------------------------------------
// (1) MFMA attention, head_dim 64, kv_len <= 320.
// Swapped QK^T (S^T = K * Q^T): after the MFMA each lane holds, for ONE query row (lane&15), four
// consecutive keys per 16-key tile -> the softmax row reductions are in-lane + two xor-shuffles, and the
// un-normalised P values are already in the register layout the P*V MFMA wants as its "B" operand (the
// k-slot permutation is applied identically to the V^T operand), so P never goes through LDS.
// ------------------------------------------------------------------------------------------------
#define AM_HD 64
#define AM_KPITCH 72          // K rows: 64 + 8 bf16
#define AM_MAXT 20            // max 16-key tiles (kv_len <= 320)
#define AM_MAXT_LONG 36       // the long-sequence instantiation: kv_len <= 576 (pre fusion: 2 x 256 patch tokens + 64 latents)

// Optional extras (gated x-attn inside the LLM, helpers.py:192-232): Q given as f32 split-K partial slabs
// (q_slabs > 0: Q points to f32, reduced while loading), keys masked by media time (text_time[q] == key/n_per_media + 1,
// rows with text_time == 0 zeroed), f32 output, early-exit control block.
template <bool XATTN, int NWAVE, bool LOOP = false, bool F16 = false, int MAXT = AM_MAXT>   // F16: q / k / v / P / o in fp16 (the vision tower's fp16 arithmetic); MAXT: key tiles of 16 held in registers
__global__ __launch_bounds__(64 * NWAVE) void attn_mfma_kernel(const void* __restrict__ Qv, const bf16_t* __restrict__ Kp,
                                                        const bf16_t* __restrict__ V, void* __restrict__ Ov,
                                                        int q_len, int kv_len, int ldq, int ldk, int ldv, int ldo,
                                                        long q_bstride, long k_bstride, long v_bstride, long o_bstride,
                                                        float scale, int q_slabs, long q_slab_stride,
                                                        const int* __restrict__ text_time, int n_per_media, int out_is_f32,
                                                        const int* ctl, const bf16_t* __restrict__ K2,
                                                        const bf16_t* __restrict__ V2, int kv1, int ld2, long bstride2, int tpw) {
  DEER_RETURN_IF_EXITED(ctl);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int kvpad = (kv_len + 31) & ~31;
  const int vpitch = kvpad + 8;
  bf16_t* Ks = reinterpret_cast<bf16_t*>(smem_raw);                 // [kvpad][72]
  bf16_t* Vt = Ks + kvpad * AM_KPITCH;                               // [64][kvpad + 8]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int h = blockIdx.y, b = blockIdx.z;
  const bf16_t* Kb = Kp + b * k_bstride + h * AM_HD;
  const bf16_t* Vb = V + b * v_bstride + h * AM_HD;
  // optional second key/value segment (keys kv1..kv_len-1; Perceiver: [media K/V of this layer ; latent K/V])
  const bf16_t* Kb2 = K2 != nullptr ? K2 + b * bstride2 + h * AM_HD : nullptr;
  const bf16_t* Vb2 = V2 != nullptr ? V2 + b * bstride2 + h * AM_HD : nullptr;

  // A workgroup serves tpw 16-query tiles of one (head, image); wave w takes tiles w, w + NWAVE, ...  tpw = NWAVE is one tile per
  // wave; env batches (>= 128 (head, image) pairs fill the chip anyway) use tpw = 9 for the ViT's 17 tiles: two workgroups per pair
  // instead of three (one of which held a single query row), so the head's K/V are staged twice instead of three times
  // (LOOP = false keeps the single-tile form: the loop costs SGPR spills, seen as +12 % on the one-environment launches)
  int q0 = (blockIdx.x * (LOOP ? tpw : NWAVE) + wave) * 16;

  // Q fragments first (MFMA "B" operand: B[k = d][n = query]; rows >= q_len are zero): their global-load latency
  // overlaps the K/V staging below instead of following the barrier
  bf16x8 qf[2];
  int tt_q = 1, klo = 0, khi = 0;
  auto load_q = [&]() {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      uint4 v = uint4{0, 0, 0, 0};
      if (q0 + c < q_len) {
        if (XATTN) {                                           // f32 split-K partials -> sum -> bf16
          const float* qp = reinterpret_cast<const float*>(Qv) + b * q_bstride + h * AM_HD + (long)(q0 + c) * ldq + ks * 32 + g * 8;
          const float4 a0 = slab_sum4(qp, q_slabs, q_slab_stride), a1 = slab_sum4(qp + 4, q_slabs, q_slab_stride);
          v = uint4{pack2x<F16>(a0.x, a0.y), pack2x<F16>(a0.z, a0.w), pack2x<F16>(a1.x, a1.y), pack2x<F16>(a1.z, a1.w)};
        } else {
          v = *reinterpret_cast<const uint4*>(reinterpret_cast<const bf16_t*>(Qv) + b * q_bstride + h * AM_HD +
                                              (long)(q0 + c) * ldq + ks * 32 + g * 8);
        }
      }
      qf[ks] = __builtin_bit_cast(bf16x8, v);
    }
    // media-time mask (x-attn only): key j is visible iff text_time[q] == j / n_per_media + 1, i.e. j in [klo, khi)
    tt_q = (XATTN && q0 + c < q_len) ? text_time[b * q_len + q0 + c] : 1;
    klo = (tt_q - 1) * n_per_media;
    khi = tt_q * n_per_media;
  };
  load_q();

  // ---- stage K (row-major) and V (transposed) for this head; rows >= kv_len are zero ----
  // two key rows per thread: K rows are copied as they are, V is transposed with 32-bit LDS stores that carry the
  // same d of two adjacent keys (half the store instructions of a per-element transpose)
  for (int idx = tid; idx < (kvpad >> 1) * 8; idx += 64 * NWAVE) {
    const int row = (idx >> 3) * 2, seg = idx & 7;
    uint4 k0 = uint4{0, 0, 0, 0}, k1 = k0, v0 = k0, v1 = k0;
    if (row < kv_len) {
      const bool s2 = row >= kv1;
      k0 = *reinterpret_cast<const uint4*>((s2 ? Kb2 + (long)(row - kv1) * ld2 : Kb + (long)row * ldk) + seg * 8);
      v0 = *reinterpret_cast<const uint4*>((s2 ? Vb2 + (long)(row - kv1) * ld2 : Vb + (long)row * ldv) + seg * 8);
    }
    if (row + 1 < kv_len) {
      const bool s2 = row + 1 >= kv1;
      k1 = *reinterpret_cast<const uint4*>((s2 ? Kb2 + (long)(row + 1 - kv1) * ld2 : Kb + (long)(row + 1) * ldk) + seg * 8);
      v1 = *reinterpret_cast<const uint4*>((s2 ? Vb2 + (long)(row + 1 - kv1) * ld2 : Vb + (long)(row + 1) * ldv) + seg * 8);
    }
    *reinterpret_cast<uint4*>(Ks + row * AM_KPITCH + seg * 8) = k0;
    *reinterpret_cast<uint4*>(Ks + (row + 1) * AM_KPITCH + seg * 8) = k1;
    const uint32_t a[4] = {v0.x, v0.y, v0.z, v0.w}, bq[4] = {v1.x, v1.y, v1.z, v1.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const uint32_t even = (a[e] & 0xffffu) | (bq[e] << 16);          // d = seg*8 + 2e   : (key row, key row+1)
      const uint32_t odd = (a[e] >> 16) | (bq[e] & 0xffff0000u);       // d = seg*8 + 2e+1
      *reinterpret_cast<uint32_t*>(Vt + (seg * 8 + 2 * e) * vpitch + row) = even;
      *reinterpret_cast<uint32_t*>(Vt + (seg * 8 + 2 * e + 1) * vpitch + row) = odd;
    }
  }
  __syncthreads();

  const int nt = kvpad >> 4;
  for (int ti = wave; ti < (LOOP ? tpw : NWAVE); ti += NWAVE, q0 += 16 * NWAVE) {
  if (q0 >= q_len) break;
  if (LOOP && ti >= NWAVE) load_q();

  // ---- S^T tiles: s[t][r] = S[q = c][key = t*16 + g*4 + r] ----
  f32x4 s[MAXT];
#pragma unroll
  for (int t = 0; t < MAXT; ++t) {
    s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (t < nt) {
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks + (t * 16 + c) * AM_KPITCH + ks * 32 + g * 8);
        s[t] = mfma16<F16>(kf, qf[ks], s[t]);
      }
    }
  }
  // ---- softmax over keys (fp32) ----
  float mx = -INFINITY;
#pragma unroll
  for (int t = 0; t < MAXT; ++t) {
    if (t < nt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = t * 16 + g * 4 + r;
        float v = (key < kv_len) ? s[t][r] * scale : -INFINITY;
        if (XATTN && key < kv_len && (key < klo || key >= khi)) v = -3.4028234663852886e38f;   // helpers.py:218
        s[t][r] = v;
        mx = fmaxf(mx, v);
      }
    }
  }
  mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < MAXT; ++t) {
    if (t < nt) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __expf(s[t][r] - mx);
        s[t][r] = p;
        sum += p;
      }
    }
  }
  sum += __shfl_xor(sum, 16, 64);
  sum += __shfl_xor(sum, 32, 64);
  const float inv = (XATTN && tt_q == 0) ? 0.f : 1.f / sum;   // helpers.py:223-229: no preceding media -> zero row

  // ---- O^T = V^T * P^T : k-slot (g, j<4) <-> key 32*ch + g*4 + j ; (g, j>=4) <-> key 32*ch + 16 + g*4 + (j-4) ----
  f32x4 o[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ch = 0; ch < MAXT / 2; ++ch) {
    if (ch * 2 < nt) {
      uint4 pw;
      pw.x = pack2x<F16>(s[2 * ch][0], s[2 * ch][1]);
      pw.y = pack2x<F16>(s[2 * ch][2], s[2 * ch][3]);
      pw.z = pack2x<F16>(s[2 * ch + 1][0], s[2 * ch + 1][1]);
      pw.w = pack2x<F16>(s[2 * ch + 1][2], s[2 * ch + 1][3]);
      const bf16x8 pf = __builtin_bit_cast(bf16x8, pw);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const bf16_t* vp = Vt + (dt * 16 + c) * vpitch + ch * 32 + g * 4;
        uint4 vw;
        const uint2 lo = *reinterpret_cast<const uint2*>(vp);
        const uint2 hi = *reinterpret_cast<const uint2*>(vp + 16);
        vw.x = lo.x; vw.y = lo.y; vw.z = hi.x; vw.w = hi.y;
        o[dt] = mfma16<F16>(__builtin_bit_cast(bf16x8, vw), pf, o[dt]);
      }
    }
  }
  // lane holds O[q = q0 + c][d = dt*16 + g*4 .. +3]
  if (q0 + c < q_len) {
    const long off = b * o_bstride + (long)(q0 + c) * ldo + h * AM_HD + g * 4;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      if (out_is_f32)
        *reinterpret_cast<float4*>(reinterpret_cast<float*>(Ov) + off + dt * 16) =
            float4{o[dt][0] * inv, o[dt][1] * inv, o[dt][2] * inv, o[dt][3] * inv};
      else
        *reinterpret_cast<uint2*>(reinterpret_cast<bf16_t*>(Ov) + off + dt * 16) =
            uint2{pack2x<F16>(o[dt][0] * inv, o[dt][1] * inv), pack2x<F16>(o[dt][2] * inv, o[dt][3] * inv)};
    }
  }
  }   // query tiles of this wave
}

// ------------------------------------------------------------------------------------------------
// (1b) ViT self-attention, round 5 (csrc/attention.hip::attn_vit_kernel): the same arithmetic as attn_mfma_kernel<false, ...> - same
// MFMAs in the same order, scores scaled, row maximum subtracted, exp, sum, P rounded to bf16, O scaled by 1 / sum - restructured
// around what the counters said about the old kernel at 16 frames (profiles/r05_b_attn_pmc_before.txt: 24.0 us, MFMA busy 8 %, the
// LDS busy 35 % of the launch with 56 % of those cycles in bank conflicts, 1 590 VALU instructions per wave):
//  * V stays ROW-major in LDS ([key][72], plain 16-byte stores, conflict-free) and the P*V operand is fetched with the gfx950 transpose
//    read ds_read_b64_tr_b16: a 16-lane group hands in the 16 eight-byte pieces of a [4 keys][16 d] block and lane c receives column c
//    (tools/tr_probe.hip) - exactly the k-slot layout the old kernel built with an 8-way bank-conflicted transposing store
//    (32-bit stores of one d of two keys: (8 * pitch / 2) % 32 == 0 puts the 8 segments of a row pair on ONE bank);
//  * all K / V pieces of a thread (9 x 16 B) are requested before the first one is stored: one memory round trip instead of three;
//  * softmax: the maximum is taken over the raw scores, exp2((s - max) * scale * log2 e) is one FMA + one v_exp per score, and only
//    the last key tile is masked against kv_len (half the VALU work per score).
// Results differ from the old kernel by the rounding of exp alone (exp2 of a fused product instead of __expf of a scaled
// difference): both kernels are compared with an fp32 reference in tests/test_hip_ops.py; DEER_ATTN_VIT=0 selects the old kernel.
// ------------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((address_space(3))) s16x4_t lds_s16x4_t;

// SEG2 (round 6): keys [0, kv1) from (Kp, V), keys [kv1, kv_len) from (K2, V2) with their own row pitch / batch stride - the Perceiver's
// [media K/V ; latent K/V] (helpers.py:51 without the concat) at 64 latents x (256 + 64) keys: NT = 20, four waves, MINW = 2 (20 K / V pieces
// + 20 score tiles per lane need more than 128 VGPRs); was attn_mfma_kernel (transposing V stores): 12.3 us per launch
template <int NWAVE, bool LOOP, int NT, bool F16, bool SEG2 = false, int MINW = 4>     // NT = key tiles of 16 (kv_len <= 16 NT); PV runs (NT + 1) / 2 chunks of 32 keys
__global__ __launch_bounds__(64 * NWAVE, MINW) void attn_vit_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ Kp,
                                                              const bf16_t* __restrict__ V, bf16_t* __restrict__ O, int q_len, int kv_len,
                                                              int ldq, int ldk, int ldv, int ldo, long q_bstride, long k_bstride,
                                                              long v_bstride, long o_bstride, float scale_log2e, int tpw,
                                                              const bf16_t* __restrict__ K2, const bf16_t* __restrict__ V2, int kv1, int ld2,
                                                              long bstride2) {
  constexpr int KROWS = NT * 16, NCH = (NT + 1) / 2, VROWS = NCH * 32;
  constexpr int NPIECE = (KROWS + VROWS) * 8, PER = (NPIECE + 64 * NWAVE - 1) / (64 * NWAVE);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* Ks = reinterpret_cast<bf16_t*>(smem_raw);                  // [KROWS][72]
  bf16_t* Vs = Ks + KROWS * AM_KPITCH;                               // [VROWS][72] row-major: read transposed
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c = lane & 15, g = lane >> 4;
  const int h = blockIdx.y, b = blockIdx.z;
  const bf16_t* Kb = Kp + b * k_bstride + h * AM_HD;
  const bf16_t* Vb = V + b * v_bstride + h * AM_HD;
  int q0 = (blockIdx.x * (LOOP ? tpw : NWAVE) + wave) * 16;

  // ---- every K / V piece of this thread in flight, then the Q fragments; rows >= kv_len are zero ----
  uint4 piece[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int p = tid + i * 64 * NWAVE;
    piece[i] = uint4{0, 0, 0, 0};
    if (p < KROWS * 8) {
      const int row = p >> 3, seg = p & 7;
      if (SEG2 && row >= kv1) {
        if (row < kv_len) piece[i] = *reinterpret_cast<const uint4*>(K2 + b * bstride2 + h * AM_HD + (long)(row - kv1) * ld2 + seg * 8);
      } else if (row < kv_len) piece[i] = *reinterpret_cast<const uint4*>(Kb + (long)row * ldk + seg * 8);
    } else if (p < NPIECE) {
      const int row = (p - KROWS * 8) >> 3, seg = p & 7;
      if (SEG2 && row >= kv1) {
        if (row < kv_len) piece[i] = *reinterpret_cast<const uint4*>(V2 + b * bstride2 + h * AM_HD + (long)(row - kv1) * ld2 + seg * 8);
      } else if (row < kv_len) piece[i] = *reinterpret_cast<const uint4*>(Vb + (long)row * ldv + seg * 8);
    }
  }
  bf16x8 qf[2];
  auto load_q = [&]() {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      uint4 v = uint4{0, 0, 0, 0};
      if (q0 + c < q_len) v = *reinterpret_cast<const uint4*>(Q + b * q_bstride + h * AM_HD + (long)(q0 + c) * ldq + ks * 32 + g * 8);
      qf[ks] = __builtin_bit_cast(bf16x8, v);
    }
  };
  load_q();
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int p = tid + i * 64 * NWAVE;
    if (p < NPIECE) *reinterpret_cast<uint4*>(Ks + (p >> 3) * AM_KPITCH + (p & 7) * 8) = piece[i];   // Vs follows Ks: piece p -> row p >> 3 of [K ; V]
  }
  __syncthreads();

  // transpose-read base of this lane: key g*4 + (c >> 2), d piece (c & 3) * 4 of a [4 keys][16 d] block
  lds_s16x4_t* vt = (lds_s16x4_t*)(Vs + (g * 4 + (c >> 2)) * AM_KPITCH + (c & 3) * 4);      // generic -> LDS address space (C-style cast)
#pragma nounroll
  for (int ti = wave; ti < (LOOP ? tpw : NWAVE); ti += NWAVE, q0 += 16 * NWAVE) {
    if (q0 >= q_len) break;
    if (LOOP && ti >= NWAVE) load_q();
    // the K fragments do not depend on the query tile: without this hipcc hoists all 34 of them out of the loop (136 VGPRs -> one
    // workgroup per CU, or 460 bytes of scratch per lane under the two-workgroup register bound)
    if (LOOP) asm volatile("" ::: "memory");
    // ---- S^T tiles: s[t][r] = S[q = c][key = t*16 + g*4 + r] (unscaled) ----
    f32x4 s[2 * NCH];
#pragma unroll
    for (int t = 0; t < 2 * NCH; ++t) s[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    // the two halves of the head dimension as the OUTER loop: 17 independent MFMAs in a row, each accumulator's second MFMA 17 issues
    // after its first (the same sum per score as t-outer: ks = 0 then ks = 1)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const bf16x8 kf = *reinterpret_cast<const bf16x8*>(Ks + (t * 16 + c) * AM_KPITCH + ks * 32 + g * 8);
        s[t] = mfma16<F16>(kf, qf[ks], s[t]);
      }
    // keys beyond kv_len live in the last tile only (kv_len > 16 (NT - 1), checked by the launcher)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if ((NT - 1) * 16 + g * 4 + r >= kv_len) s[NT - 1][r] = -INFINITY;
    // SEG2 (Perceiver): the arithmetic of attn_mfma_kernel, which this instantiation replaces - scores scaled first, __expf of the
    // difference (`scale_log2e` carries the plain scale there) - so that its results are the old kernel's bit for bit (the bf16
    // arithmetic sits 5 % under its hard-input gate: tests/test_hard_inputs.py); the ViT form keeps its fused exp2
    if constexpr (SEG2) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) s[t][r] *= scale_log2e;   // -inf stays -inf
    }
    float mx = -INFINITY;
#pragma unroll
    for (int t = 0; t < NT; ++t) mx = fmaxf(fmaxf(fmaxf(mx, s[t][0]), fmaxf(s[t][1], s[t][2])), s[t][3]);
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mneg = -mx * scale_log2e;
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float p;
        if constexpr (SEG2) p = __expf(s[t][r] - mx);
        else p = __builtin_amdgcn_exp2f(__builtin_fmaf(s[t][r], scale_log2e, mneg));
        s[t][r] = p;
        sum += p;
      }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.f / sum;
    // ---- O^T = V^T * P^T : k-slot (g, j<4) <-> key 32*ch + g*4 + j ; (g, j>=4) <-> key 32*ch + 16 + g*4 + (j-4) ----
    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ch = 0; ch < NCH; ++ch) {
      uint4 pw;
      pw.x = pack2x<F16>(s[2 * ch][0], s[2 * ch][1]);
      pw.y = pack2x<F16>(s[2 * ch][2], s[2 * ch][3]);
      pw.z = pack2x<F16>(s[2 * ch + 1][0], s[2 * ch + 1][1]);      // tile NT (odd NT: beyond the keys) holds zeros
      pw.w = pack2x<F16>(s[2 * ch + 1][2], s[2 * ch + 1][3]);
      const bf16x8 pf = __builtin_bit_cast(bf16x8, pw);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const s16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(vt + ((ch * 32) * AM_KPITCH + dt * 16) / 4);
        const s16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(vt + ((ch * 32 + 16) * AM_KPITCH + dt * 16) / 4);
        const uint2 l2 = __builtin_bit_cast(uint2, lo), h2 = __builtin_bit_cast(uint2, hi);
        const uint4 vw = uint4{l2.x, l2.y, h2.x, h2.y};
        o[dt] = mfma16<F16>(__builtin_bit_cast(bf16x8, vw), pf, o[dt]);
      }
    }
    if (q0 + c < q_len) {                                    // lane holds O[q = q0 + c][d = dt*16 + g*4 .. +3]
      bf16_t* op = O + b * o_bstride + (long)(q0 + c) * ldo + h * AM_HD + g * 4;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        *reinterpret_cast<uint2*>(op + dt * 16) = uint2{pack2x<F16>(o[dt][0] * inv, o[dt][1] * inv), pack2x<F16>(o[dt][2] * inv, o[dt][3] * inv)};
    }
  }
}

template <int NWAVE, bool LOOP, int NT, bool F16, bool SEG2 = false, int MINW = 4>
static int launch_attn_vit_nt(dim3 grid, hipStream_t st, const bf16_t* Q, const bf16_t* K, const bf16_t* V, bf16_t* O, int q_len, int kv_len, int ldq,
                              int ldk, int ldv, int ldo, long q_bstride, long k_bstride, long v_bstride, long o_bstride, float scale, int tpw,
                              const bf16_t* K2 = nullptr, const bf16_t* V2 = nullptr, int kv1 = 0, int ld2 = 0, long bstride2 = 0) {
  constexpr int smem = (NT * 16 + ((NT + 1) / 2) * 32) * AM_KPITCH * 2;
  static std::atomic<bool> attr_set{false};
  auto kern = &attn_vit_kernel<NWAVE, LOOP, NT, F16, SEG2, MINW>;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem) != hipSuccess) return DEER_ERR_LAUNCH;
    attr_set = true;
  }
  hipLaunchKernelGGL(kern, grid, dim3(64 * NWAVE), smem, st, Q, K, V, O, q_len, kv_len, ldq, ldk, ldv, ldo, q_bstride, k_bstride, v_bstride, o_bstride,
                     SEG2 ? scale : scale * 1.4426950408889634f, tpw, K2, V2, kv1, ld2, bstride2);
  DEER_LAUNCH_CHECK();
  return DEER_OK;
}

template <bool F16 = false>
static int launch_attn_mfma(const void* Q, const void* K, const void* V, void* O, int batch, int heads, int q_len, int kv_len,
                            int ldq, int ldk, int ldv, int ldo, long q_bstride, long k_bstride, long v_bstride, long o_bstride,
                            float scale, int q_slabs, long q_slab_stride, const int* text_time, int n_per_media, int out_is_f32,
                            const int* ctl, void* stream, const void* K2 = nullptr, const void* V2 = nullptr, int kv1 = -1,
                            int ld2 = 0, long bstride2 = 0) {
  // long key sequences (pre fusion, flamingo_mpt.py:585-607: the 64 latents attend 2 x 256 patch tokens + themselves = 576 keys): the same
  // kernel with 36 key tiles in registers (144 score VGPRs; one workgroup per CU: 154 KB of LDS for K and V^T) - <= 64 queries, no extras
  const bool long_kv = kv_len > AM_MAXT * 16;
  if (Q == nullptr || K == nullptr || V == nullptr || O == nullptr) return DEER_ERR_SHAPE;
  if (q_len <= 0 || kv_len <= 0 || kv_len > AM_MAXT_LONG * 16 || (ldq & 7) || (ldk & 7) || (ldv & 7) || (ldo & 3))
    return DEER_ERR_SHAPE;
  if (long_kv && (q_slabs > 0 || text_time != nullptr || out_is_f32 || ctl != nullptr)) return DEER_ERR_SHAPE;
  if (K2 == nullptr) kv1 = kv_len;                         // single segment
  else if (V2 == nullptr || kv1 < 0 || kv1 > kv_len || (ld2 & 7)) return DEER_ERR_SHAPE;
  const bf16_t* k2 = reinterpret_cast<const bf16_t*>(K2);
  const bf16_t* v2 = reinterpret_cast<const bf16_t*>(V2);
  const int kvpad = (kv_len + 31) & ~31;
  const int smem = (kvpad * AM_KPITCH + AM_HD * (kvpad + 8)) * (int)sizeof(bf16_t);
  if (long_kv) {
    static std::atomic<bool> long_attr{false};
    constexpr int max_long = (AM_MAXT_LONG * 16 * AM_KPITCH + AM_HD * (AM_MAXT_LONG * 16 + 8)) * 2;
    if (!long_attr) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_mfma_kernel<false, 4, false, F16, AM_MAXT_LONG>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              max_long) != hipSuccess)
        return DEER_ERR_LAUNCH;
      long_attr = true;
    }
    const int tiles_q = (q_len + 15) / 16;
    hipLaunchKernelGGL((attn_mfma_kernel<false, 4, false, F16, AM_MAXT_LONG>), dim3((tiles_q + 3) / 4, heads, batch), dim3(256), smem, reinterpret_cast<hipStream_t>(stream),
                       Q, reinterpret_cast<const bf16_t*>(K), reinterpret_cast<const bf16_t*>(V), O, q_len, kv_len, ldq, ldk, ldv, ldo, q_bstride, k_bstride,
                       v_bstride, o_bstride, scale, 0, 0L, nullptr, 1, 0, nullptr, k2, v2, kv1, ld2, bstride2, 4);
    DEER_LAUNCH_CHECK();
    return DEER_OK;
  }
  static std::atomic<bool> attr_set{false};
  constexpr int max_smem = (AM_MAXT * 16 * AM_KPITCH + AM_HD * (AM_MAXT * 16 + 8)) * 2;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_mfma_kernel<false, 4, false, F16>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            max_smem) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_mfma_kernel<false, 8, false, F16>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            max_smem) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_mfma_kernel<false, 8, true, F16>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            max_smem) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_mfma_kernel<true, 4, false, F16>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            max_smem) != hipSuccess)
      return DEER_ERR_LAUNCH;
    attr_set = true;
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const bf16_t* kp = reinterpret_cast<const bf16_t*>(K);
  const bf16_t* vp = reinterpret_cast<const bf16_t*>(V);
  // 8 waves (128 queries) per workgroup when there are many queries: the K/V staging of a head is shared by twice as many
  // queries and is pulled by twice as many waves (the per-CU fill rate grows with the number of waves issuing loads)
  static const bool wide_ok = [] { const char* e = getenv("DEER_ATTN_WIDE"); return e == nullptr || e[0] != '0'; }();
  const bool wide = wide_ok && q_slabs == 0 && q_len > 64;
  // env batches (>= 128 (head, image) pairs: the chip is full with ONE workgroup per pair): every wave walks several query tiles
  // and the head's K/V are staged once; at one environment (16-32 pairs) the queries stay spread over more workgroups
  static const bool loop_ok = [] { const char* e = getenv("DEER_ATTN_LOOP"); return e == nullptr || e[0] != '0'; }();
  const int nwave = wide ? 8 : 4, tiles = (q_len + 15) / 16;
  int tpw = nwave;
  if (loop_ok && wide && (long)heads * batch >= 128) {
    const int n_wg = (tiles + nwave) / (nwave + 1);          // at most one extra tile for a workgroup's first wave(s)
    tpw = (tiles + n_wg - 1) / n_wg;
  }
  dim3 grid((tiles + tpw - 1) / tpw, heads, batch);
  // ViT-L/14 self-attention (bf16 q / k / v of one segment, bf16 output, 257 keys = 17 key tiles): attn_vit_kernel
  static const bool vit_ok = [] { const char* e = getenv("DEER_ATTN_VIT"); return e == nullptr || e[0] != '0'; }();
  if (vit_ok && wide && K2 == nullptr && text_time == nullptr && !out_is_f32 && ctl == nullptr && (kv_len + 15) / 16 == 17) {
    const bf16_t* q = reinterpret_cast<const bf16_t*>(Q);
    bf16_t* o = reinterpret_cast<bf16_t*>(O);
#define DEER_VIT_ARGS grid, st, q, kp, vp, o, q_len, kv_len, ldq, ldk, ldv, ldo, q_bstride, k_bstride, v_bstride, o_bstride, scale, tpw
    // env batches (>= 128 (head, image) pairs): ONE 16-wave workgroup per pair walks all 17 query tiles - the head's K / V are staged once
    // (DEER_ATTN_W16=0: two 8-wave workgroups per pair, each staging them)
    static const bool w16 = [] { const char* e = getenv("DEER_ATTN_W16"); return e == nullptr || e[0] != '0'; }();
    if (tpw != nwave && w16) {
      grid = dim3(1, heads, batch);
      tpw = tiles;
      return launch_attn_vit_nt<16, true, 17, F16>(DEER_VIT_ARGS);
    }
    return tpw != nwave ? launch_attn_vit_nt<8, true, 17, F16>(DEER_VIT_ARGS) : launch_attn_vit_nt<8, false, 17, F16>(DEER_VIT_ARGS);
#undef DEER_VIT_ARGS
  }
  // Perceiver attention (64 latents over [256 media tokens ; 64 latents] = 20 key tiles, two segments): the same restructured kernel
  if (vit_ok && !wide && K2 != nullptr && q_slabs == 0 && text_time == nullptr && !out_is_f32 && ctl == nullptr && (kv_len + 15) / 16 == 20 &&
      kv_len > 16 * 19 && tpw == nwave)
    return launch_attn_vit_nt<4, false, 20, F16, true, 2>(grid, st, reinterpret_cast<const bf16_t*>(Q), kp, vp, reinterpret_cast<bf16_t*>(O), q_len, kv_len, ldq,
                                                          ldk, ldv, ldo, q_bstride, k_bstride, v_bstride, o_bstride, scale, tpw, k2, v2, kv1, ld2, bstride2);
#define DEER_ATTN_ARGS Q, kp, vp, O, q_len, kv_len, ldq, ldk, ldv, ldo, q_bstride, k_bstride, v_bstride, o_bstride, scale, q_slabs, \
                       q_slab_stride, text_time, n_per_media, out_is_f32, ctl, k2, v2, kv1, ld2, bstride2, tpw
  if (q_slabs > 0) {
    hipLaunchKernelGGL((attn_mfma_kernel<true, 4, false, F16>), grid, dim3(256), smem, st, DEER_ATTN_ARGS);
  } else if (wide && tpw != nwave)
    hipLaunchKernelGGL((attn_mfma_kernel<false, 8, true, F16>), grid, dim3(512), smem, st, DEER_ATTN_ARGS);
  else if (wide)
    hipLaunchKernelGGL((attn_mfma_kernel<false, 8, false, F16>), grid, dim3(512), smem, st, DEER_ATTN_ARGS);
  else
    hipLaunchKernelGGL((attn_mfma_kernel<false, 4, false, F16>), grid, dim3(256), smem, st, DEER_ATTN_ARGS);
#undef DEER_ATTN_ARGS
  DEER_LAUNCH_CHECK();
  return DEER_OK;
}

extern "C" int deer_attn_mfma_hd64(const void* Q, const void* K, const void* V, void* O, int batch, int heads,
                                   int q_len, int kv_len, int ldq, int ldk, int ldv, int ldo, long q_bstride,
                                   long k_bstride, long v_bstride, long o_bstride, float scale, void* stream) {
  return launch_attn_mfma(Q, K, V, O, batch, heads, q_len, kv_len, ldq, ldk, ldv, ldo, q_bstride, k_bstride, v_bstride, o_bstride,
                          scale, 0, 0, nullptr, 1, 0, nullptr, stream);
}

// Keys/values in two segments: keys [0, kv1) from (K1, V1), keys [kv1, kv1+kv2) from (K2, V2).  PerceiverAttention
// (helpers.py:47-73) attends over [media tokens ; latents]; the media K/V of every layer are produced up front by one
// batched GEMM (media tokens are layer-invariant), the latent K/V by the per-layer q|k|v projection of the latents.
extern "C" int deer_attn_mfma_hd64_2seg(const void* Q, const void* K1, const void* V1, const void* K2, const void* V2, void* O,
                                        int batch, int heads, int q_len, int kv1, int kv2, int ldq, int ld1, int ld2, int ldo,
                                        long q_bstride, long bstride1, long bstride2, long o_bstride, float scale,
                                        void* stream) {
  if (K2 == nullptr || V2 == nullptr || kv1 < 0 || kv2 <= 0) return DEER_ERR_SHAPE;
  return launch_attn_mfma(Q, K1, V1, O, batch, heads, q_len, kv1 + kv2, ldq, ld1, ld1, ldo, q_bstride, bstride1, bstride1, o_bstride,
                          scale, 0, 0, nullptr, 1, 0, nullptr, stream, K2, V2, kv1, ld2, bstride2);
}

// ---- the same two entry points on fp16 q / k / v with an fp16 result (round 6: the vision tower's fp16 arithmetic) ----
extern "C" int deer_attn_f16_hd64(const void* Q, const void* K, const void* V, void* O, int batch, int heads,
                                   int q_len, int kv_len, int ldq, int ldk, int ldv, int ldo, long q_bstride,
                                   long k_bstride, long v_bstride, long o_bstride, float scale, void* stream) {
  return launch_attn_mfma<true>(Q, K, V, O, batch, heads, q_len, kv_len, ldq, ldk, ldv, ldo, q_bstride, k_bstride, v_bstride, o_bstride,
                          scale, 0, 0, nullptr, 1, 0, nullptr, stream);
}

// Keys/values in two segments: keys [0, kv1) from (K1, V1), keys [kv1, kv1+kv2) from (K2, V2).  PerceiverAttention
// (helpers.py:47-73) attends over [media tokens ; latents]; the media K/V of every layer are produced up front by one
// batched GEMM (media tokens are layer-invariant), the latent K/V by the per-layer q|k|v projection of the latents.
extern "C" int deer_attn_f16_hd64_2seg(const void* Q, const void* K1, const void* V1, const void* K2, const void* V2, void* O,
                                        int batch, int heads, int q_len, int kv1, int kv2, int ldq, int ld1, int ld2, int ldo,
                                        long q_bstride, long bstride1, long bstride2, long o_bstride, float scale,
                                        void* stream) {
  if (K2 == nullptr || V2 == nullptr || kv1 < 0 || kv2 <= 0) return DEER_ERR_SHAPE;
  return launch_attn_mfma<true>(Q, K1, V1, O, batch, heads, q_len, kv1 + kv2, ldq, ld1, ld1, ldo, q_bstride, bstride1, bstride1, o_bstride,
                          scale, 0, 0, nullptr, 1, 0, nullptr, stream, K2, V2, kv1, ld2, bstride2);
}

// MaskedCrossAttention core on the MFMA kernel: q = sum_s qslab[s] (f32 [batch*T, ldqs], head h at column h*64),
// kv bf16 [batch*n_kv, ldkv] (k at column h*64, v at inner + h*64), media mask from text_time, out f32/bf16 [batch*T, ldo].
template <bool F16>
static int xattn_mfma(const float* qslab, int s_in, long slab_stride, int ldqs, const void* kv, int ldkv, int inner,
                               const int* text_time, int n_per_media, void* out, int out_is_f32, int ldo, int T, int n_kv,
                               int heads, int batch, float scale, const int* ctl, void* stream) {
  if (s_in <= 0 || n_per_media <= 0 || text_time == nullptr) return DEER_ERR_SHAPE;
  const bf16_t* kvp = reinterpret_cast<const bf16_t*>(kv);
  return launch_attn_mfma<F16>(qslab, kvp, kvp + inner, out, batch, heads, T, n_kv, ldqs, ldkv, ldkv, ldo, (long)T * ldqs,
                          (long)n_kv * ldkv, (long)n_kv * ldkv, (long)T * ldo, scale, s_in, slab_stride, text_time, n_per_media,
                          out_is_f32, ctl, stream);
}

extern "C" int deer_xattn_mfma(const float* qslab, int s_in, long slab_stride, int ldqs, const void* kv, int ldkv, int inner,
                               const int* text_time, int n_per_media, void* out, int out_is_f32, int ldo, int T, int n_kv,
                               int heads, int batch, float scale, const int* ctl, void* stream) {
  return xattn_mfma<false>(qslab, s_in, slab_stride, ldqs, kv, ldkv, inner, text_time, n_per_media, out, out_is_f32, ldo, T, n_kv, heads, batch, scale, ctl, stream);
}
extern "C" int deer_xattn_mfma_f16(const float* qslab, int s_in, long slab_stride, int ldqs, const void* kv, int ldkv, int inner,
                                   const int* text_time, int n_per_media, void* out, int out_is_f32, int ldo, int T, int n_kv,
                                   int heads, int batch, float scale, const int* ctl, void* stream) {   // fp16 K / V (round 6)
  return xattn_mfma<true>(qslab, s_in, slab_stride, ldqs, kv, ldkv, inner, text_time, n_per_media, out, out_is_f32, ldo, T, n_kv, heads, batch, scale, ctl, stream);
}

// ------------------------------------------------------------------------------------------------
// (2) gated cross-attention core (open_flamingo/src/helpers.py:184-233): per head
//     q = sum_s qslab[s][t][h*64..] * 64^-0.5 ; sim = q k^T ; mask: text_time[t] == media_time[j]
//     (only_attend_immediate_media) ; softmax ; rows with text_time == 0 zeroed ; out = attn v.
// kv: bf16 [n_kv][ldkv] with k at column h*64 and v at column inner + h*64.
// ------------------------------------------------------------------------------------------------
#define XA_MAXT 32
#define XA_MAXKV 128
__global__ __launch_bounds__(256) void xattn_small_kernel(const float* __restrict__ qslab, int s_in, long slab_stride,
                                                          int ldqs, const bf16_t* __restrict__ kv, int ldkv, int inner,
                                                          const int* __restrict__ text_time, int n_per_media,
                                                          void* __restrict__ out, int out_is_f32, int ldo, int T,
                                                          int n_kv, float scale, const int* ctl) {
  DEER_RETURN_IF_EXITED(ctl);
  __shared__ float qs[XA_MAXT][64 + 1];
  __shared__ float sim[XA_MAXT][XA_MAXKV + 1];
  __shared__ bf16_t ks[XA_MAXKV][64 + 2];
  __shared__ bf16_t vs[XA_MAXKV][64 + 2];
  const int h = blockIdx.x, tid = threadIdx.x;
  // environment blockIdx.y: its T query rows, its n_kv media rows, its text_time entries
  qslab += (long)blockIdx.y * T * ldqs;
  kv += (long)blockIdx.y * n_kv * ldkv;
  text_time += blockIdx.y * T;
  out = out_is_f32 ? (void*)(reinterpret_cast<float*>(out) + (long)blockIdx.y * T * ldo)
                   : (void*)(reinterpret_cast<bf16_t*>(out) + (long)blockIdx.y * T * ldo);
  // q: split-K reduce, 4 floats per thread
  for (int idx = tid; idx < T * 16; idx += 256) {
    const int t = idx >> 4, d4 = (idx & 15) * 4;
    const float* p = qslab + (long)t * ldqs + h * 64 + d4;
    const float4 a = slab_sum4(p, s_in, slab_stride);
    qs[t][d4] = a.x * scale; qs[t][d4 + 1] = a.y * scale; qs[t][d4 + 2] = a.z * scale; qs[t][d4 + 3] = a.w * scale;
  }
  // k, v of this head: 16-byte loads (8 bf16), 8 per media row
  for (int idx = tid; idx < n_kv * 16; idx += 256) {
    const int j = idx >> 4, part = idx & 15, isv = part >> 3, seg = (part & 7) * 8;
    const uint4 v = *reinterpret_cast<const uint4*>(kv + (long)j * ldkv + (isv ? inner : 0) + h * 64 + seg);
    bf16_t* dst = (isv ? &vs[j][seg] : &ks[j][seg]);
    const uint32_t u[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) reinterpret_cast<uint32_t*>(dst)[e] = u[e];
  }
  __syncthreads();
  for (int idx = tid; idx < T * n_kv; idx += 256) {
    const int t = idx / n_kv, j = idx - t * n_kv;
    float a = 0.f;
#pragma unroll 16
    for (int d = 0; d < 64; ++d) a += qs[t][d] * bf2f(ks[j][d]);
    const int media_time = j / n_per_media + 1;
    if (text_time[t] != media_time) a = -3.4028234663852886e38f;      // -finfo.max (helpers.py:218)
    sim[t][j] = a;
  }
  __syncthreads();
  // softmax per row: one wave per row
  const int lane = tid & 63, wave = tid >> 6;
  for (int t = wave; t < T; t += 4) {
    float mx = -INFINITY;
    for (int j = lane; j < n_kv; j += 64) mx = fmaxf(mx, sim[t][j]);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int j = lane; j < n_kv; j += 64) {
      const float p = expf(sim[t][j] - mx);
      sim[t][j] = p;
      sum += p;
    }
    sum = wave_sum(sum);
    const float inv = (text_time[t] == 0) ? 0.f : 1.f / sum;           // helpers.py:223-229
    for (int j = lane; j < n_kv; j += 64) sim[t][j] *= inv;
  }
  __syncthreads();
  for (int idx = tid; idx < T * 64; idx += 256) {
    const int t = idx >> 6, d = idx & 63;
    float a = 0.f;
    for (int j = 0; j < n_kv; ++j) a += sim[t][j] * bf2f(vs[j][d]);
    if (out_is_f32) reinterpret_cast<float*>(out)[(long)t * ldo + h * 64 + d] = a;
    else reinterpret_cast<bf16_t*>(out)[(long)t * ldo + h * 64 + d] = f2bf(a);
  }
}

extern "C" int deer_xattn_small(const float* qslab, int s_in, long slab_stride, int ldqs, const void* kv, int ldkv,
                                int inner, const int* text_time, int n_per_media, void* out, int out_is_f32, int ldo, int T,
                                int n_kv, int heads, int batch, float scale, const int* ctl, void* stream) {
  if (T <= 0 || T > XA_MAXT || n_kv <= 0 || n_kv > XA_MAXKV || s_in <= 0 || n_per_media <= 0 || batch <= 0) return DEER_ERR_SHAPE;
  hipLaunchKernelGGL(xattn_small_kernel, dim3(heads, batch), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), qslab, s_in,
                     slab_stride, ldqs, reinterpret_cast<const bf16_t*>(kv), ldkv, inner, text_time, n_per_media, out,
                     out_is_f32, ldo, T, n_kv, scale, ctl);
  DEER_LAUNCH_CHECK();
  return DEER_OK;
}

// ------------------------------------------------------------------------------------------------
// (3) MPT self-attention, two launches:
//   qkv_reduce_ln_kernel : qkv[t] = sum_s slab[s][t][0..3d); optional LayerNorm over the FULL d_model of q and
//                          of k (attn_qk_ln, weight only).  grid (T, 3) - one workgroup per (row, q|k|v).
//   mpt_attn_small_kernel: per head h: softmax(q k^T / sqrt(hd) + alibi + key-pad + causal) v, head_dim <= 128,
//                          alibi[h][j] = -(T-1-j) * 2^(-bias_max*(h+1)/H).
// ------------------------------------------------------------------------------------------------
#define MA_MAXT 32
__global__ __launch_bounds__(256) void qkv_reduce_ln_kernel(const float* __restrict__ slab, int s_in, long slab_stride,
                                                            int d, const float* __restrict__ q_ln_w,
                                                            const float* __restrict__ k_ln_w, float eps,
                                                            float* __restrict__ qkv, const int* ctl,
                                                            const int* __restrict__ cmap = nullptr, int rows_per_env = 0) {
  DEER_RETURN_IF_EXITED(ctl);
  if (cmap != nullptr && (int)blockIdx.x >= cmap[CMAP_N] * rows_per_env) return;      // rows of exited environments (compaction)
  __shared__ float red[16];
  const int t = blockIdx.x, part = blockIdx.y;
  const long base = (long)t * 3 * d + (long)part * d;
  const int n4 = d >> 2;
  float4 v[4];                                           // d <= 4096
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    v[j] = float4{0.f, 0.f, 0.f, 0.f};
    const int i4 = threadIdx.x + j * 256;
    if (i4 < n4) {
      v[j] = slab_sum4(slab + base + (long)i4 * 4, s_in, slab_stride);
    }
  }
  const float* w = (part == 0) ? q_ln_w : (part == 1 ? k_ln_w : nullptr);
  if (w != nullptr) {
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) sum += v[j].x + v[j].y + v[j].z + v[j].w;      // lanes beyond n4 hold zeros
    const float mean = block_sum(sum, red) / d;
    float var = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (threadIdx.x + j * 256 < n4) {
        const float a = v[j].x - mean, b = v[j].y - mean, c = v[j].z - mean, e = v[j].w - mean;
        var += a * a + b * b + c * c + e * e;
      }
    const float rstd = rsqrtf(block_sum(var, red) / d + eps);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int i4 = threadIdx.x + j * 256;
      if (i4 < n4) {
        const float4 g = *reinterpret_cast<const float4*>(w + (long)i4 * 4);
        v[j].x = (v[j].x - mean) * rstd * g.x; v[j].y = (v[j].y - mean) * rstd * g.y;
        v[j].z = (v[j].z - mean) * rstd * g.z; v[j].w = (v[j].w - mean) * rstd * g.w;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int i4 = threadIdx.x + j * 256;
    if (i4 < n4) *reinterpret_cast<float4*>(qkv + base + (long)i4 * 4) = v[j];
  }
}

template <bool F16 = false>
__global__ __launch_bounds__(256) void mpt_attn_small_kernel(const float* __restrict__ qkv, int d_model, int hd,
                                                             const unsigned char* __restrict__ key_mask,
                                                             float alibi_slope_base, int n_heads,
                                                             void* __restrict__ out, int out_is_f32, int ldo, int T,
                                                             const int* ctl, bf16_t* __restrict__ out_lo,
                                                             const int* __restrict__ cmap = nullptr) {
  DEER_RETURN_IF_EXITED(ctl);
  // compaction: blockIdx.y is a SLOT; the key-padding mask is indexed by the slot's environment
  if (cmap != nullptr && (int)blockIdx.y >= cmap[CMAP_N]) return;
  const int env_of_slot = cmap != nullptr ? cmap[CMAP_SLOT_ENV + blockIdx.y] : (int)blockIdx.y;
  __shared__ __attribute__((aligned(16))) float qs[MA_MAXT][128 + 4];
  __shared__ __attribute__((aligned(16))) float ks[MA_MAXT][128 + 4];
  __shared__ __attribute__((aligned(16))) float vs[MA_MAXT][128 + 4];
  __shared__ float sim[MA_MAXT][MA_MAXT + 1];
  const int h = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ld = 3 * d_model;
  qkv += (long)blockIdx.y * T * ld;                        // environment blockIdx.y
  if (key_mask != nullptr) key_mask += env_of_slot * T;
  out = out_is_f32 ? (void*)(reinterpret_cast<float*>(out) + (long)blockIdx.y * T * ldo)
                   : (void*)(reinterpret_cast<bf16_t*>(out) + (long)blockIdx.y * T * ldo);
  if (out_lo != nullptr) out_lo += (long)blockIdx.y * T * ldo;
  const int hd4 = hd >> 2;
  for (int idx = tid; idx < T * hd4; idx += 256) {
    const int t = idx / hd4, d = (idx - t * hd4) * 4;
    const float* p = qkv + (long)t * ld + h * hd + d;
    *reinterpret_cast<float4*>(&qs[t][d]) = *reinterpret_cast<const float4*>(p);
    *reinterpret_cast<float4*>(&ks[t][d]) = *reinterpret_cast<const float4*>(p + d_model);
    *reinterpret_cast<float4*>(&vs[t][d]) = *reinterpret_cast<const float4*>(p + 2 * d_model);
  }
  __syncthreads();
  const float sc = rsqrtf((float)hd);
  const float slope = exp2f(-alibi_slope_base * (float)(h + 1) / (float)n_heads);
  for (int idx = tid; idx < T * T; idx += 256) {
    const int i = idx / T, j = idx - i * T;
    float a = 0.f;
    if (j <= i)
      for (int d = 0; d < hd; d += 4) {
        const float4 qv = *reinterpret_cast<const float4*>(&qs[i][d]);
        const float4 kv = *reinterpret_cast<const float4*>(&ks[j][d]);
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
    const int t = idx / hd, d = idx - t * hd;
    float a = 0.f;
    for (int j = 0; j <= t; ++j) a += sim[t][j] * vs[j][d];
    if (out_is_f32) reinterpret_cast<float*>(out)[(long)t * ldo + h * hd + d] = a;
    else {
      const bf16_t hi = f2x<F16>(a);
      reinterpret_cast<bf16_t*>(out)[(long)t * ldo + h * hd + d] = hi;
      if (out_lo != nullptr) out_lo[(long)t * ldo + h * hd + d] = f2x<F16>(a - x2f<F16>(hi));    // second plane: a = hi + lo
    }
  }
}

// qkv_ws: f32 [T, 3*d_model] workspace holding the reduced (and q/k-normalised) qkv.
extern "C" int deer_mpt_attn_small(const float* qkvslab, int s_in, long slab_stride, int d_model, int n_heads,
                                   const float* q_ln_w, const float* k_ln_w, float eps, const unsigned char* key_mask,
                                   float alibi_bias_max, float* qkv_ws, void* out, int out_is_f32, int ldo, int T, int batch,
                                   const int* ctl, void* stream) {
  const int hd = d_model / n_heads;
  if (T <= 0 || T > MA_MAXT || hd > 128 || (hd & 3) || hd * n_heads != d_model || s_in <= 0 || (d_model & 3) || d_model > 4096 ||
      qkv_ws == nullptr || batch <= 0)
    return DEER_ERR_SHAPE;
  if ((q_ln_w == nullptr) != (k_ln_w == nullptr)) return DEER_ERR_SHAPE;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(qkv_reduce_ln_kernel, dim3(T * batch, 3), dim3(256), 0, st, qkvslab, s_in, slab_stride, d_model, q_ln_w, k_ln_w,
                     eps, qkv_ws, ctl);
  hipLaunchKernelGGL(mpt_attn_small_kernel<false>, dim3(n_heads, batch), dim3(256), 0, st, qkv_ws, d_model, hd, key_mask, alibi_bias_max,
                     n_heads, out, out_is_f32, ldo, T, ctl, static_cast<bf16_t*>(nullptr), static_cast<const int*>(nullptr));
  DEER_LAUNCH_CHECK();
  return DEER_OK;
}

// deer_mpt_attn_small_hl for an env batch with compaction: slots of `cmap` (T rows each) instead of environments
template <bool F16>
static int mpt_attn_small_hl_active(const float* qkvslab, int s_in, long slab_stride, int d_model, int n_heads, const float* q_ln_w,
                                             const float* k_ln_w, float eps, const unsigned char* key_mask, float alibi_bias_max, float* qkv_ws,
                                             void* out_hi, void* out_lo, int ldo, int T, int batch, const int* ctl, const int* cmap, void* stream) {
  const int hd = d_model / n_heads;
  if (T <= 0 || T > MA_MAXT || hd > 128 || (hd & 3) || hd * n_heads != d_model || s_in <= 0 || (d_model & 3) || d_model > 4096 ||
      qkv_ws == nullptr || batch <= 0 || out_hi == nullptr || out_lo == nullptr || cmap == nullptr)
    return DEER_ERR_SHAPE;
  if ((q_ln_w == nullptr) != (k_ln_w == nullptr)) return DEER_ERR_SHAPE;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(qkv_reduce_ln_kernel, dim3(T * batch, 3), dim3(256), 0, st, qkvslab, s_in, slab_stride, d_model, q_ln_w, k_ln_w,
                     eps, qkv_ws, ctl, cmap, T);
  hipLaunchKernelGGL(mpt_attn_small_kernel<F16>, dim3(n_heads, batch), dim3(256), 0, st, qkv_ws, d_model, hd, key_mask, alibi_bias_max,
                     n_heads, out_hi, 0, ldo, T, ctl, reinterpret_cast<bf16_t*>(out_lo), cmap);
  DEER_LAUNCH_CHECK();
  return DEER_OK;
}

// the same op with the output as two bf16 planes hi / lo [batch*T, ldo] (activation operand of deer_gemm_skinny_hl)
template <bool F16>
static int mpt_attn_small_hl(const float* qkvslab, int s_in, long slab_stride, int d_model, int n_heads,
                                      const float* q_ln_w, const float* k_ln_w, float eps, const unsigned char* key_mask,
                                      float alibi_bias_max, float* qkv_ws, void* out_hi, void* out_lo, int ldo, int T, int batch,
                                      const int* ctl, void* stream) {
  const int hd = d_model / n_heads;
  if (T <= 0 || T > MA_MAXT || hd > 128 || (hd & 3) || hd * n_heads != d_model || s_in <= 0 || (d_model & 3) || d_model > 4096 ||
      qkv_ws == nullptr || batch <= 0 || out_hi == nullptr || out_lo == nullptr)
    return DEER_ERR_SHAPE;
  if ((q_ln_w == nullptr) != (k_ln_w == nullptr)) return DEER_ERR_SHAPE;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(qkv_reduce_ln_kernel, dim3(T * batch, 3), dim3(256), 0, st, qkvslab, s_in, slab_stride, d_model, q_ln_w, k_ln_w,
                     eps, qkv_ws, ctl);
  hipLaunchKernelGGL(mpt_attn_small_kernel<F16>, dim3(n_heads, batch), dim3(256), 0, st, qkv_ws, d_model, hd, key_mask, alibi_bias_max,
                     n_heads, out_hi, 0, ldo, T, ctl, reinterpret_cast<bf16_t*>(out_lo), static_cast<const int*>(nullptr));
  DEER_LAUNCH_CHECK();
  return DEER_OK;
}

extern "C" int deer_mpt_attn_small_hl_active(const float* qkvslab, int s_in, long slab_stride, int d_model, int n_heads, const float* q_ln_w,
                                             const float* k_ln_w, float eps, const unsigned char* key_mask, float alibi_bias_max, float* qkv_ws,
                                             void* out_hi, void* out_lo, int ldo, int T, int batch, const int* ctl, const int* cmap, void* stream) {
  return mpt_attn_small_hl_active<false>(qkvslab, s_in, slab_stride, d_model, n_heads, q_ln_w, k_ln_w, eps, key_mask, alibi_bias_max, qkv_ws, out_hi, out_lo, ldo, T,
                                         batch, ctl, cmap, stream);
}
extern "C" int deer_mpt_attn_small_hl_active_f16(const float* qkvslab, int s_in, long slab_stride, int d_model, int n_heads, const float* q_ln_w,
                                                 const float* k_ln_w, float eps, const unsigned char* key_mask, float alibi_bias_max, float* qkv_ws,
                                                 void* out_hi, void* out_lo, int ldo, int T, int batch, const int* ctl, const int* cmap, void* stream) {
  return mpt_attn_small_hl_active<true>(qkvslab, s_in, slab_stride, d_model, n_heads, q_ln_w, k_ln_w, eps, key_mask, alibi_bias_max, qkv_ws, out_hi, out_lo, ldo, T,
                                        batch, ctl, cmap, stream);
}
extern "C" int deer_mpt_attn_small_hl(const float* qkvslab, int s_in, long slab_stride, int d_model, int n_heads, const float* q_ln_w, const float* k_ln_w,
                                      float eps, const unsigned char* key_mask, float alibi_bias_max, float* qkv_ws, void* out_hi, void* out_lo, int ldo,
                                      int T, int batch, const int* ctl, void* stream) {
  return mpt_attn_small_hl<false>(qkvslab, s_in, slab_stride, d_model, n_heads, q_ln_w, k_ln_w, eps, key_mask, alibi_bias_max, qkv_ws, out_hi, out_lo, ldo, T, batch,
                                  ctl, stream);
}
extern "C" int deer_mpt_attn_small_hl_f16(const float* qkvslab, int s_in, long slab_stride, int d_model, int n_heads, const float* q_ln_w,
                                          const float* k_ln_w, float eps, const unsigned char* key_mask, float alibi_bias_max, float* qkv_ws, void* out_hi,
                                          void* out_lo, int ldo, int T, int batch, const int* ctl, void* stream) {   // fp16 hi / lo planes (round 6)
  return mpt_attn_small_hl<true>(qkvslab, s_in, slab_stride, d_model, n_heads, q_ln_w, k_ln_w, eps, key_mask, alibi_bias_max, qkv_ws, out_hi, out_lo, ldo, T, batch,
                                 ctl, stream);
}
