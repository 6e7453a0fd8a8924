// Shared device helpers for the DeeR-VLA gfx950 kernels (CDNA4: wave64, MFMA 16x16x32 bf16).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>   // launch wrappers are called from several host threads (env batches in flight): their one-time attribute flags are atomics

typedef uint16_t bf16_t;  // bf16 storage type in HBM / LDS
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

#define DEER_WAVE 64

// Error codes returned by every extern "C" entry point (0 = ok).
#define DEER_OK 0
#define DEER_ERR_SHAPE 1
#define DEER_ERR_LAUNCH 2

// Device-side control blocks shared by the LLM-layer kernels and the action-head kernels (int32 words; see
// include/deer_hip.h for the ABI view).  One block of CTL_WORDS per environment of the batch, env b at ctl + b*CTL_WORDS;
// SHADOW and ALL_EXITED are batch-global and live in block 0.
#define CTL_EXIT_FLAG 0     // 1 once the exit criterion fired in this step -> later kernels return at entry
#define CTL_EXIT_LAYER 1    // layer index the step exited at
#define CTL_CUR_EXIT_ID 2   // ExitController.cur_exit_id (value_net.py:285-286,294)
#define CTL_HOLD 3          // 1 when this environment's step % steps_per_stage != 0 (reuse cur_exit_id); per environment
#define CTL_N_EVALS 4       // number of head evaluations executed this step (diagnostics)
#define CTL_SHADOW 5        // calibration mode: evaluate EVERY exit, commit at the first that fires, never stop
#define CTL_COMMITTED 6     // shadow mode: a commit already happened in this step
#define CTL_ALL_EXITED 7    // (block 0 only) 1 once EVERY environment of the batch has exited -> later kernels return at entry
#define CTL_PREV_ACTION 8   // float[8]: action_list[-1] (pose6, gripper, pad)
#define CTL_OUT_ACTION 16   // float[8]: committed action (pose6, gripper prob, gripper logit)
#define CTL_DELTAS 24       // float[16]: delta per exit slot of this step (NaN = not evaluated)
#define CTL_N_EXITED 40     // (block 0 only) number of environments that exited in this step
#define CTL_SEQ 41          // (block 0 only) sequence number of the current control step (from step_info)
#define CTL_HOST_PTR 42     // (block 0 only) 2 words: host-visible mirror (pinned, system-coherent) or 0 - see head.hip
#define CTL_EVALS_DONE 44   // (block 0 only) environments that finished the current exit check
#define CTL_PREV_REAL 45    // 1 when CTL_PREV_ACTION is the action of an exit check of THIS step (0: the pseudo action of the first check)
#define CTL_ENS_ACTION 48   // float[8]: value_net.get_ensemble_action() - mean of the last two exit-check actions of this step (pose6, gripper prob, count)
// host mirror layout (int32 words): [0] = seq*64 + number of exit checks completed in this step ("progress"),
// [1] = seq once every environment has exited ("done"), [64*(1+b) .. +64) = copy of environment b's control block at its exit
#define HOSTM_PROGRESS 0
#define HOSTM_DONE 1
#define CTL_WORDS 64

// Row map of an env batch with COMPACTION of exited environments (round 4): the trunk keeps the rows of the still-active
// environments packed at the front of its buffers ("slots"); int32 words: [0] = number of active slots, [1 + s] = environment of slot s,
// [17 + e] = slot of environment e (-1 once it has exited).  Two copies (parity) per model: a compaction writes the other one while
// the exit check of the previous exit layer may still read the old one.  NULL map = identity, every environment active.
#define CMAP_N 0
#define CMAP_SLOT_ENV 1
#define CMAP_ENV_SLOT 17     // up to 16 environments per engine (round 5)
#define CMAP_WORDS 64
#define DEER_MAX_ENVS 16
__device__ __forceinline__ int cmap_active(const int* cmap, int B) { return cmap != nullptr ? cmap[CMAP_N] : B; }
__device__ __forceinline__ int cmap_env(const int* cmap, int slot) { return cmap != nullptr ? cmap[CMAP_SLOT_ENV + slot] : slot; }
__device__ __forceinline__ int cmap_slot(const int* cmap, int env) { return cmap != nullptr ? cmap[CMAP_ENV_SLOT + env] : env; }

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// round-to-nearest-even, NaN preserved (same as torch .to(bfloat16))
// fp32 -> bf16, round-to-nearest-even, on the gfx950 packed converter (v_cvt_pk_bf16_f32: one VALU op per PAIR; the
// integer bit-trick version costs ~7 ops per value and made the fp32->hi/lo staging of the skinny GEMM VALU-bound).
typedef __bf16 deer_bf2 __attribute__((ext_vector_type(2)));
typedef float deer_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2bf(float lo, float hi) {
  return __builtin_bit_cast(uint32_t, __builtin_convertvector((deer_f2){lo, hi}, deer_bf2));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack2bf(f, 0.f) & 0xffffu); }

// ---- the vision tower's 16-bit format (round 6): bf16 or IEEE fp16 --------------------------------------------------------------------------
// The reference evaluates with fp32 weights under fp16 autocast (eval_utils.py:333, README.md:161-167): every Linear of the ViT / Perceiver
// runs on fp16 operands and leaves an fp16 result, LayerNorm / softmax / the residual stream stay f32.  v_mfma_f32_16x16x32_f16 issues at
// the bf16 rate, OpenAI CLIP weights are fp16-native, and an fp16 activation carries 11 significand bits where bf16 carries 8 - so the
// tower kernels are templates over the format (F16 = true: fp16 operands / results; storage stays `bf16_t` = 16 raw bits) and the model
// picks one per engine (deer_config.operands_f16).  The LLM trunk keeps its bf16 hi + lo planes (f32-equivalent activations).
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 deer_h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack2h(float lo, float hi) {      // v_cvt_pk_f16_f32: round-to-nearest-even, overflow -> inf like torch .half()
  return __builtin_bit_cast(uint32_t, __builtin_convertvector((deer_f2){lo, hi}, deer_h2));
}
__device__ __forceinline__ bf16_t f2h(float f) { return (bf16_t)(pack2h(f, 0.f) & 0xffffu); }
__device__ __forceinline__ float h2f(bf16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
template <bool F16> __device__ __forceinline__ uint32_t pack2x(float lo, float hi) { return F16 ? pack2h(lo, hi) : pack2bf(lo, hi); }
template <bool F16> __device__ __forceinline__ bf16_t f2x(float f) { return F16 ? f2h(f) : f2bf(f); }
template <bool F16> __device__ __forceinline__ float x2f(bf16_t v) { return F16 ? h2f(v) : bf2f(v); }
// one 16x16x32 MFMA on fragments held as 8 x 16 raw bits
template <bool F16> __device__ __forceinline__ f32x4 mfma16(bf16x8 a, bf16x8 b, f32x4 c) {
  if constexpr (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
  else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
// hi / lo split of two f32 values: hi = fmt(v), lo = fmt(v - hi) - the LLM trunk feeds every MFMA with both planes (two MFMAs per weight
// fragment), so the activation side of a product carries ~16 (bf16) / ~22 (fp16) significand bits
template <bool F16> __device__ __forceinline__ void split2(float a, float b, uint32_t& hi, uint32_t& lo) {
  hi = pack2x<F16>(a, b);
  float ha, hb;
  if constexpr (F16) { ha = h2f((bf16_t)(hi & 0xffffu)); hb = h2f((bf16_t)(hi >> 16)); }
  else { ha = __uint_as_float(hi << 16); hb = __uint_as_float(hi & 0xffff0000u); }
  lo = pack2x<F16>(a - ha, b - hb);
}
// GEMM epilogue codes shared by csrc/gemm_tiled.hip and csrc/gemm_bigm.hip (include/deer_hip.h: DEER_EPI_*).  "16" = the family's own
// 16-bit format; EPI_BF16OUT stores bf16 whatever the operands are (fp16 tower -> the bf16 K/V the trunk's x-attn reads).
enum { DEER_E_16 = 0, DEER_E_F32 = 1, DEER_E_QGELU_16 = 2, DEER_E_GELU_16 = 3, DEER_E_RESADD_F32 = 4, DEER_E_BF16OUT = 5 };
template <bool F16> __device__ __forceinline__ uint32_t pack2_epi(float lo, float hi, int epi) {
  return (F16 && epi != DEER_E_BF16OUT) ? pack2h(lo, hi) : pack2bf(lo, hi);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Block-wide sum for blockDim.x <= 1024 (red must hold >= 16 floats of LDS).
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = 0.f;
  for (int i = 0; i < nw; ++i) t += red[i];
  return t;
}
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  const int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  float t = -INFINITY;
  for (int i = 0; i < nw; ++i) t = fmaxf(t, red[i]);
  return t;
}

// Sum of s_in f32 split-K partial slabs (float4 at p + s*stride), in slab order (deterministic).  The loads are issued in batches
// of 8 independent requests: one L2 round trip per batch instead of one per slab (the consumers of the skinny GEMM - residual
// + LayerNorm rows, attention prologues, the next GEMM's staging - are latency-bound on exactly this chain).
__device__ __forceinline__ float4 slab_sum4(const float* __restrict__ p, int s_in, long stride) {
  float4 a = float4{0.f, 0.f, 0.f, 0.f};
  int s = 0;
  if (s_in == 16) {                                        // the down-projections' 16 slabs: all in flight together (summed in slab order)
    float4 v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = *reinterpret_cast<const float4*>(p + (long)u * stride);
#pragma unroll
    for (int u = 0; u < 16; ++u) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
    return a;
  }
  for (; s + 8 <= s_in; s += 8) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(p + (long)(s + u) * stride);
#pragma unroll
    for (int u = 0; u < 8; ++u) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
  }
  if (s + 4 <= s_in) {
    float4 v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(p + (long)(s + u) * stride);
#pragma unroll
    for (int u = 0; u < 4; ++u) { a.x += v[u].x; a.y += v[u].y; a.z += v[u].z; a.w += v[u].w; }
    s += 4;
  }
  for (; s < s_in; ++s) {
    const float4 v = *reinterpret_cast<const float4*>(p + (long)s * stride);
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
  }
  return a;
}

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float quick_gelu(float x) { return x / (1.f + __expf(-1.702f * x)); }
// the bf16-output GEMM epilogues: v_exp_f32 + v_rcp_f32 (1 ulp each, the result is rounded to bf16) instead of the IEEE division
// sequence - in a one-round 257 x 256 tile the activation of 68 outputs per lane is serial time at the end of the launch
__device__ __forceinline__ float quick_gelu_bf(float x) {
  return x * __builtin_amdgcn_rcpf(1.f + __builtin_amdgcn_exp2f(-2.4554669596f * x));   // 1.702 * log2(e)
}
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// Kernels on the early-exit path return at entry once the exit flag is set (device-side
// termination: no host round trip per layer).
// Phase time stamps (TOOLS ONLY: `make ktrace` builds lib/libdeer_hip_ktrace.so with -DDEER_KTRACE; the product library compiles
// KT() to nothing).  Thread 0 of workgroup (0, 0) writes the 100 MHz real-time counter to slot `base + i` of a device array installed
// with the translation unit's setter (KT_DEFINE(name) -> extern "C" deer_ktrace_set_<name>(ptr)); tools/ktrace_trunk.py reads them.
#ifdef DEER_KTRACE
#define KT_DEFINE(name)                                                                                       \
  static __device__ long long* deer_ktrace_ptr_##name = nullptr;                                                     \
  extern "C" int deer_ktrace_set_##name(long long* p) {                                                       \
    return hipMemcpyToSymbol(HIP_SYMBOL(deer_ktrace_ptr_##name), &p, sizeof(p)) == hipSuccess ? DEER_OK : DEER_ERR_LAUNCH; \
  }
#define KT(name, first_wg, slot)                                                                                    \
  do {                                                                                                        \
    if ((first_wg) && threadIdx.x == 0 && deer_ktrace_ptr_##name != nullptr) deer_ktrace_ptr_##name[slot] = (long long)__builtin_amdgcn_s_memrealtime(); \
  } while (0)
#else
#define KT_DEFINE(name)
#define KT(name, first_wg, slot) do { } while (0)
#endif

#define DEER_RETURN_IF_EXITED(ctl) \
  do {                             \
    if ((ctl) != nullptr && ((const volatile int*)(ctl))[CTL_ALL_EXITED] != 0) return; \
  } while (0)

#define DEER_LAUNCH_CHECK()                                   \
  do {                                                        \
    hipError_t e_ = hipGetLastError();                        \
    if (e_ != hipSuccess) return DEER_ERR_LAUNCH;             \
  } while (0)
