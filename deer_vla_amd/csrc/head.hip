// Action head + exit gate of DeeR-VLA as device-side kernels (robot_flamingo/models/action_head.py:408-611,
// robot_flamingo/models/value_net.py:105-133,277-297), for a BATCH of B independent environments per launch.
//
// One head evaluation = max-pool over tokens -> 4 x (LSTM cell [-> LayerNorm]) -> 2 MLP heads -> (pose6 tanh,
// gripper sigmoid) -> delta vs. the previous exit's action -> compare with the exit threshold.  The decision
// is taken ON THE DEVICE, per environment: head_final_kernel sets that environment's ctl[EXIT_FLAG]/ctl[EXIT_LAYER]
// and commits its LSTM state; once every environment of the batch has exited it sets ctl[ALL_EXITED] and every
// later kernel of the step (LLM layers and head evaluations) returns at entry.  The host reads {exit_layer, action}
// once per step - there is no host round trip per exit check (the reference syncs on `bool(value <= thr)` at every
// exit, value_net.py:293).
//
// Batching: the B environments share every weight read (an M=B GEMV instead of B M=1 GEMVs): each wave loads its weight
// rows once and keeps B accumulators.  Environments that already exited are still carried through the arithmetic (the
// weight stream is what costs) but their control block, committed action and LSTM state are left untouched.
//
// Numerics: weights bf16 (streamed from HBM / Infinity Cache), activations, LSTM state, LayerNorm statistics
// and the delta in fp32, so the exit decision differs from an fp32 reference only by summation order.
#include "common.h"
#ifdef DEER_KTRACE
KT_DEFINE(head)
#define HKT(slot) KT(head, b == 0, slot)
#define LKT(slot) KT(head, blockIdx.x == 0, slot)
#else
#define HKT(slot) do { } while (0)
#define LKT(slot) do { } while (0)
#endif
#include "../../include/deer_hip.h"

#define HB_MAX 8   // environments per LAUNCH of the GEMV kernels (accumulator registers); batches of up to DEER_MAX_ENVS run in chunks

enum { X_RAW = 0, X_POOL_MAX = 1, X_POOL_AVG = 2, X_LN = 3 };
enum { PRO_RAW = 0, PRO_LN = 1, PRO_GROUP_LN_RELU = 2, PRO_GROUP_RELU = 3 };
enum { KIND_PSEUDO = 0, KIND_CHECK = 1, KIND_COMMIT = 2 };
enum { THR_L2 = 0, THR_MEAN = 1, THR_MAX = 2, THR_COSINE = 3 };

// Whole-launch skip: every environment exited, or (stage hold, value_net.py:285-286) nobody needs this evaluation.
__device__ __forceinline__ bool head_skip(const int* ctl, int kind, int layer, int B) {
  if (ctl == nullptr) return false;
  const volatile int* c = ctl;
  if (c[CTL_ALL_EXITED] != 0) return true;
  if (kind == KIND_COMMIT) return false;
  // HOLD is per environment (the slots of an env batch are at different steps of their sub-tasks): an environment that holds its stage
  // needs no pseudo action and only the checks from its cur_exit_id on; the launch is skipped when NO environment needs it
  bool any_hold = false;
  for (int b = 0; b < B; ++b) any_hold = any_hold || c[b * CTL_WORDS + CTL_HOLD] != 0;
  if (!any_hold) return false;
  bool any = false;
  for (int b = 0; b < B; ++b) {
    const volatile int* cb = c + b * CTL_WORDS;
    if (cb[CTL_EXIT_FLAG] != 0) continue;
    if (cb[CTL_HOLD] == 0) any = true;
    else if (kind == KIND_CHECK && layer >= cb[CTL_CUR_EXIT_ID]) any = true;
  }
  return !any;
}

// ---- host-visible mirror ------------------------------------------------------------------------------------------------
// The exit decision is taken on the device, and the device never waits for the host.  So that the HOST can stop feeding
// work once the step is over (instead of replaying ~100 launches that all return at entry, 1.6 us each), the last kernel of
// every exit check publishes its verdict into pinned, system-coherent host memory: the host replays the step as one graph
// segment per exit with a look-ahead of one segment and polls these words between replays (engine.py::_step_segmented).
__device__ __forceinline__ int* host_mirror(const int* ctl0) {
  return *reinterpret_cast<int* const*>(ctl0 + CTL_HOST_PTR);
}
__device__ __forceinline__ void host_store(int* p, int v) {
  __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// called once per environment at the end of an exit check (thread 0): the last environment publishes "check `slot` complete"
__device__ __forceinline__ void check_done(int* ctl0, int slot, int B) {
  int* hm = host_mirror(ctl0);
  if (hm == nullptr) return;
  if (B == 1) {
    // one environment = one workgroup: this thread wrote the mirror words itself, and the system-scope RELEASE stores below order
    // them (an extra __threadfence_system() is a second L2 write-back + wait, ~2 us on the critical path of every exit check)
    if (((volatile int*)ctl0)[CTL_ALL_EXITED] != 0) host_store(hm + HOSTM_DONE, ctl0[CTL_SEQ]);
    host_store(hm + HOSTM_PROGRESS, ctl0[CTL_SEQ] * 64 + slot + 1);
    return;
  }
  __threadfence_system();
  if (atomicAdd(&ctl0[CTL_EVALS_DONE], 1) + 1 == B) {
    ctl0[CTL_EVALS_DONE] = 0;
    __threadfence_system();
    if (((volatile int*)ctl0)[CTL_ALL_EXITED] != 0) host_store(hm + HOSTM_DONE, ctl0[CTL_SEQ]);
    host_store(hm + HOSTM_PROGRESS, ctl0[CTL_SEQ] * 64 + slot + 1);
  }
}

typedef __attribute__((ext_vector_type(4))) unsigned int head_u32x4;
// 16-byte streaming load (weights are read once per launch: do not displace L2 lines the trunk kernels re-use)
__device__ __forceinline__ uint4 ld_stream16(const bf16_t* p) {
  const head_u32x4 v = __builtin_nontemporal_load(reinterpret_cast<const head_u32x4*>(p));
  return uint4{v.x, v.y, v.z, v.w};
}

__device__ __forceinline__ float dot8(const uint4 w, const float* x) {
  const uint32_t u[4] = {w.x, w.y, w.z, w.w};
  float a = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    a += __uint_as_float(u[i] << 16) * x[2 * i];
    a += __uint_as_float(u[i] & 0xffff0000u) * x[2 * i + 1];
  }
  return a;
}

// Weight rows are bf16 (product arithmetic) or f32 (precision = 1, csrc/precise.hip): eight consecutive weights as a register bundle
template <typename WT> struct W8;
template <> struct W8<bf16_t> {
  typedef uint4 reg;
  static __device__ __forceinline__ reg zero() { return uint4{0, 0, 0, 0}; }
  static __device__ __forceinline__ reg load(const bf16_t* p) { return *reinterpret_cast<const uint4*>(p); }
  static __device__ __forceinline__ reg load_stream(const bf16_t* p) { return ld_stream16(p); }
  static __device__ __forceinline__ float dot(const reg& w, const float* x) { return dot8(w, x); }
};
// IEEE fp16 rows (round 6: w_kind = 2 - what fp16 autocast feeds the head's Linears / LSTM; 11 significand bits where bf16 keeps 8)
struct half_t { uint16_t v; };
template <> struct W8<half_t> {
  typedef uint4 reg;
  static __device__ __forceinline__ reg zero() { return uint4{0, 0, 0, 0}; }
  static __device__ __forceinline__ reg load(const half_t* p) { return *reinterpret_cast<const uint4*>(p); }
  static __device__ __forceinline__ reg load_stream(const half_t* p) { return ld_stream16(reinterpret_cast<const bf16_t*>(p)); }
  static __device__ __forceinline__ float dot(const reg& w, const float* x) {
    const uint32_t u[4] = {w.x, w.y, w.z, w.w};
    float a = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      a += h2f((bf16_t)(u[i] & 0xffffu)) * x[2 * i];
      a += h2f((bf16_t)(u[i] >> 16)) * x[2 * i + 1];
    }
    return a;
  }
};
template <> struct W8<float> {
  struct reg { float4 a, b; };
  static __device__ __forceinline__ reg zero() { return reg{float4{0.f, 0.f, 0.f, 0.f}, float4{0.f, 0.f, 0.f, 0.f}}; }
  static __device__ __forceinline__ reg load(const float* p) { return reg{*reinterpret_cast<const float4*>(p), *reinterpret_cast<const float4*>(p + 4)}; }
  static __device__ __forceinline__ reg load_stream(const float* p) { return load(p); }
  static __device__ __forceinline__ float dot(const reg& w, const float* x) {
    return w.a.x * x[0] + w.a.y * x[1] + w.a.z * x[2] + w.a.w * x[3] + w.b.x * x[4] + w.b.y * x[5] + w.b.z * x[6] + w.b.w * x[7];
  }
};

// LayerNorm of src[0..n) into dst (LDS), optional ReLU; all threads of the block participate.
__device__ __forceinline__ void block_ln(const float* __restrict__ src, float* dst, int n, const float* __restrict__ w,
                                         const float* __restrict__ b, float eps, bool relu, float* red) {
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) s += src[i];
  const float mean = block_sum(s, red) / n;
  float v = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const float d = src[i] - mean;
    v += d * d;
  }
  const float rstd = rsqrtf(block_sum(v, red) / n + eps);
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    float y = (src[i] - mean) * rstd * w[i] + b[i];
    dst[i] = relu ? fmaxf(y, 0.f) : y;
  }
}

// LayerNorm of ONE row of n <= 2048 values (n % 4 == 0) by ONE wave: the row lives in registers, no block barrier (head_final_body runs
// the actions-head and the gripper-head LayerNorm side by side on two waves: 3.7 -> 1.2 us of every evaluation, tools/ktrace_head.py)
__device__ __forceinline__ void wave_ln_row(const float* __restrict__ src, float* dst, int n, const float* __restrict__ w,
                                            const float* __restrict__ bta, float eps, bool relu) {
  const int lane = threadIdx.x & 63;
  float4 v[8], gw[8], gb[8];
  float sum = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int i = (lane + 64 * e) * 4;
    v[e] = gw[e] = gb[e] = float4{0.f, 0.f, 0.f, 0.f};
    if (i < n) {
      v[e] = *reinterpret_cast<const float4*>(src + i);
      gw[e] = *reinterpret_cast<const float4*>(w + i);
      gb[e] = *reinterpret_cast<const float4*>(bta + i);
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) sum += v[e].x + v[e].y + v[e].z + v[e].w;
  const float mean = wave_sum(sum) / n;
  float var = 0.f;
#pragma unroll
  for (int e = 0; e < 8; ++e)
    if ((lane + 64 * e) * 4 < n) {
      const float a = v[e].x - mean, bq = v[e].y - mean, c = v[e].z - mean, dd = v[e].w - mean;
      var += a * a + bq * bq + c * c + dd * dd;
    }
  const float rstd = rsqrtf(wave_sum(var) / n + eps);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int i = (lane + 64 * e) * 4;
    if (i < n) {
      float4 y;
      y.x = (v[e].x - mean) * rstd * gw[e].x + gb[e].x; y.y = (v[e].y - mean) * rstd * gw[e].y + gb[e].y;
      y.z = (v[e].z - mean) * rstd * gw[e].z + gb[e].z; y.w = (v[e].w - mean) * rstd * gw[e].w + gb[e].w;
      if (relu) { y.x = fmaxf(y.x, 0.f); y.y = fmaxf(y.y, 0.f); y.z = fmaxf(y.z, 0.f); y.w = fmaxf(y.w, 0.f); }
      *reinterpret_cast<float4*>(dst + i) = y;
    }
  }
}

// Staging of activation rows into LDS.  Every load of a batch is requested before the first LDS store: a plain
// `for (i..) dst[i] = src[i]` loop compiles to one exposed L2 round trip per iteration, which for an 8-environment batch
// (32-64 iterations per thread) cost more than the weight stream of the whole launch.
//  * block_copy_to_lds: n contiguous floats (n % 4 == 0, 16-byte aligned), all threads, optional ReLU
//  * rows_ln_to_lds: LayerNorm of B rows (row b at src + b*stride, n <= 2048, n % 4 == 0): wave w takes rows w, w + nw, ...,
//    the row lives in registers and the statistics are reduced inside the wave - no block barrier per row.
__device__ __forceinline__ void block_copy_to_lds(const float* __restrict__ src, float* dst, int n, bool relu) {
  const int step = blockDim.x * 4;
  for (int base = threadIdx.x * 4; base < n; base += step * 8) {
    float4 t[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = base + u * step;
      t[u] = i < n ? *reinterpret_cast<const float4*>(src + i) : float4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = base + u * step;
      if (i < n) {
        float4 v = t[u];
        if (relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
        *reinterpret_cast<float4*>(dst + i) = v;
      }
    }
  }
}

#define HEAD_LN_MAXE 8    // float4 per lane: rows of up to 2048 values
__device__ __forceinline__ void rows_ln_to_lds(const float* __restrict__ src, long stride, float* dst, int n, int B,
                                               const float* __restrict__ w, const float* __restrict__ bta, float eps, bool relu) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  float4 gw[HEAD_LN_MAXE], gb[HEAD_LN_MAXE];            // gamma / beta do not depend on the row: requested once, up front
#pragma unroll
  for (int e = 0; e < HEAD_LN_MAXE; ++e) {
    const int i = (lane + 64 * e) * 4;
    gw[e] = i < n ? *reinterpret_cast<const float4*>(w + i) : float4{0.f, 0.f, 0.f, 0.f};
    gb[e] = i < n ? *reinterpret_cast<const float4*>(bta + i) : float4{0.f, 0.f, 0.f, 0.f};
  }
  for (int b = wave; b < B; b += nw) {
    const float* s = src + (long)b * stride;
    float* d = dst + b * n;
    float4 v[HEAD_LN_MAXE];
    float sum = 0.f;
#pragma unroll
    for (int e = 0; e < HEAD_LN_MAXE; ++e) {
      const int i = (lane + 64 * e) * 4;
      v[e] = i < n ? *reinterpret_cast<const float4*>(s + i) : float4{0.f, 0.f, 0.f, 0.f};
      sum += v[e].x + v[e].y + v[e].z + v[e].w;
    }
    const float mean = wave_sum(sum) / n;
    float var = 0.f;
#pragma unroll
    for (int e = 0; e < HEAD_LN_MAXE; ++e)
      if ((lane + 64 * e) * 4 < n) {
        const float a = v[e].x - mean, bq = v[e].y - mean, c = v[e].z - mean, dd = v[e].w - mean;
        var += a * a + bq * bq + c * c + dd * dd;
      }
    const float rstd = rsqrtf(wave_sum(var) / n + eps);
#pragma unroll
    for (int e = 0; e < HEAD_LN_MAXE; ++e) {
      const int i = (lane + 64 * e) * 4;
      if (i < n) {
        float4 y;
        y.x = (v[e].x - mean) * rstd * gw[e].x + gb[e].x; y.y = (v[e].y - mean) * rstd * gw[e].y + gb[e].y;
        y.z = (v[e].z - mean) * rstd * gw[e].z + gb[e].z; y.w = (v[e].w - mean) * rstd * gw[e].w + gb[e].w;
        if (relu) { y.x = fmaxf(y.x, 0.f); y.y = fmaxf(y.y, 0.f); y.z = fmaxf(y.z, 0.f); y.w = fmaxf(y.w, 0.f); }
        *reinterpret_cast<float4*>(d + i) = y;
      }
    }
  }
}

// rows (row b at src + b*stride) -> dst [B][n]: LayerNorm'ed, or copied (one block-wide copy when the rows are contiguous)
__device__ __forceinline__ void rows_to_lds(const float* __restrict__ src, long stride, float* dst, int n, int B, bool ln,
                                            const float* __restrict__ w, const float* __restrict__ bta, float eps, bool relu) {
  if (ln) {
    rows_ln_to_lds(src, stride, dst, n, B, w, bta, eps, relu);
  } else if (stride == n) {
    block_copy_to_lds(src, dst, B * n, relu);
  } else {
    for (int b = 0; b < B; ++b) block_copy_to_lds(src + (long)b * stride, dst + b * n, n, relu);
  }
}

// ---- token pooling (AdaptiveMaxPool1d / AvgPool over the T text tokens, action_head.py:480-483,519-520) --------------
// feats: [B][T][d] (env b at feats + b*T*d) -> pooled [B][d].  One launch per head evaluation, so that the 256
// workgroups of the first LSTM layer do not each re-read B*T*d features.
// key_mask (uint8 [B][T], 0 = right-padding) or NULL: instructions of an environment batch are right-padded to a common T,
// and the reference - which runs every environment alone (B = 1, padding="longest", data.py:905-919) - never sees those
// rows, so they are left out of the pool (max: skipped; avg: divided by the number of valid tokens).
__global__ __launch_bounds__(256) void head_pool_kernel(const float* __restrict__ feats, float* __restrict__ pooled, int T, int d,
                                                        int avg, int B, const unsigned char* __restrict__ key_mask,
                                                        const int* ctl, int kind, int layer, const float* __restrict__ add,
                                                        const int* __restrict__ cmap = nullptr) {
  if (head_skip(ctl, kind, layer, B)) return;
  const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= d) return;
  // compaction: the features of environment b live at its SLOT of the layer's row map; an environment that has left the trunk has no
  // rows any more - its pooled feature is zero and the rest of the evaluation ignores it (head_final returns on its EXIT_FLAG)
  const int slot = cmap != nullptr ? cmap[CMAP_ENV_SLOT + b] : b;
  if (slot < 0) {
    pooled[(long)b * d + i] = 0.f;
    return;
  }
  const float* x = feats + ((long)slot * T) * d + i;
  const unsigned char* km = key_mask != nullptr ? key_mask + b * T : nullptr;
  float a = avg ? 0.f : -INFINITY;
  int n = 0;
  for (int t = 0; t < T; ++t) {
    if (km != nullptr && km[t] == 0) continue;
    const float v = x[(long)t * d];
    a = avg ? a + v : fmaxf(a, v);
    ++n;
  }
  if (avg) a /= (float)max(n, 1);
  if (add != nullptr) a += add[(long)b * d + i];          // use_state: the embedded robot state (action_head.py:536)
  pooled[(long)b * d + i] = a;
}

extern "C" int deer_head_pool(const float* feats, float* pooled, int T, int d, int avg, int B, const unsigned char* key_mask,
                              const int* ctl, int kind, int layer, void* stream) {
  if (T <= 0 || d <= 0 || B <= 0 || B > DEER_MAX_ENVS) return DEER_ERR_SHAPE;
  hipLaunchKernelGGL(head_pool_kernel, dim3((d + 255) / 256, B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), feats, pooled, T, d,
                     avg, B, key_mask, ctl, kind, layer, static_cast<const float*>(nullptr));
  DEER_LAUNCH_CHECK();
  return DEER_OK;
}

// deer_head_pool on the hidden state of a layer whose rows are packed by the row map `cmap` (env batch with compaction)
extern "C" int deer_head_pool_active(const float* feats, float* pooled, int T, int d, int avg, int B, const unsigned char* key_mask,
                                     const int* ctl, int kind, int layer, const int* cmap, void* stream) {
  if (T <= 0 || d <= 0 || B <= 0 || B > DEER_MAX_ENVS || cmap == nullptr) return DEER_ERR_SHAPE;
  hipLaunchKernelGGL(head_pool_kernel, dim3((d + 255) / 256, B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), feats, pooled, T, d,
                     avg, B, key_mask, ctl, kind, layer, static_cast<const float*>(nullptr), cmap);
  DEER_LAUNCH_CHECK();
  return DEER_OK;
}

// deer_head_pool with the robot-state embedding added to the pooled feature (DeterministicDecoder(use_state=True), action_head.py:524-536)
extern "C" int deer_head_pool_state(const float* feats, float* pooled, int T, int d, int avg, int B, const unsigned char* key_mask,
                                    const float* state_emb, const int* ctl, int kind, int layer, void* stream) {
  if (T <= 0 || d <= 0 || B <= 0 || B > DEER_MAX_ENVS || state_emb == nullptr) return DEER_ERR_SHAPE;
  hipLaunchKernelGGL(head_pool_kernel, dim3((d + 255) / 256, B), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), feats, pooled, T, d,
                     avg, B, key_mask, ctl, kind, layer, state_emb);
  DEER_LAUNCH_CHECK();
  return DEER_OK;
}

// ---- robot-state embedding of DeterministicDecoder(use_state=True) (action_head.py:443-453,524-536) ------------------------------
// state [B][8] f32 = (arm pose robot_obs[:6], gripper opening robot_obs[-1] in {-1, +1}, pad);
// u = [relu(W_arm arm + b_arm) ; relu(E_grip[((g + 1) / 2) truncated])] (2d); out[b] = W_state u + b_state (d).  One environment per
// blockIdx.y, 16 output rows per workgroup; u is rebuilt per workgroup in LDS (6 MACs per element).
template <typename WT>
__global__ __launch_bounds__(256) void head_state_embed_kernel(const float* __restrict__ state, const float* __restrict__ w_arm,
                                                               const float* __restrict__ b_arm, const float* __restrict__ e_grip,
                                                               const WT* __restrict__ w_state, const float* __restrict__ b_state,
                                                               float* __restrict__ out, int d) {
  extern __shared__ __attribute__((aligned(16))) float lds[];      // u[2d]
  const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* st = state + b * 8;
  const int gi = min(1, max(0, (int)((st[6] + 1.0f) * 0.5f)));     // .long() truncates
  for (int i = threadIdx.x; i < d; i += 256) {
    float a = b_arm[i];
#pragma unroll
    for (int k = 0; k < 6; ++k) a += w_arm[i * 6 + k] * st[k];
    lds[i] = fmaxf(a, 0.f);
    lds[d + i] = fmaxf(e_grip[(long)gi * d + i], 0.f);
  }
  __syncthreads();
  for (int r = 0; r < 4; ++r) {
    const int n = blockIdx.x * 16 + wave * 4 + r;
    if (n >= d) break;
    float a = 0.f;
    for (int k = lane * 8; k < 2 * d; k += 512) a += W8<WT>::dot(W8<WT>::load(w_state + (long)n * 2 * d + k), lds + k);
    a = wave_sum(a);
    if (lane == 0) out[(long)b * d + n] = a + b_state[n];
  }
}

extern "C" int deer_head_state_embed(const float* state, const float* w_arm, const float* b_arm, const float* e_grip, const void* w_state,
                                     const float* b_state, float* out, int d, int B, int w_is_f32, void* stream) {
  if (d <= 0 || (d & 3) || B <= 0 || B > DEER_MAX_ENVS || 2 * d * 4 > 64 * 1024) return DEER_ERR_SHAPE;
  dim3 grid((d + 15) / 16, B);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (w_is_f32 == 1) hipLaunchKernelGGL(head_state_embed_kernel<float>, grid, dim3(256), 2 * d * 4, st, state, w_arm, b_arm, e_grip,
                       reinterpret_cast<const float*>(w_state), b_state, out, d);
    else if (w_is_f32 == 2) hipLaunchKernelGGL(head_state_embed_kernel<half_t>, grid, dim3(256), 2 * d * 4, st, state, w_arm, b_arm, e_grip,
                       reinterpret_cast<const half_t*>(w_state), b_state, out, d);
    else hipLaunchKernelGGL(head_state_embed_kernel<bf16_t>, grid, dim3(256), 2 * d * 4, st, state, w_arm, b_arm, e_grip,
                       reinterpret_cast<const bf16_t*>(w_state), b_state, out, d);
  DEER_LAUNCH_CHECK();
  return DEER_OK;
}

// ---- one LSTM layer, single time step (torch.nn.LSTM gate order i,f,g,o), B environments -------------------------
// wave -> hidden unit j: rows j, H+j, 2H+j, 3H+j of [W_ih | W_hh]; c' = s(f) c + s(i) tanh(g); h' = s(o) tanh(c').
// x_src: X_POOL_*: feats [B][T_stride rows][in_dim] (env b at x_src + b*x_bstride), pooled over the first T rows;
//        X_LN / X_RAW: [B][in_dim] (previous layer's h of each env, x_bstride = in_dim).  State tensors are [B][H].
template <typename WT>
__global__ __launch_bounds__(256) void head_lstm_layer_kernel(const float* __restrict__ x_src, long x_bstride, int x_mode, int T,
                                                              int in_dim, const float* __restrict__ ln_w,
                                                              const float* __restrict__ ln_b, const WT* __restrict__ w_ih,
                                                              const WT* __restrict__ w_hh, const float* __restrict__ b_ih,
                                                              const float* __restrict__ b_hh, const float* __restrict__ h_prev,
                                                              const float* __restrict__ c_prev, float* __restrict__ h_out,
                                                              float* __restrict__ c_out, int H, int B_all, int b0, int B, float eps,
                                                              const int* ctl, int kind, int layer, const float* __restrict__ ghh) {
  // environments b0 .. b0+B-1 of a batch of B_all (the launcher splits a batch whose activations do not fit the LDS)
  if (head_skip(ctl, kind, layer, B_all)) return;
  // ghh != NULL: W_hh h_prev + b_hh of this layer was computed once for the whole control step (deer_head_lstm_hh: h_prev is the state
  // the PREVIOUS step committed, the same for every exit check of a step) - [B_all][4H], gate-major; the launch then streams W_ih only
  x_src += (long)b0 * x_bstride;
  if (ghh != nullptr) ghh += (long)b0 * 4 * H; else h_prev += (long)b0 * H;
  c_prev += (long)b0 * H; h_out += (long)b0 * H; c_out += (long)b0 * H;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  LKT(16);
  float* xs = lds;                    // [B][in_dim]
  float* hs = lds + B * in_dim;       // [B][H]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = blockIdx.x * 4 + wave;
  const int jr = j < H ? j : H - 1;
  // The launch is latency-bound (24 KB of weights per wave): the first LSTM_PF k-steps of W_ih and the first two of W_hh are
  // issued BEFORE the activations are staged (pool / LayerNorm / copy into LDS), so the HBM round trip overlaps the staging and
  // there is one exposed memory latency per launch instead of one per k-step.
  constexpr int LSTM_PF = 4;
  // the gate constants of (environment = lane, unit j) - biases, the W_hh h term of the step, c - do not depend on the dot products:
  // requested with the weights (they used to be a second L2 round trip behind the wave sums, 0.8 us of a 5.8 us launch)
  float pgb[4] = {0.f, 0.f, 0.f, 0.f}, pgh[4] = {0.f, 0.f, 0.f, 0.f}, pc = 0.f;   // kept apart until the gates: an add here would wait for them
  if (lane < B) {
    const float* second = ghh != nullptr ? ghh + (long)lane * 4 * H : b_hh;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      pgb[q] = b_ih[q * H + jr];
      pgh[q] = second[q * H + jr];
    }
    pc = c_prev[lane * H + jr];
  }
  typename W8<WT>::reg wi[LSTM_PF][4], wh[2][4];
#pragma unroll
  for (int u = 0; u < LSTM_PF; ++u) {
    const int k = u * 512 + lane * 8;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      wi[u][q] = k < in_dim ? W8<WT>::load_stream(w_ih + ((long)q * H + jr) * in_dim + k) : W8<WT>::zero();
  }
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int k = u * 512 + lane * 8;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      wh[u][q] = (ghh == nullptr && k < H) ? W8<WT>::load_stream(w_hh + ((long)q * H + jr) * H + k) : W8<WT>::zero();
  }
  LKT(17);
  if (x_mode == X_LN || x_mode == X_RAW) {
    rows_to_lds(x_src, x_bstride, xs, in_dim, B, x_mode == X_LN, ln_w, ln_b, eps, false);
  } else {
    for (int b = 0; b < B; ++b) {
      const float* xb = x_src + b * x_bstride;
      float* xd = xs + b * in_dim;
      for (int i = threadIdx.x; i < in_dim; i += 256) {
        float a = xb[i];
        if (x_mode == X_POOL_MAX) {
          for (int t = 1; t < T; ++t) a = fmaxf(a, xb[(long)t * in_dim + i]);
        } else {
          for (int t = 1; t < T; ++t) a += xb[(long)t * in_dim + i];
          a /= (float)T;
        }
        xd[i] = a;
      }
    }
  }
  if (ghh == nullptr) block_copy_to_lds(h_prev, hs, B * H, false);
  __syncthreads();
  LKT(18);

  if (j >= H) return;
  float acc[4][HB_MAX];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int b = 0; b < HB_MAX; ++b) acc[q][b] = 0.f;
#pragma unroll
  for (int u = 0; u < LSTM_PF; ++u) {
    const int k = u * 512 + lane * 8;
    if (k < in_dim) {
#pragma unroll
      for (int b = 0; b < HB_MAX; ++b)
        if (b < B) {
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[q][b] += W8<WT>::dot(wi[u][q], xs + b * in_dim + k);
        }
    }
  }
  for (int k = LSTM_PF * 512 + lane * 8; k < in_dim; k += 512) {
    typename W8<WT>::reg w[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) w[q] = W8<WT>::load(w_ih + ((long)q * H + j) * in_dim + k);
#pragma unroll
    for (int b = 0; b < HB_MAX; ++b)
      if (b < B) {
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q][b] += W8<WT>::dot(w[q], xs + b * in_dim + k);
      }
  }
  if (ghh == nullptr) {
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int k = u * 512 + lane * 8;
    if (k < H) {
#pragma unroll
      for (int b = 0; b < HB_MAX; ++b)
        if (b < B) {
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[q][b] += W8<WT>::dot(wh[u][q], hs + b * H + k);
        }
    }
  }
  for (int k = 2 * 512 + lane * 8; k < H; k += 512) {
    typename W8<WT>::reg w[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) w[q] = W8<WT>::load(w_hh + ((long)q * H + j) * H + k);
#pragma unroll
    for (int b = 0; b < HB_MAX; ++b)
      if (b < B) {
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q][b] += W8<WT>::dot(w[q], hs + b * H + k);
      }
  }
  }
  LKT(19);
  // wave totals land in every lane; lane b then does environment b's gate arithmetic (B transcendental chains in parallel)
  float gi = 0.f, gf = 0.f, gg = 0.f, go = 0.f;
#pragma unroll
  for (int b = 0; b < HB_MAX; ++b)
    if (b < B) {
      const float t0 = wave_sum(acc[0][b]), t1 = wave_sum(acc[1][b]), t2 = wave_sum(acc[2][b]), t3 = wave_sum(acc[3][b]);
      if (lane == b) { gi = t0; gf = t1; gg = t2; go = t3; }
    }
  if (lane < B) {
    const int b = lane;
    gi += pgb[0] + pgh[0]; gf += pgb[1] + pgh[1]; gg += pgb[2] + pgh[2]; go += pgb[3] + pgh[3];
    const float c2 = sigmoidf_(gf) * pc + sigmoidf_(gi) * tanhf(gg);
    c_out[b * H + j] = c2;
    h_out[b * H + j] = sigmoidf_(go) * tanhf(c2);
  }
  LKT(20);
}

static int launch_head_lstm_layer(const float* x_src, long x_bstride, int x_mode, int T, int in_dim, const float* ln_w,
                                  const float* ln_b, const void* w_ih, const void* w_hh, const float* b_ih, const float* b_hh,
                                  const float* h_prev, const float* c_prev, float* h_out, float* c_out, int H, int B, float eps,
                                  const int* ctl, int kind, int layer, int w_is_f32, void* stream, const float* ghh) {
  if (in_dim <= 0 || (in_dim & 7) || H <= 0 || (H & 7) || x_mode < 0 || x_mode > 3 || (x_mode == X_LN && (ln_w == nullptr || in_dim > 2048)) ||
      ((x_mode == X_LN || x_mode == X_RAW) && (x_bstride & 3)) ||
      ((x_mode == X_POOL_MAX || x_mode == X_POOL_AVG) && T <= 0) || B <= 0 || B > DEER_MAX_ENVS)
    return DEER_ERR_SHAPE;
  // [B][in_dim + H] f32 of activations live in LDS: a batch that does not fit 150 KB goes in several launches (8 environments of
  // the 9B model: in_dim 4096 -> 2 x 4)
  const int per_env = (in_dim + H) * (int)sizeof(float);
  const int nb_max = std::min(HB_MAX, (150 * 1024 - 64) / per_env);   // and at most HB_MAX environments per launch (accumulator registers)
  if (nb_max < 1) return DEER_ERR_SHAPE;
  static std::atomic<bool> attr_set{false};
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&head_lstm_layer_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            150 * 1024) != hipSuccess ||
        hipFuncSetAttribute(reinterpret_cast<const void*>(&head_lstm_layer_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            150 * 1024) != hipSuccess)
      return DEER_ERR_LAUNCH;
    attr_set = true;
  }
  const int chunks = (B + nb_max - 1) / nb_max, nb_even = (B + chunks - 1) / chunks;
  for (int b0 = 0; b0 < B; b0 += nb_even) {
    const int nb = B - b0 < nb_even ? B - b0 : nb_even;
    const int smem = nb * per_env + 64;
    if (w_is_f32 == 1) hipLaunchKernelGGL(head_lstm_layer_kernel<float>, dim3((H + 3) / 4), dim3(256), smem, reinterpret_cast<hipStream_t>(stream), x_src,
                         x_bstride, x_mode, T, in_dim, ln_w, ln_b, reinterpret_cast<const float*>(w_ih),
                         reinterpret_cast<const float*>(w_hh), b_ih, b_hh, h_prev, c_prev, h_out, c_out, H, B, b0, nb, eps, ctl, kind, layer, ghh);
    else if (w_is_f32 == 2) hipLaunchKernelGGL(head_lstm_layer_kernel<half_t>, dim3((H + 3) / 4), dim3(256), smem, reinterpret_cast<hipStream_t>(stream), x_src,
                         x_bstride, x_mode, T, in_dim, ln_w, ln_b, reinterpret_cast<const half_t*>(w_ih),
                         reinterpret_cast<const half_t*>(w_hh), b_ih, b_hh, h_prev, c_prev, h_out, c_out, H, B, b0, nb, eps, ctl, kind, layer, ghh);
    else hipLaunchKernelGGL(head_lstm_layer_kernel<bf16_t>, dim3((H + 3) / 4), dim3(256), smem, reinterpret_cast<hipStream_t>(stream), x_src,
                         x_bstride, x_mode, T, in_dim, ln_w, ln_b, reinterpret_cast<const bf16_t*>(w_ih),
                         reinterpret_cast<const bf16_t*>(w_hh), b_ih, b_hh, h_prev, c_prev, h_out, c_out, H, B, b0, nb, eps, ctl, kind, layer, ghh);
  }
  DEER_LAUNCH_CHECK();
  return DEER_OK;
}

extern "C" int deer_head_lstm_layer(const float* x_src, long x_bstride, int x_mode, int T, int in_dim, const float* ln_w,
                                    const float* ln_b, const void* w_ih, const void* w_hh, const float* b_ih, const float* b_hh,
                                    const float* h_prev, const float* c_prev, float* h_out, float* c_out, int H, int B, float eps,
                                    const int* ctl, int kind, int layer, int w_is_f32, void* stream) {
  if (w_hh == nullptr || b_hh == nullptr || h_prev == nullptr) return DEER_ERR_SHAPE;
  return launch_head_lstm_layer(x_src, x_bstride, x_mode, T, in_dim, ln_w, ln_b, w_ih, w_hh, b_ih, b_hh, h_prev, c_prev, h_out, c_out, H, B, eps, ctl, kind,
                                layer, w_is_f32, stream, nullptr);
}

// The same LSTM layer with its recurrent half taken from `ghh` = W_hh h_prev + b_hh ([B][4H], deer_head_lstm_hh): every exit check of a
// control step starts from the state the previous step committed, so that half is the same for all of them - the launch streams W_ih only
// (16.8 + 3 x 8.4 MB per evaluation of the 4-layer head instead of 25.2 + 3 x 16.8).  Sums are grouped as (W_ih x + b_ih) + ghh.
extern "C" int deer_head_lstm_layer_pre(const float* x_src, long x_bstride, int x_mode, int T, int in_dim, const float* ln_w, const float* ln_b,
                                        const void* w_ih, const float* b_ih, const float* ghh, const float* c_prev, float* h_out, float* c_out, int H,
                                        int B, float eps, const int* ctl, int kind, int layer, int w_is_f32, void* stream) {
  if (ghh == nullptr) return DEER_ERR_SHAPE;
  return launch_head_lstm_layer(x_src, x_bstride, x_mode, T, in_dim, ln_w, ln_b, w_ih, nullptr, b_ih, nullptr, nullptr, c_prev, h_out, c_out, H, B, eps, ctl,
                                kind, layer, w_is_f32, stream, ghh);
}

// ghh[l][b][q*H + j] = W_hh^l[q*H + j, :] . h_state[l][b] + b_hh^l[q*H + j] for up to 8 LSTM layers in ONE launch (grid: H/4 x L; wave -> hidden
// unit j of layer l, its four gate rows; h_state [L][B][H]).  Once per control step, before the first head evaluation.
struct deer_lstm_hh_args { const void* w[8]; const float* b[8]; };
template <typename WT>
__global__ __launch_bounds__(256) void head_lstm_hh_kernel(deer_lstm_hh_args a, const float* __restrict__ h_state, float* __restrict__ ghh, int H, int B_all,
                                                           int b0, int B) {
  // environments b0 .. b0 + B - 1 of a batch of B_all (h_state [L][B_all][H], ghh [L][B_all][4H])
  extern __shared__ __attribute__((aligned(16))) float lds[];   // [B][H]
  const int l = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = blockIdx.x * 4 + wave;
  const int jr = j < H ? j : H - 1;
  const WT* w_hh = reinterpret_cast<const WT*>(a.w[l]);
  typename W8<WT>::reg wh[2][4];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int k = u * 512 + lane * 8;
#pragma unroll
    for (int q = 0; q < 4; ++q) wh[u][q] = k < H ? W8<WT>::load_stream(w_hh + ((long)q * H + jr) * H + k) : W8<WT>::zero();
  }
  block_copy_to_lds(h_state + ((long)l * B_all + b0) * H, lds, B * H, false);
  __syncthreads();
  if (j >= H) return;
  float acc[4][HB_MAX];
#pragma unroll
  for (int q = 0; q < 4; ++q)
#pragma unroll
    for (int b = 0; b < HB_MAX; ++b) acc[q][b] = 0.f;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int k = u * 512 + lane * 8;
    if (k < H) {
#pragma unroll
      for (int b = 0; b < HB_MAX; ++b)
        if (b < B) {
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[q][b] += W8<WT>::dot(wh[u][q], lds + b * H + k);
        }
    }
  }
  for (int k = 2 * 512 + lane * 8; k < H; k += 512) {
    typename W8<WT>::reg w[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) w[q] = W8<WT>::load(w_hh + ((long)q * H + j) * H + k);
#pragma unroll
    for (int b = 0; b < HB_MAX; ++b)
      if (b < B) {
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q][b] += W8<WT>::dot(w[q], lds + b * H + k);
      }
  }
  const float* bh = a.b[l];
  float* out = ghh + ((long)l * B_all + b0) * 4 * H;
#pragma unroll
  for (int b = 0; b < HB_MAX; ++b)
    if (b < B) {
      const float t0 = wave_sum(acc[0][b]), t1 = wave_sum(acc[1][b]), t2 = wave_sum(acc[2][b]), t3 = wave_sum(acc[3][b]);
      if (lane == 0) {
        float* o = out + (long)b * 4 * H + j;
        o[0] = t0 + bh[j]; o[H] = t1 + bh[H + j]; o[2 * H] = t2 + bh[2 * H + j]; o[3 * H] = t3 + bh[3 * H + j];
      }
    }
}

extern "C" int deer_head_lstm_hh(const void* const* w_hh, const float* const* b_hh, int L, const float* h_state, float* ghh, int H, int B, int w_is_f32,
                                 void* stream) {
  if (w_hh == nullptr || b_hh == nullptr || L <= 0 || L > 8 || h_state == nullptr || ghh == nullptr || H <= 0 || (H & 7) || B <= 0 || B > DEER_MAX_ENVS ||
      (size_t)HB_MAX * H * sizeof(float) > 64 * 1024)
    return DEER_ERR_SHAPE;
  deer_lstm_hh_args a{};
  for (int l = 0; l < L; ++l) {
    if (w_hh[l] == nullptr || b_hh[l] == nullptr) return DEER_ERR_SHAPE;
    a.w[l] = w_hh[l]; a.b[l] = b_hh[l];
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  for (int b0 = 0; b0 < B; b0 += HB_MAX) {                 // batches above HB_MAX environments: one launch per chunk
    const int nb = B - b0 < HB_MAX ? B - b0 : HB_MAX;
    const int smem = nb * H * (int)sizeof(float);
    if (w_is_f32 == 1) hipLaunchKernelGGL(head_lstm_hh_kernel<float>, dim3((H + 3) / 4, L), dim3(256), smem, st, a, h_state, ghh, H, B, b0, nb);
    else if (w_is_f32 == 2) hipLaunchKernelGGL(head_lstm_hh_kernel<half_t>, dim3((H + 3) / 4, L), dim3(256), smem, st, a, h_state, ghh, H, B, b0, nb);
    else hipLaunchKernelGGL(head_lstm_hh_kernel<bf16_t>, dim3((H + 3) / 4, L), dim3(256), smem, st, a, h_state, ghh, H, B, b0, nb);
  }
  DEER_LAUNCH_CHECK();
  return DEER_OK;
}

// ---- hidden Linear of both MLP heads: dst[b][g*out_dim + n] = W_g[n,:] . pro(src_b)_g + b_g[n] -------------------
// src: [B][src_stride]; pro: PRO_LN   x = LN(src_b[0..in))  (shared: LSTM output LayerNorm, action_head.py:55-56)
//      PRO_RAW x = src_b[0..in) (plain nn.LSTM); PRO_GROUP_LN_RELU x_g = relu(LN_g(src_b[g*in..])) (Linear -> LN -> ReLU,
//      action_head.py:97-103); PRO_GROUP_RELU x_g = relu(src_b[g*in..]).   dst: [B][2*out_dim].
template <typename WT>
__global__ __launch_bounds__(256) void head_fc_kernel(const float* __restrict__ src, int src_stride, int in_dim, int pro,
                                                      const float* __restrict__ lnw0, const float* __restrict__ lnb0,
                                                      const float* __restrict__ lnw1, const float* __restrict__ lnb1,
                                                      const WT* __restrict__ W0, const float* __restrict__ b0,
                                                      const WT* __restrict__ W1, const float* __restrict__ b1, int out_dim,
                                                      float* __restrict__ dst, int B, float eps, const int* ctl, int kind,
                                                      int layer, int B_all) {
  if (head_skip(ctl, kind, layer, B_all)) return;            // src / dst already point at this launch's first environment (B of B_all)
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* xs = lds;                     // [B][in_dim]
  const int grp = blockIdx.y;
  const bool grouped = (pro == PRO_GROUP_LN_RELU || pro == PRO_GROUP_RELU);
  const WT* W = grp ? W1 : W0;
  const float* bb = grp ? b1 : b0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n0 = (blockIdx.x * 4 + wave) * 2;
  const bool live = n0 < out_dim, two = (n0 + 1 < out_dim);
  // weights of the first two k-steps are requested before the LayerNorm staging (one exposed HBM latency per launch)
  typename W8<WT>::reg pw0[2], pw1[2];
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int k = u * 512 + lane * 8;
    pw0[u] = (live && k < in_dim) ? W8<WT>::load_stream(W + (long)n0 * in_dim + k) : W8<WT>::zero();
    pw1[u] = (two && k < in_dim) ? W8<WT>::load_stream(W + (long)(n0 + 1) * in_dim + k) : W8<WT>::zero();
  }
  {
    const float* s0 = src + (grouped ? (long)grp * in_dim : 0);
    const bool ln = (pro == PRO_LN || pro == PRO_GROUP_LN_RELU);
    const float* lw = (pro == PRO_GROUP_LN_RELU && grp) ? lnw1 : lnw0;
    const float* lb = (pro == PRO_GROUP_LN_RELU && grp) ? lnb1 : lnb0;
    rows_to_lds(s0, src_stride, xs, in_dim, B, ln, lw, lb, eps, pro == PRO_GROUP_LN_RELU || pro == PRO_GROUP_RELU);
  }
  __syncthreads();
  if (!live) return;
  float a0[HB_MAX], a1[HB_MAX];
#pragma unroll
  for (int b = 0; b < HB_MAX; ++b) a0[b] = a1[b] = 0.f;
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int k = u * 512 + lane * 8;
    if (k < in_dim) {
#pragma unroll
      for (int b = 0; b < HB_MAX; ++b)
        if (b < B) {
          a0[b] += W8<WT>::dot(pw0[u], xs + b * in_dim + k);
          a1[b] += W8<WT>::dot(pw1[u], xs + b * in_dim + k);
        }
    }
  }
  for (int k = 2 * 512 + lane * 8; k < in_dim; k += 512) {
    const typename W8<WT>::reg w0 = W8<WT>::load(W + (long)n0 * in_dim + k);
    const typename W8<WT>::reg w1 = two ? W8<WT>::load(W + (long)(n0 + 1) * in_dim + k) : W8<WT>::zero();
#pragma unroll
    for (int b = 0; b < HB_MAX; ++b)
      if (b < B) {
        a0[b] += W8<WT>::dot(w0, xs + b * in_dim + k);
        a1[b] += W8<WT>::dot(w1, xs + b * in_dim + k);
      }
  }
#pragma unroll
  for (int b = 0; b < HB_MAX; ++b)
    if (b < B) {
      const float s0 = wave_sum(a0[b]), s1 = wave_sum(a1[b]);
      if (lane == 0) {
        dst[(long)b * 2 * out_dim + (long)grp * out_dim + n0] = s0 + bb[n0];
        if (two) dst[(long)b * 2 * out_dim + (long)grp * out_dim + n0 + 1] = s1 + bb[n0 + 1];
      }
    }
}

extern "C" int deer_head_fc(const float* src, int src_stride, int in_dim, int pro, const float* lnw0, const float* lnb0,
                            const float* lnw1, const float* lnb1, const void* W0, const float* b0, const void* W1, const float* b1,
                            int out_dim, float* dst, int B, float eps, const int* ctl, int kind, int layer, int w_is_f32, void* stream) {
  if (in_dim <= 0 || (in_dim & 7) || out_dim <= 0 || pro < 0 || pro > 3 || B <= 0 || B > DEER_MAX_ENVS || (src_stride & 3) ||
      ((pro == PRO_LN || pro == PRO_GROUP_LN_RELU) && in_dim > 2048))
    return DEER_ERR_SHAPE;
  dim3 grid((out_dim + 7) / 8, 2);
  for (int e0 = 0; e0 < B; e0 += HB_MAX) {                 // batches above HB_MAX environments: one launch per chunk
    const int nb = B - e0 < HB_MAX ? B - e0 : HB_MAX;
    const int smem = (nb * in_dim + 16) * (int)sizeof(float);
    if (smem > 64 * 1024) return DEER_ERR_SHAPE;
    const float* s_ = src + (long)e0 * src_stride;
    float* d_ = dst + (long)e0 * 2 * out_dim;
    if (w_is_f32 == 1) hipLaunchKernelGGL(head_fc_kernel<float>, grid, dim3(256), smem, reinterpret_cast<hipStream_t>(stream), s_, src_stride, in_dim, pro, lnw0,
                         lnb0, lnw1, lnb1, reinterpret_cast<const float*>(W0), b0, reinterpret_cast<const float*>(W1), b1, out_dim, d_,
                         nb, eps, ctl, kind, layer, B);
    else if (w_is_f32 == 2) hipLaunchKernelGGL(head_fc_kernel<half_t>, grid, dim3(256), smem, reinterpret_cast<hipStream_t>(stream), s_, src_stride, in_dim, pro, lnw0,
                         lnb0, lnw1, lnb1, reinterpret_cast<const half_t*>(W0), b0, reinterpret_cast<const half_t*>(W1), b1, out_dim, d_,
                         nb, eps, ctl, kind, layer, B);
    else hipLaunchKernelGGL(head_fc_kernel<bf16_t>, grid, dim3(256), smem, reinterpret_cast<hipStream_t>(stream), s_, src_stride, in_dim, pro, lnw0,
                         lnb0, lnw1, lnb1, reinterpret_cast<const bf16_t*>(W0), b0, reinterpret_cast<const bf16_t*>(W1), b1, out_dim, d_,
                         nb, eps, ctl, kind, layer, B);
  }
  DEER_LAUNCH_CHECK();
  return DEER_OK;
}

// EXIT_FLAG of an environment, published BEHIND its EXIT_LAYER.  In an env batch the gather of a compaction layer (resadd_body.h) may
// run beside this exit check and must never see FLAG = 1 together with a stale LAYER (its workgroups would build different packings):
// agent-scope release in front of a relaxed agent-scope store of the flag, the reader loads the flags, fences (acquire) and then loads
// the layers.  One environment: nothing reads the pair concurrently (no compaction) - plain store.
__device__ __forceinline__ void publish_exit_flag(int* ctl, int B) {
  if (B > 1) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the compiler may drop the wait behind the write-back (MI355X guide, G16)
    __hip_atomic_store(ctl + CTL_EXIT_FLAG, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    ctl[CTL_EXIT_FLAG] = 1;
  }
}

// ---- output Linear (6 tanh + 1 sigmoid) + exit gate, per environment ----------------------------------------------
// One workgroup per environment.  src: [B][src_stride] (the last hidden Linear's output, [2][in_dim] per env).
// ctl: env b at ctl + b*CTL_WORDS.  State tensors h/c: [L][B][H] (LH = L, sH = H).  action_dbg: [B][64] f32 (64-float stride per
// environment: pose 6 A | gripper prob A | logit A, A = multi_step_action) or NULL.
// The work of one environment b as a device function of 512 threads: head_final_kernel runs it one workgroup per environment;
// head_fused_kernel (below) runs it as the last phase of the one-launch evaluation, with the hidden Linear's output staged in LDS
// (src0 / src1 then point into LDS) and the new LSTM state taken from the tagged exchange granules (GRAN) instead of h_tmp / c_tmp.
// lds: 2 * in_dim + 16 + 64 + 4 floats.  skip_checked: the caller already evaluated head_skip().
template <typename WT, bool GRAN>
__device__ __forceinline__ void head_final_body(int b, const float* s0, const float* s1, float* lds, int in_dim, int pro,
                                                const float* __restrict__ lnw0, const float* __restrict__ lnb0,
                                                const float* __restrict__ lnw1, const float* __restrict__ lnb1,
                                                const WT* __restrict__ Wa, const float* __restrict__ ba,
                                                const WT* __restrict__ Wg, const float* __restrict__ bg, int* ctl0,
                                                int kind, int layer, int slot, const float* __restrict__ thresholds,
                                                int force, int thr_type, int leq, const float* __restrict__ h_tmp,
                                                const float* __restrict__ c_tmp, float* __restrict__ h_state,
                                                float* __restrict__ c_state, int L, int H, int B,
                                                float* __restrict__ action_dbg, float eps, int A,
                                                float* __restrict__ act_ext, const unsigned long long* hgran,
                                                const unsigned long long* cgran) {
  if (head_skip(ctl0, kind, layer, B)) {
    // Every workgroup of a CHECK launch reports exactly once, whatever path it takes: a check nobody needed (stage hold)
    // still tells the host that this segment is over, and a workgroup that starts late may see ALL_EXITED raised by a
    // sibling of the SAME launch - if it left silently the arrival counter would never reach B and no verdict would be
    // published (seen with two processes time-slicing one GPU).
    if (kind == KIND_CHECK && threadIdx.x == 0) check_done(ctl0, slot, B);
    return;
  }
  HKT(0);
  int* ctl = (ctl0 != nullptr) ? ctl0 + b * CTL_WORDS : nullptr;
  if (ctl != nullptr) {
    // this environment already exited in this step, or (stage hold) does not need this evaluation
    if (ctl[CTL_EXIT_FLAG] != 0 ||
        (ctl[CTL_HOLD] != 0 && (kind == KIND_PSEUDO || (kind == KIND_CHECK && layer < ctl[CTL_CUR_EXIT_ID])))) {
      if (kind == KIND_CHECK && threadIdx.x == 0) check_done(ctl0, slot, B);
      return;
    }
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const WT* Wrow = (wave < 6) ? Wa + (long)wave * in_dim : Wg;
  typename W8<WT>::reg pw[2];                        // first two k-steps of this wave's output row, requested before the LayerNorm staging
#pragma unroll
  for (int u = 0; u < 2; ++u) {
    const int k = u * 512 + lane * 8;
    pw[u] = (wave < 7 && k < in_dim) ? W8<WT>::load(Wrow + k) : W8<WT>::zero();
  }
  HKT(1);
  float* xa = lds;                    // actions-head input [in_dim]
  float* xg = lds + in_dim;           // gripper-head input [in_dim]
  float* red = xg + in_dim;           // [16]
  float* outv = red + 16;             // [64]: 7 A raw outputs (A = multi_step_action <= 8), then the A gripper logits
  int* flag = reinterpret_cast<int*>(outv + 64);
  // the control block of this environment and the threshold of this check: staged by the one wave with no output row (wave 7), so that
  // thread 0's protocol section below reads LDS instead of walking a chain of dependent L2 round trips, and the host mirror leaves as
  // ONE wave store of the staged block (it used to be 64 serial loads + stores to host memory: 7.3 us of every committing evaluation)
  __shared__ int sctl[CTL_WORDS];
  __shared__ float sthr;
  __shared__ int sshadow;
  if (wave == 7 && ctl != nullptr) {
    sctl[lane] = ctl[lane];
    if (lane == 0) {
      sthr = (kind == KIND_CHECK && thresholds != nullptr && slot >= 0) ? thresholds[slot] : 0.f;
      sshadow = ctl0[CTL_SHADOW];
    }
  }
  float bias_w = 0.f;                                 // bias of this wave's output row, requested beside the weights
  if (A == 1 && wave < 7 && lane == 0) bias_w = (wave < 6) ? ba[wave] : bg[0];
  if (pro == PRO_GROUP_LN_RELU && in_dim <= 2048 && (in_dim & 3) == 0) {
    if (wave == 0) wave_ln_row(s0, xa, in_dim, lnw0, lnb0, eps, true);
    else if (wave == 1) wave_ln_row(s1, xg, in_dim, lnw1, lnb1, eps, true);
  } else if (pro == PRO_GROUP_LN_RELU) {
    block_ln(s0, xa, in_dim, lnw0, lnb0, eps, true, red);
    block_ln(s1, xg, in_dim, lnw1, lnb1, eps, true, red);
  } else if (pro == PRO_LN) {
    block_ln(s0, xa, in_dim, lnw0, lnb0, eps, false, red);
    __syncthreads();
    for (int i = threadIdx.x; i < in_dim; i += blockDim.x) xg[i] = xa[i];
  } else {
    for (int i = threadIdx.x; i < in_dim; i += blockDim.x) {
      xa[i] = (pro == PRO_GROUP_RELU) ? fmaxf(s0[i], 0.f) : s0[i];
      xg[i] = (pro == PRO_GROUP_RELU) ? fmaxf(s1[i], 0.f) : s1[i];
    }
  }
  if (threadIdx.x == 0) *flag = 0;
  __syncthreads();
  HKT(2);
  if (wave < 7) {
    const float* x = (wave < 6) ? xa : xg;
    float a = 0.f;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int k = u * 512 + lane * 8;
      if (k < in_dim) a += W8<WT>::dot(pw[u], x + k);
    }
    for (int k = 2 * 512 + lane * 8; k < in_dim; k += 512) a += W8<WT>::dot(W8<WT>::load(Wrow + k), x + k);
    a = wave_sum(a);
    if (lane == 0 && A == 1) {                         // every wave finishes its own output: tanh (pose) / logit + sigmoid (gripper) side by side
      const float r = a + bias_w;
      if (wave < 6) outv[wave] = tanhf(r);
      else { outv[6] = sigmoidf_(r); outv[7] = r; }
    }
  }
  if (A > 1) {
    // multi_step_action (action_head.py:472-473): 6 A pose rows of Wa then A gripper rows of Wg; wave w takes rows w, w + 8, ...
    for (int row = wave; row < 7 * A; row += 8) {
      const bool pose_row = row < 6 * A;
      const WT* wr = pose_row ? Wa + (long)row * in_dim : Wg + (long)(row - 6 * A) * in_dim;
      const float* x = pose_row ? xa : xg;
      float a = 0.f;
      for (int k = lane * 8; k < in_dim; k += 512) a += W8<WT>::dot(W8<WT>::load(wr + k), x + k);
      a = wave_sum(a);
      if (lane == 0) outv[row] = a + (pose_row ? ba[row] : bg[row - 6 * A]);
    }
  }
  __syncthreads();
  HKT(3);
  if (threadIdx.x == 0 && A > 1) {
    // ---- the same protocol as below over 7 A values; actions live in act_ext[b] = {previous | committed | ensemble}[64]; the first
    // executed action ([pose 0..5, gripper 0, logit 0]) is mirrored into the control block's 8-float fields ----
    const int NP = 6 * A, NA = 7 * A;
    for (int i = 0; i < A; ++i) outv[NA + i] = outv[NP + i];                      // gripper logits
    for (int i = 0; i < NP; ++i) outv[i] = tanhf(outv[i]);
    for (int i = NP; i < NA; ++i) outv[i] = sigmoidf_(outv[i]);
    if (action_dbg != nullptr)
      for (int i = 0; i < 8 * A; ++i) action_dbg[b * 64 + i] = outv[i];
    bool commit = false;
    if (ctl != nullptr) {
      float* prev = act_ext + (long)b * 256;
      float* outa = prev + 64;
      float* ens = prev + 128;
      float* deltas = reinterpret_cast<float*>(ctl + CTL_DELTAS);
      const bool hold = ctl[CTL_HOLD] != 0;
      if (kind == KIND_PSEUDO) {
        for (int i = 0; i < NA; ++i) prev[i] = outv[i];
        ctl[CTL_PREV_REAL] = 0;
      } else if (kind == KIND_CHECK && !hold) {
        float delta;                                    // value_net.py:105-117 over all 6 A pose values
        if (thr_type == THR_COSINE) {
          float na = 0.f, nb = 0.f, ab = 0.f;
          for (int i = 0; i < NP; ++i) { na += outv[i] * outv[i]; nb += prev[i] * prev[i]; }
          na = fmaxf(sqrtf(na), 1e-5f); nb = fmaxf(sqrtf(nb), 1e-5f);
          for (int i = 0; i < NP; ++i) ab += (outv[i] / na) * (prev[i] / nb);
          delta = 1.f - ab;
        } else {
          float acc = 0.f;
          for (int i = 0; i < NP; ++i) {
            const float dd = fabsf(outv[i] - prev[i]);
            if (thr_type == THR_L2) acc += dd * dd;
            else if (thr_type == THR_MEAN) acc += dd;
            else acc = fmaxf(acc, dd);
          }
          delta = (thr_type == THR_L2) ? sqrtf(acc / NP) : (thr_type == THR_MEAN ? acc / NP : acc);
        }
        if (slot >= 0 && slot < 16) deltas[slot] = delta;
        const bool two = ctl[CTL_PREV_REAL] != 0;
        for (int i = 0; i < NA; ++i) ens[i] = two ? 0.5f * (prev[i] + outv[i]) : outv[i];
        reinterpret_cast<float*>(ctl + CTL_ENS_ACTION)[7] = two ? 2.f : 1.f;
        ctl[CTL_PREV_REAL] = 1;
        for (int i = 0; i < NA; ++i) prev[i] = outv[i];
        const bool below = delta <= thresholds[slot];
        if ((below == (leq != 0)) || force) {
          ctl[CTL_CUR_EXIT_ID] = layer;
          commit = true;
        }
      } else {
        commit = true;
      }
      ctl[CTL_N_EVALS] += 1;
      const bool shadow_on = ctl0[CTL_SHADOW] != 0;
      if (commit && shadow_on) {
        if (ctl[CTL_COMMITTED] != 0) commit = false;
        else ctl[CTL_COMMITTED] = 1;
      }
      if (commit) {
        float* o8 = reinterpret_cast<float*>(ctl + CTL_OUT_ACTION);
        for (int i = 0; i < 8 * A; ++i) outa[i] = outv[i];
        for (int i = 0; i < 6; ++i) o8[i] = outv[i];
        o8[6] = outv[NP];
        o8[7] = outv[NA];
        ctl[CTL_EXIT_LAYER] = layer;
        if (!shadow_on) {
          publish_exit_flag(ctl, B);                      // EXIT_LAYER ordered in front of EXIT_FLAG (resadd_body.h: rowmap_dropped_mask)
          int* hm = host_mirror(ctl0);
          if (hm != nullptr) {
            for (int i = 0; i < CTL_WORDS; ++i) hm[CTL_WORDS * (1 + b) + i] = ctl[i];
            float* hx = reinterpret_cast<float*>(hm + CTL_WORDS * (1 + B)) + (long)b * 128;   // mirror extension: committed | ensemble
            for (int i = 0; i < 64; ++i) { hx[i] = outa[i]; hx[64 + i] = ens[i]; }
          }
          if (B == 1) {
            ctl0[CTL_ALL_EXITED] = 1;
          } else {
            __threadfence_system();
            if (atomicAdd(&ctl0[CTL_N_EXITED], 1) + 1 == B) ctl0[CTL_ALL_EXITED] = 1;
          }
        }
      }
      if (kind == KIND_CHECK) check_done(ctl0, slot, B);
    }
    *flag = commit ? 1 : 0;
  }
  if (wave == 0 && A == 1) {
    // ---- protocol section: lane 0 decides on the STAGED control block (every write goes to the device block and to the staged copy),
    //      the wave then stores the staged block into the host mirror, lane 0 publishes ----
    int commit_i = 0, mirror_i = 0;
    if (lane == 0) {
      float cur[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) cur[i] = outv[i];      // tanh(pose 0..5), gripper prob, gripper logit (MLPSigmoidHead with_logits)
      if (action_dbg != nullptr)
        for (int i = 0; i < 8; ++i) action_dbg[b * 64 + i] = cur[i];
      HKT(4);
      bool commit = false;
      if (ctl != nullptr) {
        auto wi = [&](int idx, int v) { ctl[idx] = v; sctl[idx] = v; };
        auto wf = [&](int idx, float v) { reinterpret_cast<float*>(ctl)[idx] = v; reinterpret_cast<float*>(sctl)[idx] = v; };
        const float* sprev = reinterpret_cast<const float*>(sctl + CTL_PREV_ACTION);
        const bool hold = sctl[CTL_HOLD] != 0;
        if (kind == KIND_PSEUDO) {
          for (int i = 0; i < 8; ++i) wf(CTL_PREV_ACTION + i, cur[i]);
          wi(CTL_PREV_REAL, 0);                           // not a member of action_list (value_net.py:120-123)
        } else if (kind == KIND_CHECK && !hold) {
          float prev[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) prev[i] = sprev[i];
          float delta;                                    // value_net.py:105-117
          if (thr_type == THR_COSINE) {
            float na = 0.f, nb = 0.f, ab = 0.f;
            for (int i = 0; i < 6; ++i) { na += cur[i] * cur[i]; nb += prev[i] * prev[i]; }
            na = fmaxf(sqrtf(na), 1e-5f); nb = fmaxf(sqrtf(nb), 1e-5f);
            for (int i = 0; i < 6; ++i) ab += (cur[i] / na) * (prev[i] / nb);
            delta = 1.f - ab;
          } else {
            float acc = 0.f;
            for (int i = 0; i < 6; ++i) {
              const float d = fabsf(cur[i] - prev[i]);
              if (thr_type == THR_L2) acc += d * d;
              else if (thr_type == THR_MEAN) acc += d;
              else acc = fmaxf(acc, d);
            }
            delta = (thr_type == THR_L2) ? sqrtf(acc / 6.f) : (thr_type == THR_MEAN ? acc / 6.f : acc);
          }
          if (slot >= 0 && slot < 16) wf(CTL_DELTAS + slot, delta);
          {                                               // get_ensemble_action(): mean over action_list[-2:] of this step (value_net.py:92-95)
            const bool two = sctl[CTL_PREV_REAL] != 0;
            for (int i = 0; i < 7; ++i) wf(CTL_ENS_ACTION + i, two ? 0.5f * (prev[i] + cur[i]) : cur[i]);
            wf(CTL_ENS_ACTION + 7, two ? 2.f : 1.f);
            wi(CTL_PREV_REAL, 1);
          }
          for (int i = 0; i < 8; ++i) wf(CTL_PREV_ACTION + i, cur[i]);   // action_list.append(action)
          const bool below = delta <= sthr;
          if ((below == (leq != 0)) || force) {           // value_net.py:293
            wi(CTL_CUR_EXIT_ID, layer);
            commit = true;
          }
        } else {                                          // KIND_COMMIT, or CHECK while holding a stage
          commit = true;
        }
        HKT(5);
        wi(CTL_N_EVALS, sctl[CTL_N_EVALS] + 1);
        if (commit && sshadow != 0) {
          // calibration: remember the FIRST exit that fires, keep evaluating the deeper exits (no EXIT_FLAG)
          if (sctl[CTL_COMMITTED] != 0) commit = false;
          else wi(CTL_COMMITTED, 1);
          if (commit) {
            for (int i = 0; i < 8; ++i) wf(CTL_OUT_ACTION + i, cur[i]);
            wi(CTL_EXIT_LAYER, layer);
          }
        } else if (commit) {
          for (int i = 0; i < 8; ++i) wf(CTL_OUT_ACTION + i, cur[i]);
          wi(CTL_EXIT_LAYER, layer);                      // EXIT_LAYER ordered in front of EXIT_FLAG (resadd_body.h: rowmap_dropped_mask)
          publish_exit_flag(ctl, B);
          sctl[CTL_EXIT_FLAG] = 1;
          mirror_i = host_mirror(ctl0) != nullptr ? 1 : 0;
        }
      }
      commit_i = commit ? 1 : 0;
    }
    mirror_i = __builtin_amdgcn_readfirstlane(mirror_i);
    if (mirror_i) {
      // this environment's verdict, readable by the host right away: the staged block as one wave store (lane 0's LDS writes above are
      // in program order in front of these reads; the release stores of check_done / the system fence below wait for the wave's stores)
      int* hmw = host_mirror(ctl0);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      hmw[CTL_WORDS * (1 + b) + lane] = sctl[lane];
    }
    if (lane == 0) {
      if (mirror_i || (commit_i && ctl != nullptr && sshadow == 0)) {
        // last environment of the batch to exit raises the batch-global flag (one workgroup per environment runs
        // concurrently, hence the device-scope atomic)
        if (B == 1) {
          ctl0[CTL_ALL_EXITED] = 1;
        } else {
          __threadfence_system();
          if (atomicAdd(&ctl0[CTL_N_EXITED], 1) + 1 == B) ctl0[CTL_ALL_EXITED] = 1;
        }
      }
      HKT(6);
      if (ctl != nullptr && kind == KIND_CHECK) check_done(ctl0, slot, B);
      HKT(7);
      *flag = commit_i;
    }
  }
  __syncthreads();
  HKT(8);
  if (!GRAN && *flag && h_state != nullptr && (H & 3) == 0) {   // DeterministicDecoder.hidden_state = h_n (action_head.py:555-556)
    const int h4 = H >> 2, n4 = L * h4;                   // float4 pieces, two per thread in flight (L * H = 4096: 512 threads, one round)
    for (int i0 = threadIdx.x; i0 < n4; i0 += 2 * blockDim.x) {
      float4 hv[2], cv[2];
      long o[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int i = i0 + u * blockDim.x;
        const int l = min(i, n4 - 1) / h4, q = min(i, n4 - 1) - l * h4;
        o[u] = ((long)l * B + b) * H + (long)q * 4;
        hv[u] = *reinterpret_cast<const float4*>(h_tmp + o[u]);
        cv[u] = *reinterpret_cast<const float4*>(c_tmp + o[u]);
      }
#pragma unroll
      for (int u = 0; u < 2; ++u)
        if (i0 + u * (int)blockDim.x < n4) {
          *reinterpret_cast<float4*>(h_state + o[u]) = hv[u];
          *reinterpret_cast<float4*>(c_state + o[u]) = cv[u];
        }
    }
  } else if (*flag && h_state != nullptr) {
    for (int i = threadIdx.x; i < L * H; i += blockDim.x) {
      const int l = i / H, u = i - l * H;
      const long o = ((long)l * B + b) * H + u;
      if (GRAN) {     // written by other workgroups of THIS launch: read through the device-coherent granules (every one is complete by now)
        h_state[o] = __uint_as_float((unsigned)__hip_atomic_load(hgran + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        c_state[o] = __uint_as_float((unsigned)__hip_atomic_load(cgran + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
      } else {
        h_state[o] = h_tmp[o];
        c_state[o] = c_tmp[o];
      }
    }
  }
  HKT(9);
}

template <typename WT>
__global__ __launch_bounds__(512) void head_final_kernel(const float* __restrict__ src, int src_stride, int in_dim, int pro,
                                                         const float* __restrict__ lnw0, const float* __restrict__ lnb0,
                                                         const float* __restrict__ lnw1, const float* __restrict__ lnb1,
                                                         const WT* __restrict__ Wa, const float* __restrict__ ba,
                                                         const WT* __restrict__ Wg, const float* __restrict__ bg, int* ctl0,
                                                         int kind, int layer, int slot, const float* __restrict__ thresholds,
                                                         int force, int thr_type, int leq, const float* __restrict__ h_tmp,
                                                         const float* __restrict__ c_tmp, float* __restrict__ h_state,
                                                         float* __restrict__ c_state, int L, int H, int B,
                                                         float* __restrict__ action_dbg, float eps, int A,
                                                         float* __restrict__ act_ext) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int b = blockIdx.x;
  const bool grouped = (pro == PRO_GROUP_LN_RELU || pro == PRO_GROUP_RELU);
  const float* s0 = src + (long)b * src_stride;
  const float* s1 = s0 + (grouped ? in_dim : 0);
  head_final_body<WT, false>(b, s0, s1, lds, in_dim, pro, lnw0, lnb0, lnw1, lnb1, Wa, ba, Wg, bg, ctl0, kind, layer, slot, thresholds, force,
                             thr_type, leq, h_tmp, c_tmp, h_state, c_state, L, H, B, action_dbg, eps, A, act_ext, nullptr, nullptr);
}

// ------------------------------------------------------------------------------------------------------------------------------
// One head evaluation (pool -> L LSTM layers -> hidden Linears -> output Linear + exit gate) as ONE launch (round 5).
// The eight-launch form above is a chain of dependent latency-bound kernels (64-70 us at one environment, 42-82 MB of weights = 10-16 us
// of stream): every launch boundary drains the GPU, ramps up again and pays a cold HBM round trip for its weights.  Here G workgroups of
// 8 waves stay resident for the whole evaluation; wave w owns the SAME rows the separate kernels give a wave (one LSTM hidden unit = its
// four gate rows, two rows of a hidden Linear) and the vectors that separate the phases - pooled feature, every layer's h, the hidden
// Linears' outputs: 0.5-2 K floats per environment - travel between workgroups as DATA-TAGGED 8-byte granules {f32 value, tag} written
// and read with relaxed agent-scope atomics (the guide's hand-off R2: no flag, no fence; a consumer polls the granule itself).  The tag
// is the number of evaluations the exchange buffer has served (kept beside the error word): nothing is reset between evaluations.  Every phase requests its weight
// rows BEFORE it gathers its input, so the HBM latency of a phase hides behind the hand-off it has to wait for anyway.
// The arithmetic per row (k split over the lanes in 512-column steps, in-wave butterfly sums, LayerNorm one wave per row, the output
// Linear + exit gate = head_final_body) is that of the separate kernels; only control steps with a control block use this form (the
// window / calibration path and static exits keep the separate kernels).  Every spin is bounded: a timeout raises err[0].
// Residency: G <= 128 workgroups of 512 threads - they need no particular placement and no other kernel ever waits for them.
#define HF_SPIN_LIMIT (1 << 22)

#define HF_LN_MAXE 4     // float4 per lane: LayerNorm rows of up to 1024 values (LSTM hidden size, hidden Linears)
// rows_ln_to_lds in two halves: gamma / beta are requested BEFORE the phase's hand-off wait, the rows are normalised after it (the
// arithmetic and its order are those of rows_ln_to_lds: one wave per row, in-wave sums)
struct hf_ln_regs { float4 gw[HF_LN_MAXE], gb[HF_LN_MAXE]; };
__device__ __forceinline__ void hf_ln_prefetch(hf_ln_regs& r, const float* __restrict__ w, const float* __restrict__ bta, int n) {
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int e = 0; e < HF_LN_MAXE; ++e) {
    const int i = (lane + 64 * e) * 4;
    r.gw[e] = (w != nullptr && i < n) ? *reinterpret_cast<const float4*>(w + i) : float4{0.f, 0.f, 0.f, 0.f};
    r.gb[e] = (bta != nullptr && i < n) ? *reinterpret_cast<const float4*>(bta + i) : float4{0.f, 0.f, 0.f, 0.f};
  }
}
__device__ __forceinline__ void hf_ln_apply(const hf_ln_regs& r, const float* src, long stride, float* dst, int n, int B, float eps, bool relu) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int b = wave; b < B; b += nw) {
    const float* s = src + (long)b * stride;
    float* d = dst + b * n;
    float4 v[HF_LN_MAXE];
    float sum = 0.f;
#pragma unroll
    for (int e = 0; e < HF_LN_MAXE; ++e) {
      const int i = (lane + 64 * e) * 4;
      v[e] = i < n ? *reinterpret_cast<const float4*>(s + i) : float4{0.f, 0.f, 0.f, 0.f};
      sum += v[e].x + v[e].y + v[e].z + v[e].w;
    }
    const float mean = wave_sum(sum) / n;
    float var = 0.f;
#pragma unroll
    for (int e = 0; e < HF_LN_MAXE; ++e)
      if ((lane + 64 * e) * 4 < n) {
        const float a = v[e].x - mean, bq = v[e].y - mean, c = v[e].z - mean, dd = v[e].w - mean;
        var += a * a + bq * bq + c * c + dd * dd;
      }
    const float rstd = rsqrtf(wave_sum(var) / n + eps);
#pragma unroll
    for (int e = 0; e < HF_LN_MAXE; ++e) {
      const int i = (lane + 64 * e) * 4;
      if (i < n) {
        float4 y;
        y.x = (v[e].x - mean) * rstd * r.gw[e].x + r.gb[e].x; y.y = (v[e].y - mean) * rstd * r.gw[e].y + r.gb[e].y;
        y.z = (v[e].z - mean) * rstd * r.gw[e].z + r.gb[e].z; y.w = (v[e].w - mean) * rstd * r.gw[e].w + r.gb[e].w;
        if (relu) { y.x = fmaxf(y.x, 0.f); y.y = fmaxf(y.y, 0.f); y.z = fmaxf(y.z, 0.f); y.w = fmaxf(y.w, 0.f); }
        *reinterpret_cast<float4*>(d + i) = y;
      }
    }
  }
}

__device__ __forceinline__ unsigned long long hf_pack(float v, unsigned tag) {
  return ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v);
}
__device__ __forceinline__ void hf_publish(unsigned long long* g, float v, unsigned tag) {
  __hip_atomic_store(g, hf_pack(v, tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// n tagged granules -> dst (LDS), swept by the first `gw` waves of the workgroup (the others wait at the barrier behind them).  A lane
// keeps up to 8 requests in flight; granules that have not arrived are requested AGAIN TOGETHER after a short sleep - re-polling them one
// after the other (the first versions) costs a memory-side round trip per late granule: a freshly loaded copy of the first late one says
// nothing about the copies of the others, which were loaded before they arrived (profiles/r05_e_head_eval_fused_*: 56-94 us per
// evaluation against 51-67 for the eight launches).
__device__ __forceinline__ void hf_gather(const unsigned long long* g, int n, float* dst, unsigned tag, int* err, int gw) {
  if ((int)threadIdx.x >= 64 * gw) return;
  const int step = 64 * gw;
  for (int base = threadIdx.x; base < n; base += step * 8) {
    unsigned long long v[8];
    bool ok[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) ok[u] = base + u * step >= n;
    int spins = 0;
    while (true) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (!ok[u]) v[u] = __hip_atomic_load(g + base + u * step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      bool pending = false;
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (!ok[u]) {
          if ((unsigned)(v[u] >> 32) == tag) {
            ok[u] = true;
            dst[base + u * step] = __uint_as_float((unsigned)v[u]);
          } else {
            pending = true;
          }
        }
      if (!pending) break;
      if (++spins > HF_SPIN_LIMIT || (spins > 4096 && *(volatile int*)err != 0)) { *err = 1; break; }
      __builtin_amdgcn_s_sleep(4);
    }
  }
}

// workgroup barrier that orders LDS traffic only: __syncthreads() carries a release fence, which on gfx9 is `s_waitcnt vmcnt(0)` - it would
// drain the weight rows a phase has just requested and serialise their HBM round trip with the hand-off the phase waits for next
__device__ __forceinline__ void hf_lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}

template <typename WT, int HBT>
__global__ __launch_bounds__(512) void head_fused_kernel(deer_head_fused_args a) {
  const int B = a.B, H = a.H, d = a.d, L = a.L;
  int* ctl0 = a.ctl;
  if (head_skip(ctl0, a.kind, a.layer, B)) {
    if ((int)blockIdx.x < B && a.kind == KIND_CHECK && threadIdx.x == 0) check_done(ctl0, a.slot, B);
    return;
  }
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* raw = lds;                          // [B][max_in]: gathered input of the phase
  float* xs = lds + (long)B * a.max_in;      // [B][max_in] (grouped phases: [2][B][in]): the phase's normalised input
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int TW = gridDim.x * 8, gw = blockIdx.x * 8 + wave;
  // tag of this evaluation: err[1] counts the evaluations this exchange buffer has served; the workgroup that runs the last phase bumps
  // it when everything is over (every workgroup that has any work read it before: the last phase waits for all of their results)
  const unsigned tag = ((const volatile unsigned*)a.err)[1] + 1u;
  int n_stamp = 0;
#define HF_STAMP()                                                                     \
  do {                                                                                 \
    if (a.trace != nullptr && blockIdx.x == 0 && threadIdx.x == 0 && n_stamp < 64) a.trace[n_stamp] = wall_clock64(); \
    ++n_stamp;                                                                         \
  } while (0)
  HF_STAMP();
  unsigned long long* gX0 = a.xg;
  unsigned long long* gH = gX0 + (long)B * d;
  unsigned long long* gC = gH + (long)L * B * H;
  unsigned long long* gZ = gC + (long)L * B * H;

  // ---- LSTM layer 0: this wave's four gate rows are requested before anything else ----
  constexpr int LSTM_PF = 4;
  const int j0 = gw;                                                  // first hidden unit of this wave (others: j0 + TW, ...)
  const int jr0 = j0 < H ? j0 : H - 1;
  typename W8<WT>::reg wi[LSTM_PF][4];
  {
    const WT* w_ih = reinterpret_cast<const WT*>(a.w_ih[0]);
#pragma unroll
    for (int u = 0; u < LSTM_PF; ++u) {
      const int k = u * 512 + lane * 8;
#pragma unroll
      for (int q = 0; q < 4; ++q) wi[u][q] = k < d ? W8<WT>::load_stream(w_ih + ((long)q * H + jr0) * d + k) : W8<WT>::zero();
    }
  }
  // ---- phase 0: token pool (head_pool_kernel's arithmetic).  One environment: EVERY workgroup pools all d columns for itself - T rows of
  // d floats from L2, all T loads of a thread in flight - and the first hand-off disappears; env batches spread the pool over the launch ----
  if (HBT == 1) {
    for (int i = tid; i < d; i += 512) {
      const int sl = a.cmap != nullptr ? a.cmap[CMAP_ENV_SLOT] : 0;
      float acc = 0.f;
      if (sl >= 0) {
        const float* x = a.feats + ((long)sl * a.T) * d + i;
        acc = a.avg ? 0.f : -INFINITY;
        int n = 0;
        for (int t0 = 0; t0 < a.T; t0 += 8) {
          float v[8];
          bool on[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            on[u] = t0 + u < a.T && (a.key_mask == nullptr || a.key_mask[t0 + u] != 0);
            v[u] = on[u] ? x[(long)(t0 + u) * d] : 0.f;
          }
#pragma unroll
          for (int u = 0; u < 8; ++u)
            if (on[u]) { acc = a.avg ? acc + v[u] : fmaxf(acc, v[u]); ++n; }
        }
        if (a.avg) acc /= (float)max(n, 1);
      }
      xs[i] = acc;
    }
  }
  for (int e = blockIdx.x * 512 + tid; e < B * d && HBT > 1; e += gridDim.x * 512) {
    const int b = e / d, i = e - b * d;
    const int sl = a.cmap != nullptr ? a.cmap[CMAP_ENV_SLOT + b] : b;
    float acc = 0.f;
    if (sl >= 0) {
      const float* x = a.feats + ((long)sl * a.T) * d + i;
      const unsigned char* km = a.key_mask != nullptr ? a.key_mask + b * a.T : nullptr;
      acc = a.avg ? 0.f : -INFINITY;
      int n = 0;
      for (int t = 0; t < a.T; ++t) {
        if (km != nullptr && km[t] == 0) continue;
        const float v = x[(long)t * d];
        acc = a.avg ? acc + v : fmaxf(acc, v);
        ++n;
      }
      if (a.avg) acc /= (float)max(n, 1);
    }
    hf_publish(gX0 + e, acc, tag);
  }
  HF_STAMP();                                                        // pool done (this workgroup)
  // ---- LSTM layers ----
#pragma unroll
  for (int l = 0; l < 4; ++l) {
    if (l >= L) break;
    const int in_dim = l == 0 ? d : H;
    const WT* w_ih = reinterpret_cast<const WT*>(a.w_ih[l]);
    if (l > 0) {                                                     // this layer's rows: in flight while the input is gathered
#pragma unroll
      for (int u = 0; u < LSTM_PF; ++u) {
        const int k = u * 512 + lane * 8;
#pragma unroll
        for (int q = 0; q < 4; ++q) wi[u][q] = k < in_dim ? W8<WT>::load_stream(w_ih + ((long)q * H + jr0) * in_dim + k) : W8<WT>::zero();
      }
    }
    const bool ln = l > 0 && a.lstm_ln;
    const float* b_ih = a.b_ih[l];
    const float* ghh = a.ghh + (long)l * B * 4 * H;
    const float* c_prev = a.c_prev + (long)l * B * H;
    // everything else this phase reads from memory is requested before the hand-off wait as well: the LayerNorm affine of the input and
    // (lane b) the bias / recurrent half / cell state of this wave's first hidden unit
    hf_ln_regs lnr;
    if (ln) hf_ln_prefetch(lnr, a.ln_w[l > 0 ? l - 1 : 0], a.ln_b[l > 0 ? l - 1 : 0], in_dim);
    float pb[4] = {0.f, 0.f, 0.f, 0.f}, pg[4] = {0.f, 0.f, 0.f, 0.f}, pc = 0.f;
    if (lane < B && j0 < H) {
      const float* gp = ghh + (long)lane * 4 * H + j0;
#pragma unroll
      for (int q = 0; q < 4; ++q) { pb[q] = b_ih[q * H + j0]; pg[q] = gp[q * H]; }
      pc = c_prev[lane * H + j0];
    }
    hf_lds_barrier();                                                 // the previous phase's reads of raw / xs are over
    if (l > 0 || HBT > 1) hf_gather(l == 0 ? gX0 : gH + (long)(l - 1) * B * H, B * in_dim, ln ? raw : xs, tag, a.err, a.gather_waves);
    hf_lds_barrier();
    HF_STAMP();                                                      // input gathered
    if (ln) {
      hf_ln_apply(lnr, raw, in_dim, xs, in_dim, B, a.eps, false);
      hf_lds_barrier();
    }
    HF_STAMP();                                                      // input normalised
    for (int j = j0; j < H; j += TW) {
      float acc[4][HBT];
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int b = 0; b < HBT; ++b) acc[q][b] = 0.f;
      if (j == j0) {
#pragma unroll
        for (int u = 0; u < LSTM_PF; ++u) {
          const int k = u * 512 + lane * 8;
          if (k < in_dim) {
#pragma unroll
            for (int b = 0; b < HBT; ++b)
              if (b < B) {
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[q][b] += W8<WT>::dot(wi[u][q], xs + b * in_dim + k);
              }
          }
        }
      }
      for (int k = (j == j0 ? LSTM_PF * 512 : 0) + lane * 8; k < in_dim; k += 512) {
        typename W8<WT>::reg w[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) w[q] = W8<WT>::load(w_ih + ((long)q * H + j) * in_dim + k);
#pragma unroll
        for (int b = 0; b < HBT; ++b)
          if (b < B) {
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q][b] += W8<WT>::dot(w[q], xs + b * in_dim + k);
          }
      }
      float gi = 0.f, gf = 0.f, gg = 0.f, go = 0.f;
#pragma unroll
      for (int b = 0; b < HBT; ++b)
        if (b < B) {
          const float t0 = wave_sum(acc[0][b]), t1 = wave_sum(acc[1][b]), t2 = wave_sum(acc[2][b]), t3 = wave_sum(acc[3][b]);
          if (lane == b) { gi = t0; gf = t1; gg = t2; go = t3; }
        }
      if (lane < B) {
        const int b = lane;
        if (j != j0) {
          const float* gp = ghh + (long)b * 4 * H + j;
#pragma unroll
          for (int q = 0; q < 4; ++q) { pb[q] = b_ih[q * H + j]; pg[q] = gp[q * H]; }
          pc = c_prev[b * H + j];
        }
        gi += pb[0] + pg[0]; gf += pb[1] + pg[1]; gg += pb[2] + pg[2]; go += pb[3] + pg[3];
        const float c2 = sigmoidf_(gf) * pc + sigmoidf_(gi) * tanhf(gg);
        const float h2 = sigmoidf_(go) * tanhf(c2);
        const long o = ((long)l * B + b) * H + j;
        a.c_tmp[o] = c2;
        a.h_tmp[o] = h2;
        hf_publish(gC + o, c2, tag);
        hf_publish(gH + o, h2, tag);
      }
    }
    HF_STAMP();                                                      // layer published (this workgroup)
  }
  HF_STAMP();                                                        // LSTM layers published (this workgroup)
  // ---- hidden Linears of both MLP heads ----
  const float* lnw_last = L == 4 ? a.ln_w[3] : L == 3 ? a.ln_w[2] : L == 2 ? a.ln_w[1] : a.ln_w[0];   // LayerNorm of the last LSTM layer's output
  const float* lnb_last = L == 4 ? a.ln_b[3] : L == 3 ? a.ln_b[2] : L == 2 ? a.ln_b[1] : a.ln_b[0];
  long zoff = 0;
  int in_dim = H;
#pragma unroll
  for (int fi = 0; fi < 3; ++fi) {
    if (fi >= a.n_fc) break;
    const int out_dim = a.fc_dim[fi];
    const int PG = (out_dim + 1) / 2;                                // row pairs per group
    const int p0 = gw;
    const bool live0 = p0 < 2 * PG;
    const int grp0 = live0 ? p0 / PG : 0, n00 = live0 ? 2 * (p0 - grp0 * PG) : 0;
    const WT* W0 = reinterpret_cast<const WT*>(grp0 ? a.fw[fi][1] : a.fw[fi][0]);
    typename W8<WT>::reg pw0[2], pw1[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int k = u * 512 + lane * 8;
      pw0[u] = (live0 && k < in_dim) ? W8<WT>::load_stream(W0 + (long)n00 * in_dim + k) : W8<WT>::zero();
      pw1[u] = (live0 && n00 + 1 < out_dim && k < in_dim) ? W8<WT>::load_stream(W0 + (long)(n00 + 1) * in_dim + k) : W8<WT>::zero();
    }
    const float* bb0 = grp0 ? a.fb[fi][1] : a.fb[fi][0];
    const float pbias0 = live0 ? bb0[n00] : 0.f, pbias1 = (live0 && n00 + 1 < out_dim) ? bb0[n00 + 1] : 0.f;
    hf_ln_regs lnr0, lnr1;                                           // LayerNorm affine of the phase's input, requested before the wait
    if (fi == 0) {
      if (a.lstm_ln) hf_ln_prefetch(lnr0, lnw_last, lnb_last, H);
    } else if (a.mlp_ln) {
      hf_ln_prefetch(lnr0, a.fln_w[fi > 0 ? fi - 1 : 0][0], a.fln_b[fi > 0 ? fi - 1 : 0][0], in_dim);
      hf_ln_prefetch(lnr1, a.fln_w[fi > 0 ? fi - 1 : 0][1], a.fln_b[fi > 0 ? fi - 1 : 0][1], in_dim);
    }
    hf_lds_barrier();
    if (fi == 0) {                                                   // shared input: [LN of] the last LSTM layer's h
      hf_gather(gH + (long)(L - 1) * B * H, B * H, a.lstm_ln ? raw : xs, tag, a.err, a.gather_waves);
      hf_lds_barrier();
      if (a.lstm_ln) {
        hf_ln_apply(lnr0, raw, H, xs, H, B, a.eps, false);
        hf_lds_barrier();
      }
    } else {                                                         // grouped: x_g = relu([LN_g](z[b][g * in ..])) -> xs[g][b][in]
      hf_gather(gZ + zoff - (long)B * 2 * in_dim, B * 2 * in_dim, raw, tag, a.err, a.gather_waves);
      hf_lds_barrier();
      if (a.mlp_ln) {
        hf_ln_apply(lnr0, raw, 2 * in_dim, xs, in_dim, B, a.eps, true);
        hf_ln_apply(lnr1, raw + in_dim, 2 * in_dim, xs + (long)B * in_dim, in_dim, B, a.eps, true);
      } else {
#pragma unroll
        for (int g = 0; g < 2; ++g) rows_to_lds(raw + g * in_dim, 2 * in_dim, xs + (long)g * B * in_dim, in_dim, B, false, nullptr, nullptr, a.eps, true);
      }
      hf_lds_barrier();
    }
    for (int p = p0; p < 2 * PG; p += TW) {
      const int grp = p / PG, n0 = 2 * (p - grp * PG);
      const bool two = n0 + 1 < out_dim;
      const WT* W = reinterpret_cast<const WT*>(grp ? a.fw[fi][1] : a.fw[fi][0]);
      const float* bb = grp ? a.fb[fi][1] : a.fb[fi][0];
      const float* x = xs + (fi == 0 ? 0 : (long)grp * B * in_dim);
      float a0[HBT], a1[HBT];
#pragma unroll
      for (int b = 0; b < HBT; ++b) a0[b] = a1[b] = 0.f;
      if (p == p0) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const int k = u * 512 + lane * 8;
          if (k < in_dim) {
#pragma unroll
            for (int b = 0; b < HBT; ++b)
              if (b < B) {
                a0[b] += W8<WT>::dot(pw0[u], x + b * in_dim + k);
                a1[b] += W8<WT>::dot(pw1[u], x + b * in_dim + k);
              }
          }
        }
      }
      for (int k = (p == p0 ? 2 * 512 : 0) + lane * 8; k < in_dim; k += 512) {
        const typename W8<WT>::reg w0 = W8<WT>::load(W + (long)n0 * in_dim + k);
        const typename W8<WT>::reg w1 = two ? W8<WT>::load(W + (long)(n0 + 1) * in_dim + k) : W8<WT>::zero();
#pragma unroll
        for (int b = 0; b < HBT; ++b)
          if (b < B) {
            a0[b] += W8<WT>::dot(w0, x + b * in_dim + k);
            a1[b] += W8<WT>::dot(w1, x + b * in_dim + k);
          }
      }
#pragma unroll
      for (int b = 0; b < HBT; ++b)
        if (b < B) {
          const float s0 = wave_sum(a0[b]), s1 = wave_sum(a1[b]);
          if (lane == 0) {
            unsigned long long* z = gZ + zoff + (long)b * 2 * out_dim + (long)grp * out_dim + n0;
            hf_publish(z, s0 + (p == p0 ? pbias0 : bb[n0]), tag);
            if (two) hf_publish(z + 1, s1 + (p == p0 ? pbias1 : bb[n0 + 1]), tag);
          }
        }
    }
    HF_STAMP();                                                      // hidden Linear published (this workgroup)
    zoff += (long)B * 2 * out_dim;
    in_dim = out_dim;
  }
  // ---- output Linear + exit gate: workgroup b < B serves environment b (head_final_body) ----
  const int b = blockIdx.x;
  if (b >= B) return;
  __syncthreads();
  int pro;
  const float *s0, *s1, *lw0, *lb0, *lw1, *lb1;
  if (a.n_fc == 0) {
    hf_gather(gH + ((long)(L - 1) * B + b) * H, H, raw, tag, a.err, a.gather_waves);
    pro = a.lstm_ln ? PRO_LN : PRO_RAW;
    s0 = s1 = raw;
    lw0 = lw1 = lnw_last; lb0 = lb1 = lnb_last;
  } else {
    hf_gather(gZ + zoff - (long)B * 2 * in_dim + (long)b * 2 * in_dim, 2 * in_dim, raw, tag, a.err, a.gather_waves);
    pro = a.mlp_ln ? PRO_GROUP_LN_RELU : PRO_GROUP_RELU;
    s0 = raw; s1 = raw + in_dim;
    const int nf = a.n_fc;
    lw0 = nf == 3 ? a.fln_w[2][0] : nf == 2 ? a.fln_w[1][0] : a.fln_w[0][0]; lb0 = nf == 3 ? a.fln_b[2][0] : nf == 2 ? a.fln_b[1][0] : a.fln_b[0][0];
    lw1 = nf == 3 ? a.fln_w[2][1] : nf == 2 ? a.fln_w[1][1] : a.fln_w[0][1]; lb1 = nf == 3 ? a.fln_b[2][1] : nf == 2 ? a.fln_b[1][1] : a.fln_b[0][1];
  }
  __syncthreads();
  HF_STAMP();                                                        // final input gathered
  head_final_body<WT, true>(b, s0, s1, xs, in_dim, pro, lw0, lb0, lw1, lb1, reinterpret_cast<const WT*>(a.Wa), a.ba, reinterpret_cast<const WT*>(a.Wg),
                            a.bg, ctl0, a.kind, a.layer, a.slot, a.thresholds, a.force, a.thr_type, a.leq, a.h_tmp, a.c_tmp, a.h_state, a.c_state, L,
                            H, B, a.action_dbg, a.eps, a.A, a.act_ext, gH, gC);
  __syncthreads();
  HF_STAMP();                                                        // done
  if (b == 0 && threadIdx.x == 0) ((volatile unsigned*)a.err)[1] = tag;      // one environment: workgroup 0 is the only one that gets here
}

// granules the exchange buffer of deer_head_fused needs (8 bytes each) for B environments
extern "C" long deer_head_fused_granules(int B, int d, int H, int L, int n_fc, const int* fc_dim) {
  long n = (long)B * d + 2L * L * B * H;
  for (int i = 0; i < n_fc; ++i) n += (long)B * 2 * fc_dim[i];
  return n;
}

// One evaluation of the head as one launch.  Arguments: the union of the separate kernels' (deer_head_pool / deer_head_lstm_layer_pre /
// deer_head_fc / deer_head_final_multi); ghh [L][B][4H] from deer_head_lstm_hh; xg: deer_head_fused_granules() x 8 bytes, zeroed once;
// err: int32, raised when a hand-off timed out.  Needs a control block (the tag comes from its sequence word), bf16 or f32 weights,
// L <= 4, n_fc <= 3, B <= 8 and B * 2 * max(d, H, 2 * fc_dim) * 4 <= 150 KB of LDS; returns DEER_ERR_SHAPE otherwise (the caller then
// uses the separate kernels).
extern "C" int deer_head_fused(const deer_head_fused_args* args, int w_is_f32, int n_workgroups, void* stream) {
  if (args == nullptr) return DEER_ERR_SHAPE;
  deer_head_fused_args a = *args;
  if (w_is_f32) return DEER_ERR_SHAPE;                    // bf16 weights only: the fp32 / fp16 arithmetics keep the separate kernels
  // one environment only for now: the 8-environment instantiation (32 accumulators + the prefetched rows + the LayerNorm staging) spills
  // 273 VGPRs under hipcc - env batches keep the separate kernels
  if (args->B != 1) return DEER_ERR_SHAPE;
  if (a.ctl == nullptr || a.xg == nullptr || a.err == nullptr || a.ghh == nullptr || a.B <= 0 || a.B > HB_MAX || a.L <= 0 || a.L > 4 || a.n_fc < 0 ||
      a.n_fc > 3 || (a.d & 7) || (a.H & 7) || a.d > 2048 || a.H > 1024 || a.kind == KIND_COMMIT || a.A < 1 || a.A > 8)
    return DEER_ERR_SHAPE;
  if (a.kind == KIND_CHECK && (a.thresholds == nullptr || a.slot < 0 || a.slot > 61)) return DEER_ERR_SHAPE;
  int max_in = a.d > a.H ? a.d : a.H;
  for (int i = 0; i < a.n_fc; ++i) {
    if ((a.fc_dim[i] & 7) || a.fc_dim[i] > 1024) return DEER_ERR_SHAPE;
    if (2 * a.fc_dim[i] > max_in) max_in = 2 * a.fc_dim[i];
  }
  a.max_in = max_in;
  static const int gwaves = [] { const char* e = getenv("DEER_HF_GATHER_WAVES"); const int v = e ? atoi(e) : 8; return v < 1 ? 1 : (v > 8 ? 8 : v); }();
  a.gather_waves = gwaves;
  const int smem = a.B * 2 * max_in * 4 + 512;
  if (smem > 150 * 1024) return DEER_ERR_SHAPE;
  if (n_workgroups < a.B) n_workgroups = a.B;
  if (n_workgroups > 256) n_workgroups = 256;
  static std::atomic<bool> attr_set{false};
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&head_fused_kernel<bf16_t, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess ||
        false)
      return DEER_ERR_LAUNCH;
    attr_set = true;
  }
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL((head_fused_kernel<bf16_t, 1>), dim3(n_workgroups), dim3(512), smem, st, a);
  DEER_LAUNCH_CHECK();
  return DEER_OK;
}

// A = multi_step_action (action_head.py:472-473): Wa has 6 A rows, Wg A rows; A > 1 needs act_ext ([B][4][64] f32: previous /
// committed / ensemble action of 7 A values + A logits) when ctl is given.  action_dbg: [B][64] (pose 6 A | gripper A | logit A).
extern "C" int deer_head_final_multi(const float* src, int src_stride, int in_dim, int pro, const float* lnw0, const float* lnb0,
                                     const float* lnw1, const float* lnb1, const void* Wa, const float* ba, const void* Wg, const float* bg,
                                     int* ctl, int kind, int layer, int slot, const float* thresholds, int force, int thr_type, int leq,
                                     const float* h_tmp, const float* c_tmp, float* h_state, float* c_state, int L, int H, int B,
                                     float* action_dbg, float eps, int w_is_f32, int A, float* act_ext, void* stream);

extern "C" int deer_head_final(const float* src, int src_stride, int in_dim, int pro, const float* lnw0, const float* lnb0,
                               const float* lnw1, const float* lnb1, const void* Wa, const float* ba, const void* Wg, const float* bg,
                               int* ctl, int kind, int layer, int slot, const float* thresholds, int force, int thr_type, int leq,
                               const float* h_tmp, const float* c_tmp, float* h_state, float* c_state, int L, int H, int B,
                               float* action_dbg, float eps, int w_is_f32, void* stream) {
  return deer_head_final_multi(src, src_stride, in_dim, pro, lnw0, lnb0, lnw1, lnb1, Wa, ba, Wg, bg, ctl, kind, layer, slot, thresholds, force,
                               thr_type, leq, h_tmp, c_tmp, h_state, c_state, L, H, B, action_dbg, eps, w_is_f32, 1, nullptr, stream);
}

extern "C" int deer_head_final_multi(const float* src, int src_stride, int in_dim, int pro, const float* lnw0, const float* lnb0,
                                     const float* lnw1, const float* lnb1, const void* Wa, const float* ba, const void* Wg, const float* bg,
                                     int* ctl, int kind, int layer, int slot, const float* thresholds, int force, int thr_type, int leq,
                                     const float* h_tmp, const float* c_tmp, float* h_state, float* c_state, int L, int H, int B,
                                     float* action_dbg, float eps, int w_is_f32, int A, float* act_ext, void* stream) {
  if (in_dim <= 0 || (in_dim & 7) || pro < 0 || pro > 3 || kind < 0 || kind > 2 || B <= 0 || B > DEER_MAX_ENVS) return DEER_ERR_SHAPE;
  if (kind == KIND_CHECK && (thresholds == nullptr || slot < 0 || ctl == nullptr)) return DEER_ERR_SHAPE;
  if (A < 1 || A > 8 || (A > 1 && ctl != nullptr && act_ext == nullptr)) return DEER_ERR_SHAPE;
  const int smem = (2 * in_dim + 16 + 64 + 4) * (int)sizeof(float);
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  if (w_is_f32 == 1) hipLaunchKernelGGL(head_final_kernel<float>, dim3(B), dim3(512), smem, st, src, src_stride, in_dim, pro, lnw0, lnb0, lnw1, lnb1,
                       reinterpret_cast<const float*>(Wa), ba, reinterpret_cast<const float*>(Wg), bg, ctl, kind, layer, slot,
                       thresholds, force, thr_type, leq, h_tmp, c_tmp, h_state, c_state, L, H, B, action_dbg, eps, A, act_ext);
    else if (w_is_f32 == 2) hipLaunchKernelGGL(head_final_kernel<half_t>, dim3(B), dim3(512), smem, st, src, src_stride, in_dim, pro, lnw0, lnb0, lnw1, lnb1,
                       reinterpret_cast<const half_t*>(Wa), ba, reinterpret_cast<const half_t*>(Wg), bg, ctl, kind, layer, slot,
                       thresholds, force, thr_type, leq, h_tmp, c_tmp, h_state, c_state, L, H, B, action_dbg, eps, A, act_ext);
    else hipLaunchKernelGGL(head_final_kernel<bf16_t>, dim3(B), dim3(512), smem, st, src, src_stride, in_dim, pro, lnw0, lnb0, lnw1, lnb1,
                       reinterpret_cast<const bf16_t*>(Wa), ba, reinterpret_cast<const bf16_t*>(Wg), bg, ctl, kind, layer, slot,
                       thresholds, force, thr_type, leq, h_tmp, c_tmp, h_state, c_state, L, H, B, action_dbg, eps, A, act_ext);
  DEER_LAUNCH_CHECK();
  return DEER_OK;
}

// ---- per-step control-block reset (start of every control step) ----------------------------------------
// step_info: device int32[4] written by the host before the step: {hold (1 iff cur_step % steps_per_stage != 0), step sequence
// number, host mirror pointer lo, hi (0 = no mirror)}.
__global__ void ctl_begin_step_kernel(int* ctl0, const int* hold_src, int B, int* cmap) {
  const int b = blockIdx.x;
  if (cmap != nullptr && b == 0 && threadIdx.x < 2) {      // both copies of the row map start as the identity: every environment active
    int* cm = cmap + threadIdx.x * CMAP_WORDS;
    cm[CMAP_N] = B;
    for (int e = 0; e < DEER_MAX_ENVS; ++e) {
      cm[CMAP_SLOT_ENV + e] = e;
      cm[CMAP_ENV_SLOT + e] = e < B ? e : -1;
    }
  }
  int* ctl = ctl0 + b * CTL_WORDS;
  if (threadIdx.x == 0) {
    ctl[CTL_EXIT_FLAG] = 0;
    ctl[CTL_EXIT_LAYER] = -1;
    ctl[CTL_N_EVALS] = 0;
    ctl[CTL_COMMITTED] = 0;
    ctl[CTL_PREV_REAL] = 0;                             // reset_actions() after every step of the ensembling harness (eval_utils.py:461)
    // ... which also empties action_list: a step in which no exit check runs (hold steps, steps_per_stage > 1) has NO ensemble action -
    // count 0 makes get_ensemble_action() assert like the reference's `assert len(self.action_list) > 0` (value_net.py:93) instead of
    // handing back the previous step's mean
    ctl[CTL_ENS_ACTION + 7] = 0;
    // step_info[0] = hold MASK: bit b set iff environment b is inside a stage (its own step % steps_per_stage != 0, value_net.py:285-286);
    // one environment: 0 / 1 as before
    ctl[CTL_HOLD] = (hold_src != nullptr) ? ((hold_src[0] >> b) & 1) : 0;
    if (b == 0) {
      ctl[CTL_SEQ] = (hold_src != nullptr) ? hold_src[1] : 0;
      ctl[CTL_HOST_PTR] = (hold_src != nullptr) ? hold_src[2] : 0;
      ctl[CTL_HOST_PTR + 1] = (hold_src != nullptr) ? hold_src[3] : 0;
      ctl[CTL_ALL_EXITED] = 0;
      ctl[CTL_N_EXITED] = 0;
      ctl[CTL_EVALS_DONE] = 0;
    }
  }
  if (threadIdx.x < 16) reinterpret_cast<float*>(ctl + CTL_DELTAS)[threadIdx.x] = __int_as_float(0x7fc00000);
}

extern "C" int deer_ctl_begin_step(int* ctl, const int* hold_src, int B, void* stream) {
  if (ctl == nullptr || B <= 0 || B > DEER_MAX_ENVS) return DEER_ERR_SHAPE;
  hipLaunchKernelGGL(ctl_begin_step_kernel, dim3(B), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), ctl, hold_src, B, static_cast<int*>(nullptr));
  DEER_LAUNCH_CHECK();
  return DEER_OK;
}

// the same, also resetting the two copies of the row map (2 x CMAP_WORDS int32) of an env batch with compaction to the identity
extern "C" int deer_ctl_begin_step_map(int* ctl, const int* hold_src, int B, int* cmap, void* stream) {
  if (ctl == nullptr || B <= 0 || B > DEER_MAX_ENVS || cmap == nullptr) return DEER_ERR_SHAPE;
  hipLaunchKernelGGL(ctl_begin_step_kernel, dim3(B), dim3(64), 0, reinterpret_cast<hipStream_t>(stream), ctl, hold_src, B, cmap);
  DEER_LAUNCH_CHECK();
  return DEER_OK;
}
