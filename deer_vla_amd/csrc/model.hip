// The native spine (include/deer_model.h): weight-arena layout + ingestion by reference state-dict name, workspace layout,
// and the ORDER in which the gfx950 kernels are enqueued for one DeeR-VLA control step:
//   2x ViT-L/14 (robot_flamingo/models/flamingo_mpt.py:556-583; open_clip arithmetic SURVEY App. B.2)
//   -> 2x PerceiverResampler (open_flamingo/src/helpers.py:107-132) -> media tokens (flamingo_mpt.py:661)
//   -> MPT layers with gated x-attn (open_flamingo/src/flamingo_lm.py:46-83, helpers.py:260-279)
//   -> per-exit LSTM action head + exit gate (robot_flamingo/models/action_head.py:499-611, value_net.py:120-133,277-297).
// Host code only (plus two small conversion kernels for weight ingestion); every FLOP of the step is in the other .hip files.
#include "common.h"
#include "../../include/deer_hip.h"
#include "../../include/deer_model.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <unordered_map>
#include <vector>

namespace {

constexpr float kEps = 1e-5f;
constexpr int kVitHeadLayers = 3;   // ViT blocks in the first piece of a vision chain (short graph: the GPU idles until it is submitted)
constexpr int kMaxRows = 512;       // LLM rows (n_envs * T): 16 environments x 32 tokens (data.py:905-919: max_length = 32); the trunk GEMM runs them in blocks of 128
constexpr int kMaxSplit = 32;
// LLM rows ABOVE which the trunk projections run on deer_gemm_skinny_hl (pre-split hi/lo activation planes, LDS-DMA ring, 128-column
// workgroups, GELU/slab reduction once per layer).  Default 0 = always (r03, full-depth step as one graph: 3.79 -> 3.59 ms at one
// environment, 4.84 -> 4.39 at two, 5.56 -> 4.94 at three, 8.90 -> 8.01 at eight); DEER_SKINNY_HL_MIN=128 selects deer_gemm_skinny
// (f32 activation split inside the kernel) everywhere.  The fp32 arithmetic (two weight planes) always uses deer_gemm_skinny.
int hl_min_rows() {
  static const int v = [] { const char* e = getenv("DEER_SKINNY_HL_MIN"); return e ? atoi(e) : 0; }();
  return v;
}

// ---- weight ingestion kernels -------------------------------------------------------------------------------------------
template <typename SrcT>
__device__ __forceinline__ float ld_as_f32(const SrcT* p, long i);
template <>
__device__ __forceinline__ float ld_as_f32<float>(const float* p, long i) { return p[i]; }
template <>
__device__ __forceinline__ float ld_as_f32<bf16_t>(const bf16_t* p, long i) { return bf2f(p[i]); }

// dst[r][c] (pitch dst_pitch, element type f32 (DST = 0), bf16 (1) or fp16 (2)) = src[r][c], r < rows, c < cols; columns cols..dst_cols-1 = 0
template <typename SrcT, int DST>
__global__ void ingest_rows_kernel(const SrcT* __restrict__ src, void* __restrict__ dst, long rows, int cols, int dst_cols, long dst_pitch) {
  const long total = rows * dst_cols;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long r = i / dst_cols;
    const int c = (int)(i - r * dst_cols);
    const float v = c < cols ? ld_as_f32<SrcT>(src, r * cols + c) : 0.f;
    if (DST == 1) reinterpret_cast<bf16_t*>(dst)[r * dst_pitch + c] = f2bf(v);
    else if (DST == 2) reinterpret_cast<bf16_t*>(dst)[r * dst_pitch + c] = f2h(v);
    else reinterpret_cast<float*>(dst)[r * dst_pitch + c] = v;
  }
}

// row-major W[N,K] (f32 or bf16) -> MFMA-fragment order Wp[N/16][K/32][64 lanes][8] bf16 (the layout deer_gemm_skinny streams)
// LO: the second plane of the fp32 arithmetic, bf16(w - bf16(w)) - together with the hi plane ~16 mantissa bits of every weight
// F16: fp16 fragments (deer_config.operands_f16: the whole product arithmetic on fp16 operands)
template <typename SrcT, bool LO = false, bool F16 = false>
__global__ void ingest_pack_kernel(const SrcT* __restrict__ W, bf16_t* __restrict__ Wp, int N, int K) {
  const long total = (long)(N >> 4) * (K >> 5) * 64;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int lane = (int)(idx & 63);
    const long tk = idx >> 6;
    const int kt = (int)(tk % (K >> 5));
    const int tile = (int)(tk / (K >> 5));
    const int n = tile * 16 + (lane & 15), k = kt * 32 + (lane >> 4) * 8;
    uint32_t o[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      float a = ld_as_f32<SrcT>(W, (long)n * K + k + 2 * e), b = ld_as_f32<SrcT>(W, (long)n * K + k + 2 * e + 1);
      if (LO) {
        a -= bf2f(f2bf(a));
        b -= bf2f(f2bf(b));
      }
      o[e] = pack2x<F16>(a, b);
    }
    *reinterpret_cast<uint4*>(Wp + idx * 8) = uint4{o[0], o[1], o[2], o[3]};
  }
}

enum SlotKind { SK_F32 = 0, SK_BF16 = 1, SK_PACK = 2, SK_F16 = 3 };   // SK_F16: fp16 rows (vision-tower GEMM operands, deer_config.operands_f16)

struct Slot {
  int kind;
  long rows;
  int cols;          // source row length
  int dst_cols;      // destination row length (>= cols: zero padded)
  long dst_pitch;    // destination row pitch (elements)
  size_t dst[2];     // arena offsets (second destination optional: dst[1] == SIZE_MAX when unused)
  long dst2_pitch;
  bool required;
  bool loaded;
};

struct Layout {
  size_t cur = 0;
  size_t add(size_t bytes) {
    const size_t o = cur;
    cur += (bytes + 255) & ~size_t(255);
    return o;
  }
};

struct VitLayerW { size_t ln1w, ln1b, wqkv, bqkv, wo, bo, ln2w, ln2b, wfc, bfc, wpr, bpr; };
struct PercLayerW { size_t nlw, nlb, wqkv, wo, fnw, fnb, w1, w2; };
// one PerceiverResampler weight set ("perceiver." for both cameras, or + "perceiver_gripper." when sep_resampler)
struct PercSet { size_t latents, normw, normb, nm_w, nm_b, wkv_all; std::vector<PercLayerW> L; };
struct XattnW { size_t nw, nb, wq, wo, ag, fg, fnw, fnb, w1, w2; int kv_index; };
struct LlmLayerW {
  bool has_xa;
  XattnW xa;
  size_t ln1w, ln1b, wqkv, qlnw, klnw, wo, ln2w, ln2b, wup, wdown;
  std::string ln1b_name, ln2b_name;
};
struct LstmW { size_t wih, whh, bih, bhh, lnw, lnb; };
struct FcW { size_t w[2], b[2], lnw[2], lnb[2]; };
// one DeterministicDecoder: extra_exit (the head of the exit criterion and of every action without layerwise_exit_eval) or one of
// the per-layer heads lm_exit_modules.j / lm_head (flamingo_mpt.py:236-244,450-457), each with its own LSTM state
struct HeadW {
  std::vector<LstmW> lstm;
  std::vector<FcW> fc;
  size_t wa, ba, wg, bg;
  size_t h_state = SIZE_MAX, c_state = SIZE_MAX;    // workspace offsets of this head's LSTM state (per-layer heads)
  int layer = -1;                                   // exit layer the head serves (per-layer heads)
};

// f32 activation buffers of the fp32-activation arithmetic (deer_config.precision = 1; batched single-stream schedule only)
struct PreciseWS {
  size_t im2col, xn, qkv, ao, h, mln, mkv, latln, pqkv, pao, pln, ph, kv_all;
};

// activation buffers of the vision tower for n camera frames (offsets into the workspace)
struct VisionWS {
  int n = 0, first = 0;                       // number of frames, index of the first one in the step's frame list
  size_t im2col, patch_out, vx, v_ln, v_qkv, v_ao, v_h, v_slab, p_lat, p_mln, p_mkv, p_latln, p_qkv, p_ao, p_ln, p_h;
  size_t vis_x, vis_x_f32;                    // row ranges of the step's media-token buffers
  int vit_split[2], perc_split[2];
  int cam = 0;                                // Perceiver weight set of this chain's frames (sep_resampler: 1 = gripper camera)
  size_t out_bf = SIZE_MAX, out_f32 = SIZE_MAX;   // sep_resampler: contiguous media tokens of the chain, scattered into the env-major buffers
};

int pick_split(long M, long N, long K, int target_blocks = 320, int max_split = 8) {
  // split-K factor for a projection whose 64x64 output tiles alone cannot fill 256 CUs (measured optimum 2,2 for ViT-L at 514 rows)
  const long blocks = ((M + 63) / 64) * ((N + 63) / 64);
  int S = 1;
  while (S * 2 <= max_split && blocks * S * 2 <= target_blocks && K % (S * 2 * 64) == 0 && K / (S * 2) >= 128) S *= 2;
  // env batches (>= 2048 rows): the long-K projection (ViT c_proj, K = 4096) is 128x128-tiled into only 136-264 workgroups of 64
  // K-steps each; two K halves measured -17 % at 2056 rows, -13 % at 3084, -7 % at 4112 (tools/bench_splitk.py), the short-K
  // out_proj gains nothing
  if (S == 1 && M >= 2048 && K >= 4096 && K % 128 == 0) S = 2;
  return S;
}

}  // namespace

struct ProfRec { std::string name; hipEvent_t e0, e1; double flops, bytes; };

struct deer_model {
  deer_config c;
  int wkind = 0;                    // head weight rows: 0 = bf16, 1 = f32 (precision = 1), 2 = fp16 (operands_f16) - csrc/head.hip
  bool f16 = false;                 // every 16-bit operand of the product arithmetic in IEEE fp16 (deer_config.operands_f16)
  // derived
  int P, tok, W, kpad, nl, p_inner, Lp, d, xinner, n_xattn, H, Lh, B, N, n_fc;
  bool fuse_pre = false;            // deer_config.fusion_pre: ONE Perceiver call per environment over both cameras' patch tokens (flamingo_mpt.py:585-607)
  int n_media = 0;                  // media tokens per environment seen by the gated x-attn: 2 nl (post fusion) or nl (pre fusion)
  int fc_dims[3];
  // arena
  Layout al;
  std::unordered_map<std::string, Slot> slots;
  std::vector<std::string> slot_order;
  size_t conv, cls, pos, ln_pre_w, ln_pre_b;
  std::vector<VitLayerW> vit;
  PercSet ps[2];                    // [camera]; ps[1] is only built with sep_resampler
  size_t w_arm = SIZE_MAX, b_arm = SIZE_MAX, e_grip = SIZE_MAX, w_state = SIZE_MAX, b_state = SIZE_MAX;   // use_state head weights
  size_t state_in = SIZE_MAX, state_emb = SIZE_MAX;
  size_t wte, wkv_all;
  std::vector<LlmLayerW> llm;
  std::vector<LstmW> lstm;         // extra_exit
  std::vector<FcW> fc;
  size_t wa, ba, wg, bg;
  std::vector<HeadW> lw;            // layerwise_exit_eval: per-layer heads in the reference's registration order (lm_exit_modules.0.., lm_head)
  size_t hf_trace = SIZE_MAX;
  size_t hf_xg = SIZE_MAX, hf_err = SIZE_MAX;   // one-launch head evaluation (csrc/head.hip: deer_head_fused): exchange granules, error word
  bool head_fused = false;          // one-launch head evaluation (measured: 65-71 us against 51 for the eight launches - an experiment like the
                                    // persistent trunk layer; DEER_HEAD_FUSED=1 / deer_model_set_head_fused switch it on)
  int head_wgs = 128;               // DEER_HEAD_WGS: resident workgroups of the one-launch evaluation
  size_t act_ext = SIZE_MAX;        // multi_step_action > 1: [B][4][64] f32 = previous / committed / ensemble action of 7 A values (+ A gripper logits)
  // workspace
  Layout wl;
  std::unordered_map<std::string, std::pair<size_t, size_t>> ws_named;
  VisionWS vws;
  PreciseWS hp{};
  std::unordered_map<size_t, size_t> pack_lo;   // arena offset of a packed weight -> offset of its lo plane (precision = 1)
  std::vector<VisionWS> chains;
  size_t img, vis_x, vis_x_f32, kv_all, ids, key_mask, text_time, x, xn, ao, slab_a, slab_b, qkv_ws, hidden, h_state, c_state, h_tmp,
      c_tmp, h_shadow, c_shadow, pooled, ctl, step_info, thresholds, action_dbg;
  size_t x2, cmap;                  // env batch with compaction of exited environments: second residual-stream buffer, two row maps (common.h CMAP_*)
  size_t pl_state;                  // csrc/persistent_layer.hip (DEER_PERSISTENT_LAYER=1): 192 barrier words, then the error word
  size_t ln_stats;                  // one-environment trunk (csrc/trunk_r16.hip): q/k LayerNorm moments per 32-column group
  size_t xn_hl, ao_hl, h_hl;        // bf16 hi / lo planes of the trunk activations (env batches > kHlMinRows rows: deer_gemm_skinny_hl)
  long hl_plane_d, hl_plane_h;      // elements per plane: rows * d, rows * max_n
  size_t slab_a_elems, slab_b_elems;
  size_t z_fc[3];
  char* arena = nullptr;
  char* ws = nullptr;
  // controller
  std::vector<int> exit_ids;
  int ctl_max_layer = 0, thr_type = 0, leq = 1;
  bool persistent_layer = false;    // N1 experiment: one launch per trunk layer (DEER_PERSISTENT_LAYER=1 / deer_model_set_persistent_layer)
  size_t ghh = 0;
  bool head_pre = true;             // recurrent half of the LSTM head once per control step (DEER_HEAD_PRE=0: every evaluation streams W_hh again)
  bool ghh_valid = false;           // the pre-pass result matches h_state: set by deer_begin_step, cleared by deer_model_head_state_changed
  bool compact = true;              // env batches: compaction of exited environments (DEER_COMPACT=0 / deer_model_set_compaction)
  // per-call overrides of the coarse operators
  const void* img_override = nullptr;
  const float* tokens_override = nullptr;
  const void* media_override = nullptr;
  const long long* ids_override = nullptr;
  const unsigned char* mask_override = nullptr;
  const float* thr_override = nullptr;
  // profiler
  bool prof_on = false;
  std::vector<ProfRec> prof;
  std::vector<hipEvent_t> ev_pool;
  size_t ev_next = 0;

  template <typename T> T* A(size_t off) const { return off == SIZE_MAX ? nullptr : reinterpret_cast<T*>(arena + off); }
  template <typename T> T* Wk(size_t off) const { return off == SIZE_MAX ? nullptr : reinterpret_cast<T*>(ws + off); }
  bool loaded(const std::string& n) const {
    auto it = slots.find(n);
    return it != slots.end() && it->second.loaded;
  }
};

namespace {

hipEvent_t get_event(deer_model* m) {
  if (m->ev_next == m->ev_pool.size()) {
    hipEvent_t e;
    (void)hipEventCreate(&e);
    m->ev_pool.push_back(e);
  }
  return m->ev_pool[m->ev_next++];
}

// bracket one kernel-launching call with two events when the profiler is on
struct Bracket {
  deer_model* m;
  hipStream_t st;
  bool on;
  Bracket(deer_model* m_, const char* name, double flops, double bytes, void* stream) : m(m_), st((hipStream_t)stream), on(m_->prof_on) {
    if (!on) return;
    ProfRec r{name, get_event(m), get_event(m), flops, bytes};
    (void)hipEventRecord(r.e0, st);
    m->prof.push_back(r);
  }
  ~Bracket() {
    if (on) (void)hipEventRecord(m->prof.back().e1, st);
  }
};

// a failing launch wrapper is exceptional: say which one (the C ABI itself only returns the code)
#define DEER_TRY(expr)                                                                              \
  do {                                                                                              \
    const int rc_ = (expr);                                                                         \
    if (rc_ != DEER_OK) {                                                                           \
      fprintf(stderr, "deer_model: %s -> %d (%s:%d)\n", #expr, rc_, __FILE__, __LINE__);             \
      return rc_;                                                                                   \
    }                                                                                               \
  } while (0)

// ---- arena layout -----------------------------------------------------------------------------------------------------
size_t add_slot(deer_model* m, const std::string& name, int kind, long rows, int cols, bool required = true, int dst_cols = -1,
                size_t existing = SIZE_MAX, long pitch = -1) {
  if (dst_cols < 0) dst_cols = cols;
  if (pitch < 0) pitch = dst_cols;
  const size_t esz = kind == SK_F32 ? 4 : 2;
  const size_t off = existing != SIZE_MAX ? existing : m->al.add((size_t)rows * pitch * esz);
  Slot s{kind, rows, cols, dst_cols, pitch, {off, SIZE_MAX}, 0, required, false};
  if (kind == SK_PACK && m->c.precision) {   // fp32 arithmetic: lo plane of the packed weight, streamed by a second skinny launch
    s.dst[1] = m->al.add((size_t)rows * pitch * 2);
    m->pack_lo[off] = s.dst[1];
  }
  m->slots[name] = s;
  m->slot_order.push_back(name);
  return off;
}

void build_arena(deer_model* m) {
  const deer_config& c = m->c;
  const int W = m->W;
  const std::string v = "vision_encoder.visual.";
  const int kk = 3 * c.patch_size * c.patch_size;
  // GEMM operands of the vision tower / Perceiver / x-attn K|V projection: bf16 for the product arithmetic, f32 for the fp32 one
  const int GK = c.precision ? SK_F32 : (c.operands_f16 ? SK_F16 : SK_BF16);
  const size_t we = c.precision ? 4 : 2;
  m->conv = add_slot(m, v + "conv1.weight", GK, W, kk, true, m->kpad);
  m->cls = add_slot(m, v + "class_embedding", SK_F32, 1, W);
  m->pos = add_slot(m, v + "positional_embedding", SK_F32, m->tok, W);
  m->ln_pre_w = add_slot(m, v + "ln_pre.weight", SK_F32, 1, W);
  m->ln_pre_b = add_slot(m, v + "ln_pre.bias", SK_F32, 1, W);
  m->vit.resize(c.vit_layers);
  for (int l = 0; l < c.vit_layers; ++l) {
    const std::string p = v + "transformer.resblocks." + std::to_string(l) + ".";
    VitLayerW& L = m->vit[l];
    L.ln1w = add_slot(m, p + "ln_1.weight", SK_F32, 1, W);
    L.ln1b = add_slot(m, p + "ln_1.bias", SK_F32, 1, W);
    L.wqkv = add_slot(m, p + "attn.in_proj_weight", GK, 3 * W, W);
    L.bqkv = add_slot(m, p + "attn.in_proj_bias", SK_F32, 1, 3 * W);
    L.wo = add_slot(m, p + "attn.out_proj.weight", GK, W, W);
    L.bo = add_slot(m, p + "attn.out_proj.bias", SK_F32, 1, W);
    L.ln2w = add_slot(m, p + "ln_2.weight", SK_F32, 1, W);
    L.ln2b = add_slot(m, p + "ln_2.bias", SK_F32, 1, W);
    L.wfc = add_slot(m, p + "mlp.c_fc.weight", GK, c.vit_mlp, W);
    L.bfc = add_slot(m, p + "mlp.c_fc.bias", SK_F32, 1, c.vit_mlp);
    L.wpr = add_slot(m, p + "mlp.c_proj.weight", GK, W, c.vit_mlp);
    L.bpr = add_slot(m, p + "mlp.c_proj.bias", SK_F32, 1, W);
  }
  // ---- Perceiver: the media tokens are the same in every layer, so norm_media / to_kv of ALL layers are stacked (applied up
  // front by one LayerNorm pass + one batched GEMM); the latents are projected to q | k | v by [to_q ; to_kv] in one GEMM
  const int inner = m->p_inner, Lp = m->Lp;
  for (int cam = 0; cam < (c.sep_resampler ? 2 : 1); ++cam) {
    const std::string pp = cam == 0 ? "perceiver." : "perceiver_gripper.";      // flamingo_mpt.py:132-134
    PercSet& S = m->ps[cam];
    S.latents = add_slot(m, pp + "latents", SK_F32, m->nl, W);
    S.normw = add_slot(m, pp + "norm.weight", SK_F32, 1, W);
    S.normb = add_slot(m, pp + "norm.bias", SK_F32, 1, W);
    S.nm_w = m->al.add((size_t)Lp * W * 4);
    S.nm_b = m->al.add((size_t)Lp * W * 4);
    S.wkv_all = m->al.add((size_t)Lp * 2 * inner * W * we);
    S.L.resize(Lp);
    for (int l = 0; l < Lp; ++l) {
      const std::string a = pp + "layers." + std::to_string(l) + ".0.", f = pp + "layers." + std::to_string(l) + ".1.";
      PercLayerW& L = S.L[l];
      add_slot(m, a + "norm_media.weight", SK_F32, 1, W, true, -1, S.nm_w + (size_t)l * W * 4);
      add_slot(m, a + "norm_media.bias", SK_F32, 1, W, true, -1, S.nm_b + (size_t)l * W * 4);
      L.nlw = add_slot(m, a + "norm_latents.weight", SK_F32, 1, W);
      L.nlb = add_slot(m, a + "norm_latents.bias", SK_F32, 1, W);
      L.wqkv = m->al.add((size_t)3 * inner * W * we);
      add_slot(m, a + "to_q.weight", GK, inner, W, true, -1, L.wqkv);
      add_slot(m, a + "to_kv.weight", GK, 2 * inner, W, true, -1, L.wqkv + (size_t)inner * W * we);
      m->slots[a + "to_kv.weight"].dst[1] = S.wkv_all + (size_t)l * 2 * inner * W * we;
      m->slots[a + "to_kv.weight"].dst2_pitch = W;
      L.wo = add_slot(m, a + "to_out.weight", GK, W, inner);
      L.fnw = add_slot(m, f + "0.weight", SK_F32, 1, W);
      L.fnb = add_slot(m, f + "0.bias", SK_F32, 1, W);
      L.w1 = add_slot(m, f + "1.weight", GK, (long)c.perc_ff_mult * W, W);
      L.w2 = add_slot(m, f + "3.weight", GK, W, c.perc_ff_mult * W);
    }
  }
  // ---- LLM: projections pre-packed in MFMA-fragment order; to_kv of all x-attn layers concatenated (media is layer-invariant)
  const int d = m->d, xin = m->xinner;
  m->wte = add_slot(m, "lang_encoder.transformer.wte.weight", c.precision ? SK_F32 : (c.operands_f16 ? SK_F16 : SK_BF16), c.vocab_size, d);
  m->n_xattn = 0;
  for (int n = 0; n < c.n_layers; ++n)
    if ((n + 1) % c.cross_attn_every_n_layers == 0) ++m->n_xattn;
  const int GKx = c.precision ? SK_F32 : (c.operands_f16 ? SK_F16 : SK_BF16);   // x-attn to_kv: the media K/V GEMM belongs to the tower's arithmetic
  const size_t wex = c.precision ? 4 : 2;
  m->wkv_all = m->n_xattn ? m->al.add((size_t)m->n_xattn * 2 * xin * W * wex) : SIZE_MAX;
  m->llm.resize(c.n_layers);
  int kv_index = 0;
  const char* ln1 = c.mpt7b_names ? "norm_1" : "ln_1";
  const char* ln2 = c.mpt7b_names ? "norm_2" : "ln_2";
  const char* up = c.mpt7b_names ? "ffn.up_proj" : "mlp.mlp_up";
  const char* down = c.mpt7b_names ? "ffn.down_proj" : "mlp.mlp_down";
  for (int n = 0; n < c.n_layers; ++n) {
    const std::string blk = "lang_encoder.transformer.blocks." + std::to_string(n) + ".";
    LlmLayerW& L = m->llm[n];
    L.has_xa = (n + 1) % c.cross_attn_every_n_layers == 0;
    if (L.has_xa) {
      const std::string x = blk + "gated_cross_attn_layer.";
      XattnW& X = L.xa;
      X.nw = add_slot(m, x + "attn.norm.weight", SK_F32, 1, d);
      X.nb = add_slot(m, x + "attn.norm.bias", SK_F32, 1, d);
      X.wq = add_slot(m, x + "attn.to_q.weight", SK_PACK, xin, d);
      add_slot(m, x + "attn.to_kv.weight", GKx, 2 * xin, W, true, -1, m->wkv_all + (size_t)kv_index * 2 * xin * W * wex);
      X.kv_index = kv_index++;
      X.wo = add_slot(m, x + "attn.to_out.weight", SK_PACK, d, xin);
      X.ag = add_slot(m, x + "attn_gate", SK_F32, 1, 1);
      X.fg = add_slot(m, x + "ff_gate", SK_F32, 1, 1);
      X.fnw = add_slot(m, x + "ff.0.weight", SK_F32, 1, d);
      X.fnb = add_slot(m, x + "ff.0.bias", SK_F32, 1, d);
      X.w1 = add_slot(m, x + "ff.1.weight", SK_PACK, (long)c.xattn_ff_mult * d, d);
      X.w2 = add_slot(m, x + "ff.3.weight", SK_PACK, d, c.xattn_ff_mult * d);
    }
    const std::string mm = blk + "decoder_layer.";
    L.ln1w = add_slot(m, mm + ln1 + ".weight", SK_F32, 1, d);
    L.ln1b_name = mm + ln1 + ".bias";
    L.ln1b = add_slot(m, L.ln1b_name, SK_F32, 1, d, false);          // no_bias models strip it (mosaic_gpt_3b.py:147-153)
    L.wqkv = add_slot(m, mm + "attn.Wqkv.weight", SK_PACK, 3L * d, d);
    L.qlnw = L.klnw = SIZE_MAX;
    if (c.attn_qk_ln) {
      L.qlnw = add_slot(m, mm + "attn.q_ln.weight", SK_F32, 1, d);
      L.klnw = add_slot(m, mm + "attn.k_ln.weight", SK_F32, 1, d);
    }
    L.wo = add_slot(m, mm + "attn.out_proj.weight", SK_PACK, d, d);
    L.ln2w = add_slot(m, mm + ln2 + ".weight", SK_F32, 1, d);
    L.ln2b_name = mm + ln2 + ".bias";
    L.ln2b = add_slot(m, L.ln2b_name, SK_F32, 1, d, false);
    L.wup = add_slot(m, mm + up + ".weight", SK_PACK, (long)c.mlp_ratio * d, d);
    L.wdown = add_slot(m, mm + down + ".weight", SK_PACK, d, c.mlp_ratio * d);
  }
  // ---- action head (DeterministicDecoder, action_head.py:408-497): LN-LSTM = [LSTM, LN, Dropout] x L under rnn.layers.{3l, 3l+1}
  const std::string p = "extra_exit.";
  const int HK = c.precision ? SK_F32 : (c.operands_f16 ? SK_F16 : SK_BF16);      // head weights: bf16 / fp16, or f32 for the fp32 arithmetic
  const int H = m->H;
  if (c.use_state) {                                   // action_head.py:443-453
    m->w_arm = add_slot(m, p + "embed_arm_state.0.weight", SK_F32, d, 6);
    m->b_arm = add_slot(m, p + "embed_arm_state.0.bias", SK_F32, 1, d);
    m->e_grip = add_slot(m, p + "embed_gripper_state.0.weight", SK_F32, 2, d);
    m->w_state = add_slot(m, p + "embed_state.weight", HK, d, 2 * d);
    m->b_state = add_slot(m, p + "embed_state.bias", SK_F32, 1, d);
  }
  auto add_head = [&](const std::string& p, std::vector<LstmW>& lstm, std::vector<FcW>& fc, size_t& wa, size_t& ba, size_t& wg, size_t& bg) {
  lstm.resize(m->Lh);
  int in_f = d;
  for (int l = 0; l < m->Lh; ++l) {
    std::string r, sfx;
    if (c.lstm_layernorm) { r = p + "rnn.layers." + std::to_string(3 * l) + "."; sfx = "_l0"; }
    else { r = p + "rnn."; sfx = "_l" + std::to_string(l); }
    LstmW& Lw = lstm[l];
    Lw.wih = add_slot(m, r + "weight_ih" + sfx, HK, 4L * H, in_f);
    Lw.whh = add_slot(m, r + "weight_hh" + sfx, HK, 4L * H, H);
    Lw.bih = add_slot(m, r + "bias_ih" + sfx, SK_F32, 1, 4 * H);
    Lw.bhh = add_slot(m, r + "bias_hh" + sfx, SK_F32, 1, 4 * H);
    Lw.lnw = Lw.lnb = SIZE_MAX;
    if (c.lstm_layernorm) {
      Lw.lnw = add_slot(m, p + "rnn.layers." + std::to_string(3 * l + 1) + ".weight", SK_F32, 1, H);
      Lw.lnb = add_slot(m, p + "rnn.layers." + std::to_string(3 * l + 1) + ".bias", SK_F32, 1, H);
    }
    in_f = H;
  }
  // MLP heads, dropout_mode='layerwise' (action_head.py:86-116): [Drop, (Lin, LN|Id, ReLU, Drop) x n_hidden, Lin, Tanh|Sigmoid]
  fc.resize(m->n_fc);
  const char* heads[2] = {"actions", "gripper"};
  for (int g = 0; g < 2; ++g) {
    int cur = H;
    for (int i = 0; i < m->n_fc; ++i) {
      const std::string li = p + heads[g] + ".mlp." + std::to_string(1 + 4 * i) + ".";
      const std::string ni = p + heads[g] + ".mlp." + std::to_string(2 + 4 * i) + ".";
      fc[i].w[g] = add_slot(m, li + "weight", HK, m->fc_dims[i], cur);
      fc[i].b[g] = add_slot(m, li + "bias", SK_F32, 1, m->fc_dims[i]);
      fc[i].lnw[g] = fc[i].lnb[g] = SIZE_MAX;
      if (c.mlp_layernorm) {
        fc[i].lnw[g] = add_slot(m, ni + "weight", SK_F32, 1, m->fc_dims[i]);
        fc[i].lnb[g] = add_slot(m, ni + "bias", SK_F32, 1, m->fc_dims[i]);
      }
      cur = m->fc_dims[i];
    }
    const std::string lo = p + heads[g] + ".mlp." + std::to_string(1 + 4 * m->n_fc) + ".";
    const int A = std::max(1, c.multi_step_action);    // action_head.py:472-473: out_features * multi_step_action outputs per head
    const int n_out = g == 0 ? 6 * A : A;
    (g == 0 ? wa : wg) = add_slot(m, lo + "weight", HK, n_out, cur);
    (g == 0 ? ba : bg) = add_slot(m, lo + "bias", SK_F32, 1, n_out);
  }
  };
  add_head(p, m->lstm, m->fc, m->wa, m->ba, m->wg, m->bg);
  if (c.layerwise_exit_eval) {   // one head per internal exit + lm_head for the last layer (flamingo_mpt.py:236-244 with multi_exit = True)
    int j = 0;
    for (int i = c.exit_interval - 1; i < c.n_layers - 1; i += c.exit_interval, ++j) {
      m->lw.emplace_back();
      HeadW& h = m->lw.back();
      h.layer = i;
      add_head("lm_exit_modules." + std::to_string(j) + ".", h.lstm, h.fc, h.wa, h.ba, h.wg, h.bg);
    }
    m->lw.emplace_back();
    HeadW& h = m->lw.back();
    h.layer = c.n_layers - 1;
    add_head("lm_head.", h.lstm, h.fc, h.wa, h.ba, h.wg, h.bg);
  }
}

// ---- workspace layout -------------------------------------------------------------------------------------------------
void build_vision_ws(deer_model* m, VisionWS& ws, int n, int first, const VisionWS* splits) {
  const deer_config& c = m->c;
  const int P = m->P, W = m->W, nl = m->nl, inner = m->p_inner, Lp = m->Lp;
  const long R = (long)n * (P + 1);
  ws.n = n;
  ws.first = first;
  ws.im2col = m->wl.add((size_t)n * P * m->kpad * 2);
  ws.patch_out = m->wl.add((size_t)n * P * W * 4);
  ws.vx = m->wl.add((size_t)R * W * 4);
  ws.v_ln = m->wl.add((size_t)R * W * 2);
  ws.v_qkv = m->wl.add((size_t)R * 3 * W * 2);
  ws.v_ao = m->wl.add((size_t)R * W * 2);
  ws.v_h = m->wl.add((size_t)R * c.vit_mlp * 2);
  if (splits == nullptr) {
    ws.vit_split[0] = pick_split(R, W, W);
    ws.vit_split[1] = pick_split(R, W, c.vit_mlp);
    ws.perc_split[0] = pick_split((long)n * nl, W, inner);
    ws.perc_split[1] = pick_split((long)n * nl, W, (long)c.perc_ff_mult * W);
    if (const char* ov = getenv("DEER_VIT_SPLIT")) {
      int v[4];
      if (sscanf(ov, "%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3]) == 4) {
        ws.vit_split[0] = v[0]; ws.vit_split[1] = v[1]; ws.perc_split[0] = v[2]; ws.perc_split[1] = v[3];
      }
    }
  } else {   // same summation order as the batched schedule: the results of all schedules are bit-identical
    memcpy(ws.vit_split, splits->vit_split, sizeof(ws.vit_split));
    memcpy(ws.perc_split, splits->perc_split, sizeof(ws.perc_split));
  }
  const long slab = std::max((long)std::max(ws.vit_split[0], ws.vit_split[1]) * R, (long)std::max(ws.perc_split[0], ws.perc_split[1]) * n * nl);
  ws.v_slab = m->wl.add((size_t)slab * W * 4);
  ws.p_lat = m->wl.add((size_t)n * nl * W * 4);
  ws.p_mln = m->wl.add((size_t)Lp * n * P * W * 2);
  ws.p_mkv = m->wl.add((size_t)Lp * n * P * 2 * inner * 2);
  ws.p_latln = m->wl.add((size_t)n * nl * W * 2);
  ws.p_qkv = m->wl.add((size_t)n * nl * 3 * inner * 2);
  ws.p_ao = m->wl.add((size_t)n * nl * inner * 2);
  ws.p_ln = m->wl.add((size_t)n * nl * W * 2);
  ws.p_h = m->wl.add((size_t)n * nl * c.perc_ff_mult * W * 2);
  ws.vis_x = m->vis_x + (size_t)first * nl * W * 2;
  ws.vis_x_f32 = m->vis_x_f32 + (size_t)first * nl * W * 4;
}

size_t named(deer_model* m, const char* name, size_t bytes) {
  const size_t off = m->wl.add(bytes);
  m->ws_named[name] = {off, bytes};
  return off;
}

void build_workspace(deer_model* m) {
  const deer_config& c = m->c;
  const int N = m->N, W = m->W, nl = m->nl, d = m->d, B = m->B, S = c.image_size;
  m->img = named(m, "img", (size_t)N * 3 * S * S * (c.precision ? 4 : 2));        // camera frames: bf16, or f32 for the fp32 arithmetic
  m->vis_x = named(m, "vis_x", (size_t)N * nl * W * 2);
  m->vis_x_f32 = named(m, "vis_x_f32", (size_t)N * nl * W * 4);
  build_vision_ws(m, m->vws, N, 0, nullptr);
  m->ws_named["vx"] = {m->vws.vx, (size_t)N * (m->P + 1) * W * 4};
  // two chains of camera frames on two streams hide each other's launch boundaries while a kernel cannot fill the chip by itself
  // (1-7 environments); from 16 frames on every GEMM of ONE chain is a full round of frame tiles (gemm_bigm.hip: 16 x 16 = 256
  // workgroups) - measured at 8 environments: batched 945 -> 950, four batches in flight 1102 -> 1172 steps/s
  int n_ch = c.n_chains > 0 ? c.n_chains : (N >= 16 ? 1 : 2);
  if (const char* e = getenv("DEER_CHAINS")) n_ch = atoi(e);
  n_ch = std::max(1, std::min(N, n_ch));
  if (c.sep_resampler) n_ch = 2;                       // camera-major frames: chain 0 = every env's rgb frame, chain 1 = the gripper frames
  if (c.fusion_pre) n_ch = 1;                          // the Perceiver joins the two cameras of an environment: one chain holds both
  const int per = (N + n_ch - 1) / n_ch;
  for (int ch = 0; ch < n_ch; ++ch) {
    const int lo = ch * per, hi = std::min(N, (ch + 1) * per);
    if (hi <= lo) break;
    m->chains.emplace_back();
    VisionWS& cw = m->chains.back();
    build_vision_ws(m, cw, hi - lo, lo, &m->vws);
    if (c.sep_resampler) {
      cw.cam = ch;
      cw.out_bf = m->wl.add((size_t)(hi - lo) * nl * W * 2);
      cw.out_f32 = m->wl.add((size_t)(hi - lo) * nl * W * 4);
    }
  }
  m->kv_all = named(m, "kv_all", (size_t)N * nl * std::max(m->n_xattn, 1) * 2 * m->xinner * 2);
  if (c.precision) {   // f32 twins of the vision tower's activations and of the x-attn K/V (csrc/precise.hip)
    const long R = (long)N * (m->P + 1), inner = m->p_inner, NL = (long)N * nl, NP = (long)N * m->P;
    m->hp.im2col = m->wl.add((size_t)NP * m->kpad * 4);
    m->hp.xn = m->wl.add((size_t)R * W * 4);
    m->hp.qkv = m->wl.add((size_t)R * 3 * W * 4);
    m->hp.ao = m->wl.add((size_t)R * W * 4);
    m->hp.h = m->wl.add((size_t)R * c.vit_mlp * 4);
    m->hp.mln = m->wl.add((size_t)NP * W * 4);
    m->hp.mkv = m->wl.add((size_t)NP * 2 * inner * 4);
    m->hp.latln = m->wl.add((size_t)NL * W * 4);
    m->hp.pqkv = m->wl.add((size_t)NL * 3 * inner * 4);
    m->hp.pao = m->wl.add((size_t)NL * inner * 4);
    m->hp.pln = m->wl.add((size_t)NL * W * 4);
    m->hp.ph = m->wl.add((size_t)NL * c.perc_ff_mult * W * 4);
    m->hp.kv_all = named(m, "kv_all_f32", (size_t)NL * std::max(m->n_xattn, 1) * 2 * m->xinner * 4);
  }
  const int T = std::min(B * c.max_text_len, kMaxRows);
  m->ids = named(m, "ids", (size_t)T * 8);
  m->key_mask = named(m, "key_mask", (size_t)T);
  m->text_time = named(m, "text_time", (size_t)T * 4);
  m->x = named(m, "x", (size_t)T * d * 4);
  m->xn = named(m, "xn", (size_t)T * d * 4);
  m->ao = named(m, "ao", (size_t)T * std::max(d, m->xinner) * 4);
  const long max_n = std::max(std::max((long)c.mlp_ratio * d, (long)c.xattn_ff_mult * d), 3L * d);
  const int mpad = 16 * ((T + 15) / 16);
  const int planes = c.precision ? 2 : 1;
  m->slab_a_elems = (size_t)planes * kMaxSplit * mpad * d;
  m->slab_b_elems = (size_t)planes * 16 * mpad * max_n;
  m->slab_a = named(m, "slab_a", m->slab_a_elems * 4);
  m->slab_b = named(m, "slab_b", m->slab_b_elems * 4);
  m->qkv_ws = named(m, "qkv_ws", (size_t)T * 3 * d * 4);
  m->hl_plane_d = (long)T * std::max(d, m->xinner);
  m->hl_plane_h = (long)T * max_n;
  m->xn_hl = m->wl.add((size_t)2 * m->hl_plane_d * 2);
  m->ao_hl = m->wl.add((size_t)2 * m->hl_plane_d * 2);
  m->h_hl = m->wl.add((size_t)2 * m->hl_plane_h * 2);
  m->x2 = m->wl.add((size_t)T * d * 4);
  m->cmap = named(m, "cmap", (size_t)2 * CMAP_WORDS * 4);
  m->pl_state = named(m, "pl_state", (size_t)(16 * 12 + 16 + 64 * 64) * 4);     // persistent-layer experiment: barrier words + error word
  m->ln_stats = m->wl.add((size_t)(3 * d / 32 + 1) * 16 * 2 * 4);
  m->hidden = named(m, "hidden", (size_t)c.n_layers * T * d * 4);
  const size_t st = (size_t)m->Lh * B * m->H * 4;
  m->h_state = named(m, "h_state", st);
  m->c_state = named(m, "c_state", st);
  m->h_tmp = named(m, "h_tmp", st);
  m->c_tmp = named(m, "c_tmp", st);
  m->h_shadow = named(m, "h_shadow", st);
  m->ghh = named(m, "lstm_ghh", 4 * st);                              // W_hh h_state + b_hh of every LSTM layer, once per control step (deer_begin_step)
  m->c_shadow = named(m, "c_shadow", st);
  for (int i = 0; i < m->n_fc; ++i) m->z_fc[i] = m->wl.add((size_t)B * 2 * m->fc_dims[i] * 4);
  m->pooled = named(m, "pooled", (size_t)B * d * 4);
  if (c.use_state) {
    m->state_in = named(m, "state_in", (size_t)B * 8 * 4);
    m->state_emb = named(m, "state_emb", (size_t)B * d * 4);
  }
  m->ctl = named(m, "ctl", (size_t)B * CTL_WORDS * 4);
  m->step_info = named(m, "step_info", 16);
  m->thresholds = named(m, "thresholds", 16 * 4);
  // every head evaluation's outputs: [B][64] f32 = pose (6 A) | gripper prob (A) | gripper logit (A), A = multi_step_action (A = 1: the
  // familiar [pose6, prob, logit])
  m->action_dbg = named(m, "action_dbg", (size_t)B * 64 * 4);
  m->act_ext = named(m, "act_ext", (size_t)B * 4 * 64 * 4);
  m->hf_xg = named(m, "head_fused_xg", (size_t)deer_head_fused_granules(B, d, m->H, m->Lh, m->n_fc, m->fc_dims) * 8);
  m->hf_err = named(m, "head_fused_err", 256);
  m->hf_trace = named(m, "head_fused_trace", 64 * 8);
  if (!m->lw.empty()) {                                                // per-layer heads: own LSTM state, ONE dense block
    // host view: [head][2 (h, c)][Lh][B][H], heads in registration order (dense whatever st is: st is a multiple of 32 bytes, H % 8 == 0)
    const size_t base = m->wl.add(2 * st * m->lw.size());
    size_t k = 0;
    for (HeadW& h : m->lw) {
      h.h_state = base + (2 * k) * st;
      h.c_state = base + (2 * k + 1) * st;
      ++k;
    }
    m->ws_named["lw_state"] = {base, 2 * st * m->lw.size()};
  }
}

// ---- launch helpers ---------------------------------------------------------------------------------------------------
int gemm(deer_model* m, const void* A, const void* Wt, void* C, long M, long N, long K, int epi, const float* bias, void* st,
         int lda = -1, int ldc = -1) {
  if (lda < 0) lda = (int)K;
  if (ldc < 0) ldc = (int)N;
  const double out_b = (epi == DEER_EPI_F32 || epi == DEER_EPI_RESADD_F32) ? 4.0 : 2.0;
  // the profiler's class name stays "deer_gemm_bf16_nt" for both 16-bit formats (one kernel family; bench.py's roofline keys on it)
  Bracket b(m, "deer_gemm_bf16_nt", 2.0 * M * N * K, 2.0 * (M * K + N * K) + out_b * M * N, st);
  if (m->c.operands_f16) return deer_gemm_f16_nt(A, lda, 0, Wt, (int)K, bias, C, ldc, 0, (int)M, (int)N, (int)K, 1, epi, nullptr, 0, nullptr, st);
  return deer_gemm_bf16_nt(A, lda, 0, Wt, (int)K, bias, C, ldc, 0, (int)M, (int)N, (int)K, 1, epi, nullptr, 0, nullptr, st);
}

int gemm_splitk(deer_model* m, const void* A, const void* Wt, float* slab, long M, long N, long K, int S, void* st) {
  Bracket b(m, "deer_gemm_bf16_nt", 2.0 * M * N * K, 2.0 * (M * K + N * K) + 4.0 * S * M * N, st);
  if (m->c.operands_f16) return deer_gemm_f16_nt_splitk(A, (int)K, Wt, (int)K, slab, (int)M, (int)N, (int)K, S, 0, nullptr, st);
  return deer_gemm_bf16_nt_splitk(A, (int)K, Wt, (int)K, slab, (int)M, (int)N, (int)K, S, 0, nullptr, st);
}

// x += sum_s slab[s] + bias, then (optionally) LayerNorm -> 16-bit / f32: closes a split-K projection of the vision tower
int vresadd(deer_model* m, float* x, const float* slab, int S, long rows, int C, const float* bias, const float* gamma, const float* beta,
            void* out_bf, float* out_f32, void* st) {
  Bracket b(m, "deer_resadd_ln", 0, 4.0 * rows * C * (S + 2) + (out_bf ? 2.0 : 0.0) * rows * C, st);
  if (m->c.operands_f16)
    return deer_resadd_ln_f16(x, slab, S, rows * C, nullptr, bias, gamma, beta, gamma ? out_bf : nullptr, gamma ? out_f32 : nullptr, nullptr,
                              (int)rows, C, kEps, nullptr, st);
  return deer_resadd_ln(x, slab, S, rows * C, nullptr, bias, gamma, beta, gamma ? out_bf : nullptr, gamma ? out_f32 : nullptr, nullptr,
                        (int)rows, C, kEps, nullptr, st);
}

int ln_rows(deer_model* m, const float* x, const float* gamma, const float* beta, void* out_bf, long rows, int C, void* st) {
  Bracket b(m, "deer_layernorm_rows", 8.0 * rows * C, 6.0 * rows * C, st);
  if (m->c.operands_f16) return deer_layernorm_rows_f16(x, C, 0, (int)rows, 1, gamma, beta, out_bf, nullptr, C, 0, C, kEps, st);
  return deer_layernorm_rows(x, C, 0, (int)rows, 1, gamma, beta, out_bf, nullptr, C, 0, C, kEps, st);
}

const void* step_images(const deer_model* m) { return m->img_override ? m->img_override : m->Wk<void>(m->img); }

int gemmf(deer_model* m, const float* A, long lda, const float* Wt, const float* bias, float* C, long ldc, long M, long N, long K, int epi, void* st);

int patch_embed(deer_model* m, const VisionWS& ws, void* st) {
  const deer_config& c = m->c;
  const int P = m->P, W = m->W;
  const long R = (long)ws.n * (P + 1);
  const char* img = reinterpret_cast<const char*>(step_images(m)) + (size_t)ws.first * 3 * c.image_size * c.image_size * (c.precision ? 4 : 2);
  if (c.precision) {   // f32 frames -> f32 patches -> exact-f32 MFMA conv (csrc/precise.hip)
    {
      Bracket b(m, "deer_vit_im2col", 0, 0, st);
      DEER_TRY(deer_vit_im2col_f32(reinterpret_cast<const float*>(img), ws.n, c.image_size, c.patch_size, m->Wk<float>(m->hp.im2col), m->kpad, st));
    }
    DEER_TRY(gemmf(m, m->Wk<float>(m->hp.im2col), m->kpad, m->A<float>(m->conv), nullptr, m->Wk<float>(ws.patch_out), W, (long)ws.n * P, W, m->kpad, 0, st));
  } else {
    {
      Bracket b(m, "deer_vit_im2col", 0, 0, st);
      if (c.operands_f16) DEER_TRY(deer_vit_im2col_f16(img, 2, ws.n, c.image_size, c.patch_size, m->Wk<void>(ws.im2col), m->kpad, st));   // fp16 frames
      else DEER_TRY(deer_vit_im2col(img, 1, ws.n, c.image_size, c.patch_size, m->Wk<void>(ws.im2col), m->kpad, st));
    }
    DEER_TRY(gemm(m, m->Wk<void>(ws.im2col), m->A<void>(m->conv), m->Wk<void>(ws.patch_out), (long)ws.n * P, W, m->kpad, DEER_EPI_F32, nullptr, st));
  }
  {
    Bracket b(m, "deer_vit_embed_lnpre", 0, 0, st);
    DEER_TRY(deer_vit_embed_lnpre(m->Wk<float>(ws.patch_out), m->A<float>(m->cls), m->A<float>(m->pos), m->A<float>(m->ln_pre_w),
                                  m->A<float>(m->ln_pre_b), m->Wk<float>(ws.vx), ws.n, P, W, kEps, st));
  }
  // out_proj / c_proj run split-K into f32 slabs; the slab reduction, bias, residual add and the NEXT LayerNorm are one launch
  // (deer_resadd_ln), so a block is 7 launches and no projection leaves CUs idle
  return ln_rows(m, m->Wk<float>(ws.vx), m->A<float>(m->vit[0].ln1w), m->A<float>(m->vit[0].ln1b), m->Wk<void>(ws.v_ln), R, W, st);
}

// sep_resampler: the chain's media tokens land in a contiguous staging buffer and are scattered into the env-major media buffers
// (env b: [rgb latents ; gripper latents], flamingo_mpt.py:661)
int scatter_media(deer_model* m, const VisionWS& ws, void* st) {
  if (ws.out_bf == SIZE_MAX) return DEER_OK;
  const size_t row_bf = (size_t)m->nl * m->W * 2, row_f = (size_t)m->nl * m->W * 4;
  const int b0 = ws.first - ws.cam * m->B;              // first environment of this chain (camera-major frame order)
  hipStream_t s = (hipStream_t)st;
  if (hipMemcpy2DAsync(m->Wk<char>(m->vis_x) + ((size_t)b0 * 2 + ws.cam) * row_bf, 2 * row_bf, m->Wk<char>(ws.out_bf), row_bf, row_bf, ws.n,
                       hipMemcpyDeviceToDevice, s) != hipSuccess) return DEER_ERR_LAUNCH;
  if (hipMemcpy2DAsync(m->Wk<char>(m->vis_x_f32) + ((size_t)b0 * 2 + ws.cam) * row_f, 2 * row_f, m->Wk<char>(ws.out_f32), row_f, row_f, ws.n,
                       hipMemcpyDeviceToDevice, s) != hipSuccess) return DEER_ERR_LAUNCH;
  return DEER_OK;
}

int perceiver(deer_model* m, const VisionWS& ws, void* st) {
  const deer_config& c = m->c;
  const PercSet& PS = m->ps[ws.cam];
  void* out_bf = ws.out_bf != SIZE_MAX ? m->Wk<void>(ws.out_bf) : m->Wk<void>(ws.vis_x);
  float* out_f32 = ws.out_f32 != SIZE_MAX ? m->Wk<float>(ws.out_f32) : m->Wk<float>(ws.vis_x_f32);
  const int P = m->P, W = m->W, tok = m->tok, nl = m->nl, inner = m->p_inner, Lp = m->Lp;
  // pre fusion (flamingo_mpt.py:598-602): the patch tokens of an environment's two frames are ONE media sequence of 2 P tokens for ONE set
  // of latents.  The media side below works on frames (its LayerNorm / K|V rows are written contiguously: [frame][P] = [env][2 P]); the
  // latent side on `N` latent sets attending `PM` media tokens each.
  const int NF = ws.n, N = m->fuse_pre ? ws.n / 2 : ws.n, PM = m->fuse_pre ? 2 * P : P;
  // Media side once for all layers: one LayerNorm pass with every layer's norm_media affine, one batched GEMM with every layer's
  // to_kv.  Per layer only the 64 latents move: q|k|v projection, attention over [media K/V ; latent K/V] (two segments,
  // helpers.py:51 without the concat), to_out and the FF - each residual projection split-K, closed by the reducer that also
  // applies the NEXT LayerNorm.
  {
    Bracket b(m, "deer_broadcast_rows", 0, 0, st);
    DEER_TRY(deer_broadcast_rows(m->A<float>(PS.latents), m->Wk<float>(ws.p_lat), (long)nl * W, N, st));
  }
  const float* tokens = m->tokens_override ? m->tokens_override + (size_t)ws.first * P * W : m->Wk<float>(ws.vx) + W;   // skip the cls row
  const long tok_bstride = m->tokens_override ? (long)P * W : (long)tok * W;
  {
    Bracket b(m, "deer_layernorm_rows", 8.0 * NF * P * W, (4.0 + 2.0 * Lp) * NF * P * W, st);
    DEER_TRY((c.operands_f16 ? deer_layernorm_rows_multi_f16 : deer_layernorm_rows_multi)(tokens, W, tok_bstride, P, NF, m->A<float>(PS.nm_w), m->A<float>(PS.nm_b), Lp, W,
                                       m->Wk<void>(ws.p_mln), (long)NF * P * W, W, (long)P * W, W, kEps, st));
  }
  {
    Bracket b(m, "deer_gemm_bf16_nt", 2.0 * Lp * NF * P * 2 * inner * W, 2.0 * Lp * ((double)NF * P * W + 2.0 * inner * W + (double)NF * P * 2 * inner), st);
    DEER_TRY((c.operands_f16 ? deer_gemm_f16_nt_wbatch : deer_gemm_bf16_nt_wbatch)(m->Wk<void>(ws.p_mln), W, (long)NF * P * W, m->A<void>(PS.wkv_all), W, 2L * inner * W, nullptr,
                                      m->Wk<void>(ws.p_mkv), 2 * inner, (long)NF * P * 2 * inner, NF * P, 2 * inner, W, Lp, DEER_EPI_BF16, 0,
                                      nullptr, st));
  }
  DEER_TRY(ln_rows(m, m->Wk<float>(ws.p_lat), m->A<float>(PS.L[0].nlw), m->A<float>(PS.L[0].nlb), m->Wk<void>(ws.p_latln), (long)N * nl, W, st));
  const int Pa = ws.perc_split[0], Pf = ws.perc_split[1];
  char* qkv = m->Wk<char>(ws.p_qkv);
  for (int li = 0; li < Lp; ++li) {
    const PercLayerW& L = PS.L[li];
    DEER_TRY(gemm(m, m->Wk<void>(ws.p_latln), m->A<void>(L.wqkv), qkv, (long)N * nl, 3 * inner, W, DEER_EPI_BF16, nullptr, st));
    const char* mkv = m->Wk<char>(ws.p_mkv) + (size_t)li * NF * P * 2 * inner * 2;
    {
      Bracket b(m, "deer_attn_mfma_hd64", 4.0 * N * c.perc_heads * nl * (PM + nl) * 64, 0, st);
      DEER_TRY((c.operands_f16 ? deer_attn_f16_hd64_2seg : deer_attn_mfma_hd64_2seg)(qkv, mkv, mkv + (size_t)inner * 2, qkv + (size_t)inner * 2, qkv + (size_t)2 * inner * 2, m->Wk<void>(ws.p_ao), N,
                                        c.perc_heads, nl, PM, nl, 3 * inner, 2 * inner, 3 * inner, inner, (long)nl * 3 * inner, (long)PM * 2 * inner,
                                        (long)nl * 3 * inner, (long)nl * inner, 1.0f / sqrtf((float)c.perc_dim_head), st));
    }
    DEER_TRY(gemm_splitk(m, m->Wk<void>(ws.p_ao), m->A<void>(L.wo), m->Wk<float>(ws.v_slab), (long)N * nl, W, inner, Pa, st));
    DEER_TRY(vresadd(m, m->Wk<float>(ws.p_lat), m->Wk<float>(ws.v_slab), Pa, (long)N * nl, W, nullptr, m->A<float>(L.fnw), m->A<float>(L.fnb),
                     m->Wk<void>(ws.p_ln), nullptr, st));
    DEER_TRY(gemm(m, m->Wk<void>(ws.p_ln), m->A<void>(L.w1), m->Wk<void>(ws.p_h), (long)N * nl, (long)c.perc_ff_mult * W, W, DEER_EPI_GELU_BF16, nullptr, st));
    DEER_TRY(gemm_splitk(m, m->Wk<void>(ws.p_h), m->A<void>(L.w2), m->Wk<float>(ws.v_slab), (long)N * nl, W, (long)c.perc_ff_mult * W, Pf, st));
    if (li + 1 < Lp) {
      const PercLayerW& nx = PS.L[li + 1];
      DEER_TRY(vresadd(m, m->Wk<float>(ws.p_lat), m->Wk<float>(ws.v_slab), Pf, (long)N * nl, W, nullptr, m->A<float>(nx.nlw), m->A<float>(nx.nlb),
                       m->Wk<void>(ws.p_latln), nullptr, st));
    } else {   // closing perceiver.norm -> media tokens (bf16 for the K/V GEMM, f32 kept)
      DEER_TRY(vresadd(m, m->Wk<float>(ws.p_lat), m->Wk<float>(ws.v_slab), Pf, (long)N * nl, W, nullptr, m->A<float>(PS.normw),
                       m->A<float>(PS.normb), out_bf, out_f32, st));
    }
  }
  return scatter_media(m, ws, st);
}

int vit_blocks(deer_model* m, const VisionWS& ws, int lo, int hi, void* st) {
  const deer_config& c = m->c;
  const int W = m->W, tok = m->tok, H = c.vit_heads, N = ws.n;
  const long R = (long)N * tok;
  const int So = ws.vit_split[0], Sp = ws.vit_split[1];
  char* qkv = m->Wk<char>(ws.v_qkv);
  for (int li = lo; li < hi; ++li) {
    const VitLayerW& L = m->vit[li];
    DEER_TRY(gemm(m, m->Wk<void>(ws.v_ln), m->A<void>(L.wqkv), qkv, R, 3 * W, W, DEER_EPI_BF16, m->A<float>(L.bqkv), st));
    {
      Bracket b(m, "deer_attn_mfma_hd64", 4.0 * N * H * tok * tok * 64, 0, st);
      DEER_TRY((c.operands_f16 ? deer_attn_f16_hd64 : deer_attn_mfma_hd64)(qkv, qkv + (size_t)W * 2, qkv + (size_t)2 * W * 2, m->Wk<void>(ws.v_ao), N, H, tok, tok, 3 * W, 3 * W, 3 * W, W,
                                   (long)tok * 3 * W, (long)tok * 3 * W, (long)tok * 3 * W, (long)tok * W, 0.125f, st));
    }
    DEER_TRY(gemm_splitk(m, m->Wk<void>(ws.v_ao), m->A<void>(L.wo), m->Wk<float>(ws.v_slab), R, W, W, So, st));
    DEER_TRY(vresadd(m, m->Wk<float>(ws.vx), m->Wk<float>(ws.v_slab), So, R, W, m->A<float>(L.bo), m->A<float>(L.ln2w), m->A<float>(L.ln2b),
                     m->Wk<void>(ws.v_ln), nullptr, st));
    DEER_TRY(gemm(m, m->Wk<void>(ws.v_ln), m->A<void>(L.wfc), m->Wk<void>(ws.v_h), R, c.vit_mlp, W, DEER_EPI_QGELU_BF16, m->A<float>(L.bfc), st));
    DEER_TRY(gemm_splitk(m, m->Wk<void>(ws.v_h), m->A<void>(L.wpr), m->Wk<float>(ws.v_slab), R, W, c.vit_mlp, Sp, st));
    if (li + 1 < c.vit_layers) {
      const VitLayerW& nx = m->vit[li + 1];
      DEER_TRY(vresadd(m, m->Wk<float>(ws.vx), m->Wk<float>(ws.v_slab), Sp, R, W, m->A<float>(L.bpr), m->A<float>(nx.ln1w), m->A<float>(nx.ln1b),
                       m->Wk<void>(ws.v_ln), nullptr, st));
    } else {
      DEER_TRY(vresadd(m, m->Wk<float>(ws.vx), m->Wk<float>(ws.v_slab), Sp, R, W, m->A<float>(L.bpr), nullptr, nullptr, nullptr, nullptr, st));
    }
  }
  return DEER_OK;
}


// ---- fp32-activation arithmetic of the vision tower (deer_config.precision = 1, csrc/precise.hip) --------------------------
int gemmf(deer_model* m, const float* A, long lda, const float* Wt, const float* bias, float* C, long ldc, long M, long N, long K, int epi, void* st) {
  Bracket b(m, "deer_gemm_f32_nt", 2.0 * M * N * K, 4.0 * M * K + 4.0 * N * K + 4.0 * M * N, st);
  return deer_gemm_f32_nt(A, (int)lda, Wt, (int)K, bias, C, (int)ldc, (int)M, (int)N, (int)K, epi, st);
}

int ln_rows_f32(deer_model* m, const float* x, const float* gamma, const float* beta, float* out, long rows, int C, void* st) {
  Bracket b(m, "deer_layernorm_rows", 8.0 * rows * C, 8.0 * rows * C, st);
  return deer_layernorm_rows(x, C, 0, (int)rows, 1, gamma, beta, nullptr, out, C, 0, C, kEps, st);
}

int vit_blocks_f32(deer_model* m, const VisionWS& ws, int lo, int hi, void* st) {
  const deer_config& c = m->c;
  const int W = m->W, tok = m->tok, H = c.vit_heads, N = ws.n;
  const long R = (long)N * tok;
  float* vx = m->Wk<float>(ws.vx);
  float* xn = m->Wk<float>(m->hp.xn);
  float* qkv = m->Wk<float>(m->hp.qkv);
  float* ao = m->Wk<float>(m->hp.ao);
  float* h = m->Wk<float>(m->hp.h);
  for (int li = lo; li < hi; ++li) {
    const VitLayerW& L = m->vit[li];
    if (li == 0) DEER_TRY(ln_rows_f32(m, vx, m->A<float>(L.ln1w), m->A<float>(L.ln1b), xn, R, W, st));
    DEER_TRY(gemmf(m, xn, W, m->A<float>(L.wqkv), m->A<float>(L.bqkv), qkv, 3 * W, R, 3 * W, W, 0, st));
    {
      Bracket b(m, "deer_attn_f32", 4.0 * N * H * tok * tok * 64, 0, st);
      DEER_TRY(deer_attn_f32(qkv, qkv + W, qkv + 2 * W, nullptr, nullptr, ao, N, H, tok, tok, 0, 3 * W, 3 * W, 0, W, (long)tok * 3 * W, (long)tok * 3 * W, 0,
                             (long)tok * W, 0.125f, st));
    }
    DEER_TRY(gemmf(m, ao, W, m->A<float>(L.wo), m->A<float>(L.bo), vx, W, R, W, W, 3, st));                       // x += attn(ln_1(x))
    DEER_TRY(ln_rows_f32(m, vx, m->A<float>(L.ln2w), m->A<float>(L.ln2b), xn, R, W, st));
    DEER_TRY(gemmf(m, xn, W, m->A<float>(L.wfc), m->A<float>(L.bfc), h, c.vit_mlp, R, c.vit_mlp, W, 1, st));      // QuickGELU
    DEER_TRY(gemmf(m, h, c.vit_mlp, m->A<float>(L.wpr), m->A<float>(L.bpr), vx, W, R, W, c.vit_mlp, 3, st));      // x += mlp(ln_2(x))
    if (li + 1 < c.vit_layers) {
      const VitLayerW& nx = m->vit[li + 1];
      DEER_TRY(ln_rows_f32(m, vx, m->A<float>(nx.ln1w), m->A<float>(nx.ln1b), xn, R, W, st));
    }
  }
  return DEER_OK;
}

int perceiver_f32(deer_model* m, const VisionWS& ws, void* st) {
  const deer_config& c = m->c;
  const PercSet& PS = m->ps[ws.cam];
  void* out_bf = ws.out_bf != SIZE_MAX ? m->Wk<void>(ws.out_bf) : m->Wk<void>(ws.vis_x);
  float* out_f32 = ws.out_f32 != SIZE_MAX ? m->Wk<float>(ws.out_f32) : m->Wk<float>(ws.vis_x_f32);
  const int P = m->P, W = m->W, tok = m->tok, nl = m->nl, inner = m->p_inner, Lp = m->Lp;
  const int NF = ws.n, N = m->fuse_pre ? ws.n / 2 : ws.n, PM = m->fuse_pre ? 2 * P : P;      // pre fusion: see perceiver()
  const long NL = (long)N * nl, NP = (long)NF * P, ffw = (long)c.perc_ff_mult * W;
  {
    Bracket b(m, "deer_broadcast_rows", 0, 0, st);
    DEER_TRY(deer_broadcast_rows(m->A<float>(PS.latents), m->Wk<float>(ws.p_lat), (long)nl * W, N, st));
  }
  const float* tokens = m->tokens_override ? m->tokens_override + (size_t)ws.first * P * W : m->Wk<float>(ws.vx) + W;   // skip the cls row
  const long tok_bstride = m->tokens_override ? (long)P * W : (long)tok * W;
  float* lat = m->Wk<float>(ws.p_lat);
  float* mln = m->Wk<float>(m->hp.mln);
  float* mkv = m->Wk<float>(m->hp.mkv);
  float* latln = m->Wk<float>(m->hp.latln);
  float* pqkv = m->Wk<float>(m->hp.pqkv);
  float* pao = m->Wk<float>(m->hp.pao);
  float* pln = m->Wk<float>(m->hp.pln);
  float* ph = m->Wk<float>(m->hp.ph);
  const float scale = 1.0f / sqrtf((float)c.perc_dim_head);
  for (int li = 0; li < Lp; ++li) {
    const PercLayerW& L = PS.L[li];
    {   // norm_media of this layer on the (layer-invariant) media tokens, then its to_kv (helpers.py:47-56)
      Bracket b(m, "deer_layernorm_rows", 8.0 * NP * W, 8.0 * NP * W, st);
      DEER_TRY(deer_layernorm_rows(tokens, W, tok_bstride, P, NF, m->A<float>(PS.nm_w) + (size_t)li * W, m->A<float>(PS.nm_b) + (size_t)li * W,
                                   nullptr, mln, W, (long)P * W, W, kEps, st));
    }
    DEER_TRY(gemmf(m, mln, W, m->A<float>(PS.wkv_all) + (size_t)li * 2 * inner * W, nullptr, mkv, 2 * inner, NP, 2 * inner, W, 0, st));
    DEER_TRY(ln_rows_f32(m, lat, m->A<float>(L.nlw), m->A<float>(L.nlb), latln, NL, W, st));
    DEER_TRY(gemmf(m, latln, W, m->A<float>(L.wqkv), nullptr, pqkv, 3 * inner, NL, 3 * inner, W, 0, st));
    {
      Bracket b(m, "deer_attn_f32", 4.0 * N * c.perc_heads * nl * (PM + nl) * 64, 0, st);
      DEER_TRY(deer_attn_f32(pqkv, mkv, mkv + inner, pqkv + inner, pqkv + 2 * inner, pao, N, c.perc_heads, nl, PM, nl, 3 * inner, 2 * inner, 3 * inner, inner,
                             (long)nl * 3 * inner, (long)PM * 2 * inner, (long)nl * 3 * inner, (long)nl * inner, scale, st));
    }
    DEER_TRY(gemmf(m, pao, inner, m->A<float>(L.wo), nullptr, lat, W, NL, W, inner, 3, st));
    DEER_TRY(ln_rows_f32(m, lat, m->A<float>(L.fnw), m->A<float>(L.fnb), pln, NL, W, st));
    DEER_TRY(gemmf(m, pln, W, m->A<float>(L.w1), nullptr, ph, ffw, NL, ffw, W, 2, st));
    DEER_TRY(gemmf(m, ph, ffw, m->A<float>(L.w2), nullptr, lat, W, NL, W, ffw, 3, st));
  }
  // closing perceiver.norm -> media tokens: f32 (this arithmetic) and bf16 (kept current for readers of "vis_x")
  {
    Bracket b(m, "deer_layernorm_rows", 8.0 * NL * W, 10.0 * NL * W, st);
    DEER_TRY(deer_layernorm_rows(lat, W, 0, (int)NL, 1, m->A<float>(PS.normw), m->A<float>(PS.normb), out_bf, out_f32, W, 0, W, kEps, st));
  }
  return scatter_media(m, ws, st);
}

int media_kv_f32(deer_model* m, void* st) {
  if (!m->n_xattn) return DEER_OK;
  if (m->media_override != nullptr) return DEER_ERR_SHAPE;      // a bf16 media tensor cannot feed the f32 arithmetic: use the model's own
  return gemmf(m, m->Wk<float>(m->vis_x_f32), m->W, m->A<float>(m->wkv_all), nullptr, m->Wk<float>(m->hp.kv_all), (long)m->n_xattn * 2 * m->xinner,
               (long)m->B * m->n_media, (long)m->n_xattn * 2 * m->xinner, m->W, 0, st);
}

int media_kv(deer_model* m, void* st) {
  if (m->c.precision) return media_kv_f32(m, st);
  if (!m->n_xattn) return DEER_OK;
  const void* media = m->media_override ? m->media_override : m->Wk<void>(m->vis_x);
  // the K / V leave in the arithmetic's own 16-bit format (fp16 engines: fp16 K / V for the *_f16 x-attn kernels)
  return gemm(m, media, m->A<void>(m->wkv_all), m->Wk<void>(m->kv_all), (long)m->B * m->n_media, (long)m->n_xattn * 2 * m->xinner, m->W, DEER_EPI_BF16,
              nullptr, st);
}

// ---- LLM ----------------------------------------------------------------------------------------------------------------
struct Pending { const float* slab; int S; long stride; const float* gate; };

int skinny(deer_model* m, const void* Wp, long N, long K, int R, float* out_slab, size_t out_elems, const void* A, int lda, const float* a_slab, int s_in,
           int a_mode, const int* ctl, void* st, int* S_out, long* stride_out) {
  if (R > 128) return DEER_ERR_SHAPE;      // only the hi/lo-plane kernel runs row blocks (bf16 arithmetic, d % 64 == 0)
  const int S = deer_skinny_splitk(R, (int)N, (int)K);
  const int mpad = 16 * ((R + 15) / 16);
  const int planes = m->c.precision ? 2 : 1;
  if ((size_t)planes * S * mpad * N > out_elems) return DEER_ERR_SHAPE;
  *S_out = planes * S;
  *stride_out = (long)mpad * N;
  {
    Bracket b(m, "deer_gemm_skinny", 2.0 * R * N * K, 2.0 * N * K, st);    // algorithmic bytes = the bf16 weights, once
    DEER_TRY((m->f16 ? deer_gemm_skinny_f16 : deer_gemm_skinny)(A, lda, a_slab, s_in, (long)mpad * K, a_mode, Wp, out_slab, R, (int)N, (int)K, S, ctl, st));
  }
  if (planes == 2) {   // fp32 arithmetic: W = hi + lo (two bf16 planes); the lo plane's products go into S more slabs of the same consumer
    auto it = m->pack_lo.find((size_t)(reinterpret_cast<const char*>(Wp) - m->arena));
    if (it == m->pack_lo.end()) return DEER_ERR_SHAPE;
    Bracket b(m, "deer_gemm_skinny", 2.0 * R * N * K, 2.0 * N * K, st);
    DEER_TRY((m->f16 ? deer_gemm_skinny_f16 : deer_gemm_skinny)(A, lda, a_slab, s_in, (long)mpad * K, a_mode, m->A<void>(it->second), out_slab + (size_t)S * mpad * N, R, (int)N, (int)K, S, ctl, st));
  }
  return DEER_OK;
}

// ---- env batch with COMPACTION of exited environments ("compact active rows after each exit layer") --------------------
// The reference stops every environment at its own layer (mosaic_gpt_3b.py:438-443, one environment per process).  In an env batch the
// rows of an environment that has exited leave the trunk: two layers after an exit check (the check of layer e runs beside layer e + 1;
// its verdict is complete before layer e + 2 starts: the step driver waits for it) the first row operation of layer e + 2 GATHERS the
// surviving environments' rows to the front of the other residual-stream buffer and publishes the new row map; every later kernel of the
// layer works on the active slots only (the trunk GEMM runs the body compiled for that many row tiles).  Per-environment results are
// bit-identical to the uncompacted batch: no arithmetic depends on a row's position.  DEER_COMPACT=0 disables it.
struct RowCtx {
  float* x = nullptr;             // residual stream of this layer
  const int* cmap = nullptr;      // row map of this layer (nullptr: no compaction - all rows, identity)
  const float* x_in = nullptr;    // gather source (first row operation of a compaction layer), else nullptr
  const int* cmap_old = nullptr;
  int T = 0;
  int drop_upto = -1;             // gather: environments that exited at a layer <= drop_upto leave (the verdicts ordered before this layer)
};

bool block_hl(const deer_model* m, int R);
// ... not with layerwise_exit_eval: the per-layer heads read hidden_states[layer] in ENVIRONMENT order after the step, when that layer's
// row map has long been overwritten by later compaction layers - an ablation mode, not a throughput path
bool compact_on(const deer_model* m) { return m->compact && m->B > 1 && !m->c.precision && !m->c.layerwise_exit_eval; }
// ... on the hi/lo-plane path of BOTH halves of a layer (every K a multiple of 64)
bool compact_active(const deer_model* m, int R) { return compact_on(m) && block_hl(m, R) && ((((long)m->c.xattn_ff_mult * m->d) & 63) == 0); }

struct PlanRow { int need_pseudo, is_exit, slot; };
std::vector<PlanRow> dynamic_plan(const deer_model* m);
bool is_compaction_layer(const deer_model* m, int layer);
int cmap_parity(const deer_model* m, int layer);

bool use_hl(const deer_model* m, int R) { return !m->c.precision && R > hl_min_rows(); }

// the MPT block of a layer runs on the hi/lo-plane kernels (every K of its four projections is d or mlp_ratio*d)
bool block_hl(const deer_model* m, int R) { return use_hl(m, R) && (m->d & 63) == 0 && (((long)m->c.mlp_ratio * m->d) & 63) == 0; }

int trunk_splitk(const deer_model* m, int R, long N, long K) {
  return block_hl(m, R) ? deer_skinny_hl_splitk(R, (int)N, (int)K) : deer_skinny_splitk(R, (int)N, (int)K);
}

// env-batch form: activation as bf16 hi / lo planes
int skinny_hl(deer_model* m, const void* Wp, long N, long K, int R, float* out_slab, size_t out_elems, const bf16_t* hi, const bf16_t* lo, int lda,
              const int* ctl, void* st, int* S_out, long* stride_out, const RowCtx* rc = nullptr) {
  const int S = deer_skinny_hl_splitk(R, (int)N, (int)K);
  const int mpad = 16 * ((R + 15) / 16);
  if ((size_t)S * mpad * N > out_elems) return DEER_ERR_SHAPE;
  *S_out = S;
  *stride_out = (long)mpad * N;
  Bracket b(m, "deer_gemm_skinny_hl", 2.0 * R * N * K, 2.0 * N * K, st);    // algorithmic bytes = the bf16 weights, once
  if (rc != nullptr && rc->cmap != nullptr)
    return (m->f16 ? deer_gemm_skinny_hl_active_f16 : deer_gemm_skinny_hl_active)(hi, lo, lda, Wp, out_slab, R, (int)N, (int)K, S, mpad, ctl, rc->cmap, rc->T, st);
  if (R > 128) return (m->f16 ? deer_gemm_skinny_hl_rows_f16 : deer_gemm_skinny_hl_rows)(hi, lo, lda, Wp, out_slab, R, (int)N, (int)K, S, mpad, ctl, st);
  return (m->f16 ? deer_gemm_skinny_hl_f16 : deer_gemm_skinny_hl)(hi, lo, lda, Wp, out_slab, R, (int)N, (int)K, S, ctl, st);
}

// GELU(sum of the up-projection's slabs) -> hi / lo planes [R][C] (the activation of the down-projection)
int gelu_split(deer_model* m, const float* slab, int S, long stride, int R, long C, const int* ctl, void* st, const RowCtx* rc = nullptr) {
  Bracket b(m, "deer_slab_gelu_split", 0, 4.0 * S * R * C + 4.0 * R * C, st);
  bf16_t* h = m->Wk<bf16_t>(m->h_hl);
  if (rc != nullptr && rc->cmap != nullptr)
    return (m->f16 ? deer_slab_gelu_split_active_f16 : deer_slab_gelu_split_active)(slab, S, stride, 1, h, h + m->hl_plane_h, R, (int)C, ctl, rc->cmap, rc->T, st);
  return (m->f16 ? deer_slab_gelu_split_f16 : deer_slab_gelu_split)(slab, S, stride, 1, h, h + m->hl_plane_h, R, (int)C, ctl, st);
}

// resadd with the LayerNorm output as hi / lo planes in xn_hl (rows x d)
int resadd_split(deer_model* m, int R, const Pending* p, const float* gamma, const float* beta, float* x_copy, const int* ctl, void* st,
                 const RowCtx* rc = nullptr) {
  Bracket b(m, "deer_resadd_ln", 0, 4.0 * R * m->d * ((p ? p->S : 0) + 3), st);
  bf16_t* h = m->Wk<bf16_t>(m->xn_hl);
  if (rc != nullptr && rc->cmap != nullptr)
    return (m->f16 ? deer_resadd_ln_rows_f16 : deer_resadd_ln_rows)(rc->x, p ? p->slab : nullptr, p ? p->S : 0, p ? p->stride : 0, p ? p->gate : nullptr, gamma, beta, h, h + m->hl_plane_d,
                               nullptr, x_copy, R, m->d, kEps, ctl, rc->cmap, rc->T, rc->x_in, rc->cmap_old, m->B, rc->drop_upto, st);
  return (m->f16 ? deer_resadd_ln_split_f16 : deer_resadd_ln_split)(rc ? rc->x : m->Wk<float>(m->x), p ? p->slab : nullptr, p ? p->S : 0, p ? p->stride : 0, p ? p->gate : nullptr, nullptr, gamma,
                              beta, h, h + m->hl_plane_d, nullptr, x_copy, R, m->d, kEps, ctl, st);
}

int resadd(deer_model* m, int R, const Pending* p, const float* gamma, const float* beta, float* x_copy, const int* ctl, void* st,
           const RowCtx* rc = nullptr) {
  Bracket b(m, "deer_resadd_ln", 0, 4.0 * R * m->d * ((p ? p->S : 0) + 3), st);
  if (rc != nullptr && rc->cmap != nullptr)
    return (m->f16 ? deer_resadd_ln_rows_f16 : deer_resadd_ln_rows)(rc->x, p ? p->slab : nullptr, p ? p->S : 0, p ? p->stride : 0, p ? p->gate : nullptr, gamma, beta, nullptr, nullptr,
                               gamma ? m->Wk<float>(m->xn) : nullptr, x_copy, R, m->d, kEps, ctl, rc->cmap, rc->T, rc->x_in, rc->cmap_old, m->B, rc->drop_upto, st);
  return deer_resadd_ln(rc ? rc->x : m->Wk<float>(m->x), p ? p->slab : nullptr, p ? p->S : 0, p ? p->stride : 0, p ? p->gate : nullptr, nullptr, gamma, beta,
                        nullptr, gamma ? m->Wk<float>(m->xn) : nullptr, x_copy, R, m->d, kEps, ctl, st);
}

// the residual branch a layer leaves un-applied when it is not finalized: the down-projection slabs (shape-determined)
Pending pending_of_down(const deer_model* m, int R) {
  const long K = (long)m->c.mlp_ratio * m->d;
  const int S = trunk_splitk(m, R, m->d, K) * (m->c.precision ? 2 : 1);
  return Pending{m->Wk<float>(m->slab_a), S, (long)(16 * ((R + 15) / 16)) * m->d, nullptr};
}

// One environment, <= 16 text rows: the twelve-launch layer (csrc/trunk_r16.hip) - the wide projections reduce their K split inside
// the workgroup and leave final results (GELU planes / q|k|v + LayerNorm moments), so the two deer_slab_gelu_split passes and the
// slab-reducing q/k LayerNorm launch of the fifteen-launch form below disappear; LayerNorm outputs travel as fragment-ordered bf16
// planes (coalesced operand reads).  DEER_TRUNK_R16=0 selects the form below (env batches, long instructions, 9B).
bool r16_ok(const deer_model* m, int T) {
  static const bool on = [] { const char* e = getenv("DEER_TRUNK_R16"); return e == nullptr || e[0] != '0'; }();
  const deer_config& c = m->c;
  // d = 4096 (MPT-7B, round 5): no q/k LayerNorm there - the wide GEMM's 16-column form has no moments epilogue
  return on && !c.precision && m->B == 1 && T <= 16 && (m->d == 2048 || m->d == 256 || (m->d == 4096 && !c.attn_qk_ln)) && block_hl(m, T) && (((long)c.xattn_ff_mult * m->d) & 63) == 0 &&
         (m->d & 127) == 0 && m->d / c.n_heads <= 128 && ((m->d / c.n_heads) & 3) == 0;
}

int resadd_packed(deer_model* m, int R, const Pending* p, const float* gamma, const float* beta, float* x_copy, const int* ctl, void* st) {
  Bracket b(m, "deer_resadd_ln", 0, 4.0 * R * m->d * ((p ? p->S : 0) + 3), st);
  bf16_t* h = m->Wk<bf16_t>(m->xn_hl);
  return (m->f16 ? deer_resadd_ln_packed_f16 : deer_resadd_ln_packed)(m->Wk<float>(m->x), p ? p->slab : nullptr, p ? p->S : 0, p ? p->stride : 0, p ? p->gate : nullptr, nullptr, gamma, beta, h,
                               h + m->hl_plane_d, nullptr, x_copy, R, m->d, kEps, ctl, st);
}

int wide_gemm(deer_model* m, const void* Wp, long N, int epi, float* out_f32, bf16_t* out_hi, bf16_t* out_lo, long ldo, int T, const int* ctl, void* st) {
  Bracket b(m, "deer_trunk_wide_gemm", 2.0 * T * N * m->d, 2.0 * N * m->d, st);      // algorithmic bytes = the bf16 weights, once
  const bf16_t* h = m->Wk<bf16_t>(m->xn_hl);
  return (m->f16 ? deer_trunk_wide_gemm_f16 : deer_trunk_wide_gemm)(h, h + m->hl_plane_d, Wp, (int)N, m->d, epi, out_f32, out_hi, out_lo, (int)ldo, m->Wk<float>(m->ln_stats), T, ctl, st);
}

int llm_layer_r16(deer_model* m, int i, int T, bool use_mask, bool pending_in, bool finalize, bool use_ctl, void* st) {
  const deer_config& c = m->c;
  const LlmLayerW& L = m->llm[i];
  const int d = m->d, xin = m->xinner;
  const int* ctl = use_ctl ? m->Wk<int>(m->ctl) : nullptr;
  float* slab_a = m->Wk<float>(m->slab_a);
  float* qkv = m->Wk<float>(m->qkv_ws);
  const bf16_t* xh = m->Wk<bf16_t>(m->xn_hl);
  bf16_t* hh = m->Wk<bf16_t>(m->h_hl);
  bf16_t* hl = hh + m->hl_plane_h;
  bf16_t* aoh = m->Wk<bf16_t>(m->ao_hl);
  bf16_t* aol = aoh + m->hl_plane_d;
  const long ffw = (long)c.mlp_ratio * d;
  const size_t rows_cap = std::min(m->B * c.max_text_len, kMaxRows);
  Pending pend{}, *pp = nullptr;
  float* prev_hidden = nullptr;
  if (pending_in) {
    pend = pending_of_down(m, T);
    pp = &pend;
    prev_hidden = m->Wk<float>(m->hidden) + (size_t)(i - 1) * rows_cap * d;
  }
  int S;
  long stride;
  // N1 experiment (csrc/persistent_layer.hip): the whole layer as ONE persistent launch - same device functions, same order, bit-identical
  // results, SLOWER; never the default (a persistent launch needs all of its workgroups resident: one engine per GPU)
  if (m->persistent_layer && !m->f16 && L.has_xa && c.xattn_ff_mult == 4 && c.mlp_ratio == 4) {
    const XattnW& X = L.xa;
    deer_trunk_layer_args a{};
    a.T = T; a.d = d; a.xinner = xin; a.heads = c.xattn_heads; a.n_heads = c.n_heads; a.ffw = (int)ffw; a.n_kv = m->n_media; a.n_per_media = m->n_media;
    a.ld_kv = m->n_xattn * 2 * xin; a.NS = ((d >> 4) + 31) / 32; a.pending_s = pending_in ? pend.S : 0; a.qk_ln = c.attn_qk_ln ? 1 : 0;
    a.s_w2 = deer_skinny_hl_splitk(T, d, (int)ffw); a.s_wo = deer_skinny_hl_splitk(T, d, d); a.s_down = a.s_w2;
    if (pending_in && (pend.S != a.s_down || pend.stride != 16L * d)) return DEER_ERR_SHAPE;
    a.eps = kEps; a.xattn_scale = 1.0f / sqrtf((float)c.xattn_dim_head); a.alibi_bias_max = (float)c.alibi_bias_max;
    a.x = m->Wk<float>(m->x); a.slab_a = slab_a; a.qkv = qkv; a.stats = m->Wk<float>(m->ln_stats); a.prev_hidden = prev_hidden;
    a.hidden_out = finalize ? m->Wk<float>(m->hidden) + (size_t)i * rows_cap * d : nullptr;
    a.xn_hi = const_cast<bf16_t*>(xh); a.xn_lo = const_cast<bf16_t*>(xh) + m->hl_plane_d; a.h_hi = hh; a.h_lo = hl; a.ao_hi = aoh; a.ao_lo = aol;
    a.kv = m->Wk<char>(m->kv_all) + (size_t)X.kv_index * 2 * xin * 2; a.text_time = m->Wk<int>(m->text_time);
    a.key_mask = use_mask ? (m->mask_override ? m->mask_override : m->Wk<unsigned char>(m->key_mask)) : nullptr;
    a.x_nw = m->A<float>(X.nw); a.x_nb = m->A<float>(X.nb); a.x_ag = m->A<float>(X.ag); a.x_fnw = m->A<float>(X.fnw); a.x_fnb = m->A<float>(X.fnb);
    a.x_fg = m->A<float>(X.fg); a.ln1w = m->A<float>(L.ln1w); a.ln1b = m->loaded(L.ln1b_name) ? m->A<float>(L.ln1b) : nullptr;
    a.ln2w = m->A<float>(L.ln2w); a.ln2b = m->loaded(L.ln2b_name) ? m->A<float>(L.ln2b) : nullptr; a.qlnw = m->A<float>(L.qlnw); a.klnw = m->A<float>(L.klnw);
    a.x_wq = m->A<void>(X.wq); a.x_wo = m->A<void>(X.wo); a.x_w1 = m->A<void>(X.w1); a.x_w2 = m->A<void>(X.w2);
    a.wqkv = m->A<void>(L.wqkv); a.wo = m->A<void>(L.wo); a.wup = m->A<void>(L.wup); a.wdown = m->A<void>(L.wdown);
    a.barrier = m->Wk<unsigned>(m->pl_state); a.error = m->Wk<int>(m->pl_state) + 16 * 12;
    a.trace = i < 64 ? a.error + 16 + 64 * i : nullptr;   // per layer: who arrived last at every barrier, and when (tools/persistent_layer_check.py)
    Bracket b(m, "deer_trunk_layer_persistent", 2.0 * T * (2.0 * d * xin + 8.0 * d * d + 12.0 * d * d), 2.0 * (2.0 * d * xin + 20.0 * d * d), st);
    return deer_trunk_layer_persistent(&a, st);
  }
  if (L.has_xa) {   // gated x-attn (helpers.py:260-279)
    const XattnW& X = L.xa;
    DEER_TRY(resadd_packed(m, T, pp, m->A<float>(X.nw), m->A<float>(X.nb), prev_hidden, ctl, st));
    prev_hidden = nullptr;
    const int n_media = m->n_media;
    const char* kv = m->Wk<char>(m->kv_all) + (size_t)X.kv_index * 2 * xin * 2;
    if ((size_t)c.xattn_heads * 16 * d > m->slab_a_elems) return DEER_ERR_SHAPE;
    {
      Bracket b(m, "deer_xattn_fused", 2.0 * T * d * xin * 2, 2.0 * 2 * xin * d, st);
      DEER_TRY((m->f16 ? deer_xattn_fused_packed_f16 : deer_xattn_fused_packed)(xh, xh + m->hl_plane_d, d, m->A<void>(X.wq), kv, m->n_xattn * 2 * xin, xin, m->Wk<int>(m->text_time), n_media, n_media,
                                       m->A<void>(X.wo), slab_a, 16L * d, T, c.xattn_heads, 1.0f / sqrtf((float)c.xattn_dim_head), ctl, st));
    }
    pend = Pending{slab_a, c.xattn_heads, 16L * d, m->A<float>(X.ag)};
    const long xff = (long)c.xattn_ff_mult * d;
    DEER_TRY(resadd_packed(m, T, &pend, m->A<float>(X.fnw), m->A<float>(X.fnb), nullptr, ctl, st));
    DEER_TRY(wide_gemm(m, m->A<void>(X.w1), xff, 1, nullptr, hh, hl, xff, T, ctl, st));                       // GELU(ff.1(.)) as hi / lo planes
    DEER_TRY(skinny_hl(m, m->A<void>(X.w2), d, xff, T, slab_a, m->slab_a_elems, hh, hl, (int)xff, ctl, st, &S, &stride));
    pend = Pending{slab_a, S, stride, m->A<float>(X.fg)};
    pp = &pend;
  }
  // MPT block
  const float* ln1b = m->loaded(L.ln1b_name) ? m->A<float>(L.ln1b) : nullptr;
  const float* ln2b = m->loaded(L.ln2b_name) ? m->A<float>(L.ln2b) : nullptr;
  DEER_TRY(resadd_packed(m, T, pp, m->A<float>(L.ln1w), ln1b, prev_hidden, ctl, st));
  DEER_TRY(wide_gemm(m, m->A<void>(L.wqkv), 3L * d, c.attn_qk_ln ? 2 : 0, qkv, nullptr, nullptr, 3L * d, T, ctl, st));   // final q|k|v (+ LayerNorm moments)
  {
    Bracket b(m, "deer_trunk_mpt_attn", 0, 0, st);
    const unsigned char* km = m->mask_override ? m->mask_override : m->Wk<unsigned char>(m->key_mask);
    DEER_TRY((m->f16 ? deer_trunk_mpt_attn_f16 : deer_trunk_mpt_attn)(qkv, m->Wk<float>(m->ln_stats), d, c.n_heads, m->A<float>(L.qlnw), m->A<float>(L.klnw), kEps, use_mask ? km : nullptr,
                                 (float)c.alibi_bias_max, aoh, aol, d, T, ctl, st));
  }
  DEER_TRY(skinny_hl(m, m->A<void>(L.wo), d, d, T, slab_a, m->slab_a_elems, aoh, aol, d, ctl, st, &S, &stride));
  pend = Pending{slab_a, S, stride, nullptr};
  DEER_TRY(resadd_packed(m, T, &pend, m->A<float>(L.ln2w), ln2b, nullptr, ctl, st));
  DEER_TRY(wide_gemm(m, m->A<void>(L.wup), ffw, 1, nullptr, hh, hl, ffw, T, ctl, st));                         // GELU(mlp_up(.)) as hi / lo planes
  DEER_TRY(skinny_hl(m, m->A<void>(L.wdown), d, ffw, T, slab_a, m->slab_a_elems, hh, hl, (int)ffw, ctl, st, &S, &stride));
  if (finalize) {   // hidden_states[i] = output of layer i (mosaic_gpt_3b.py:424-427)
    pend = Pending{slab_a, S, stride, nullptr};
    DEER_TRY(resadd(m, T, &pend, nullptr, nullptr, m->Wk<float>(m->hidden) + (size_t)i * rows_cap * d, ctl, st));
  }
  return DEER_OK;
}

// FlamingoLayer.forward (flamingo_lm.py:46-83): gated x-attn (helpers.py:260-279) then the MPT block on
// R = n_envs*T rows
int llm_layer(deer_model* m, int i, int T, bool use_mask, bool pending_in, bool finalize, bool use_ctl, void* st) {
  const deer_config& c = m->c;
  const LlmLayerW& L = m->llm[i];
  if (r16_ok(m, T)) return llm_layer_r16(m, i, T, use_mask, pending_in, finalize, use_ctl, st);
  const int d = m->d, B = m->B, R = B * T, xin = m->xinner;
  const int* ctl = use_ctl ? m->Wk<int>(m->ctl) : nullptr;
  float* slab_a = m->Wk<float>(m->slab_a);
  float* slab_b = m->Wk<float>(m->slab_b);
  float* xn = m->Wk<float>(m->xn);
  float* ao = m->Wk<float>(m->ao);
  const bool hl = block_hl(m, R);                          // env batch of 4+ environments: activations as bf16 hi / lo planes
  const bf16_t* xh = m->Wk<bf16_t>(m->xn_hl);
  const bf16_t* hh = m->Wk<bf16_t>(m->h_hl);
  bf16_t* aoh = m->Wk<bf16_t>(m->ao_hl);
  // env batch with compaction of exited environments (hi/lo-plane path, dynamic steps): this layer's residual-stream buffer and row map;
  // the first row operation of a compaction layer gathers the surviving rows from the previous layer's buffer / map
  RowCtx rcx, *rc = nullptr;
  bool gather = false;
  if (use_ctl && compact_active(m, R)) {
    int* maps = m->Wk<int>(m->cmap);
    const int par = cmap_parity(m, i);
    rcx.x = par ? m->Wk<float>(m->x2) : m->Wk<float>(m->x);
    rcx.cmap = maps + par * CMAP_WORDS;
    rcx.T = T;
    rc = &rcx;
    gather = is_compaction_layer(m, i);
  }
  auto first_row_op = [&](RowCtx& first) {                  // the context of the layer's first row operation
    first = rcx;
    if (gather) {
      const int parp = cmap_parity(m, i - 1);
      first.x_in = parp ? m->Wk<float>(m->x2) : m->Wk<float>(m->x);
      first.cmap_old = m->Wk<int>(m->cmap) + parp * CMAP_WORDS;
      first.drop_upto = i - 2;
    }
  };
  RowCtx rfirst;
  if (rc != nullptr) first_row_op(rfirst);
  Pending pend{}, *pp = nullptr;
  // a layer that was not finalized leaves its last residual branch to this layer's first row op, which then ALSO writes the
  // completed x out as hidden_states[i-1] (no extra launch): every hidden state up to the exit layer is real
  float* prev_hidden = nullptr;
  if (pending_in) {
    pend = pending_of_down(m, R);
    pp = &pend;
    prev_hidden = m->Wk<float>(m->hidden) + (size_t)(i - 1) * std::min(B * c.max_text_len, kMaxRows) * d;
  }
  int S;
  long stride;
  if (L.has_xa) {
    const XattnW& X = L.xa;
    DEER_TRY(resadd(m, R, pp, m->A<float>(X.nw), m->A<float>(X.nb), prev_hidden, ctl, st, rc ? &rfirst : nullptr));
    prev_hidden = nullptr;
    static const bool fused = [] { const char* e = getenv("DEER_XATTN_FUSED"); return e == nullptr || e[0] != '0'; }();
    const int n_media = m->n_media;
    const char* kv = m->Wk<char>(m->kv_all) + (size_t)X.kv_index * 2 * xin * 2;
    if (c.precision) {
      // fp32-activation arithmetic: q projection (f32 activations as bf16 hi + lo), fp32 attention over the f32 K/V, output projection
      DEER_TRY(skinny(m, m->A<void>(X.wq), xin, d, R, slab_b, m->slab_b_elems, xn, d, nullptr, 0, DEER_A_F32, ctl, st, &S, &stride));
      {
        Bracket b(m, "deer_xattn_f32", 0, 0, st);
        DEER_TRY(deer_xattn_f32(slab_b, S, stride, xin, m->Wk<float>(m->hp.kv_all) + (size_t)X.kv_index * 2 * xin, m->n_xattn * 2 * xin, xin,
                                m->Wk<int>(m->text_time), n_media, ao, xin, T, n_media, c.xattn_heads, B, 1.0f / sqrtf((float)c.xattn_dim_head), ctl, st));
      }
      DEER_TRY(skinny(m, m->A<void>(X.wo), d, xin, R, slab_a, m->slab_a_elems, ao, xin, nullptr, 0, DEER_A_F32, ctl, st, &S, &stride));
    } else if (fused && T <= 32 && (d & 127) == 0) {
      // to_q -> attention -> to_out in ONE launch; output = one f32 slab per head (xattn_fused.hip)
      const int mpad = 16 * ((R + 15) / 16);
      if ((size_t)c.xattn_heads * mpad * d > m->slab_a_elems) return DEER_ERR_SHAPE;
      Bracket b(m, "deer_xattn_fused", 2.0 * R * d * xin * 2, 2.0 * 2 * xin * d, st);
      if (rc != nullptr)
        DEER_TRY((m->f16 ? deer_xattn_fused_active_f16 : deer_xattn_fused_active)(xn, d, m->A<void>(X.wq), kv, m->n_xattn * 2 * xin, xin, m->Wk<int>(m->text_time), n_media, n_media, m->A<void>(X.wo),
                                         slab_a, (long)mpad * d, T, c.xattn_heads, B, 1.0f / sqrtf((float)c.xattn_dim_head), ctl, rc->cmap, st));
      else
        DEER_TRY((m->f16 ? deer_xattn_fused_f16 : deer_xattn_fused)(xn, d, m->A<void>(X.wq), kv, m->n_xattn * 2 * xin, xin, m->Wk<int>(m->text_time), n_media, n_media, m->A<void>(X.wo),
                                  slab_a, (long)mpad * d, T, c.xattn_heads, B, 1.0f / sqrtf((float)c.xattn_dim_head), ctl, st));
      S = c.xattn_heads;
      stride = (long)mpad * d;
    } else {
      DEER_TRY(skinny(m, m->A<void>(X.wq), xin, d, R, slab_b, m->slab_b_elems, xn, d, nullptr, 0, DEER_A_F32, ctl, st, &S, &stride));
      {
        Bracket b(m, "deer_xattn_mfma", 0, 0, st);
        DEER_TRY((m->f16 ? deer_xattn_mfma_f16 : deer_xattn_mfma)(slab_b, S, stride, xin, kv, m->n_xattn * 2 * xin, xin, m->Wk<int>(m->text_time), n_media, ao, 1, xin, T, n_media,
                                 c.xattn_heads, B, 1.0f / sqrtf((float)c.xattn_dim_head), ctl, st));
      }
      DEER_TRY(skinny(m, m->A<void>(X.wo), d, xin, R, slab_a, m->slab_a_elems, ao, xin, nullptr, 0, DEER_A_F32, ctl, st, &S, &stride));
    }
    pend = Pending{slab_a, S, stride, m->A<float>(X.ag)};
    const long xff = (long)c.xattn_ff_mult * d;
    if (hl && (xff & 63) == 0) {
      DEER_TRY(resadd_split(m, R, &pend, m->A<float>(X.fnw), m->A<float>(X.fnb), nullptr, ctl, st, rc));
      DEER_TRY(skinny_hl(m, m->A<void>(X.w1), xff, d, R, slab_b, m->slab_b_elems, xh, xh + m->hl_plane_d, d, ctl, st, &S, &stride, rc));
      DEER_TRY(gelu_split(m, slab_b, S, stride, R, xff, ctl, st, rc));
      DEER_TRY(skinny_hl(m, m->A<void>(X.w2), d, xff, R, slab_a, m->slab_a_elems, hh, hh + m->hl_plane_h, (int)xff, ctl, st, &S, &stride, rc));
    } else {
      DEER_TRY(resadd(m, R, &pend, m->A<float>(X.fnw), m->A<float>(X.fnb), nullptr, ctl, st));
      DEER_TRY(skinny(m, m->A<void>(X.w1), xff, d, R, slab_b, m->slab_b_elems, xn, d, nullptr, 0, DEER_A_F32, ctl, st, &S, &stride));
      DEER_TRY(skinny(m, m->A<void>(X.w2), d, xff, R, slab_a, m->slab_a_elems, nullptr, 0, slab_b, S, DEER_A_SLABS_GELU, ctl, st, &S, &stride));
    }
    pend = Pending{slab_a, S, stride, m->A<float>(X.fg)};
    pp = &pend;
  }
  const float* ln1b = m->loaded(L.ln1b_name) ? m->A<float>(L.ln1b) : nullptr;
  const float* ln2b = m->loaded(L.ln2b_name) ? m->A<float>(L.ln2b) : nullptr;
  const unsigned char* km = m->mask_override ? m->mask_override : m->Wk<unsigned char>(m->key_mask);
  const long ffw = (long)c.mlp_ratio * d;
  if (hl) {
    // (without an x-attn in this layer the ln_1 row operation is the layer's first one: it does the gathering)
    DEER_TRY(resadd_split(m, R, pp, m->A<float>(L.ln1w), ln1b, prev_hidden, ctl, st, rc ? (L.has_xa ? rc : &rfirst) : nullptr));
    DEER_TRY(skinny_hl(m, m->A<void>(L.wqkv), 3L * d, d, R, slab_b, m->slab_b_elems, xh, xh + m->hl_plane_d, d, ctl, st, &S, &stride, rc));
    {
      Bracket b(m, "deer_mpt_attn_small", 0, 0, st);
      if (rc != nullptr)
        DEER_TRY((m->f16 ? deer_mpt_attn_small_hl_active_f16 : deer_mpt_attn_small_hl_active)(slab_b, S, stride, d, c.n_heads, m->A<float>(L.qlnw), m->A<float>(L.klnw), kEps, use_mask ? km : nullptr,
                                               (float)c.alibi_bias_max, m->Wk<float>(m->qkv_ws), aoh, aoh + m->hl_plane_d, d, T, B, ctl, rc->cmap, st));
      else
        DEER_TRY((m->f16 ? deer_mpt_attn_small_hl_f16 : deer_mpt_attn_small_hl)(slab_b, S, stride, d, c.n_heads, m->A<float>(L.qlnw), m->A<float>(L.klnw), kEps, use_mask ? km : nullptr,
                                        (float)c.alibi_bias_max, m->Wk<float>(m->qkv_ws), aoh, aoh + m->hl_plane_d, d, T, B, ctl, st));
    }
    DEER_TRY(skinny_hl(m, m->A<void>(L.wo), d, d, R, slab_a, m->slab_a_elems, aoh, aoh + m->hl_plane_d, d, ctl, st, &S, &stride, rc));
    pend = Pending{slab_a, S, stride, nullptr};
    DEER_TRY(resadd_split(m, R, &pend, m->A<float>(L.ln2w), ln2b, nullptr, ctl, st, rc));
    DEER_TRY(skinny_hl(m, m->A<void>(L.wup), ffw, d, R, slab_b, m->slab_b_elems, xh, xh + m->hl_plane_d, d, ctl, st, &S, &stride, rc));
    DEER_TRY(gelu_split(m, slab_b, S, stride, R, ffw, ctl, st, rc));
    DEER_TRY(skinny_hl(m, m->A<void>(L.wdown), d, ffw, R, slab_a, m->slab_a_elems, hh, hh + m->hl_plane_h, (int)ffw, ctl, st, &S, &stride, rc));
  } else {
    DEER_TRY(resadd(m, R, pp, m->A<float>(L.ln1w), ln1b, prev_hidden, ctl, st));
    DEER_TRY(skinny(m, m->A<void>(L.wqkv), 3L * d, d, R, slab_b, m->slab_b_elems, xn, d, nullptr, 0, DEER_A_F32, ctl, st, &S, &stride));
    {
      Bracket b(m, "deer_mpt_attn_small", 0, 0, st);
      DEER_TRY(deer_mpt_attn_small(slab_b, S, stride, d, c.n_heads, m->A<float>(L.qlnw), m->A<float>(L.klnw), kEps, use_mask ? km : nullptr,
                                   (float)c.alibi_bias_max, m->Wk<float>(m->qkv_ws), ao, 1, d, T, B, ctl, st));
    }
    DEER_TRY(skinny(m, m->A<void>(L.wo), d, d, R, slab_a, m->slab_a_elems, ao, d, nullptr, 0, DEER_A_F32, ctl, st, &S, &stride));
    pend = Pending{slab_a, S, stride, nullptr};
    DEER_TRY(resadd(m, R, &pend, m->A<float>(L.ln2w), ln2b, nullptr, ctl, st));
    DEER_TRY(skinny(m, m->A<void>(L.wup), ffw, d, R, slab_b, m->slab_b_elems, xn, d, nullptr, 0, DEER_A_F32, ctl, st, &S, &stride));
    DEER_TRY(skinny(m, m->A<void>(L.wdown), d, ffw, R, slab_a, m->slab_a_elems, nullptr, 0, slab_b, S, DEER_A_SLABS_GELU, ctl, st, &S, &stride));
  }
  if (finalize) {   // hidden_states[i] = output of layer i (mosaic_gpt_3b.py:424-427)
    pend = Pending{slab_a, S, stride, nullptr};
    const size_t rows_cap = std::min(B * c.max_text_len, kMaxRows);
    DEER_TRY(resadd(m, R, &pend, nullptr, nullptr, m->Wk<float>(m->hidden) + (size_t)i * rows_cap * d, ctl, st, rc));
  }
  return DEER_OK;
}

// One DeterministicDecoder evaluation on hidden_states[layer] (action_head.py:499-611) followed by the exit gate
// (value_net.py:120-133,277-297).  kind: PSEUDO (prev action from layer i-1, value_net.py:122-125), CHECK (delta <= threshold ->
// exit + commit LSTM state), COMMIT (static exit_id / committing call)
int head_eval(deer_model* m, int layer, int T, int kind, int slot, bool force, bool use_ctl, bool shadow, bool no_ctl_final, const float* feats,
              bool use_mask, void* st, int head = 0) {
  const deer_config& c = m->c;
  const int H = m->H, d = m->d, B = m->B;
  // head 0: extra_exit; head k > 0: the k-th per-layer head of layerwise_exit_eval (own weights, own LSTM state, no pre-pass)
  if (head < 0 || head > (int)m->lw.size()) return DEER_ERR_SHAPE;
  const HeadW* hw = head > 0 ? &m->lw[head - 1] : nullptr;
  const std::vector<LstmW>& lstm_w = hw ? hw->lstm : m->lstm;
  const std::vector<FcW>& fc_w = hw ? hw->fc : m->fc;
  const size_t wa = hw ? hw->wa : m->wa, ba = hw ? hw->ba : m->ba, wg = hw ? hw->wg : m->wg, bg = hw ? hw->bg : m->bg;
  const int* ctl = use_ctl ? m->Wk<int>(m->ctl) : nullptr;
  const size_t rows_cap = std::min(B * c.max_text_len, kMaxRows);
  const bool feats_default = feats == nullptr;
  if (feats == nullptr) feats = m->Wk<float>(m->hidden) + (size_t)layer * rows_cap * d;     // [B*T, d]: env b owns rows b*T .. b*T+T-1
  const unsigned char* km = use_mask ? (m->mask_override ? m->mask_override : m->Wk<unsigned char>(m->key_mask)) : nullptr;
  // ---- control steps of one environment: the whole evaluation as ONE launch (csrc/head.hip: head_fused_kernel) ----
  if (m->head_fused && !m->f16 && head == 0 && use_ctl && !no_ctl_final && feats_default && kind != DEER_KIND_COMMIT && B == 1 && !c.precision && !c.use_state &&
      m->head_pre && m->ghh_valid && m->Lh <= 4 && m->n_fc <= 3 && d <= 2048 && H <= 1024) {
    deer_head_fused_args a{};
    a.feats = feats; a.T = T; a.d = d; a.avg = c.pooling_avg; a.key_mask = km;
    a.cmap = compact_active(m, B * T) ? m->Wk<int>(m->cmap) + cmap_parity(m, layer) * CMAP_WORDS : nullptr;
    a.B = B; a.H = H; a.L = m->Lh; a.n_fc = m->n_fc; a.lstm_ln = c.lstm_layernorm; a.mlp_ln = c.mlp_layernorm;
    for (int i = 0; i < 3; ++i) a.fc_dim[i] = m->fc_dims[i];
    for (int l = 0; l < m->Lh; ++l) {
      a.w_ih[l] = m->A<void>(m->lstm[l].wih); a.b_ih[l] = m->A<float>(m->lstm[l].bih);
      a.ln_w[l] = c.lstm_layernorm ? m->A<float>(m->lstm[l].lnw) : nullptr; a.ln_b[l] = c.lstm_layernorm ? m->A<float>(m->lstm[l].lnb) : nullptr;
    }
    a.ghh = m->Wk<float>(m->ghh); a.c_prev = m->Wk<float>(m->c_state); a.h_tmp = m->Wk<float>(m->h_tmp); a.c_tmp = m->Wk<float>(m->c_tmp);
    for (int i = 0; i < m->n_fc; ++i)
      for (int g = 0; g < 2; ++g) {
        a.fw[i][g] = m->A<void>(m->fc[i].w[g]); a.fb[i][g] = m->A<float>(m->fc[i].b[g]);
        a.fln_w[i][g] = c.mlp_layernorm ? m->A<float>(m->fc[i].lnw[g]) : nullptr; a.fln_b[i][g] = c.mlp_layernorm ? m->A<float>(m->fc[i].lnb[g]) : nullptr;
      }
    a.Wa = m->A<void>(m->wa); a.ba = m->A<float>(m->ba); a.Wg = m->A<void>(m->wg); a.bg = m->A<float>(m->bg);
    a.ctl = m->Wk<int>(m->ctl); a.kind = kind; a.layer = layer; a.slot = slot;
    a.thresholds = m->thr_override ? m->thr_override : m->Wk<float>(m->thresholds);
    a.force = force ? 1 : 0; a.thr_type = m->thr_type; a.leq = m->leq;
    a.h_state = m->Wk<float>(shadow ? m->h_shadow : m->h_state); a.c_state = m->Wk<float>(shadow ? m->c_shadow : m->c_state);
    a.action_dbg = m->Wk<float>(m->action_dbg); a.eps = kEps; a.A = std::max(1, c.multi_step_action); a.act_ext = m->Wk<float>(m->act_ext);
    a.xg = m->Wk<unsigned long long>(m->hf_xg); a.err = m->Wk<int>(m->hf_err);
    static const bool hf_trace_on = [] { const char* e = getenv("DEER_HF_TRACE"); return e != nullptr && e[0] == '1'; }();
    a.trace = hf_trace_on ? m->Wk<unsigned long long>(m->hf_trace) : nullptr;
    Bracket b(m, "deer_head_fused", 0, 2.0 * (4.0 * H * (d + (m->Lh - 1.0) * H)), st);
    const int rc = deer_head_fused(&a, 0, m->head_wgs, st);
    if (rc != DEER_ERR_SHAPE) return rc;                       // DEER_ERR_SHAPE: a shape the one-launch form does not take -> separate kernels
  }
  float* pooled = m->Wk<float>(m->pooled);
  {
    Bracket b(m, "deer_head_pool", 0, 0, st);
    // env batch with compaction: hidden_states[layer] is packed by that layer's row map
    const bool cm = use_ctl && feats_default && compact_active(m, B * T);
    if (c.use_state) DEER_TRY(deer_head_pool_state(feats, pooled, T, d, c.pooling_avg, B, km, m->Wk<float>(m->state_emb), ctl, kind, layer, st));
    else if (cm) DEER_TRY(deer_head_pool_active(feats, pooled, T, d, c.pooling_avg, B, km, ctl, kind, layer, m->Wk<int>(m->cmap) + cmap_parity(m, layer) * CMAP_WORDS, st));
    else DEER_TRY(deer_head_pool(feats, pooled, T, d, c.pooling_avg, B, km, ctl, kind, layer, st));
  }
  float* h_tmp = m->Wk<float>(m->h_tmp);
  float* c_tmp = m->Wk<float>(m->c_tmp);
  const float* h_prev = m->Wk<float>(hw ? hw->h_state : m->h_state);
  const float* c_prev = m->Wk<float>(hw ? hw->c_state : m->c_state);
  const size_t lst = (size_t)B * H;
  for (int l = 0; l < m->Lh; ++l) {
    const LstmW& Lw = lstm_w[l];
    const float* src;
    long bstride;
    int mode, in_dim;
    const float *lnw = nullptr, *lnb = nullptr;
    if (l == 0) { src = pooled; bstride = d; mode = DEER_X_RAW; in_dim = d; }
    else {
      src = h_tmp + (l - 1) * lst; bstride = H; in_dim = H;
      if (c.lstm_layernorm) { mode = DEER_X_LN; lnw = m->A<float>(lstm_w[l - 1].lnw); lnb = m->A<float>(lstm_w[l - 1].lnb); }
      else mode = DEER_X_RAW;
    }
    // control steps (the features are this step's hidden states): the recurrent half comes from deer_begin_step's pre-pass; window mode
    // (explicit features, the state moves from frame to frame inside one call) keeps the fused form
    // ... and so does an evaluation enqueued after the host changed h_state without a deer_begin_step in between (reset,
    // reset_env, a manual commit - deer_model_head_state_changed): the pre-pass result would be stale
    const bool pre = m->head_pre && m->Lh <= 8 && feats_default && m->ghh_valid && head == 0;
    Bracket b(m, "deer_head_lstm_layer", 0, 2.0 * 4 * H * (in_dim + (pre ? 0 : H)), st);
    if (pre)
      DEER_TRY(deer_head_lstm_layer_pre(src, bstride, mode, T, in_dim, lnw, lnb, m->A<void>(Lw.wih), m->A<float>(Lw.bih), m->Wk<float>(m->ghh) + 4 * l * lst,
                                        c_prev + l * lst, h_tmp + l * lst, c_tmp + l * lst, H, B, kEps, ctl, kind, layer, m->wkind, st));
    else
      DEER_TRY(deer_head_lstm_layer(src, bstride, mode, T, in_dim, lnw, lnb, m->A<void>(Lw.wih), m->A<void>(Lw.whh), m->A<float>(Lw.bih), m->A<float>(Lw.bhh),
                                    h_prev + l * lst, c_prev + l * lst, h_tmp + l * lst, c_tmp + l * lst, H, B, kEps, ctl, kind, layer, m->wkind, st));
  }
  const float* src = h_tmp + (m->Lh - 1) * lst;
  int in_dim = H, sstride = H, pro = DEER_PRO_RAW;
  const float* ln[4] = {nullptr, nullptr, nullptr, nullptr};
  if (c.lstm_layernorm) { pro = DEER_PRO_LN; ln[0] = m->A<float>(lstm_w.back().lnw); ln[1] = m->A<float>(lstm_w.back().lnb); }
  for (int fi = 0; fi < m->n_fc; ++fi) {
    const FcW& F = fc_w[fi];
    const int dim = m->fc_dims[fi];
    float* z = m->Wk<float>(m->z_fc[fi]);
    {
      Bracket b(m, "deer_head_fc", 0, 0, st);
      DEER_TRY(deer_head_fc(src, sstride, in_dim, pro, ln[0], ln[1], ln[2], ln[3], m->A<void>(F.w[0]), m->A<float>(F.b[0]), m->A<void>(F.w[1]),
                            m->A<float>(F.b[1]), dim, z, B, kEps, ctl, kind, layer, m->wkind, st));
    }
    src = z; in_dim = dim; sstride = 2 * dim;
    if (c.mlp_layernorm) { pro = DEER_PRO_GROUP_LN_RELU; ln[0] = m->A<float>(F.lnw[0]); ln[1] = m->A<float>(F.lnb[0]); ln[2] = m->A<float>(F.lnw[1]); ln[3] = m->A<float>(F.lnb[1]); }
    else { pro = DEER_PRO_GROUP_RELU; ln[0] = ln[1] = ln[2] = ln[3] = nullptr; }
  }
  Bracket b(m, "deer_head_final", 0, 0, st);
  const float* thr = m->thr_override ? m->thr_override : m->Wk<float>(m->thresholds);
  return deer_head_final_multi(src, sstride, in_dim, pro, ln[0], ln[1], ln[2], ln[3], m->A<void>(wa), m->A<float>(ba), m->A<void>(wg), m->A<float>(bg),
                               no_ctl_final ? nullptr : m->Wk<int>(m->ctl), kind, layer, slot, thr, force ? 1 : 0, m->thr_type, m->leq, h_tmp, c_tmp,
                               m->Wk<float>(shadow ? m->h_shadow : m->h_state), m->Wk<float>(shadow ? m->c_shadow : m->c_state), m->Lh, H, B,
                               m->Wk<float>(m->action_dbg), kEps, m->wkind, std::max(1, c.multi_step_action), m->Wk<float>(m->act_ext), st);
}

// per layer of the dynamic step: (need_pseudo, is_exit, exit slot) - mosaic_gpt_3b.py:397-443, value_net.py:122-125
std::vector<PlanRow> dynamic_plan(const deer_model* m) {
  std::vector<PlanRow> plan;
  auto find = [&](int v) { return (int)(std::find(m->exit_ids.begin(), m->exit_ids.end(), v) - m->exit_ids.begin()); };
  const int ne = (int)m->exit_ids.size();
  for (int i = 0; i < m->c.n_layers; ++i) {
    const bool need_pseudo = find(i + 1) < ne && (i + 1) - m->c.exit_interval < 0 && (i + 1) <= m->ctl_max_layer;
    const int k = find(i);
    const bool is_exit = k < ne && i <= m->ctl_max_layer;
    plan.push_back(PlanRow{need_pseudo ? 1 : 0, is_exit ? 1 : 0, is_exit ? k : -1});
    if (i >= m->ctl_max_layer) break;
  }
  return plan;
}

// compaction layers of the dynamic step: two layers after every exit check (its verdict is complete before that layer starts)
bool is_compaction_layer(const deer_model* m, int layer) {
  if (layer < 2) return false;
  const std::vector<PlanRow> plan = dynamic_plan(m);
  return layer < (int)plan.size() && plan[layer - 2].is_exit != 0;
}

int cmap_parity(const deer_model* m, int layer) {
  const std::vector<PlanRow> plan = dynamic_plan(m);
  int n = 0;
  for (int l = 2; l <= layer && l < (int)plan.size(); ++l) n += plan[l - 2].is_exit ? 1 : 0;
  return n & 1;
}

int embed(deer_model* m, int T, void* st) {
  Bracket b(m, "deer_embed_tokens", 0, 0, st);
  const long long* ids = m->ids_override ? m->ids_override : m->Wk<long long>(m->ids);
  if (m->c.precision)
    return deer_embed_tokens_f32(ids, m->A<float>(m->wte), m->Wk<float>(m->x), m->Wk<int>(m->text_time), T, m->B, m->d, m->c.vocab_size, m->c.media_token_id, st);
  return (m->f16 ? deer_embed_tokens_f16 : deer_embed_tokens)(ids, m->A<void>(m->wte), m->Wk<float>(m->x), m->Wk<int>(m->text_time), T, m->B, m->d, m->c.vocab_size, m->c.media_token_id, st);
}

int llm_dynamic(deer_model* m, int T, bool use_mask, bool shadow, void* st) {
  if (m->c.use_state) return DEER_ERR_SHAPE;            // the reference's dynamic exit raises with use_state (value_net.py:122-129)
  const std::vector<PlanRow> plan = dynamic_plan(m);
  bool pending = false;
  DEER_TRY(embed(m, T, st));
  for (int i = 0; i < (int)plan.size(); ++i) {
    const bool fin = plan[i].need_pseudo || plan[i].is_exit;
    DEER_TRY(llm_layer(m, i, T, use_mask, pending, fin, true, st));
    pending = !fin;
    // rows that are right-padding of an ENV BATCH stay out of the token pool (an independent single-environment run never has
    // them); with one environment the pool covers all T rows like the reference's (action_head.py:519-520)
    const bool pm = use_mask && m->B > 1;
    if (plan[i].need_pseudo) DEER_TRY(head_eval(m, i, T, DEER_KIND_PSEUDO, -1, false, true, false, false, nullptr, pm, st));
    if (plan[i].is_exit)
      DEER_TRY(head_eval(m, i, T, DEER_KIND_CHECK, plan[i].slot, i >= m->ctl_max_layer, true, shadow, false, nullptr, pm, st));
  }
  return DEER_OK;
}

// DeterministicDecoder(use_state=True): the embedding of this step's robot state ("state_in"), once per step
int state_embed(deer_model* m, void* st) {
  if (!m->c.use_state) return DEER_OK;
  Bracket b(m, "deer_head_state_embed", 0, 0, st);
  return deer_head_state_embed(m->Wk<float>(m->state_in), m->A<float>(m->w_arm), m->A<float>(m->b_arm), m->A<float>(m->e_grip), m->A<void>(m->w_state),
                               m->A<float>(m->b_state), m->Wk<float>(m->state_emb), m->d, m->B, m->wkind, st);
}

int llm_static(deer_model* m, int T, bool use_mask, int exit_id, void* st) {
  DEER_TRY(state_embed(m, st));
  DEER_TRY(embed(m, T, st));
  for (int i = 0; i <= exit_id; ++i) DEER_TRY(llm_layer(m, i, T, use_mask, false, true, false, st));
  return head_eval(m, exit_id, T, DEER_KIND_COMMIT, -1, false, false, false, false, nullptr, use_mask && m->B > 1, st);
}

int vision(deer_model* m, const VisionWS& ws, int part, bool with_kv, void* st) {
  const int nv = m->c.vit_layers;
  const int n_head = std::min(kVitHeadLayers, nv - 1);
  const int lo = part == 2 ? n_head : 0, hi = part == 1 ? n_head : nv;
  if (part != 2) DEER_TRY(patch_embed(m, ws, st));
  DEER_TRY(m->c.precision ? vit_blocks_f32(m, ws, lo, hi, st) : vit_blocks(m, ws, lo, hi, st));
  if (part != 1) {
    DEER_TRY(m->c.precision ? perceiver_f32(m, ws, st) : perceiver(m, ws, st));
    if (with_kv) DEER_TRY(media_kv(m, st));
  }
  return DEER_OK;
}

}  // namespace

// ====================================================================================================================
extern "C" {

int deer_model_create(const deer_config* cfg, deer_model** out) {
  if (cfg == nullptr || out == nullptr) return DEER_ERR_SHAPE;
  const deer_config& c = *cfg;
  if (c.n_envs < 1 || c.n_envs > DEER_MAX_ENVS || c.max_text_len < 1 || c.n_layers < 1 || c.vit_layers < 1 || c.perc_depth < 1) return DEER_ERR_SHAPE;
  if (c.image_size % c.patch_size) return DEER_ERR_SHAPE;
  if (c.vit_width % c.vit_heads || c.vit_width / c.vit_heads != 64 || c.perc_dim_head != 64 || c.xattn_dim_head != 64) return DEER_ERR_SHAPE;
  if (c.d_model % 32 || c.d_model / c.n_heads > 128 || 2 * c.perc_latents > 128) return DEER_ERR_SHAPE;
  if (c.mlp_num_hidden_layers < 0 || c.mlp_num_hidden_layers > 3 || c.lstm_num_layers < 1) return DEER_ERR_SHAPE;
  if (c.multi_step_action < 0 || c.multi_step_action > 8 || (c.layerwise_exit_eval && c.use_state)) return DEER_ERR_SHAPE;
  if (c.operands_f16 && c.precision) return DEER_ERR_SHAPE;     // the fp32 arithmetic has no 16-bit tower
  if (c.fusion_pre && c.sep_resampler) return DEER_ERR_SHAPE;   // pre fusion has ONE resampler for both cameras (flamingo_mpt.py:602)
  deer_model* m = new deer_model();
  m->c = c;
  m->f16 = c.operands_f16 != 0;
  m->wkind = c.precision ? 1 : (m->f16 ? 2 : 0);
  const int g = c.image_size / c.patch_size;
  m->P = g * g;
  m->tok = m->P + 1;
  m->W = c.vit_width;
  m->kpad = (3 * c.patch_size * c.patch_size + 63) / 64 * 64;
  m->nl = c.perc_latents;
  m->p_inner = c.perc_heads * c.perc_dim_head;
  m->Lp = c.perc_depth;
  m->d = c.d_model;
  m->xinner = c.xattn_heads * c.xattn_dim_head;
  m->H = c.head_hidden;
  m->Lh = c.lstm_num_layers;
  m->B = c.n_envs;
  m->N = 2 * c.n_envs;
  m->fuse_pre = c.fusion_pre != 0;
  m->n_media = m->fuse_pre ? m->nl : 2 * m->nl;
  m->n_fc = c.mlp_num_hidden_layers;
  const int dims[3] = {1024, 512, 256};                      // action_head.py:87-89
  for (int i = 0; i < 3; ++i) m->fc_dims[i] = dims[i];
  build_arena(m);
  build_workspace(m);
  // default controller: exits every exit_interval layers + the last layer (flamingo_mpt.py:239-250,268-270)
  for (int i = c.exit_interval - 1; i < c.n_layers - 1; i += c.exit_interval) m->exit_ids.push_back(i);
  m->exit_ids.push_back(c.n_layers - 1);
  m->ctl_max_layer = m->exit_ids.back();
  if (const char* e = getenv("DEER_COMPACT")) m->compact = e[0] != '0';
  if (const char* e = getenv("DEER_HEAD_PRE")) m->head_pre = e[0] != '0';
  if (const char* e = getenv("DEER_HEAD_FUSED")) m->head_fused = e[0] != '0';
  if (const char* e = getenv("DEER_HEAD_WGS")) m->head_wgs = std::max(1, std::min(256, atoi(e)));
  if (const char* e = getenv("DEER_PERSISTENT_LAYER")) m->persistent_layer = e[0] == '1';
  *out = m;
  return DEER_OK;
}

int deer_model_set_persistent_layer(deer_model* m, int on) {
  if (m == nullptr) return DEER_ERR_SHAPE;
  m->persistent_layer = on != 0;  // takes effect for pieces enqueued / captured from now on; one engine per GPU only (see csrc/persistent_layer.hip)
  return DEER_OK;
}

// the host wrote h_state (episode reset, shadow copy, manual commit): head evaluations enqueued before the next deer_begin_step fall back
// to the fused LSTM kernel, which reads h_state itself
int deer_model_head_state_changed(deer_model* m) {
  m->ghh_valid = false;
  return DEER_OK;
}

// control steps of one environment: the head evaluation as one launch (DEER_HEAD_FUSED=1; off by default: measured slower) or as the eight separate kernels
int deer_model_set_head_fused(deer_model* m, int on) {
  if (m == nullptr) return DEER_ERR_SHAPE;
  m->head_fused = on != 0;   // takes effect for pieces enqueued / captured from now on
  return DEER_OK;
}

int deer_model_set_compaction(deer_model* m, int on) {
  if (m == nullptr) return DEER_ERR_SHAPE;
  m->compact = on != 0;          // takes effect for pieces enqueued / captured from now on
  return DEER_OK;
}

void deer_model_destroy(deer_model* m) {
  if (m == nullptr) return;
  for (hipEvent_t e : m->ev_pool) (void)hipEventDestroy(e);
  delete m;
}

long deer_model_arena_bytes(const deer_model* m) { return (long)m->al.cur; }
long deer_model_workspace_bytes(const deer_model* m) { return (long)m->wl.cur; }

int deer_model_bind(deer_model* m, void* arena, void* workspace) {
  if (arena == nullptr || workspace == nullptr || ((uintptr_t)arena & 255) || ((uintptr_t)workspace & 255)) return DEER_ERR_SHAPE;
  m->arena = reinterpret_cast<char*>(arena);
  m->ws = reinterpret_cast<char*>(workspace);
  return DEER_OK;
}

int deer_model_share_weights(deer_model* m, const deer_model* src) {
  // a second model object (e.g. another n_envs: the window-mode / env-batch sibling) over the SAME weight arena: the arena layout
  // depends on the shape description only, so the already-ingested tensors are simply adopted
  if (src == nullptr || src->arena == nullptr || m->al.cur != src->al.cur || m->slots.size() != src->slots.size()) return DEER_ERR_SHAPE;
  for (auto& kv : m->slots) {
    auto it = src->slots.find(kv.first);
    if (it == src->slots.end() || it->second.dst[0] != kv.second.dst[0]) return DEER_ERR_SHAPE;
    kv.second.loaded = it->second.loaded;
  }
  m->arena = src->arena;
  return DEER_OK;
}

int deer_model_knows_tensor(const deer_model* m, const char* name) { return m->slots.count(name) ? 1 : 0; }

int deer_model_load_tensor(deer_model* m, const char* name, const void* src, int src_is_bf16, long numel, void* stream) {
  if (m->arena == nullptr || src == nullptr) return DEER_ERR_SHAPE;
  auto it = m->slots.find(name);
  if (it == m->slots.end()) return DEER_ERR_SHAPE;
  Slot& s = it->second;
  if (numel != s.rows * s.cols) return DEER_ERR_SHAPE;
  hipStream_t st = reinterpret_cast<hipStream_t>(stream);
  const long total = s.rows * s.dst_cols;
  const int blocks = (int)std::min<long>((total + 255) / 256, 4096);
  if (s.kind == SK_PACK) {
    if ((s.rows & 15) || (s.cols & 31)) return DEER_ERR_SHAPE;
    const long tot = (s.rows >> 4) * (s.cols >> 5) * 64;
    const int pb = (int)std::min<long>((tot + 255) / 256, 4096);
    if (m->c.operands_f16) {
      if (src_is_bf16) hipLaunchKernelGGL((ingest_pack_kernel<bf16_t, false, true>), dim3(pb), dim3(256), 0, st, (const bf16_t*)src, m->A<bf16_t>(s.dst[0]), (int)s.rows, s.cols);
      else hipLaunchKernelGGL((ingest_pack_kernel<float, false, true>), dim3(pb), dim3(256), 0, st, (const float*)src, m->A<bf16_t>(s.dst[0]), (int)s.rows, s.cols);
    } else if (src_is_bf16) hipLaunchKernelGGL(ingest_pack_kernel<bf16_t>, dim3(pb), dim3(256), 0, st, (const bf16_t*)src, m->A<bf16_t>(s.dst[0]), (int)s.rows, s.cols);
    else hipLaunchKernelGGL(ingest_pack_kernel<float>, dim3(pb), dim3(256), 0, st, (const float*)src, m->A<bf16_t>(s.dst[0]), (int)s.rows, s.cols);
    if (s.dst[1] != SIZE_MAX) {
      if (src_is_bf16) hipLaunchKernelGGL((ingest_pack_kernel<bf16_t, true>), dim3(pb), dim3(256), 0, st, (const bf16_t*)src, m->A<bf16_t>(s.dst[1]), (int)s.rows, s.cols);
      else hipLaunchKernelGGL((ingest_pack_kernel<float, true>), dim3(pb), dim3(256), 0, st, (const float*)src, m->A<bf16_t>(s.dst[1]), (int)s.rows, s.cols);
    }
  } else {
    for (int k = 0; k < 2; ++k) {
      if (s.dst[k] == SIZE_MAX) continue;
      void* dst = m->A<void>(s.dst[k]);
      const long pitch = k == 0 ? s.dst_pitch : s.dst2_pitch;
      if (s.kind == SK_BF16) {
        if (src_is_bf16) hipLaunchKernelGGL((ingest_rows_kernel<bf16_t, 1>), dim3(blocks), dim3(256), 0, st, (const bf16_t*)src, dst, s.rows, s.cols, s.dst_cols, pitch);
        else hipLaunchKernelGGL((ingest_rows_kernel<float, 1>), dim3(blocks), dim3(256), 0, st, (const float*)src, dst, s.rows, s.cols, s.dst_cols, pitch);
      } else if (s.kind == SK_F16) {
        if (src_is_bf16) hipLaunchKernelGGL((ingest_rows_kernel<bf16_t, 2>), dim3(blocks), dim3(256), 0, st, (const bf16_t*)src, dst, s.rows, s.cols, s.dst_cols, pitch);
        else hipLaunchKernelGGL((ingest_rows_kernel<float, 2>), dim3(blocks), dim3(256), 0, st, (const float*)src, dst, s.rows, s.cols, s.dst_cols, pitch);
      } else {
        if (src_is_bf16) hipLaunchKernelGGL((ingest_rows_kernel<bf16_t, 0>), dim3(blocks), dim3(256), 0, st, (const bf16_t*)src, dst, s.rows, s.cols, s.dst_cols, pitch);
        else hipLaunchKernelGGL((ingest_rows_kernel<float, 0>), dim3(blocks), dim3(256), 0, st, (const float*)src, dst, s.rows, s.cols, s.dst_cols, pitch);
      }
    }
  }
  DEER_LAUNCH_CHECK();
  s.loaded = true;
  return DEER_OK;
}

int deer_model_missing_tensors(const deer_model* m, char* buf, int buflen) {
  int n = 0;
  std::string out;
  for (const std::string& name : m->slot_order) {
    const Slot& s = m->slots.at(name);
    if (s.required && !s.loaded) {
      ++n;
      out += name;
      out += '\n';
    }
  }
  if (buf != nullptr && buflen > 0) {
    strncpy(buf, out.c_str(), (size_t)buflen - 1);
    buf[buflen - 1] = 0;
  }
  return n;
}

int deer_model_buffer(const deer_model* m, int which, const char* name, long* offset, long* bytes) {
  if (which == 1) {
    auto it = m->ws_named.find(name);
    if (it == m->ws_named.end()) return DEER_ERR_SHAPE;
    if (offset) *offset = (long)it->second.first;
    if (bytes) *bytes = (long)it->second.second;
    return DEER_OK;
  }
  auto it = m->slots.find(name);
  if (it == m->slots.end()) return DEER_ERR_SHAPE;
  const Slot& s = it->second;
  if (offset) *offset = (long)s.dst[0];
  if (bytes) *bytes = (long)(s.rows * s.dst_pitch * (s.kind == SK_F32 ? 4 : 2));
  return DEER_OK;
}

int deer_model_configure_exit(deer_model* m, const int* exit_ids, int n_exit, int max_layer, int thr_type, int leq) {
  if (exit_ids == nullptr || n_exit <= 0 || n_exit > 16 || thr_type < 0 || thr_type > 3) return DEER_ERR_SHAPE;
  for (int i = 0; i < n_exit; ++i)
    if (exit_ids[i] < 0 || exit_ids[i] >= m->c.n_layers || (i && exit_ids[i] <= exit_ids[i - 1])) return DEER_ERR_SHAPE;
  const int cm = std::min(max_layer - 1, exit_ids[n_exit - 1]);               // value_net.py:173
  // some exit must be reachable, and the deepest reachable layer must BE an exit: otherwise no check would ever be forced and a
  // step would end without a verdict (the reference raises KeyError on thresholds[i] in that situation)
  bool ok = false;
  for (int i = 0; i < n_exit; ++i) ok = ok || exit_ids[i] == cm;
  if (!ok) return DEER_ERR_SHAPE;
  m->exit_ids.assign(exit_ids, exit_ids + n_exit);
  m->ctl_max_layer = cm;
  m->thr_type = thr_type;
  m->leq = leq ? 1 : 0;
  return DEER_OK;
}

int deer_model_real_num_exit(const deer_model* m) {
  int n = 0;
  for (int e : m->exit_ids) n += e <= m->ctl_max_layer ? 1 : 0;
  return n;
}

int deer_model_n_chains(const deer_model* m) { return (int)m->chains.size(); }

int deer_dynamic_plan(const deer_model* m, int* need_pseudo, int* is_exit, int* slot, int cap) {
  const std::vector<PlanRow> plan = dynamic_plan(m);
  for (int i = 0; i < (int)plan.size() && i < cap; ++i) {
    if (need_pseudo) need_pseudo[i] = plan[i].need_pseudo;
    if (is_exit) is_exit[i] = plan[i].is_exit;
    if (slot) slot[i] = plan[i].slot;
  }
  return (int)plan.size();
}

// ---- pieces -------------------------------------------------------------------------------------------------------------
int deer_begin_step(deer_model* m, const int* step_info, void* stream) {
  if (m->ws == nullptr) return DEER_ERR_SHAPE;
  if (m->head_pre && m->Lh <= 8) {
    // every head evaluation of this step starts from the LSTM state the previous step committed (action_head.py:560-575 with
    // update_hidden_state=False until the committing call): its recurrent half W_hh h + b_hh is computed here, once, for all layers
    Bracket b(m, "deer_head_lstm_hh", 0, 2.0 * m->Lh * 4 * m->H * m->H, stream);
    const void* w[8];
    const float* bb[8];
    for (int l = 0; l < m->Lh; ++l) { w[l] = m->A<void>(m->lstm[l].whh); bb[l] = m->A<float>(m->lstm[l].bhh); }
    DEER_TRY(deer_head_lstm_hh(w, bb, m->Lh, m->Wk<float>(m->h_state), m->Wk<float>(m->ghh), m->H, m->B, m->wkind, stream));
    m->ghh_valid = true;
  }
  Bracket b(m, "deer_ctl_begin_step", 0, 0, stream);
  if (compact_on(m)) return deer_ctl_begin_step_map(m->Wk<int>(m->ctl), step_info ? step_info : m->Wk<int>(m->step_info), m->B, m->Wk<int>(m->cmap), stream);
  return deer_ctl_begin_step(m->Wk<int>(m->ctl), step_info ? step_info : m->Wk<int>(m->step_info), m->B, stream);
}

int deer_vision(deer_model* m, int chain, int part, int with_kv, void* stream) {
  if (m->ws == nullptr || part < 0 || part > 2 || chain >= (int)m->chains.size()) return DEER_ERR_SHAPE;
  if (chain < 0 && m->c.sep_resampler) {                // the batched pass would mix the cameras: run the two camera chains in turn
    for (const VisionWS& cw : m->chains) DEER_TRY(vision(m, cw, part, false, stream));
    return (with_kv && part != 1) ? media_kv(m, stream) : DEER_OK;
  }
  return vision(m, chain < 0 ? m->vws : m->chains[chain], part, with_kv != 0, stream);
}

int deer_media_kv(deer_model* m, void* stream) { return m->ws ? media_kv(m, stream) : DEER_ERR_SHAPE; }

int deer_head_state(deer_model* m, void* stream) { return m->ws ? state_embed(m, stream) : DEER_ERR_SHAPE; }

int deer_llm_embed(deer_model* m, int T, void* stream) {
  if (m->ws == nullptr || T <= 0 || T > m->c.max_text_len || m->B * T > kMaxRows) return DEER_ERR_SHAPE;
  return embed(m, T, stream);
}

int deer_llm_layer(deer_model* m, int layer, int T, int use_mask, int pending_in, int finalize, int use_ctl, void* stream) {
  if (m->ws == nullptr || layer < 0 || layer >= m->c.n_layers || T <= 0 || T > m->c.max_text_len || m->B * T > kMaxRows) return DEER_ERR_SHAPE;
  return llm_layer(m, layer, T, use_mask != 0, pending_in != 0, finalize != 0, use_ctl != 0, stream);
}

int deer_head_eval(deer_model* m, int layer, int T, int kind, int slot, int force, int use_ctl, int shadow, int no_ctl_final, const float* feats,
                   int use_mask, void* stream) {
  if (m->ws == nullptr || layer < 0 || layer >= m->c.n_layers || T <= 0 || kind < 0 || kind > 2) return DEER_ERR_SHAPE;
  return head_eval(m, layer, T, kind, slot, force != 0, use_ctl != 0, shadow != 0, no_ctl_final != 0, feats, use_mask != 0, stream);
}

// layerwise_exit_eval (flamingo_mpt.py:450-457): the action of exit layer `layer` from that layer's OWN head (head = 1 + its index in the
// reference's registration order: lm_exit_modules.0 .., lm_head last) on hidden_states[layer], starting from that head's own LSTM state
// (workspace "lw_state").  The outputs land in "action_dbg", the new state in h_tmp / c_tmp: the caller commits it for the environments
// that exited at this layer.  deer_model_layerwise_head: head index of an exit layer, or -1.
int deer_model_layerwise_head(const deer_model* m, int layer) {
  for (size_t k = 0; k < m->lw.size(); ++k)
    if (m->lw[k].layer == layer) return (int)k + 1;
  return -1;
}

int deer_head_eval_layerwise(deer_model* m, int head, int layer, int T, int use_mask, void* stream) {
  if (m->ws == nullptr || head < 1 || head > (int)m->lw.size() || layer < 0 || layer >= m->c.n_layers || T <= 0) return DEER_ERR_SHAPE;
  return head_eval(m, layer, T, DEER_KIND_COMMIT, -1, false, false, false, true, nullptr, use_mask != 0, stream, head);
}

int deer_step_enqueue(deer_model* m, int T, int use_mask, int exit_id, int shadow, const int* step_info, void* stream) {
  if (m->ws == nullptr || T <= 0 || T > m->c.max_text_len || m->B * T > kMaxRows || exit_id >= m->c.n_layers) return DEER_ERR_SHAPE;
  DEER_TRY(deer_begin_step(m, step_info, stream));
  if (m->c.sep_resampler) {                             // one chain per camera (own Perceiver weights), same stream
    for (const VisionWS& cw : m->chains) DEER_TRY(vision(m, cw, 0, false, stream));
    DEER_TRY(media_kv(m, stream));
  } else {
    DEER_TRY(vision(m, m->vws, 0, true, stream));
  }
  return exit_id < 0 ? llm_dynamic(m, T, use_mask != 0, shadow != 0, stream) : llm_static(m, T, use_mask != 0, exit_id, stream);
}

// ---- the three coarse operators ----------------------------------------------------------------------------
int deer_vit_l14_encode(deer_model* m, const void* images_bf16, int n_images, float* tokens_out, void* stream) {
  if (m->ws == nullptr || images_bf16 == nullptr || n_images != m->N) return DEER_ERR_SHAPE;
  m->img_override = images_bf16;
  int rc = patch_embed(m, m->vws, stream);
  if (rc == DEER_OK) rc = m->c.precision ? vit_blocks_f32(m, m->vws, 0, m->c.vit_layers, stream) : vit_blocks(m, m->vws, 0, m->c.vit_layers, stream);
  m->img_override = nullptr;
  if (rc != DEER_OK) return rc;
  if (tokens_out != nullptr) {   // patch tokens x[:, 1:] of every image, without ln_post (flamingo_mpt.py:580)
    const size_t row = (size_t)m->W * 4;
    for (int n = 0; n < n_images; ++n)
      if (hipMemcpyAsync(tokens_out + (size_t)n * m->P * m->W, m->Wk<float>(m->vws.vx) + ((size_t)n * m->tok + 1) * m->W, row * m->P, hipMemcpyDeviceToDevice,
                         (hipStream_t)stream) != hipSuccess)
        return DEER_ERR_LAUNCH;
  }
  return DEER_OK;
}

int deer_perceiver_resample(deer_model* m, const float* tokens, int n_images, void* media_bf16_out, float* media_f32_out, void* stream) {
  if (m->ws == nullptr || n_images != m->N) return DEER_ERR_SHAPE;
  m->tokens_override = tokens;
  int rc = DEER_OK;
  if (m->c.sep_resampler) {                             // frames camera-major: one pass per camera with its own weight set
    for (const VisionWS& cw : m->chains)
      if (rc == DEER_OK) rc = m->c.precision ? perceiver_f32(m, cw, stream) : perceiver(m, cw, stream);
  } else {
    rc = m->c.precision ? perceiver_f32(m, m->vws, stream) : perceiver(m, m->vws, stream);
  }
  m->tokens_override = nullptr;
  if (rc != DEER_OK) return rc;
  const size_t n = (size_t)m->B * m->n_media * m->W;
  if (media_bf16_out && hipMemcpyAsync(media_bf16_out, m->Wk<void>(m->vis_x), n * 2, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) return DEER_ERR_LAUNCH;
  if (media_f32_out && hipMemcpyAsync(media_f32_out, m->Wk<void>(m->vis_x_f32), n * 4, hipMemcpyDeviceToDevice, (hipStream_t)stream) != hipSuccess) return DEER_ERR_LAUNCH;
  return DEER_OK;
}

int deer_llm_early_exit(deer_model* m, const long long* ids, const unsigned char* key_mask, int T, const void* media_bf16, int exit_id, int shadow,
                        const float* thresholds, const int* step_info, void* stream) {
  if (m->ws == nullptr || ids == nullptr || T <= 0 || T > m->c.max_text_len || m->B * T > kMaxRows || exit_id >= m->c.n_layers) return DEER_ERR_SHAPE;
  m->ids_override = ids;
  m->mask_override = key_mask;
  m->media_override = media_bf16;
  m->thr_override = thresholds;
  int rc = deer_begin_step(m, step_info, stream);
  if (rc == DEER_OK) rc = media_kv(m, stream);
  if (rc == DEER_OK) rc = exit_id < 0 ? llm_dynamic(m, T, key_mask != nullptr, shadow != 0, stream) : llm_static(m, T, key_mask != nullptr, exit_id, stream);
  m->ids_override = nullptr;
  m->mask_override = nullptr;
  m->media_override = nullptr;
  m->thr_override = nullptr;
  return rc;
}

// ---- profiler -------------------------------------------------------------------------------------------------------------
int deer_prof_enable(deer_model* m, int on) {
  m->prof_on = on != 0;
  if (on) {
    m->prof.clear();
    m->ev_next = 0;
  }
  return DEER_OK;
}

int deer_prof_count(const deer_model* m) { return (int)m->prof.size(); }

int deer_prof_get(deer_model* m, int i, char* name, int name_len, float* us, double* flops, double* bytes) {
  if (i < 0 || i >= (int)m->prof.size()) return DEER_ERR_SHAPE;
  const ProfRec& r = m->prof[i];
  if (name && name_len > 0) {
    strncpy(name, r.name.c_str(), (size_t)name_len - 1);
    name[name_len - 1] = 0;
  }
  float ms = 0.f;
  if (hipEventSynchronize(r.e1) != hipSuccess || hipEventElapsedTime(&ms, r.e0, r.e1) != hipSuccess) return DEER_ERR_LAUNCH;
  if (us) *us = ms * 1e3f;
  if (flops) *flops = r.flops;
  if (bytes) *bytes = r.bytes;
  return DEER_OK;
}

}  // extern "C"
