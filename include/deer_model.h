/* deer_model.h - the native spine of the DeeR-VLA early-exit forward path on MI355X: a model object that owns the weight
 * arena layout, the workspace, the device-side exit control blocks and the ORDER in which the gfx950 kernels of
 * deer_hip.h are enqueued.  C ABI (plain pointers / sizes / hipStream_t as void*, no torch types).
 *
 * What it replaces in the reference (paths relative to the reference repo):
 *   deer_vit_l14_encode      open_clip ViT-L/14 `vision_encoder.visual(x)[1]`          robot_flamingo/models/flamingo_mpt.py:556-583
 *   deer_perceiver_resample  PerceiverResampler.forward                                open_flamingo/src/helpers.py:107-132
 *   deer_llm_early_exit      FlamingoLMMixin.forward -> MosaicGPT.forward layer loop   open_flamingo/src/flamingo_lm.py:204-233,
 *                            with the exit controller + action head at the exits       mosaic_gpt_3b.py:274-449, value_net.py:120-133,
 *                                                                                      277-297, action_head.py:499-611,
 *                                                                                      flamingo_mpt.py:443-461
 * (SURVEY.md §8b "What the HIP extension exports").  The piece-wise entry points below them are what a host uses to feed
 * the step as graph pieces (deer_vla_amd/engine.py) - same kernels, same order.
 *
 * Every function only ENQUEUES on `stream` (graph-capturable; no allocation, no synchronisation) unless it says "host only".
 * Return codes: 0 ok, DEER_ERR_SHAPE (1) invalid argument / state, DEER_ERR_LAUNCH (2) a kernel launch failed.
 */
#ifndef DEER_MODEL_H
#define DEER_MODEL_H
#ifdef __cplusplus
extern "C" {
#endif

/* Shape description of one model (mirrors deer_vla_amd/config.py::DeerConfig; the reference scatters these over
 * robot_flamingo/models/factory.py:13-26, the HF config.json of the MPT repos, open_clip's ViT-L-14 config and the DeeR
 * checkpoint dict, robot_flamingo/eval/eval_calvin.py:455-476). */
typedef struct deer_config {
  int image_size, patch_size, vit_width, vit_layers, vit_heads, vit_mlp;
  int perc_depth, perc_heads, perc_dim_head, perc_latents, perc_ff_mult;
  int vocab_size, d_model, n_heads, n_layers, mlp_ratio, attn_qk_ln, alibi_bias_max;
  int cross_attn_every_n_layers, xattn_heads, xattn_dim_head, xattn_ff_mult, media_token_id;
  int mpt7b_names;          /* 1: norm_1 / ffn.up_proj / ffn.down_proj parameter names (modeling_gpt_9b.py), 0: ln_1 / mlp.mlp_up */
  int exit_interval;
  int head_hidden, lstm_num_layers, lstm_layernorm, mlp_layernorm, mlp_num_hidden_layers, pooling_avg;
  int n_envs;               /* environments evaluated per control step (1..16) */
  int max_text_len;         /* longest instruction (tokens); n_envs * T must be <= 512 rows (128 with precision = 1) */
  int n_chains;             /* independent vision chains of the two-stream schedule (0 = default 2) */
  int precision;            /* 0: 16-bit MFMA operands (the product path; format: operands_f16 below); 1: fp32 arithmetic -
                             * f32 activations and f32 (or bf16 hi + lo) weight copies everywhere (csrc/precise.hip; single-stream schedule;
                             * ~1/10 of the vision tower's throughput, 1.8x the arena) */
  int use_state;            /* DeterministicDecoder(use_state=True), action_head.py:524-536: the embedded robot state (workspace buffer
                             * "state_in", [n_envs][8] f32) is added to the pooled feature.  Static exit_id only: the reference's dynamic
                             * exit raises with it (value_net.py:122-129 calls the head without a state tensor). */
  int sep_resampler;        /* flamingo_mpt.py:132-134,656-659: the gripper camera has its own PerceiverResampler ("perceiver_gripper.*").
                             * The camera frames are then ordered camera-major ([rgb of every env ; gripper of every env]) and every vision
                             * chain holds the frames of ONE camera. */
  int multi_step_action;    /* A (0 / 1 = one action): the heads emit 6 A pose + A gripper values per call (action_head.py:458,472-473); the
                             * exit criterion's delta runs over all 6 A pose values (value_net.py:105-133); A <= 8 */
  int layerwise_exit_eval;  /* flamingo_mpt.py:236-244,253,450-457: per-layer heads "lm_exit_modules.j.*" / "lm_head.*" are ingested next to
                             * "extra_exit.*", each with its own LSTM state; deer_head_eval_layerwise evaluates one of them */
  int operands_f16;         /* round 6 (precision = 0 only): the 16-bit FORMAT of the whole path.  1 = every 16-bit MFMA operand is IEEE
                             * fp16 - weights of the vision tower / trunk / head, LayerNorm
                             * and GEMM results of the tower, the trunk's hi + lo activation planes, media and trunk K/V, the embedding
                             * table, the camera frames ("img") - the reference's evaluation arithmetic (fp32 weights under fp16 autocast,
                             * eval_utils.py:333); accumulation, LayerNorm / softmax statistics, residual streams, LSTM state stay f32.
                             * 0 = the same kernels on bf16 operands (a `--precision bf16` / amp_bf16 reference run).
                             * Python: precision = "fp16" (default) | "bf16" | "fp32"  (deer_vla_amd/_abi.py PRECISIONS). */
  int fusion_pre;           /* round 6: 1 = `fusion_mode='pre'` (flamingo_mpt.py:378-379,585-607): the patch tokens of an environment's two frames are
                             * ONE media sequence (2 x 256 tokens) for ONE PerceiverResampler call -> perc_latents media tokens per environment for
                             * the gated x-attn (post fusion, the default and every released checkpoint: one call per frame, 2 x perc_latents) */
} deer_config;

typedef struct deer_model deer_model;

/* ---- construction (host only) ---------------------------------------------------------------------------------- */
int deer_model_create(const deer_config* cfg, deer_model** out);
void deer_model_destroy(deer_model* m);
long deer_model_arena_bytes(const deer_model* m);      /* device bytes for the weights (bf16 GEMM operands, f32 norms/biases) */
long deer_model_workspace_bytes(const deer_model* m);  /* device bytes for activations, slabs, LSTM state, control blocks */
/* arena / workspace: device allocations of at least the sizes above, 256-byte aligned, owned by the caller */
int deer_model_bind(deer_model* m, void* arena, void* workspace);
/* Ingest one tensor of the REFERENCE's state dict (names of SURVEY.md §8b "Weight/ckpt format", e.g.
 * "vision_encoder.visual.transformer.resblocks.3.attn.in_proj_weight", "perceiver.layers.0.0.to_kv.weight",
 * "lang_encoder.transformer.blocks.5.decoder_layer.attn.Wqkv.weight", "extra_exit.rnn.layers.0.weight_ih_l0").
 * src: DEVICE pointer, f32 (src_is_bf16 = 0) or bf16 (1), `numel` elements in the reference's layout; the model converts
 * and re-lays it out (rounded once to the operand format - fp16 or bf16, see operands_f16; MFMA-fragment packing for the LLM projections; conv1 reshaped + zero-padded; Perceiver
 * to_q/to_kv stacked; x-attn to_kv of all layers concatenated).  Returns DEER_ERR_SHAPE for an unknown name / wrong size. */
int deer_model_load_tensor(deer_model* m, const char* name, const void* src, int src_is_bf16, long numel, void* stream);
/* host only: adopt the weight arena (and the loaded state) of another model built from the same shape description with a
 * different n_envs / max_text_len - one copy of the weights serves the single-environment engine, the env-batch engine and the
 * window-mode (calibration) engine.  `m` still needs its own workspace (deer_model_bind with arena = src's arena, or call this
 * after binding the workspace). */
int deer_model_share_weights(deer_model* m, const deer_model* src);
int deer_model_knows_tensor(const deer_model* m, const char* name);              /* host only: 1 if `name` is a parameter of this model */
int deer_model_missing_tensors(const deer_model* m, char* buf, int buflen);      /* host only: count of REQUIRED tensors not loaded; names (newline separated) into buf */
/* host only: location of a named buffer (bytes from the base).  which = 0 arena, 1 workspace.  Workspace names: "img", "vx",
 * "vis_x", "vis_x_f32", "kv_all", "ids", "key_mask", "text_time", "x", "hidden", "h_state", "c_state", "h_tmp", "c_tmp",
 * "h_shadow", "c_shadow", "pooled", "ctl", "thresholds", "step_info", "action_dbg" ([n_envs][64]: pose 6 A | gripper A | logit A),
 * "act_ext" (multi_step_action > 1: [n_envs][4][64] previous / committed / ensemble action), "lw_state" (layerwise_exit_eval). */
int deer_model_buffer(const deer_model* m, int which, const char* name, long* offset, long* bytes);

/* ---- exit controller configuration (ExitController.__init__ / _set_threshold_value, value_net.py:164-183) ------- */
/* host only.  max_layer as given to the reference's controller (the controller uses min(max_layer - 1, last exit)). */
/* Env batches (n_envs > 1): compaction of the rows of exited environments in the trunk (csrc/model.hip).  On by default
 * (DEER_COMPACT=0 at creation turns it off); changing it affects what is enqueued / captured afterwards.  Results per environment are
 * bit-identical either way. */
int deer_model_set_compaction(deer_model* m, int on);
/* one-environment control steps: head evaluation as ONE launch (deer_head_fused; an experiment, off by default: measured slower than the
 * eight separate kernels) or as the eight separate kernels (default) */
int deer_model_set_head_fused(deer_model* m, int on);
/* N1 experiment (csrc/persistent_layer.hip): every trunk layer of a one-environment step as ONE persistent launch (12 phases, device-wide
 * barriers) instead of twelve launches.  Bit-identical, slower, and only safe with ONE engine per GPU: off unless DEER_PERSISTENT_LAYER=1. */
int deer_model_set_persistent_layer(deer_model* m, int on);
int deer_model_configure_exit(deer_model* m, const int* exit_ids, int n_exit, int max_layer, int thr_type, int leq);
int deer_model_real_num_exit(const deer_model* m);

/* ---- the three coarse operators of SURVEY.md §8b ------------------------------------------------------------------ */
/* images: bf16 [n_images,3,S,S] (already CLIP-normalised); tokens_out: f32 [n_images,256,W] patch tokens (x[:,1:], no ln_post)
 * or NULL to leave them in the workspace only (the Perceiver reads them there). */
int deer_vit_l14_encode(deer_model* m, const void* images_bf16, int n_images, float* tokens_out, void* stream);   /* images: f32 when precision = 1, fp16 when operands_f16 */
/* (operands_f16: every `*_bf16` media / image argument below carries IEEE fp16 instead - the model's 16-bit format) */
/* tokens: f32 [n_images,256,W] or NULL (= the workspace tokens of the last deer_vit_l14_encode); media_bf16_out / media_f32_out:
 * [n_images*64, W] latents of every image in image order (rgb, gripper per environment = the post-fusion concat of
 * flamingo_mpt.py:661; with fusion_pre [n_envs*64, W]) or NULL to leave them in the workspace. */
int deer_perceiver_resample(deer_model* m, const float* tokens, int n_images, void* media_bf16_out, float* media_f32_out,
                            void* stream);
/* ids: int64 [n_envs,T] device; key_mask: uint8 [n_envs,T] (0 = padding) or NULL; media_bf16: [n_envs*128, W] or NULL (workspace);
 * exit_id >= 0: static exit (flamingo_mpt.py:446-461), exit_id < 0: dynamic exit with the configured controller;
 * thresholds: device f32[16] or NULL (= the model's own "thresholds" buffer); step_info: device-visible int32[4]
 * {hold mask (bit b: environment b is inside a stage, its step % steps_per_stage != 0), sequence number, host mirror ptr lo, hi} or NULL.  Results: control blocks ("ctl": exit layer, action, deltas per
 * environment), hidden states ("hidden"), committed LSTM state ("h_state"/"c_state"). */
int deer_llm_early_exit(deer_model* m, const long long* ids, const unsigned char* key_mask, int T, const void* media_bf16,
                        int exit_id, int shadow, const float* thresholds, const int* step_info, void* stream);

/* ---- pieces (the same work in host-schedulable units; engine.py replays them as HIP-graph pieces) ------------------ */
/* resets the control blocks AND computes the recurrent half (W_hh h + b_hh of every LSTM layer) of this step's head evaluations from the
 * committed LSTM state.  deer_head_eval(feats = NULL) uses that pre-pass result while it is valid: call deer_model_head_state_changed()
 * after writing h_state from the host (episode reset, manual commit) - evaluations enqueued before the next deer_begin_step then stream
 * W_hh themselves (fused kernel, same result up to summation order). */
int deer_begin_step(deer_model* m, const int* step_info, void* stream);
int deer_model_head_state_changed(deer_model* m);
/* chain: -1 = all camera frames batched on one stream, c >= 0 = chain c of the multi-stream schedule (its own workspace);
 * part: 0 whole tower, 1 patch embedding + first blocks, 2 the rest + Perceiver; media_kv: also project K|V of every x-attn layer */
int deer_vision(deer_model* m, int chain, int part, int media_kv, void* stream);
int deer_media_kv(deer_model* m, void* stream);
/* use_state models: the embedding of this step's robot state (workspace "state_in") for the head evaluations of the step; no-op otherwise */
int deer_head_state(deer_model* m, void* stream);
int deer_llm_embed(deer_model* m, int T, void* stream);
/* pending_in: the previous layer left its last residual branch un-applied (it was not finalized); finalize: write hidden[i] */
int deer_llm_layer(deer_model* m, int layer, int T, int use_mask, int pending_in, int finalize, int use_ctl, void* stream);
/* one DeterministicDecoder evaluation + exit gate.  feats: NULL = hidden[layer]; kind DEER_KIND_*; slot = exit index */
int deer_head_eval(deer_model* m, int layer, int T, int kind, int slot, int force, int use_ctl, int shadow, int no_ctl_final,
                   const float* feats, int use_mask, void* stream);
/* layerwise_exit_eval: head index (1-based, the reference's registration order) of an exit layer or -1; one evaluation of that head on
 * hidden[layer] from ITS LSTM state (workspace "lw_state": [head][h, c][L][n_envs][H]); outputs in "action_dbg", new state in h_tmp / c_tmp */
int deer_model_layerwise_head(const deer_model* m, int layer);
int deer_head_eval_layerwise(deer_model* m, int head, int layer, int T, int use_mask, void* stream);
/* whole control step on ONE stream (vision batched): begin + vision + media K/V + trunk + heads */
int deer_step_enqueue(deer_model* m, int T, int use_mask, int exit_id, int shadow, const int* step_info, void* stream);
/* layers of the dynamic step: returns n and fills need_pseudo[i], is_exit[i], slot[i] for i < n (host only) */
int deer_dynamic_plan(const deer_model* m, int* need_pseudo, int* is_exit, int* slot, int cap);
int deer_model_n_chains(const deer_model* m);

/* ---- native driver of the pipelined dynamic step (csrc/step_driver.hip) -------------------------------------------------
 * The host binding captures the pieces above as HIP graphs once (two graphs per vision chain, one per trunk layer of the dynamic
 * plan, one per head evaluation) and hands the hipGraphExec_t handles over; deer_step_plan_run then submits them, keeps at most
 * `lookahead` trunk layers in flight beyond an undecided exit check, polls the pinned verdict mirror the device writes
 * (deer_hip.h DEER_HOSTM_*) and stops at the exit - replaces the host loop of the reference (mosaic_gpt_3b.py:397-443, one
 * `bool(value <= thr)` sync per exit) without a stream synchronisation.  step_info_pinned: pinned int32[4] = {hold, seq, mirror
 * pointer lo, hi} that the chain-0 head graph's deer_begin_step reads; host_mirror: pinned int32[(1 + n_envs) * 64]. */
typedef struct deer_step_plan deer_step_plan;
int deer_step_plan_create(int n_chains, void* const* chain_head, void* const* chain_tail, int n_pieces, void* const* main_graphs,
                          void* const* head_graphs, const int* is_exit, int lookahead, int n_envs, int* step_info_pinned,
                          int* host_mirror, deer_step_plan** out);
/* returns 0, 2 (HIP error) or 3 (no verdict within 20 s); ctl_out: host int32[n_envs * 64]; pieces_out: NULL or trunk pieces submitted */
int deer_step_plan_run(deer_step_plan* p, int hold, int seq, void* main_stream, void* const* chain_streams, void* head_stream,
                       int* ctl_out, int* pieces_out);
void deer_step_plan_destroy(deer_step_plan* p);

/* ---- in-situ profiler: HIP events around every kernel-launching call of the spine (bench.py roofline pass) --------- */
int deer_prof_enable(deer_model* m, int on);                      /* host only; on = 1 clears the record list */
int deer_prof_count(const deer_model* m);
int deer_prof_get(deer_model* m, int i, char* name, int name_len, float* us, double* flops, double* bytes);  /* syncs the events */

#ifdef __cplusplus
}
#endif
#endif /* DEER_MODEL_H */
