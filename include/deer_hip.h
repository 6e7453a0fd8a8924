/* deer_hip.h - C ABI of libdeer_hip.so: the MI355X (gfx950) kernels behind DeeR-VLA's early-exit forward path.
 *
 * The reference (yueyang130/DeeR-VLA) has no C ABI: its "plugin interface" for this path is a Python object
 * protocol whose compute is implicit vendor kernels reached through torch ops.  Each entry
 * point below replaces one such implicit op and cites the reference call site it stands for (paths relative
 * to the reference repo).  The Python host in deer_vla_amd/ binds these with ctypes (deer_vla_amd/_abi.py)
 * and keeps the reference's Python surface (create_model_and_transforms / MPTFlamingo.forward / the
 * mosaic_gpt multi-exit lang_encoder.forward / ExitController) on top - see INTEGRATION.md.
 *
 * Conventions
 *   - all pointers are DEVICE pointers unless noted; `stream` is a hipStream_t passed as void*
 *   - bf16 data is passed as `const void*` (uint16 storage); f32 as float*
 *   - every function only ENQUEUES work on `stream` (graph-capturable, no allocation, no sync) and
 *     returns 0 on success, DEER_ERR_SHAPE (1) for an invalid argument, DEER_ERR_LAUNCH (2) if the launch failed
 *   - `ctl` is the device control block array (int32[DEER_CTL_WORDS] per environment); kernels on the early-exit path
 *     return at entry once ctl[DEER_CTL_ALL_EXITED] != 0 (device-side termination, no host round trip per layer); the
 *     exit checks also publish their verdict into a pinned host mirror (DEER_CTL_HOST_PTR) so that the host can stop
 *     enqueueing the remaining graph pieces of the step without the device ever waiting for it
 */
#ifndef DEER_HIP_H
#define DEER_HIP_H
#ifdef __cplusplus
extern "C" {
#endif

#define DEER_OK 0
#define DEER_ERR_SHAPE 1
#define DEER_ERR_LAUNCH 2

/* control block layout (32-bit words); one block per environment of the batch (env b at ctl + b*DEER_CTL_WORDS);
 * SHADOW and ALL_EXITED are batch-global (block 0); HOLD is per environment: deer_ctl_begin_step spreads bit b of step_info[0] into block b */
#define DEER_CTL_EXIT_FLAG 0
#define DEER_CTL_EXIT_LAYER 1
#define DEER_CTL_CUR_EXIT_ID 2
#define DEER_CTL_HOLD 3
#define DEER_CTL_N_EVALS 4
#define DEER_CTL_SHADOW 5      /* calibration: evaluate every exit, commit at the first that fires, never stop */
#define DEER_CTL_COMMITTED 6
#define DEER_CTL_ALL_EXITED 7  /* block 0: every environment of the batch has exited */
#define DEER_CTL_PREV_ACTION 8   /* float[8] */
#define DEER_CTL_OUT_ACTION 16   /* float[8]: pose[6], gripper prob, gripper logit */
#define DEER_CTL_DELTAS 24       /* float[16] */
#define DEER_CTL_N_EXITED 40   /* block 0: environments exited so far in this step */
#define DEER_CTL_SEQ 41        /* block 0: sequence number of the current control step (from step_info) */
#define DEER_CTL_HOST_PTR 42   /* block 0, 2 words: pinned host mirror the verdicts are published to, or 0 */
#define DEER_CTL_EVALS_DONE 44 /* block 0: environments that finished the current exit check */
#define DEER_CTL_PREV_REAL 45  /* PREV_ACTION is an exit-check action of this step (member of value_net.action_list), not the pseudo action */
#define DEER_CTL_ENS_ACTION 48 /* float[8]: ActionValueNet.get_ensemble_action (value_net.py:92-95): mean of the last two exit-check actions
                                * of this step - pose[6], gripper prob, number of actions averaged (1 or 2) */
/* host mirror (int32 words, pinned + system-coherent): [0] = seq*64 + exit checks completed this step, [1] = seq once every
 * environment exited, [64*(1+b), +64) = environment b's control block at its exit.  Lets the host stop enqueueing the
 * remaining graph segments of a step without the device ever waiting for the host. */
#define DEER_HOSTM_PROGRESS 0
#define DEER_HOSTM_DONE 1
#define DEER_CTL_WORDS 64

/* GEMM epilogues (deer_gemm_bf16_nt) */
#define DEER_EPI_BF16 0
#define DEER_EPI_F32 1
#define DEER_EPI_QGELU_BF16 2
#define DEER_EPI_GELU_BF16 3
#define DEER_EPI_RESADD_F32 4
#define DEER_EPI_BF16OUT 5   /* bf16 store whatever the operand format (deer_gemm_f16_nt: fp16 tower -> bf16 media K/V for the trunk's x-attn) */
/* skinny-GEMM A operand sources */
#define DEER_A_BF16 0
#define DEER_A_SLABS_GELU 1
#define DEER_A_SLABS 2
#define DEER_A_F32 3
/* head input modes / prologues / evaluation kinds / delta types */
#define DEER_X_RAW 0
#define DEER_X_POOL_MAX 1
#define DEER_X_POOL_AVG 2
#define DEER_X_LN 3
#define DEER_PRO_RAW 0
#define DEER_PRO_LN 1
#define DEER_PRO_GROUP_LN_RELU 2
#define DEER_PRO_GROUP_RELU 3
#define DEER_KIND_PSEUDO 0
#define DEER_KIND_CHECK 1
#define DEER_KIND_COMMIT 2
#define DEER_THR_L2 0
#define DEER_THR_MEAN 1
#define DEER_THR_MAX 2
#define DEER_THR_COSINE 3

/* ---- MFMA GEMM, large M:  C = epi(A[M,K] * W[N,K]^T + bias) -----------------------------------------------
 * Replaces the nn.Linear / nn.Conv2d(patch embed) calls inside open_clip's ViT-L/14 (called at
 * robot_flamingo/models/flamingo_mpt.py:580), PerceiverAttention/FeedForward (open_flamingo/src/helpers.py:47-52,
 * 15-22) and MaskedCrossAttention.to_kv (helpers.py:190).  A, W bf16; bias f32 or NULL; C bf16 or f32 per `epi`;
 * DEER_EPI_RESADD_F32: C(f32) += tanh(*gate or 1) * (A W^T + bias).  batch: A += z*strideA, C += z*strideC.
 * tile: 0 auto (the selector of csrc/gemm_tiled.hip: by M, N, K and the epilogue), 1 64x64, 2 64x128, 3 128x128; 4..79 name one kernel
 * instantiation each for tests and microbenchmarks (LDS-ring tiles 4-46, 16-wave 32-column rings 51-62, frame tiles = one 257-row camera
 * frame per row tile: 63-73 on 16 waves, 74 / 75 on 8, 76-79 on 4 waves (whole frames only, K % 64 == 0, no erf-GELU epilogue); a tile
 * that does not take the shape / epilogue returns DEER_ERR_SHAPE).  Needs K%8==0, N%16==0. */
int deer_gemm_bf16_nt(const void* A, int lda, long strideA, const void* W, int ldw, const float* bias, void* C, int ldc,
                      long strideC, int M, int N, int K, int batch, int epi, const float* gate, int tile, const int* ctl,
                      void* stream);

/* Batched form with one weight matrix per batch entry: C[z] = epi(A[z] W[z]^T) (Perceiver to_kv of every layer applied to
 * that layer's norm_media(x), helpers.py:47-55: the media tokens do not change across layers). */
int deer_gemm_bf16_nt_wbatch(const void* A, int lda, long strideA, const void* W, int ldw, long strideW, const float* bias,
                             void* C, int ldc, long strideC, int M, int N, int K, int batch, int epi, int tile, const int* ctl,
                             void* stream);
/* Split-K form of deer_gemm_bf16_nt for the latency-bound ViT / Perceiver projections that end in a residual add
 * (open_clip c_proj / out_proj, helpers.py:71 to_out, :22 FeedForward out): slab[s][M][N] (f32) = A[:, Ks] W[:, Ks]^T;
 * the consumer (deer_resadd_ln) sums the slabs, adds the bias and applies the residual + following LayerNorm. */
int deer_gemm_bf16_nt_splitk(const void* A, int lda, const void* W, int ldw, float* slab, int M, int N, int K, int splitk,
                             int tile, const int* ctl, void* stream);

/* ---- the vision tower's fp16 arithmetic (round 6) --------------------------------------------------------------------------------
 * The reference's evaluation runs fp32 weights under fp16 autocast (robot_flamingo/eval/eval_utils.py:333 `torch.cuda.amp.autocast(enabled=
 * self.amp)`, README.md:161-167 `--precision fp32 --amp 1`): every nn.Linear / conv of open_clip's ViT-L/14 and of the PerceiverResampler
 * computes on fp16 operands and returns fp16, LayerNorm / softmax / residual adds run in f32.  The same kernels as the bf16 family above,
 * instantiated on v_mfma_f32_16x16x32_f16 (same issue rate): A, W IEEE fp16; DEER_EPI_BF16 / QGELU_BF16 / GELU_BF16 store FP16 here,
 * DEER_EPI_BF16OUT stores bf16; f32 accumulation and f32 epilogue arithmetic as before.  Same arguments, same tile codes. */
int deer_gemm_f16_nt(const void* A, int lda, long strideA, const void* W, int ldw, const float* bias, void* C, int ldc,
                     long strideC, int M, int N, int K, int batch, int epi, const float* gate, int tile, const int* ctl,
                     void* stream);
int deer_gemm_f16_nt_wbatch(const void* A, int lda, long strideA, const void* W, int ldw, long strideW, const float* bias,
                            void* C, int ldc, long strideC, int M, int N, int K, int batch, int epi, int tile, const int* ctl,
                            void* stream);
int deer_gemm_f16_nt_splitk(const void* A, int lda, const void* W, int ldw, float* slab, int M, int N, int K, int splitk,
                            int tile, const int* ctl, void* stream);
/* deer_attn_mfma_hd64 / _2seg on fp16 q / k / v with an fp16 result (P rounded to fp16 for the P V MFMA; scores, softmax, O in f32) */
int deer_attn_f16_hd64(const void* Q, const void* K, const void* V, void* O, int batch, int heads, int q_len, int kv_len,
                       int ldq, int ldk, int ldv, int ldo, long q_bstride, long k_bstride, long v_bstride, long o_bstride,
                       float scale, void* stream);
int deer_attn_f16_hd64_2seg(const void* Q, const void* K1, const void* V1, const void* K2, const void* V2, void* O, int batch,
                            int heads, int q_len, int kv1, int kv2, int ldq, int ld1, int ld2, int ldo, long q_bstride, long bstride1,
                            long bstride2, long o_bstride, float scale, void* stream);
/* deer_layernorm_rows / _multi / deer_resadd_ln with the 16-bit LayerNorm output in fp16 (the A operand of deer_gemm_f16_nt) */
int deer_layernorm_rows_f16(const float* x, long in_rstride, long in_bstride, int rows_per_batch, int batch, const float* gamma,
                            const float* beta, void* out_f16, float* out_f32, long out_rstride, long out_bstride, int C, float eps,
                            void* stream);
int deer_layernorm_rows_multi_f16(const float* x, long in_rstride, long in_bstride, int rows_per_batch, int batch,
                                  const float* gamma, const float* beta, int n_sets, long param_stride, void* out_f16,
                                  long out_set_stride, long out_rstride, long out_bstride, int C, float eps, void* stream);
int deer_resadd_ln_f16(float* x, const float* slab, int s_in, long slab_stride, const float* gate, const float* bias,
                       const float* gamma, const float* beta, void* out_f16, float* out_f32, float* x_copy, int T, int d, float eps,
                       const int* ctl, void* stream);
/* im2col with fp16 patches; img_kind: 0 = f32 frames, 1 = bf16, 2 = fp16 (copied bit for bit) */
int deer_vit_im2col_f16(const void* img, int img_kind, int N, int S, int patch, void* out_f16, int Kpad, void* stream);

/* ---- MFMA GEMM, M <= 128 (weight-streaming): part[ks][Mpad][N] = A[:, Kslice ks] * W[:, Kslice ks]^T ------
 * Replaces the bias-free nn.Linear calls of the MPT GPTBlock (EXTERNAL; constructed mosaic_gpt_3b.py:104-106,
 * called :413-417) and of GatedCrossAttentionBlock (helpers.py:188,231,15-22) at T<=32 text tokens.
 * Wp = deer_pack_weight_mfma16(W).  Mpad = M rounded up to a multiple of 16.  a_mode: DEER_A_BF16 reads A (bf16 [M,lda]);
 * DEER_A_F32 reads A (f32 [M,lda]); DEER_A_SLABS(_GELU) reads sum_s Aslab[s*slab_stride_in + m*K + k] (optionally
 * through exact GELU).  f32 sources are fed to the MFMA as bf16 hi + bf16 lo (two MFMAs per weight fragment; free
 * on this HBM-bound kernel), so the activation keeps ~16 mantissa bits.
 * Partials are reduced by the consumer (deer_resadd_ln / deer_*_attn_small / next deer_gemm_skinny). */
int deer_gemm_skinny(const void* A, int lda, const float* Aslab, int s_in, long slab_stride_in, int a_mode, const void* Wp,
                     float* part, int M, int N, int K, int splitk, const int* ctl, void* stream);
int deer_skinny_splitk(int M, int N, int K);                       /* host helper: suggested split-K */
/* The same Linears for an ENV BATCH of 4-8 environments (49..128 rows), where MFMA time, LDS-read time and the weight stream are
 * the same order: the activation arrives pre-split as two bf16 planes A_hi = bf16(a), A_lo = bf16(a - A_hi), each [>= M rows, lda]
 * (written by deer_resadd_ln_split / deer_mpt_attn_small_hl / deer_slab_gelu_split), and is staged - together with the packed
 * weight fragments - by LDS-DMA into a ring (counted vmcnt + one raw barrier per 64 K-columns).  128-column workgroups:
 * splitk = deer_skinny_hl_splitk(M, N, K) slabs, half as many as deer_gemm_skinny.  K % (64 * splitk) == 0.  part[ks][Mpad][N];
 * rows M..Mpad-1 are written as zeros. */
int deer_gemm_skinny_hl(const void* A_hi, const void* A_lo, int lda, const void* Wp, float* part, int M, int N, int K, int splitk,
                        const int* ctl, void* stream);
int deer_skinny_hl_splitk(int M, int N, int K);
/* the same for up to 512 rows (8 environments x 32-token instructions, data.py:905-919): row blocks of <= 128 rows, one launch each,
 * part[ks][slab_rows][N] with slab_rows >= 16 * ceil(M / 16) */
int deer_gemm_skinny_hl_rows(const void* A_hi, const void* A_lo, int lda, const void* Wp, float* part, int M, int N, int K, int splitk,
                             int slab_rows, const int* ctl, void* stream);
/* hi / lo planes [rows, C] of act(sum_s slab[s]) (act = exact GELU when gelu != 0): the activation of the down-projections
 * (GELU between mlp_up / mlp_down of the MPT block and ff.1 / ff.3 of the gated x-attn block, helpers.py:15-22), computed ONCE
 * instead of once per column group of the consumer. */
int deer_slab_gelu_split(const float* slab, int s_in, long slab_stride, int gelu, void* out_hi, void* out_lo, int rows, int C,
                         const int* ctl, void* stream);
int deer_pack_weight_mfma16(const void* W, void* Wp, int N, int K, void* stream);  /* [N,K] -> [N/16][K/32][64][8] */

/* ---- attention ----------------------------------------------------------------------------------------------
 * deer_attn_mfma_hd64: softmax(scale * Q K^T) V per (batch, head), head_dim 64, kv_len <= 320 (<= 576 through a 36-key-tile instantiation of the same
 * kernel, one workgroup per CU: pre fusion's two-segment call over 2 x 256 patch tokens + 64 latents).  Replaces
 * nn.MultiheadAttention inside the open_clip ViT blocks and the einsum/softmax of PerceiverAttention
 * (helpers.py:53-63).  Q,K,V,O bf16 with row strides ld* and batch strides *_bstride (elements); head h lives at
 * column h*64. */
int deer_attn_mfma_hd64(const void* Q, const void* K, const void* V, void* O, int batch, int heads, int q_len, int kv_len,
                        int ldq, int ldk, int ldv, int ldo, long q_bstride, long k_bstride, long v_bstride, long o_bstride,
                        float scale, void* stream);
/* Same with the keys/values in two segments ([0,kv1) from K1/V1 with row stride ld1, [kv1,kv1+kv2) from K2/V2 with ld2):
 * PerceiverAttention's kv = [media ; latents] (helpers.py:51) without materialising the concatenation. */
int deer_attn_mfma_hd64_2seg(const void* Q, const void* K1, const void* V1, const void* K2, const void* V2, void* O, int batch,
                             int heads, int q_len, int kv1, int kv2, int ldq, int ld1, int ld2, int ldo, long q_bstride,
                             long bstride1, long bstride2, long o_bstride, float scale, void* stream);
/* deer_xattn_mfma: MaskedCrossAttention core (helpers.py:192-232) on the MFMA attention kernel: q = sum of f32 split-K slabs
 * [batch*T, ldqs], kv bf16 [batch*n_kv, ldkv] (k at col h*64, v at col inner+h*64), key j visible iff
 * text_time[t] == j/n_per_media + 1, rows with text_time == 0 zeroed; out f32 or bf16 [batch*T, ldo]. */
int deer_xattn_mfma(const float* qslab, int s_in, long slab_stride, int ldqs, const void* kv, int ldkv, int inner,
                    const int* text_time, int n_per_media, void* out, int out_is_f32, int ldo, int T, int n_kv, int heads,
                    int batch, float scale, const int* ctl, void* stream);
/* deer_xattn_fused: to_q -> masked cross-attention -> to_out of MaskedCrossAttention (helpers.py:184-233) as ONE launch.  xn: f32
 * [batch*T, d] (= attn.norm(x)); Wq_p / Wo_p: to_q [inner,d] / to_out [d,inner] packed by deer_pack_weight_mfma16; kv as above;
 * out: f32 [heads][slab_stride]: slab h = head h's contribution to y = Attn(x) for all rows (the consumer deer_resadd_ln sums
 * the `heads` slabs, s_in = heads).  T <= 32, n_kv <= 128, d % 128 == 0. */
int deer_xattn_fused(const float* xn, int d, const void* Wq_p, const void* kv, int ldkv, int inner, const int* text_time,
                     int n_per_media, int n_kv, const void* Wo_p, float* out, long slab_stride, int T, int heads, int batch,
                     float scale, const int* ctl, void* stream);
/* deer_xattn_small: the same op as fp32 VALU code (kept as a second implementation for cross-checks) (helpers.py:192-232): q from split-K slabs (x scale), kv bf16
 * [n_kv, ldkv] (k at col h*64, v at col inner+h*64), mask text_time[t] == j/n_per_media + 1, rows with
 * text_time == 0 zeroed; out bf16 or f32 [T, ldo]. */
int deer_xattn_small(const float* qslab, int s_in, long slab_stride, int ldqs, const void* kv, int ldkv, int inner,
                     const int* text_time, int n_per_media, void* out, int out_is_f32, int ldo, int T, int n_kv, int heads,
                     int batch, float scale, const int* ctl, void* stream);
/* deer_mpt_attn_small: MPT attention core (attn bias built at mosaic_gpt_3b.py:158-219): qkv from
 * split-K slabs [T,3d]; optional q/k LayerNorm over d_model (weights f32 or NULL); ALiBi slope
 * 2^(-alibi_bias_max*(h+1)/H); causal; key_mask (uint8[T], 0 = padded) or NULL; qkv_ws = f32 [T,3d] workspace;
 * out bf16 or f32 [T, ldo].  Two launches: slab reduce + q/k LayerNorm (one workgroup per row and q|k|v), then one
 * workgroup per head. */
int deer_mpt_attn_small(const float* qkvslab, int s_in, long slab_stride, int d_model, int n_heads, const float* q_ln_w,
                        const float* k_ln_w, float eps, const unsigned char* key_mask, float alibi_bias_max, float* qkv_ws,
                        void* out, int out_is_f32, int ldo, int T, int batch, const int* ctl, void* stream);
/* the same op, output as two bf16 planes hi = bf16(o), lo = bf16(o - hi), each [batch*T, ldo] (operand of deer_gemm_skinny_hl) */
int deer_mpt_attn_small_hl(const float* qkvslab, int s_in, long slab_stride, int d_model, int n_heads, const float* q_ln_w,
                           const float* k_ln_w, float eps, const unsigned char* key_mask, float alibi_bias_max, float* qkv_ws,
                           void* out_hi, void* out_lo, int ldo, int T, int batch, const int* ctl, void* stream);

/* ---- the trunk at ONE environment, <= 16 text rows (csrc/trunk_r16.hip) ----
 * deer_trunk_wide_gemm: the wide bias-free Linears of a layer (helpers.py:15-22 ff.1; MPT block, SURVEY App. B.1: Wqkv, mlp_up) with the K
 * split INSIDE the workgroup, so the result leaves final.  a_hi / a_lo: the LayerNorm output as bf16 planes in MFMA-fragment order
 * [K/32][64][8] (deer_resadd_ln_packed); Wp = packed [N, K] (deer_pack_weight_mfma16); N % 32 == 0, K = 256 or 2048, T <= 16.
 * epi 0: out_f32 [T, ldo]; 1: exact GELU (helpers.py:19 / the MPT MLP) -> row-major bf16 hi / lo planes [T, ldo] (operand of
 * deer_gemm_skinny_hl for ff.3 / mlp_down); 2: out_f32 + stats [N/32][16][2] = (mean, centred sum of squares) of every row over each
 * 32-column group (combined by deer_trunk_mpt_attn into the q / k LayerNorm over d_model). */
int deer_trunk_wide_gemm(const void* a_hi, const void* a_lo, const void* Wp, int N, int K, int epi, float* out_f32, void* out_hi, void* out_lo,
                         int ldo, float* stats, int T, const int* ctl, void* stream);
/* deer_trunk_mpt_attn: MPT attention core like deer_mpt_attn_small on FINAL f32 q|k|v [T, 3 d_model] (no slabs); the q / k LayerNorm over
 * d_model (attn_qk_ln, weights or NULL) from the 32-column moments `stats` of deer_trunk_wide_gemm(epi 2); out as row-major bf16 hi / lo
 * planes [T, ldo].  d_model % 256 == 0, T <= 16. */
int deer_trunk_mpt_attn(const float* qkv, const float* stats, int d_model, int n_heads, const float* q_ln_w, const float* k_ln_w, float eps,
                        const unsigned char* key_mask, float alibi_bias_max, void* out_hi, void* out_lo, int ldo, int T, const int* ctl,
                        void* stream);
/* deer_resadd_ln_split with the two planes in MFMA-fragment order [d/32][64 lanes][8], lane = 16 * (k % 32 / 8) + row (T <= 16): the operand
 * of deer_trunk_wide_gemm / deer_xattn_fused_packed reads back as one contiguous 1 KiB per k-tile and plane. */
int deer_resadd_ln_packed(float* x, const float* slab, int s_in, long slab_stride, const float* gate, const float* bias, const float* gamma,
                          const float* beta, void* out_hi, void* out_lo, float* out_f32, float* x_copy, int T, int d, float eps, const int* ctl,
                          void* stream);
/* deer_xattn_fused for ONE environment of <= 16 rows with LN(x) as fragment-ordered bf16 hi / lo planes (deer_resadd_ln_packed) */
int deer_xattn_fused_packed(const void* x_hi, const void* x_lo, int d, const void* Wq_p, const void* kv, int ldkv, int inner, const int* text_time,
                            int n_per_media, int n_kv, const void* Wo_p, float* out, long slab_stride, int T, int heads, float scale,
                            const int* ctl, void* stream);

/* N1 experiment (csrc/persistent_layer.hip): ONE gated x-attn + MPT decoder layer of the one-environment trunk (flamingo_lm.py:46-83) as ONE
 * persistent launch of 256 workgroups - the twelve phases are the device functions of the kernels above, a device-wide barrier at every
 * seam.  Bit-identical to the twelve-launch layer; slower; selected by DEER_PERSISTENT_LAYER=1 only.  All buffers as the
 * spine lays them out: x f32 [T, d] (in / out), slab_a f32 (>= 16 slabs of [16, d]), qkv f32 [T, 3d], stats [3d/32][16][2], xn / h / ao
 * planes (xn in MFMA-fragment order), kv bf16 media K|V of this layer, weights packed (deer_pack_weight_mfma16).  pending_s > 0: the previous
 * layer's mlp_down slabs (in slab_a) are folded in first; hidden_out != NULL: x += this layer's mlp_down slabs and the result is copied out
 * (hidden_states[i]); else they stay pending in slab_a (s_down slabs).  barrier: 192 uint32 (zero before the FIRST launch; cumulative after), error: int32, raised when a
 * barrier times out. */
typedef struct deer_trunk_layer_args {
  int T, d, xinner, heads, n_heads, ffw, n_kv, n_per_media, ld_kv, NS, pending_s, qk_ln, s_w2, s_wo, s_down;
  float eps, xattn_scale, alibi_bias_max;
  float *x, *slab_a, *qkv, *stats, *prev_hidden, *hidden_out;
  void *xn_hi, *xn_lo, *h_hi, *h_lo, *ao_hi, *ao_lo;
  const void* kv;
  const int* text_time;
  const unsigned char* key_mask;
  const float *x_nw, *x_nb, *x_ag, *x_fnw, *x_fnb, *x_fg, *ln1w, *ln1b, *ln2w, *ln2b, *qlnw, *klnw;
  const void *x_wq, *x_wo, *x_w1, *x_w2, *wqkv, *wo, *wup, *wdown;
  unsigned* barrier;
  int* error;
  int* trace;   /* NULL or 16 rows x 4 int32: row e = {clock lo, clock hi (100 MHz), workgroup} of the LAST arrival at barrier e, row 0 = entry of workgroup 0; [3] = timeouts at e */
} deer_trunk_layer_args;
int deer_trunk_layer_persistent(const deer_trunk_layer_args* a, void* stream);

/* deer_resadd_ln for the env-batch vision tower (d == 1024, bf16 LayerNorm output, no control block): rows_per_wg = 2 or 4 rows per workgroup,
 * all loads of all rows in flight before the first use, one pair of block reductions for the R rows.  Per row bit-identical to deer_resadd_ln,
 * which selects it from 2048 rows on (open_clip ResidualAttentionBlock residual + ln_1 / ln_2 of the next op). */
int deer_resadd_ln_multirow(float* x, const float* slab, int s_in, long slab_stride, const float* gate, const float* bias, const float* gamma,
                            const float* beta, void* out_bf16, int T, int d, float eps, int rows_per_wg, void* stream);

/* ---- env batch with COMPACTION of exited environments (the reference stops every environment at its own layer,
 * mosaic_gpt_3b.py:438-443): the rows of the still-active environments stay packed at the front of the trunk's buffers.  Row map `cmap`
 * (int32, CMAP_WORDS = 64 per copy): [0] = active slots, [1 + s] = environment of slot s, [17 + e] = slot of environment e or -1 (up to 16 environments).  The
 * *_active / *_rows entry points are the kernels above restricted to the active slots (rows_per_env rows each); per-environment inputs
 * (media K/V, text_time, key mask, control blocks) stay in environment order and are reached through the map. ---- */
int deer_resadd_ln_rows(float* x, const float* slab, int s_in, long slab_stride, const float* gate, const float* gamma, const float* beta, void* out_bf16,
                        void* out_lo, float* out_f32, float* x_copy, int T_rows, int d, float eps, const int* ctl, const int* cmap, int rows_per_env,
                        const float* x_in, const int* cmap_old, int B, int drop_upto, void* stream);   /* x_in != NULL: gather the surviving rows (first row op of a compaction layer: environments that exited at a layer <= drop_upto leave), publish the new map into cmap */
int deer_gemm_skinny_hl_active(const void* A_hi, const void* A_lo, int lda, const void* Wp, float* part, int M, int N, int K, int splitk, int slab_rows,
                               const int* ctl, const int* cmap, int rows_per_env, void* stream);
int deer_slab_gelu_split_active(const float* slab, int s_in, long slab_stride, int gelu, void* out_hi, void* out_lo, int rows, int C, const int* ctl,
                                const int* cmap, int rows_per_env, void* stream);
int deer_mpt_attn_small_hl_active(const float* qkvslab, int s_in, long slab_stride, int d_model, int n_heads, const float* q_ln_w, const float* k_ln_w,
                                  float eps, const unsigned char* key_mask, float alibi_bias_max, float* qkv_ws, void* out_hi, void* out_lo, int ldo, int T,
                                  int batch, const int* ctl, const int* cmap, void* stream);
int deer_xattn_fused_active(const float* xn, int d, const void* Wq_p, const void* kv, int ldkv, int inner, const int* text_time, int n_per_media, int n_kv,
                            const void* Wo_p, float* out, long slab_stride, int T, int heads, int batch, float scale, const int* ctl, const int* cmap,
                            void* stream);
int deer_head_pool_active(const float* feats, float* pooled, int T, int d, int avg, int B, const unsigned char* key_mask, const int* ctl, int kind,
                          int layer, const int* cmap, void* stream);
int deer_ctl_begin_step_map(int* ctl, const int* step_info, int B, int* cmap, void* stream);   /* deer_ctl_begin_step + both map copies := identity */

/* (deer_head_*: w_is_f32 = 1 when the weight pointers are f32 - the fp32 arithmetic keeps the head's weights in f32) */

/* ---- fp32-activation arithmetic (csrc/precise.hip; deer_config.precision = 1): the second arithmetic of the path, for
 * north_star's "1e-3 fp32" clause.  Activations AND weights stay f32 end to end (the arena keeps f32 / hi+lo copies in this mode). */
/* every nn.Linear / conv of the ViT, the Perceiver and the media K/V projection: C f32 [M,N] (op)= A f32 [M,K] * W[N,K]^T + bias with
 * the exact-f32 MFMA.  epi: 0 store, 1 QuickGELU, 2 exact GELU, 3 C += (residual add).  K % 4 == 0, N % 4 == 0. */
int deer_gemm_f32_nt(const float* A, int lda, const float* W, int ldw, const float* bias, float* C, int ldc, int M, int N, int K,
                     int epi, void* stream);
/* open_clip MHA core / PerceiverAttention core (helpers.py:47-73) in fp32: q [batch][q_len][ldq] (head h at column h*64), keys and
 * values in one or two segments ([batch][kv_s][ld_s], pointers at the K / V column offset), out f32 [batch][q_len][ldo].
 * head_dim 64, kv1 + kv2 <= 352 with the keys of a head resident in LDS; up to ~700 keys (pre fusion: 576) through a chunked form. */
int deer_attn_f32(const float* Q, const float* K1, const float* V1, const float* K2, const float* V2, float* O, int batch, int heads,
                  int q_len, int kv1, int kv2, int ldq, int ld1, int ld2, int ldo, long q_bstride, long bstride1, long bstride2,
                  long o_bstride, float scale, void* stream);
/* MaskedCrossAttention core (helpers.py:192-232) with f32 K/V: q from split-K slabs, kv f32 [batch][n_kv][ldkv] (k at column h*64, v at
 * inner + h*64), media-time mask from text_time, out f32 [batch][T][ldo]. */
int deer_xattn_f32(const float* qslab, int s_in, long slab_stride, int ldqs, const float* kv, int ldkv, int inner, const int* text_time,
                   int n_per_media, float* out, int ldo, int T, int n_kv, int heads, int batch, float scale, const int* ctl, void* stream);

/* deer_embed_tokens with an f32 table */
int deer_embed_tokens_f32(const long long* ids, const float* wte, float* x, int* text_time, int T, int batch, int d, int vocab,
                          int media_id, void* stream);
/* im2col of the camera frames in f32 (deer_vit_im2col keeps bf16): out f32 [N*P, Kpad] */
int deer_vit_im2col_f32(const float* img, int N, int S, int patch, float* out, int Kpad, void* stream);

/* ---- row ops ------------------------------------------------------------------------------------------------
 * deer_layernorm_rows: nn.LayerNorm (ViT ln_1/ln_2, helpers.py:32-33,17,132), f32 in, bf16 and/or f32 out. */
int deer_layernorm_rows(const float* x, long in_rstride, long in_bstride, int rows_per_batch, int batch, const float* gamma,
                        const float* beta, void* out_bf16, float* out_f32, long out_rstride, long out_bstride, int C,
                        float eps, void* stream);
/* deer_layernorm_rows_multi: one pass of statistics, n_sets affine outputs out[s] = LN(x)*gamma[s]+beta[s] (Perceiver
 * norm_media of all layers on the same media tokens, helpers.py:32,47). */
int deer_layernorm_rows_multi(const float* x, long in_rstride, long in_bstride, int rows_per_batch, int batch,
                              const float* gamma, const float* beta, int n_sets, long param_stride, void* out_bf16,
                              long out_set_stride, long out_rstride, long out_bstride, int C, float eps, void* stream);
/* deer_resadd_ln: x += tanh(*gate or 1) * (sum_s slab[s] + bias or 0); optional copy of x (hidden_states[i], mosaic_gpt_3b.py:424-427);
 * optional LayerNorm -> bf16 and/or f32 (helpers.py:267-279 gated residuals; MPT block residuals + ln_1/ln_2). */
int deer_resadd_ln(float* x, const float* slab, int s_in, long slab_stride, const float* gate, const float* bias,
                   const float* gamma,
                   const float* beta, void* out_bf16, float* out_f32, float* x_copy, int T, int d, float eps, const int* ctl,
                   void* stream);
/* deer_resadd_ln with the LayerNorm output ALSO as two bf16 planes hi = bf16(y), lo = bf16(y - hi) (operand of deer_gemm_skinny_hl);
 * out_f32 optional. */
int deer_resadd_ln_split(float* x, const float* slab, int s_in, long slab_stride, const float* gate, const float* bias,
                         const float* gamma, const float* beta, void* out_hi, void* out_lo, float* out_f32, float* x_copy, int T,
                         int d, float eps, const int* ctl, void* stream);
/* ViT patch embedding (open_clip conv1 + class/positional embedding + ln_pre) */
int deer_vit_im2col(const void* img, int img_is_bf16, int N, int S, int patch, void* out_bf16, int Kpad, void* stream);
int deer_vit_embed_lnpre(const float* patch, const float* cls, const float* pos, const float* ln_w, const float* ln_b,
                         float* x, int N, int P, int W, float eps, void* stream);
/* wte lookup (mosaic_gpt_3b.py:341) + text_time = cumsum(ids == media_token_id) (flamingo_lm.py:211, helpers.py:208) */
int deer_embed_tokens(const long long* ids, const void* wte_bf16, float* x, int* text_time, int T, int batch, int d, int vocab,
                      int media_id, void* stream);
int deer_broadcast_rows(const float* src, float* dst, long n, int batch, void* stream);   /* helpers.py:128 */

/* ---- action head + exit gate (robot_flamingo/models/action_head.py:499-611, value_net.py:105-133,277-297) -----
 * All three evaluate a BATCH of B <= 16 independent environments (weights read once per 8 environments: batches above 8 run in two halves): features [B][T][d],
 * LSTM state tensors [L][B][H], control blocks ctl + b*DEER_CTL_WORDS, exit decision per environment. */
int deer_head_pool(const float* feats, float* pooled, int T, int d, int avg, int B, const unsigned char* key_mask, const int* ctl,
                   int kind, int layer, void* stream);
/* DeterministicDecoder(use_state=True) (action_head.py:443-453,524-536): deer_head_state_embed computes, per environment, the embedding
 * W_state [relu(W_arm arm + b_arm) ; relu(E_grip[(g + 1) / 2])] + b_state of the robot state (state [B][8] f32: arm pose robot_obs[:6],
 * gripper opening robot_obs[-1], pad); deer_head_pool_state is deer_head_pool with that embedding added to the pooled feature. */
int deer_head_state_embed(const float* state, const float* w_arm, const float* b_arm, const float* e_grip, const void* w_state,
                          const float* b_state, float* out, int d, int B, int w_is_f32, void* stream);
int deer_head_pool_state(const float* feats, float* pooled, int T, int d, int avg, int B, const unsigned char* key_mask,
                         const float* state_emb, const int* ctl, int kind, int layer, void* stream);   /* AdaptiveMax/AvgPool1d over the text tokens (action_head.py:480-483,
                   519-520); key_mask (uint8 [B][T], 0 = right-padding of an env batch, data.py:905-919) or NULL */
int deer_head_lstm_layer(const float* x_src, long x_bstride, int x_mode, int T, int in_dim, const float* ln_w, const float* ln_b,
                         const void* w_ih, const void* w_hh, const float* b_ih, const float* b_hh, const float* h_prev,
                         const float* c_prev, float* h_out, float* c_out, int H, int B, float eps, const int* ctl, int kind,
                         int layer, int w_is_f32, void* stream);
/* The recurrent half of all LSTM layers once per control step: ghh[l][b][q*H + j] = W_hh^l[q*H + j, :] . h_state[l][b] + b_hh^l (L <= 8 layers in
 * one launch; h_state [L][B][H]).  Every head evaluation of a step (pseudo / exit checks / committing call) starts from the state the previous
 * step committed (action_head.py:560-575, update_hidden_state=False until the committing call), so this half is common to all of them. */
int deer_head_lstm_hh(const void* const* w_hh, const float* const* b_hh, int L, const float* h_state, float* ghh, int H, int B, int w_is_f32,
                      void* stream);
/* deer_head_lstm_layer with its recurrent half taken from ghh ([B][4H] of this layer): streams W_ih only. */
int deer_head_lstm_layer_pre(const float* x_src, long x_bstride, int x_mode, int T, int in_dim, const float* ln_w, const float* ln_b,
                             const void* w_ih, const float* b_ih, const float* ghh, const float* c_prev, float* h_out, float* c_out, int H, int B,
                             float eps, const int* ctl, int kind, int layer, int w_is_f32, void* stream);
int deer_head_fc(const float* src, int src_stride, int in_dim, int pro, const float* lnw0, const float* lnb0, const float* lnw1,
                 const float* lnb1, const void* W0, const float* b0, const void* W1, const float* b1, int out_dim, float* dst,
                 int B, float eps, const int* ctl, int kind, int layer, int w_is_f32, void* stream);
int deer_head_final(const float* src, int src_stride, int in_dim, int pro, const float* lnw0, const float* lnb0, const float* lnw1,
                    const float* lnb1, const void* Wa, const float* ba, const void* Wg, const float* bg, int* ctl, int kind,
                    int layer, int slot, const float* thresholds, int force, int thr_type, int leq, const float* h_tmp,
                    const float* c_tmp, float* h_state, float* c_state, int L, int H, int B, float* action_dbg, float eps,
                    int w_is_f32, void* stream);
/* the same with multi_step_action = A (action_head.py:458,472-473: Wa [6 A, in_dim], Wg [A, in_dim]; the criterion's delta over all 6 A pose
 * values; ModelWrapper executes the first multi_execution of the A actions, eval_utils.py:466-475).  action_dbg: [B][64] f32 = pose 6 A |
 * gripper A | logit A.  A > 1 with a control block: act_ext [B][4][64] f32 holds the previous / committed / ensemble action (the 8-float
 * fields of the control block then carry the FIRST action), and a host mirror carries [B][128] f32 (committed | ensemble) behind its
 * (1 + B) * 64 words. */
int deer_head_final_multi(const float* src, int src_stride, int in_dim, int pro, const float* lnw0, const float* lnb0, const float* lnw1,
                          const float* lnb1, const void* Wa, const float* ba, const void* Wg, const float* bg, int* ctl, int kind,
                          int layer, int slot, const float* thresholds, int force, int thr_type, int leq, const float* h_tmp,
                          const float* c_tmp, float* h_state, float* c_state, int L, int H, int B, float* action_dbg, float eps,
                          int w_is_f32, int multi_step_action, float* act_ext, void* stream);
/* ---- one head evaluation as ONE launch (csrc/head.hip: head_fused_kernel; replaces the eight launches above on control steps of one
 * environment: pool -> L <= 4 LSTM layers (recurrent half from deer_head_lstm_hh) -> <= 3 hidden Linears -> output Linear + exit gate).
 * G workgroups stay resident; the vectors between the phases travel as data-tagged 8-byte granules (relaxed agent-scope atomics), the
 * tag = number of evaluations the exchange buffer has served.  xg: deer_head_fused_granules() * 8 bytes, zeroed once; err: int32[2]
 * zeroed once = {raised when a hand-off timed out, evaluation counter}.  Returns DEER_ERR_SHAPE for what it does not take (no control block, COMMIT kind, f32 weights,
 * more than one environment, d or H > 2048): the caller then uses the separate kernels.  Same arithmetic per row as those kernels. */
typedef struct deer_head_fused_args {
  const float* feats; int T, d, avg; const unsigned char* key_mask; const int* cmap;
  int B, H, L, n_fc, lstm_ln, mlp_ln;
  int fc_dim[3];
  const void* w_ih[4]; const float* b_ih[4]; const float* ln_w[4]; const float* ln_b[4];   /* ln_*[l]: LayerNorm of layer l's OUTPUT (lstm_ln) */
  const float* ghh; const float* c_prev; float* h_tmp; float* c_tmp;
  const void* fw[3][2]; const float* fb[3][2]; const float* fln_w[3][2]; const float* fln_b[3][2];
  const void* Wa; const float* ba; const void* Wg; const float* bg;
  int* ctl; int kind, layer, slot; const float* thresholds; int force, thr_type, leq;
  float* h_state; float* c_state; float* action_dbg; float eps; int A; float* act_ext;
  unsigned long long* xg; int* err; int max_in, gather_waves;                               /* max_in, gather_waves: filled by the launcher */
  unsigned long long* trace;                                                                /* NULL, or 64 x u64: 100 MHz time stamps of workgroup 0 at the phase boundaries (tools/bench_head_eval.py) */
} deer_head_fused_args;
long deer_head_fused_granules(int B, int d, int H, int L, int n_fc, const int* fc_dim);
int deer_head_fused(const deer_head_fused_args* args, int w_is_f32, int n_workgroups, void* stream);
/* ExitController.set_timestep (eval_utils.py:662-663) + per-step reset.  step_info: device int32[4] = {hold, step sequence
 * number, host mirror pointer lo, hi} or NULL (no stage hold, no mirror). */
int deer_ctl_begin_step(int* ctl, const int* step_info, int B, void* stream);

/* ---- camera-frame preprocessing (robot_flamingo/data/data.py:898-902 + open_clip eval transform, factory.py:109-112) ----------
 * uint8 frames [N][H][W][3] -> Resize(S, bicubic, shorter side) -> CenterCrop(S) -> /255 -> Normalize(mean, std) as [N][3][S][S] bf16
 * and/or f32, bit-exact w.r.t. PIL's two-pass fixed-point resampler (the reference runs PIL on the host).  tmp: device scratch of
 * deer_preprocess_scratch_bytes(N, H, W, S) bytes; mean / std_: HOST float[3]. */
int deer_preprocess_frames(const unsigned char* src, int N, int H, int W, int S, const float* mean, const float* std_,
                           unsigned char* tmp, void* out_bf16, float* out_f32, void* stream);
/* deer_preprocess_frames with the 16-bit output in IEEE fp16 (the frame format of a fp16 engine) */
int deer_preprocess_frames_f16(const unsigned char* src, int N, int H, int W, int S, const float* mean, const float* std_,
                           unsigned char* tmp, void* out_bf16, float* out_f32, void* stream);
long deer_preprocess_scratch_bytes(int N, int H, int W, int S);

/* keeps `stream` busy for ~us microseconds (profiling aid: lets the host enqueue ahead of the GPU) */
int deer_spin_us(int us, void* stream);

/* library identification: returns the gfx arch string the kernels were compiled for ("gfx950") */
const char* deer_hip_arch(void);
int deer_hip_abi_version(void);

/* ---- the LLM trunk and the head on IEEE fp16 operands (round 6) ------------------------------------------------------------------------
 * The reference's evaluation arithmetic is fp32 weights under fp16 autocast (robot_flamingo/eval/eval_utils.py:333, README.md:161-167):
 * every nn.Linear of the MPT blocks (mosaic_gpt_3b.py:413-417), of GatedCrossAttentionBlock (helpers.py:188,231,15-22) and of the action
 * head multiplies fp16-ROUNDED weights.  bf16-rounded weights are 2.6e-2 from that on the action at the full 3B size, fp16-rounded ones
 * 9e-4 (tools/amp_difference.py full --parts, profiles/r06_*).  Every trunk entry point above therefore has a twin that reads weights
 * packed as fp16 (deer_model_load_tensor with deer_config.operands_f16) and activation planes fp16 hi / lo on v_mfma_f32_16x16x32_f16 -
 * same arguments, same kernels, same speed (the trunk is HBM-bound; same bytes).  K / V of the gated x-attn arrive as fp16
 * (deer_gemm_f16_nt with DEER_EPI_BF16 = "the family's 16-bit format").  The head entry points take the weight kind in their
 * `w_is_f32` argument: 0 = bf16 rows, 1 = f32 rows, 2 = fp16 rows. */
int deer_gemm_skinny_f16(const void* A, int lda, const float* Aslab, int s_in, long slab_stride_in, int a_mode, const void* Wp,
                     float* part, int M, int N, int K, int splitk, const int* ctl, void* stream);
int deer_gemm_skinny_hl_f16(const void* A_hi, const void* A_lo, int lda, const void* Wp, float* part, int M, int N, int K, int splitk,
                        const int* ctl, void* stream);
int deer_gemm_skinny_hl_rows_f16(const void* A_hi, const void* A_lo, int lda, const void* Wp, float* part, int M, int N, int K, int splitk,
                             int slab_rows, const int* ctl, void* stream);
int deer_gemm_skinny_hl_active_f16(const void* A_hi, const void* A_lo, int lda, const void* Wp, float* part, int M, int N, int K, int splitk, int slab_rows,
                               const int* ctl, const int* cmap, int rows_per_env, void* stream);
int deer_slab_gelu_split_f16(const float* slab, int s_in, long slab_stride, int gelu, void* out_hi, void* out_lo, int rows, int C,
                         const int* ctl, void* stream);
int deer_slab_gelu_split_active_f16(const float* slab, int s_in, long slab_stride, int gelu, void* out_hi, void* out_lo, int rows, int C, const int* ctl,
                                const int* cmap, int rows_per_env, void* stream);
int deer_resadd_ln_split_f16(float* x, const float* slab, int s_in, long slab_stride, const float* gate, const float* bias,
                         const float* gamma, const float* beta, void* out_hi, void* out_lo, float* out_f32, float* x_copy, int T,
                         int d, float eps, const int* ctl, void* stream);
int deer_resadd_ln_rows_f16(float* x, const float* slab, int s_in, long slab_stride, const float* gate, const float* gamma, const float* beta, void* out_bf16,
                        void* out_lo, float* out_f32, float* x_copy, int T_rows, int d, float eps, const int* ctl, const int* cmap, int rows_per_env,
                        const float* x_in, const int* cmap_old, int B, int drop_upto, void* stream);
int deer_resadd_ln_packed_f16(float* x, const float* slab, int s_in, long slab_stride, const float* gate, const float* bias, const float* gamma,
                          const float* beta, void* out_hi, void* out_lo, float* out_f32, float* x_copy, int T, int d, float eps, const int* ctl,
                          void* stream);
int deer_trunk_wide_gemm_f16(const void* a_hi, const void* a_lo, const void* Wp, int N, int K, int epi, float* out_f32, void* out_hi, void* out_lo,
                         int ldo, float* stats, int T, const int* ctl, void* stream);
int deer_trunk_mpt_attn_f16(const float* qkv, const float* stats, int d_model, int n_heads, const float* q_ln_w, const float* k_ln_w, float eps,
                        const unsigned char* key_mask, float alibi_bias_max, void* out_hi, void* out_lo, int ldo, int T, const int* ctl,
                        void* stream);
int deer_xattn_fused_f16(const float* xn, int d, const void* Wq_p, const void* kv, int ldkv, int inner, const int* text_time,
                     int n_per_media, int n_kv, const void* Wo_p, float* out, long slab_stride, int T, int heads, int batch,
                     float scale, const int* ctl, void* stream);
int deer_xattn_fused_active_f16(const float* xn, int d, const void* Wq_p, const void* kv, int ldkv, int inner, const int* text_time, int n_per_media, int n_kv,
                            const void* Wo_p, float* out, long slab_stride, int T, int heads, int batch, float scale, const int* ctl, const int* cmap,
                            void* stream);
int deer_xattn_fused_packed_f16(const void* x_hi, const void* x_lo, int d, const void* Wq_p, const void* kv, int ldkv, int inner, const int* text_time,
                            int n_per_media, int n_kv, const void* Wo_p, float* out, long slab_stride, int T, int heads, float scale,
                            const int* ctl, void* stream);
int deer_xattn_mfma_f16(const float* qslab, int s_in, long slab_stride, int ldqs, const void* kv, int ldkv, int inner,
                    const int* text_time, int n_per_media, void* out, int out_is_f32, int ldo, int T, int n_kv, int heads,
                    int batch, float scale, const int* ctl, void* stream);
int deer_mpt_attn_small_hl_f16(const float* qkvslab, int s_in, long slab_stride, int d_model, int n_heads, const float* q_ln_w,
                           const float* k_ln_w, float eps, const unsigned char* key_mask, float alibi_bias_max, float* qkv_ws,
                           void* out_hi, void* out_lo, int ldo, int T, int batch, const int* ctl, void* stream);
int deer_mpt_attn_small_hl_active_f16(const float* qkvslab, int s_in, long slab_stride, int d_model, int n_heads, const float* q_ln_w, const float* k_ln_w,
                                  float eps, const unsigned char* key_mask, float alibi_bias_max, float* qkv_ws, void* out_hi, void* out_lo, int ldo, int T,
                                  int batch, const int* ctl, const int* cmap, void* stream);
int deer_embed_tokens_f16(const long long* ids, const void* wte_bf16, float* x, int* text_time, int T, int batch, int d, int vocab,
                      int media_id, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DEER_HIP_H */
