"""ctypes binding of libdeer_hip.so (C ABI declared in include/deer_hip.h).

The product path has NO CPU fallback: if the HIP library is missing or fails to load, ``lib()`` raises.
``build()`` (re)compiles it in-tree with hipcc for gfx950 (works without a GPU; used by
``__graft_entry__.build()``).
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from ctypes import c_char_p, c_float, c_int, c_long, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libdeer_hip.so")
if os.environ.get("DEER_HIP_LIB"):       # tools only (tools/ktrace_trunk.py: the same library built with phase time stamps, `make ktrace`)
    LIB_PATH = os.path.join(_HERE, "lib", os.environ["DEER_HIP_LIB"])
CSRC = os.path.join(_HERE, "csrc")

P, I, L, F = c_void_p, c_int, c_long, c_float

# name -> argtypes (restype is int unless listed in _RESTYPE); mirrors include/deer_hip.h one to one
SIGNATURES = {
    "deer_gemm_bf16_nt": [P, I, L, P, I, P, P, I, L, I, I, I, I, I, P, I, P, P],
    "deer_gemm_bf16_nt_wbatch": [P, I, L, P, I, L, P, P, I, L, I, I, I, I, I, I, P, P],
    "deer_gemm_bf16_nt_splitk": [P, I, P, I, P, I, I, I, I, I, P, P],
    # round 6: the vision tower's fp16 arithmetic (same arguments as the bf16 family)
    "deer_gemm_f16_nt": [P, I, L, P, I, P, P, I, L, I, I, I, I, I, P, I, P, P],
    "deer_gemm_f16_nt_wbatch": [P, I, L, P, I, L, P, P, I, L, I, I, I, I, I, I, P, P],
    "deer_gemm_f16_nt_splitk": [P, I, P, I, P, I, I, I, I, I, P, P],
    "deer_attn_f16_hd64": [P, P, P, P, I, I, I, I, I, I, I, I, L, L, L, L, F, P],
    "deer_attn_f16_hd64_2seg": [P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, L, L, L, L, F, P],
    "deer_layernorm_rows_f16": [P, L, L, I, I, P, P, P, P, L, L, I, F, P],
    "deer_layernorm_rows_multi_f16": [P, L, L, I, I, P, P, I, L, P, L, L, L, I, F, P],
    "deer_resadd_ln_f16": [P, P, I, L, P, P, P, P, P, P, P, I, I, F, P, P],
    "deer_vit_im2col_f16": [P, I, I, I, I, P, I, P],
    # round 6: the trunk on fp16 operands (twins of the bf16 entry points, same arguments)
    "deer_gemm_skinny_f16": [P, I, P, I, L, I, P, P, I, I, I, I, P, P],
    "deer_gemm_skinny_hl_f16": [P, P, I, P, P, I, I, I, I, P, P],
    "deer_gemm_skinny_hl_rows_f16": [P, P, I, P, P, I, I, I, I, I, P, P],
    "deer_gemm_skinny_hl_active_f16": [P, P, I, P, P, I, I, I, I, I, P, P, I, P],
    "deer_slab_gelu_split_f16": [P, I, L, I, P, P, I, I, P, P],
    "deer_slab_gelu_split_active_f16": [P, I, L, I, P, P, I, I, P, P, I, P],
    "deer_resadd_ln_split_f16": [P, P, I, L, P, P, P, P, P, P, P, P, I, I, F, P, P],
    "deer_resadd_ln_rows_f16": [P, P, I, L, P, P, P, P, P, P, P, I, I, F, P, P, I, P, P, I, I, P],
    "deer_resadd_ln_packed_f16": [P, P, I, L, P, P, P, P, P, P, P, P, I, I, F, P, P],
    "deer_trunk_wide_gemm_f16": [P, P, P, I, I, I, P, P, P, I, P, I, P, P],
    "deer_trunk_mpt_attn_f16": [P, P, I, I, P, P, F, P, F, P, P, I, I, P, P],
    "deer_xattn_fused_f16": [P, I, P, P, I, I, P, I, I, P, P, L, I, I, I, F, P, P],
    "deer_xattn_fused_active_f16": [P, I, P, P, I, I, P, I, I, P, P, L, I, I, I, F, P, P, P],
    "deer_xattn_fused_packed_f16": [P, P, I, P, P, I, I, P, I, I, P, P, L, I, I, F, P, P],
    "deer_xattn_mfma_f16": [P, I, L, I, P, I, I, P, I, P, I, I, I, I, I, I, F, P, P],
    "deer_mpt_attn_small_hl_f16": [P, I, L, I, I, P, P, F, P, F, P, P, P, I, I, I, P, P],
    "deer_mpt_attn_small_hl_active_f16": [P, I, L, I, I, P, P, F, P, F, P, P, P, I, I, I, P, P, P],
    "deer_embed_tokens_f16": [P, P, P, P, I, I, I, I, I, P],
    "deer_gemm_skinny": [P, I, P, I, L, I, P, P, I, I, I, I, P, P],
    "deer_skinny_splitk": [I, I, I],
    "deer_gemm_skinny_hl": [P, P, I, P, P, I, I, I, I, P, P],
    "deer_skinny_hl_splitk": [I, I, I],
    "deer_gemm_skinny_hl_rows": [P, P, I, P, P, I, I, I, I, I, P, P],
    "deer_slab_gelu_split": [P, I, L, I, P, P, I, I, P, P],
    "deer_pack_weight_mfma16": [P, P, I, I, P],
    "deer_attn_mfma_hd64": [P, P, P, P, I, I, I, I, I, I, I, I, L, L, L, L, F, P],
    "deer_attn_mfma_hd64_2seg": [P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, L, L, L, L, F, P],
    "deer_xattn_mfma": [P, I, L, I, P, I, I, P, I, P, I, I, I, I, I, I, F, P, P],
    "deer_xattn_fused": [P, I, P, P, I, I, P, I, I, P, P, L, I, I, I, F, P, P],
    "deer_xattn_small": [P, I, L, I, P, I, I, P, I, P, I, I, I, I, I, I, F, P, P],
    "deer_mpt_attn_small": [P, I, L, I, I, P, P, F, P, F, P, P, I, I, I, I, P, P],
    "deer_mpt_attn_small_hl": [P, I, L, I, I, P, P, F, P, F, P, P, P, I, I, I, P, P],
    "deer_trunk_wide_gemm": [P, P, P, I, I, I, P, P, P, I, P, I, P, P],
    "deer_resadd_ln_packed": [P, P, I, L, P, P, P, P, P, P, P, P, I, I, F, P, P],
    "deer_xattn_fused_packed": [P, P, I, P, P, I, I, P, I, I, P, P, L, I, I, F, P, P],
    "deer_trunk_mpt_attn": [P, P, I, I, P, P, F, P, F, P, P, I, I, P, P],
    "deer_resadd_ln_rows": [P, P, I, L, P, P, P, P, P, P, P, I, I, F, P, P, I, P, P, I, I, P],
    "deer_gemm_skinny_hl_active": [P, P, I, P, P, I, I, I, I, I, P, P, I, P],
    "deer_slab_gelu_split_active": [P, I, L, I, P, P, I, I, P, P, I, P],
    "deer_mpt_attn_small_hl_active": [P, I, L, I, I, P, P, F, P, F, P, P, P, I, I, I, P, P, P],
    "deer_xattn_fused_active": [P, I, P, P, I, I, P, I, I, P, P, L, I, I, I, F, P, P, P],
    "deer_head_pool_active": [P, P, I, I, I, I, P, P, I, I, P, P],
    "deer_ctl_begin_step_map": [P, P, I, P, P],
    "deer_trunk_layer_persistent": [P, P],
    "deer_layernorm_rows": [P, L, L, I, I, P, P, P, P, L, L, I, F, P],
    "deer_layernorm_rows_multi": [P, L, L, I, I, P, P, I, L, P, L, L, L, I, F, P],
    "deer_resadd_ln": [P, P, I, L, P, P, P, P, P, P, P, I, I, F, P, P],
    "deer_resadd_ln_multirow": [P, P, I, L, P, P, P, P, P, I, I, F, I, P],
    "deer_resadd_ln_split": [P, P, I, L, P, P, P, P, P, P, P, P, I, I, F, P, P],
    "deer_vit_im2col": [P, I, I, I, I, P, I, P],
    "deer_vit_embed_lnpre": [P, P, P, P, P, P, I, I, I, F, P],
    "deer_embed_tokens_f32": [P, P, P, P, I, I, I, I, I, P],
    "deer_vit_im2col_f32": [P, I, I, I, P, I, P],
    "deer_gemm_f32_nt": [P, I, P, I, P, P, I, I, I, I, I, P],
    "deer_attn_f32": [P, P, P, P, P, P, I, I, I, I, I, I, I, I, I, L, L, L, L, F, P],
    "deer_xattn_f32": [P, I, L, I, P, I, I, P, I, P, I, I, I, I, I, F, P, P],
    "deer_embed_tokens": [P, P, P, P, I, I, I, I, I, P],
    "deer_broadcast_rows": [P, P, L, I, P],
    "deer_head_pool": [P, P, I, I, I, I, P, P, I, I, P],
    "deer_head_state_embed": [P, P, P, P, P, P, P, I, I, I, P],
    "deer_head_pool_state": [P, P, I, I, I, I, P, P, P, I, I, P],
    "deer_head_lstm_layer": [P, L, I, I, I, P, P, P, P, P, P, P, P, P, P, I, I, F, P, I, I, I, P],
    "deer_head_lstm_hh": [P, P, I, P, P, I, I, I, P],
    "deer_head_lstm_layer_pre": [P, L, I, I, I, P, P, P, P, P, P, P, P, I, I, F, P, I, I, I, P],
    "deer_head_fc": [P, I, I, I, P, P, P, P, P, P, P, P, I, P, I, F, P, I, I, I, P],
    "deer_head_final": [P, I, I, I, P, P, P, P, P, P, P, P, P, I, I, I, P, I, I, I, P, P, P, P, I, I, I, P, F, I, P],
    "deer_head_final_multi": [P, I, I, I, P, P, P, P, P, P, P, P, P, I, I, I, P, I, I, I, P, P, P, P, I, I, I, P, F, I, I, P, P],
    "deer_head_fused_granules": [I, I, I, I, I, P],
    "deer_head_fused": [P, I, I, P],
    "deer_ctl_begin_step": [P, P, I, P],
    "deer_preprocess_frames": [P, I, I, I, I, P, P, P, P, P, P],
    "deer_preprocess_frames_f16": [P, I, I, I, I, P, P, P, P, P, P],
    "deer_preprocess_scratch_bytes": [I, I, I, I],
    "deer_spin_us": [I, P],
    "deer_hip_arch": [],
    "deer_hip_abi_version": [],
    # ---- the native spine (include/deer_model.h) ----
    "deer_model_create": [P, P],
    "deer_model_destroy": [P],
    "deer_model_arena_bytes": [P],
    "deer_model_workspace_bytes": [P],
    "deer_model_bind": [P, P, P],
    "deer_model_load_tensor": [P, c_char_p, P, I, L, P],
    "deer_model_share_weights": [P, P],
    "deer_model_knows_tensor": [P, c_char_p],
    "deer_model_missing_tensors": [P, P, I],
    "deer_model_buffer": [P, I, c_char_p, P, P],
    "deer_model_configure_exit": [P, P, I, I, I, I],
    "deer_model_real_num_exit": [P],
    "deer_model_set_compaction": [P, I],
    "deer_model_set_head_fused": [P, I],
    "deer_model_head_state_changed": [P],
    "deer_model_set_persistent_layer": [P, I],
    "deer_vit_l14_encode": [P, P, I, P, P],
    "deer_perceiver_resample": [P, P, I, P, P, P],
    "deer_llm_early_exit": [P, P, P, I, P, I, I, P, P, P],
    "deer_begin_step": [P, P, P],
    "deer_vision": [P, I, I, I, P],
    "deer_media_kv": [P, P],
    "deer_head_state": [P, P],
    "deer_llm_embed": [P, I, P],
    "deer_llm_layer": [P, I, I, I, I, I, I, P],
    "deer_head_eval": [P, I, I, I, I, I, I, I, I, P, I, P],
    "deer_model_layerwise_head": [P, I],
    "deer_head_eval_layerwise": [P, I, I, I, I, P],
    "deer_step_enqueue": [P, I, I, I, I, P, P],
    "deer_dynamic_plan": [P, P, P, P, I],
    "deer_model_n_chains": [P],
    "deer_prof_enable": [P, I],
    "deer_prof_count": [P],
    "deer_prof_get": [P, I, P, I, P, P, P],
    "deer_step_plan_create": [I, P, P, I, P, P, P, I, I, P, P, P],
    "deer_step_plan_run": [P, I, I, P, P, P, P, P],
    "deer_step_plan_destroy": [P],
}
_RESTYPE = {"deer_head_fused_granules": c_long, "deer_hip_arch": c_char_p, "deer_model_arena_bytes": c_long, "deer_model_workspace_bytes": c_long, "deer_preprocess_scratch_bytes": c_long,
            "deer_model_destroy": None, "deer_step_plan_destroy": None}

# constants of include/deer_hip.h
CTL_EXIT_FLAG, CTL_EXIT_LAYER, CTL_CUR_EXIT_ID, CTL_HOLD, CTL_N_EVALS, CTL_SHADOW, CTL_COMMITTED, CTL_ALL_EXITED = 0, 1, 2, 3, 4, 5, 6, 7
CTL_PREV_ACTION, CTL_OUT_ACTION, CTL_DELTAS, CTL_WORDS = 8, 16, 24, 64
CTL_N_EXITED, CTL_SEQ, CTL_HOST_PTR, CTL_EVALS_DONE = 40, 41, 42, 44
CTL_PREV_REAL, CTL_ENS_ACTION = 45, 48
HOSTM_PROGRESS, HOSTM_DONE = 0, 1
EPI_BF16, EPI_F32, EPI_QGELU_BF16, EPI_GELU_BF16, EPI_RESADD_F32, EPI_BF16OUT = 0, 1, 2, 3, 4, 5
A_BF16, A_SLABS_GELU, A_SLABS, A_F32 = 0, 1, 2, 3
X_RAW, X_POOL_MAX, X_POOL_AVG, X_LN = 0, 1, 2, 3
PRO_RAW, PRO_LN, PRO_GROUP_LN_RELU, PRO_GROUP_RELU = 0, 1, 2, 3
KIND_PSEUDO, KIND_CHECK, KIND_COMMIT = 0, 1, 2
THR_TYPES = {"L2": 0, "mean": 1, "max": 2, "cosine": 3}

_lib = None


def max_trunk_rows(cfg, precision: str = "fp16") -> int:
    """LLM rows (n_envs * T) one engine takes: 512 in the bf16 arithmetic (16 environments x the reference's max_length = 32 tokens,
    data.py:905-919; the hi/lo-plane trunk GEMM runs them in row blocks of 128), 128 in the fp32 arithmetic or when d_model % 64 != 0
    (one launch of deer_gemm_skinny)."""
    return 512 if (precision != "fp32" and cfg.d_model % 64 == 0) else 128


MAX_ENVS = 16        # environments per engine (csrc/common.h: DEER_MAX_ENVS)


def skinny_mpad(M: int) -> int:
    """Row count of one f32 partial slab of deer_gemm_skinny: M rounded up to whole 16-row MFMA tiles."""
    return 16 * ((M + 15) // 16)


class DeerConfigC(ctypes.Structure):
    """``deer_config`` of include/deer_model.h (field order must match)."""
    _fields_ = [(n, ctypes.c_int) for n in (
        "image_size", "patch_size", "vit_width", "vit_layers", "vit_heads", "vit_mlp",
        "perc_depth", "perc_heads", "perc_dim_head", "perc_latents", "perc_ff_mult",
        "vocab_size", "d_model", "n_heads", "n_layers", "mlp_ratio", "attn_qk_ln", "alibi_bias_max",
        "cross_attn_every_n_layers", "xattn_heads", "xattn_dim_head", "xattn_ff_mult", "media_token_id",
        "mpt7b_names", "exit_interval",
        "head_hidden", "lstm_num_layers", "lstm_layernorm", "mlp_layernorm", "mlp_num_hidden_layers", "pooling_avg",
        "n_envs", "max_text_len", "n_chains", "precision", "use_state", "sep_resampler", "multi_step_action", "layerwise_exit_eval",
        "operands_f16", "fusion_pre")]


# precision of an engine -> (deer_config.precision, deer_config.operands_f16):
#   "fp16"  the product arithmetic on IEEE fp16 operands (default): fp16 weights everywhere, fp16 results in the vision tower, fp16 hi + lo
#           activation planes in the trunk, f32 accumulation / LayerNorm / softmax / LSTM state - the reference's evaluation arithmetic
#           (fp32 weights under fp16 autocast, eval_utils.py:333, README.md:161-167)
#   "bf16"  the same kernels on bf16 operands - a `--precision bf16` / amp_bf16 reference run (eval_calvin.py:559-560)
#   "fp32"  f32 activations and f32 (or bf16 hi + lo) weights end to end (csrc/precise.hip): the parity arithmetic, 4-5x slower
PRECISIONS = {"fp16": (0, 1), "bf16": (0, 0), "fp32": (1, 0)}
DEFAULT_PRECISION = "fp16"


def resolve_precision(precision) -> str:
    """None -> DEER_PRECISION or the default; validates the name"""
    if precision is None:
        precision = os.environ.get("DEER_PRECISION", DEFAULT_PRECISION)
    if precision not in PRECISIONS:
        raise ValueError(f"precision must be one of {sorted(PRECISIONS)}, got {precision!r}")
    return precision


def torch_dtype16(precision: str):
    """torch dtype of the 16-bit tensors an engine of this precision exchanges (camera frames, media tokens, K / V, wte): f32 for "fp32" """
    import torch
    return {"fp16": torch.float16, "bf16": torch.bfloat16, "fp32": torch.float32}[precision]


def config_to_c(cfg, n_envs: int, max_text_len: int, n_chains: int = 0, precision: str = DEFAULT_PRECISION) -> DeerConfigC:
    c = DeerConfigC()
    for k in ("image_size", "patch_size", "vit_width", "vit_layers", "vit_heads", "vit_mlp", "perc_depth", "perc_heads",
              "perc_dim_head", "perc_latents", "perc_ff_mult", "vocab_size", "d_model", "n_heads", "mlp_ratio",
              "alibi_bias_max", "cross_attn_every_n_layers", "xattn_heads", "xattn_dim_head", "xattn_ff_mult",
              "media_token_id", "exit_interval", "head_hidden", "lstm_num_layers", "mlp_num_hidden_layers"):
        setattr(c, k, int(getattr(cfg, k)))
    c.n_layers = cfg.n_layers
    c.attn_qk_ln = 1 if cfg.attn_qk_ln else 0
    c.mpt7b_names = 1 if cfg.llm_name == "mpt_9b" else 0
    c.lstm_layernorm = 1 if cfg.lstm_layernorm else 0
    c.mlp_layernorm = 1 if cfg.mlp_layernorm else 0
    c.pooling_avg = 0 if cfg.pooling == "max" else 1
    c.n_envs, c.max_text_len, c.n_chains = n_envs, max_text_len, n_chains
    if precision not in PRECISIONS:
        raise ValueError(f"precision must be one of {sorted(PRECISIONS)}, got {precision!r}")
    c.precision, c.operands_f16 = PRECISIONS[precision]
    c.use_state = 1 if getattr(cfg, "use_state", False) else 0
    c.sep_resampler = 1 if getattr(cfg, "sep_resampler", False) else 0
    c.multi_step_action = int(getattr(cfg, "multi_step_action", 1))
    c.layerwise_exit_eval = 1 if getattr(cfg, "layerwise_exit_eval", False) else 0
    c.fusion_pre = 1 if getattr(cfg, "fusion_mode", "post") == "pre" else 0
    return c


class DeerHipError(RuntimeError):
    pass


def build(force: bool = False) -> str:
    """Compile every HIP source for gfx950 into deer_vla_amd/lib/libdeer_hip.so (hipcc; no GPU needed)."""
    if force:
        subprocess.run(["make", "-C", CSRC, "clean"], check=True, capture_output=True)
    r = subprocess.run(["make", "-C", CSRC, "-j8"], capture_output=True, text=True)
    if r.returncode != 0:
        raise DeerHipError("hipcc build of libdeer_hip.so failed:\n" + r.stdout[-4000:] + r.stderr[-4000:])
    if not os.path.isfile(LIB_PATH):
        raise DeerHipError("build finished but %s is missing" % LIB_PATH)
    return LIB_PATH


def lib() -> ctypes.CDLL:
    """Load libdeer_hip.so (once).  Raises if it is not there: the engine has no other compute path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise DeerHipError(
            f"{LIB_PATH} not found - build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc, gfx950).  deer_vla_amd has no CPU fallback.")
    # torch bundles its own libamdhip64; it must be loaded FIRST so that this library binds to the same HIP runtime
    # (two runtimes in one process = streams/pointers of one are invalid in the other -> launch errors)
    import torch  # noqa: F401
    try:
        l = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # missing libamdhip64 etc.
        raise DeerHipError(f"cannot load {LIB_PATH}: {e}") from e
    for name, argtypes in SIGNATURES.items():
        fn = getattr(l, name, None)
        if fn is None:
            raise DeerHipError(f"{LIB_PATH} does not export {name} (stale build?)")
        fn.argtypes = argtypes
        fn.restype = _RESTYPE.get(name, c_int)
    _lib = l
    return l


def check(rc: int, what: str):
    if rc != 0:
        raise DeerHipError(f"{what} failed with code {rc} ({ {1: 'invalid shape/argument', 2: 'kernel launch error'}.get(rc, '?')})")


def ptr(t, byte_offset: int = 0):
    """Device pointer of a torch tensor (None -> NULL)."""
    if t is None:
        return None
    return c_void_p(t.data_ptr() + byte_offset)
