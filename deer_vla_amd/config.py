"""Shape/config description of the DeeR-VLA hot path (one control step).

The reference has no single config object: hyper-parameters are scattered over
``robot_flamingo/models/factory.py:13-26`` (``mpt_dict``), the HF ``config.json`` of the
MPT repos (EXTERNAL), open_clip's ``ViT-L-14`` model config (EXTERNAL) and the DeeR
checkpoint dict (``robot_flamingo/eval/eval_calvin.py:455-476``).  ``DeerConfig`` gathers
exactly the numbers the hot path needs so that the oracle, the synthetic-weight
generator and the HIP engine all agree on shapes.
"""
from __future__ import annotations

from dataclasses import dataclass, field, asdict
from typing import List, Optional


@dataclass
class DeerConfig:
    # ---- CLIP ViT visual tower (open_clip "ViT-L-14") -------------
    image_size: int = 224
    patch_size: int = 14
    vit_width: int = 1024
    vit_layers: int = 24
    vit_heads: int = 16            # head_dim = vit_width // vit_heads (64 for ViT-L/14)
    vit_mlp: int = 4096
    # ---- Perceiver resampler (open_flamingo/src/helpers.py:68-132) ---------------------
    perc_depth: int = 6
    perc_heads: int = 8
    perc_dim_head: int = 64
    perc_latents: int = 64
    perc_ff_mult: int = 4
    # ---- MPT decoder (mosaic_gpt_3b.py / modeling_gpt_9b.py) ------
    llm_name: str = "mpt_dolly_3b"  # key of factory.py:13-26 mpt_dict
    vocab_size: int = 50280         # gpt-neox-20b tokenizer (50277) + <|endofchunk|>, <image>, <PAD> (factory.py:120-126,159)
    d_model: int = 2048
    n_heads: int = 16
    n_layers_total: int = 24        # depth of the pretrained LLM before truncation
    mlp_ratio: int = 4
    attn_qk_ln: bool = True         # MPT-1B: LayerNorm over d_model on q and k
    alibi_bias_max: int = 8
    cross_attn_every_n_layers: int = 1   # factory.py:17 (3B) / :23 (9B -> 4)
    xattn_heads: int = 8            # helpers.py:141-142 defaults
    xattn_dim_head: int = 64
    xattn_ff_mult: int = 4
    media_token_id: int = 50278
    eoc_token_id: int = 50277
    # ---- early exit structure (flamingo_mpt.py:191-259) ---------------------------------
    early_exit_layer: int = 11      # LLM is truncated to early_exit_layer+1 layers
    exit_interval: int = 2
    # ---- action head (action_head.py:408-497) -------------------------------------------
    head_hidden: int = 1024
    lstm_num_layers: int = 4
    lstm_layernorm: bool = True
    mlp_layernorm: bool = True
    mlp_num_hidden_layers: int = 2
    pooling: str = "max"
    window_size: int = 12
    # ---- variants the reference parses from checkpoint names (eval_calvin.py:355-377); off in the released DeeR checkpoints --------
    use_state: bool = False         # DeterministicDecoder adds an embedding of the robot state to the pooled feature (action_head.py:524-536)
    sep_resampler: bool = False     # the gripper camera has its own PerceiverResampler weights (flamingo_mpt.py:132-134,656-659)
    # ---- harness ablations / checkpoint-name variants (round 5) ---------------------------------------------------------------------
    multi_step_action: int = 1      # the heads emit 6 * A pose values and A gripper values per call (action_head.py:458,472-473; "Nstep" in a
                                    # checkpoint name, eval_calvin.py:384-387); ModelWrapper executes the first multi_execution of them
    fusion_mode: str = "post"       # "post" (released checkpoints): one PerceiverResampler call per camera, 2 x perc_latents media tokens
                                    # (flamingo_mpt.py:609-668); "pre" (round 6): both cameras' patch tokens through ONE call, perc_latents
                                    # media tokens (flamingo_mpt.py:585-607)
    layerwise_exit_eval: bool = False   # the action of exit layer k comes from that layer's OWN head lm_exits[k] / lm_head (flamingo_mpt.py:253-261,
                                        # 450-457; eval_calvin.py:330,530,539) instead of extra_exit; the exit DECISION stays with extra_exit

    supports_use_state = True       # read by factory.create_model_and_transforms (keywords without this marker raise)
    supports_sep_resampler = True
    MAX_MULTI_STEP = 8              # csrc/head.hip: 7 * A outputs per head evaluation, A <= 8

    # ------------------------------------------------------------------ derived
    @property
    def n_patches(self) -> int:
        g = self.image_size // self.patch_size
        return g * g

    @property
    def vit_tokens(self) -> int:
        return self.n_patches + 1

    @property
    def vit_head_dim(self) -> int:
        return self.vit_width // self.vit_heads

    @property
    def n_layers(self) -> int:
        """Layers actually built (flamingo_mpt.py:198 deletes the rest)."""
        return self.early_exit_layer + 1

    @property
    def head_dim(self) -> int:
        return self.d_model // self.n_heads

    @property
    def n_media(self) -> int:
        """Media tokens seen by the gated x-attn: rgb + gripper latents (flamingo_mpt.py:661); one set of latents with pre fusion (:602)."""
        return self.perc_latents if self.fusion_mode == "pre" else 2 * self.perc_latents

    @property
    def mlp_hidden_dims(self) -> List[int]:
        return [1024, 512, 256][: self.mlp_num_hidden_layers]   # action_head.py:87-89

    def exit_ids(self) -> List[int]:
        """``MPTFlamingo.get_all_exit_idx`` (flamingo_mpt.py:239-250,268-270)."""
        ids = list(range(self.exit_interval - 1, self.early_exit_layer, self.exit_interval))
        return ids + [self.n_layers - 1]

    def layerwise_heads(self) -> List[tuple]:
        """(state-dict prefix, exit layer) of the per-layer heads, in the order the reference registers them: ``lm_exit_modules.j`` is the
        head of the j-th internal exit (``self.lm_exits`` is a plain dict, ``nn.ModuleList(self.lm_exits.values())`` carries the
        parameters: flamingo_mpt.py:239-244), ``lm_head`` serves the last layer (:453-454)."""
        ids = self.exit_ids()
        return [(f"lm_exit_modules.{j}.", e) for j, e in enumerate(ids[:-1])] + [("lm_head.", ids[-1])]

    def has_xattn(self, layer_idx: int) -> bool:
        """flamingo_lm.py:176: x-attn on layers where (idx+1) % every_n == 0."""
        return (layer_idx + 1) % self.cross_attn_every_n_layers == 0

    def to_dict(self):
        return asdict(self)


def deer_3b(max_layer: int = 12, **kw) -> DeerConfig:
    """OpenFlamingo-3B / MPT-1B; released ckpts are ``layer_11`` + ``intv=2`` (README.md:141).
    ``early_exit_layer = min(ckpt.early_exit_layer, max_layer)`` (eval_calvin.py:529)."""
    return DeerConfig(early_exit_layer=min(11, max_layer), **kw)


def deer_9b(max_layer: int = 12, **kw) -> DeerConfig:
    """OpenFlamingo-9B / MPT-7B (factory.py:20-25): d=4096, 32 heads, x-attn every 4th layer."""
    base = dict(llm_name="mpt_9b", d_model=4096, n_heads=32, n_layers_total=32, attn_qk_ln=False,
                cross_attn_every_n_layers=4,
                early_exit_layer=min(15, max_layer))
    base.update(kw)
    return DeerConfig(**base)


def deer_tiny(**kw) -> DeerConfig:
    """Reduced dims for oracle-speed parity tests.  Keeps every structural feature (two cameras,
    qk-LN, ALiBi, x-attn every layer, 4-layer LN-LSTM head, interval-2 exits) and the head dims
    the kernels are specialised on (head_dim 64 vision / 128 LLM)."""
    base = dict(image_size=56, patch_size=14, vit_width=128, vit_layers=2, vit_heads=2, vit_mlp=512,
                perc_depth=2, perc_latents=64, d_model=256, n_heads=2, n_layers_total=8,
                vocab_size=515, media_token_id=513, eoc_token_id=512, early_exit_layer=5,
                head_hidden=128)
    base.update(kw)
    return DeerConfig(**base)
