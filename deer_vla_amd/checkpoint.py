"""Checkpoint loader for real weights: OpenFlamingo ``.pt`` + DeeR ``.pth`` (+ the open_clip ViT-L/14 and HF MPT
state dicts they sit on) -> the native engine's weight arena.

What the reference does (robot_flamingo/eval/eval_calvin.py):
  * hyper-parameters from the checkpoint FILE NAME (:355-421; window size ``ws_<n>`` in eval_ckpts.py:64-67) and from the
    checkpoint DICT (:455-476, written by ``save_ckpt`` train_utils.py:27-60);
  * ``early_exit_layer = min(ckpt.early_exit_layer, max_layer)`` (:529), negative values count from the LLM depth (:478-484);
  * ``model.load_state_dict(torch.load(openflamingo_checkpoint), strict=False)`` (:541-543), then
    ``ddp_model.load_state_dict(checkpoint["model_state_dict"], False)`` (:572-578): keys carry DDP's ``module.`` prefix and
    only trainable tensors are stored (``get_checkpoint``, train_utils.py:631-638) - plus, as a side effect of that function
    walking ``named_parameters`` (which lists shared tensors once), the alias names of frozen tensors, e.g. the whole MPT
    trunk under ``lang_encoder.old_decoder_blocks.N.*`` (tests/golden/ckpt_meta.json, generated from the reference).
Nothing here touches a kernel: tensors are handed, by reference name, to ``deer_model_load_tensor`` (csrc/model.hip), which
converts and re-lays them out on the device.
"""
from __future__ import annotations

import os
import re
from typing import Dict, Iterable, List, Optional, Tuple

import torch

from .config import DeerConfig, deer_3b, deer_9b
from .synthetic import param_shapes

LLM_DEPTH = {"mpt_3b": 24, "mpt_dolly_3b": 24, "mpt_9b": 32}               # eval_calvin.py:478-484


def args_from_checkpoint_name(path: str) -> Dict:
    """eval_calvin.py:355-421 + eval_ckpts.py:64-67: what the evaluation scripts read out of the checkpoint's file name."""
    name = os.path.basename(path)
    attrs = name.split("_")
    a: Dict = {"window_size": 12, "multi_step_action": 1, "llm_name": "mpt_3b"}
    if "ws" in attrs:                                                      # eval_ckpts.py:64-67
        a["window_size"] = int(attrs[attrs.index("ws") + 1])
    a["use_state"] = "state" in path                                       # eval_calvin.py:356-357 (tests the whole path)
    a["real_data"] = "real" in path
    m = re.search(r"aug_(\d+)_(\d+)", path)
    a["rgb_pad"], a["gripper_pad"] = (int(m.group(1)), int(m.group(2))) if m else (-1, -1)
    a["sep_resampler"] = "sep" in path
    a["sep_lm_head"] = "lm_head" in path
    a["residual"] = "res_" in path
    a["tcp_rel"] = "tcp" in path
    if "step" in path.split("_"):
        sp = path.split("_")
        a["multi_step_action"] = int(sp[sp.index("step") - 1])
    a["text_aug"] = "text_aug" in path
    a["traj_cons"] = "traj_cons" in path
    if "difws" in path:
        sp = path.split("_")
        ix = sp.index("difws")
        a["dif_ws"], a["min_window_size"], a["max_window_size"] = True, int(sp[ix + 1]), int(sp[ix + 2])
        a["window_size"] = a["max_window_size"]
    if "latent" in path:
        sp = path.split("_")
        a["global_latent"] = int(sp[sp.index("latent") + 1])
    a["no_image_patch"] = "no_image_patch" in path
    if "gpt" in path:
        sp = path.split("_")
        a["decoder_type"], a["hidden_size"] = "gpt", int(sp[sp.index("gpt") + 1])
    for n in ["mpt_3b", "mpt_4b", "mpt_9b", "mpt_dolly_3b", "mpt_base_4b"]:      # first match wins (eval_calvin.py:417-420)
        if n in path:
            a["llm_name"] = n
            break
    return a


CKPT_DEFAULTS = [("head_type", "deterministic"), ("early_exit_layer", -1), ("multi_exit", False), ("exit_interval", 1),
                 ("exit_dropout", 0.0), ("lstm_dropout", 0.0), ("dropout_mode", "wo_last"), ("mlp_layernorm", False),
                 ("lstm_layernorm", False), ("mlp_num_hidden_layers", 3), ("lstm_num_layers", 4), ("pooling", "max")]


def args_from_checkpoint_dict(ckpt: Dict, llm_name: str, max_layer: Optional[int] = None) -> Dict:
    """eval_calvin.py:455-489 ``readout_args`` with its defaults, the ``layernorm`` compatibility key, negative
    ``early_exit_layer`` and the default ``max_layer``."""
    a = {k: ckpt.get(k, d) for k, d in CKPT_DEFAULTS}
    if "layernorm" in ckpt:                                                 # for compatibility with old code (:471-472)
        a["mlp_layernorm"] = ckpt["layernorm"]
    if a["early_exit_layer"] < 0:
        if llm_name not in LLM_DEPTH:
            raise NotImplementedError(llm_name)
        a["early_exit_layer"] += LLM_DEPTH[llm_name]
    a["max_layer"] = a["early_exit_layer"] + 1 if max_layer in (None, -1) else max_layer
    return a


def config_from_args(name_args: Dict, ckpt_args: Dict, trunk: Optional[DeerConfig] = None) -> DeerConfig:
    """DeerConfig of the model the reference would build: ``early_exit_layer=min(ckpt.early_exit_layer, max_layer)``
    (eval_calvin.py:529), head structure from the checkpoint dict (:530-539), trunk from ``mpt_dict[llm_name]``
    (factory.py:13-26) - or from ``trunk`` (a DeerConfig whose ViT / Perceiver / MPT sizes are used as they are: reduced-size
    test fixtures; a real checkpoint never needs it)."""
    if ckpt_args["head_type"] != "deterministic":
        raise NotImplementedError("only the deterministic LSTM head of the released DeeR checkpoints is built natively")
    llm = name_args["llm_name"]
    kw = dict(exit_interval=ckpt_args["exit_interval"], lstm_layernorm=bool(ckpt_args["lstm_layernorm"]),
              mlp_layernorm=bool(ckpt_args["mlp_layernorm"]), lstm_num_layers=ckpt_args["lstm_num_layers"],
              mlp_num_hidden_layers=ckpt_args["mlp_num_hidden_layers"], pooling=ckpt_args["pooling"],
              window_size=name_args["window_size"],
              use_state=bool(name_args.get("use_state", False)), sep_resampler=bool(name_args.get("sep_resampler", False)),
              multi_step_action=int(name_args.get("multi_step_action", 1)))          # "Nstep" in the file name (eval_calvin.py:384-387)
    ee = min(ckpt_args["early_exit_layer"], ckpt_args["max_layer"])
    if trunk is not None:
        from dataclasses import replace
        return replace(trunk, early_exit_layer=ee, llm_name=llm if llm in LLM_DEPTH else trunk.llm_name, **kw)
    if llm == "mpt_9b":
        cfg = deer_9b(**kw)
    elif llm in ("mpt_3b", "mpt_dolly_3b"):
        cfg = deer_3b(**kw)
        cfg.llm_name = llm
    else:
        raise NotImplementedError(f"{llm}: only the MPT-1B (3B) and MPT-7B (9B) OpenFlamingo trunks are built natively")
    cfg.early_exit_layer = ee
    return cfg


def apply_hf_mpt_config(cfg: DeerConfig, hf_config) -> DeerConfig:
    """The fields of the HF MPT repo's ``config.json`` that decide the block's ARITHMETIC, read from the file the reference loads the
    model with (``AutoModelForCausalLM.from_pretrained(lang_encoder_path, trust_remote_code=True)``, factory.py:134-136; the repos are
    un-vendored and un-pinned) instead of being assumed: both published layouts are understood - MosaicGPT
    (mpt-1b-redpajama-200b[-dolly]: flat ``attn_qk_ln``, ``alibi_bias_max``, ``mlp_ratio``) and MPT (mpt-7b: ``attn_config.qk_ln``,
    ``attn_config.alibi_bias_max``, ``expansion_ratio``).  ``hf_config``: path of the json file or the parsed dict.  Returns a new
    DeerConfig; raises on anything this engine's MPT block does not implement (it would otherwise be silently ignored)."""
    import dataclasses
    import json
    if isinstance(hf_config, (str, bytes)) or hasattr(hf_config, "__fspath__"):
        with open(hf_config) as f:
            c = json.load(f)
    else:
        c = dict(hf_config)
    ac = c.get("attn_config") or {}

    def pick(flat, nested, default=None):
        return c[flat] if flat in c else ac.get(nested, default)
    unsupported = {
        # ABSENT keys take the HF classes' own defaults (MPTConfig: alibi False, no_bias False, qk_ln False), not the values this engine
        # happens to support: an underspecified config.json would build learned positions and biases in the reference
        "alibi": (pick("alibi", "alibi", False), True, "positions come from ALiBi only (mosaic_gpt_3b.py:342-343: no learned wpe is added)"),
        "clip_qkv": (pick("attn_clip_qkv", "clip_qkv"), None, "qkv clamping is not built"),
        "softmax_scale": (pick("softmax_scale", "softmax_scale"), None, "the block scales by 1/sqrt(head_dim)"),
        "prefix_lm": (bool(pick("prefix_lm", "prefix_lm", False)), False, "causal attention only"),
        "attn_uses_sequence_id": (bool(pick("attn_uses_sequence_id", "attn_uses_sequence_id", False)), False, "no packed sequences on this path"),
        "no_bias": (bool(c.get("no_bias", False)), True, "Linear biases are not ingested (the released MPT checkpoints are bias-free)"),
    }
    for name, (got, want, why) in unsupported.items():
        if got != want:
            raise NotImplementedError(f"HF MPT config: {name}={got!r} is not implemented by deer_vla_amd ({why})")
    norm = c.get("norm_type", "low_precision_layernorm")
    if norm not in ("low_precision_layernorm", "layernorm"):
        raise NotImplementedError(f"HF MPT config: norm_type={norm!r} (only LayerNorm variants are built)")
    upd = dict(d_model=int(c["d_model"]), n_heads=int(c["n_heads"]), n_layers_total=int(c["n_layers"]),
               mlp_ratio=int(c["mlp_ratio"] if "mlp_ratio" in c else c.get("expansion_ratio", cfg.mlp_ratio)),
               attn_qk_ln=bool(pick("attn_qk_ln", "qk_ln", False)), alibi_bias_max=int(pick("alibi_bias_max", "alibi_bias_max", 8)))
    if upd["n_layers_total"] <= cfg.early_exit_layer:
        raise ValueError(f"HF MPT config has {upd['n_layers_total']} layers, the DeeR checkpoint exits at layer {cfg.early_exit_layer}")
    return dataclasses.replace(cfg, **upd)


def canonical_key(k: str) -> str:
    """One reference state-dict key -> the canonical name the engine ingests (deer_vla_amd.synthetic.param_shapes):
    strip DDP's ``module.``; fold the two alias registrations (flamingo_lm.py:160-176) into the block-local names."""
    if k.startswith("module."):
        k = k[len("module."):]
    for alias, mid in (("lang_encoder.gated_cross_attn_layers.", "gated_cross_attn_layer"),
                       ("lang_encoder.old_decoder_blocks.", "decoder_layer")):
        if k.startswith(alias):
            n, tail = k[len(alias):].split(".", 1)
            return f"lang_encoder.transformer.blocks.{n}.{mid}.{tail}"
    return k


def map_open_clip_keys(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """open_clip ``create_model_and_transforms("ViT-L-14", pretrained="openai")`` state dict (keys ``visual.*``, plus the unused
    text tower) -> ``vision_encoder.visual.*`` as the Flamingo module registers it (factory.py:109-114, flamingo_mpt.py:93)."""
    return {"vision_encoder." + k: v for k, v in sd.items() if k.startswith("visual.")}


def map_hf_mpt_keys(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """HF MPT checkpoint (``transformer.wte.weight``, ``transformer.blocks.N.*``; SURVEY App. B.1 parameter names) -> the names
    inside DeeR, where every block sits behind the FlamingoLayer wrapper (flamingo_lm.py:15-16): ``...blocks.N.decoder_layer.*``"""
    out = {}
    for k, v in sd.items():
        m = re.match(r"transformer\.blocks\.(\d+)\.(.+)", k)
        if m:
            out[f"lang_encoder.transformer.blocks.{m.group(1)}.decoder_layer.{m.group(2)}"] = v
        elif k == "transformer.wte.weight":
            out["lang_encoder.transformer.wte.weight"] = v
    return out


def assemble_state_dict(cfg: DeerConfig, sources: Iterable[Tuple[str, Dict[str, torch.Tensor]]]) -> Tuple[Dict[str, torch.Tensor], Dict]:
    """Later sources override earlier ones (the reference loads OpenFlamingo first, the DeeR checkpoint second, both
    strict=False).  Returns (canonical state dict, report: which source every tensor came from, ignored keys, missing)."""
    want = param_shapes(cfg)
    sd: Dict[str, torch.Tensor] = {}
    origin: Dict[str, str] = {}
    ignored: Dict[str, List[str]] = {}
    for tag, src in sources:
        for k, v in src.items():
            ck = canonical_key(k)
            if ck not in want or not torch.is_tensor(v):
                ignored.setdefault(tag, []).append(k)                       # lm_head.*, lm_exit_modules.*, ln_f, text tower, layers beyond early_exit_layer ...
                continue
            shape = tuple(want[ck][0])
            if tuple(v.shape) != shape:
                if ck == "lang_encoder.transformer.wte.weight" and v.shape[1] == shape[1]:
                    t = torch.zeros(shape, dtype=v.dtype)                    # vocabulary resized for the added tokens (factory.py:120-126,159)
                    n = min(shape[0], v.shape[0])
                    t[:n] = v[:n]
                    v = t
                elif tuple(d for d in v.shape if d != 1) == tuple(d for d in shape if d != 1):
                    v = v.reshape(shape)                                     # singleton dims only: class_embedding (W,) vs (1, W), gates () vs (1,)
                else:                                                        # same numel, other layout (e.g. (in, out) vs (out, in)): never reshaped silently
                    raise RuntimeError(f"size mismatch for {k} ({tag}): {tuple(v.shape)} vs {shape}")
            sd[ck] = v.detach()
            origin[ck] = tag
    kinds = {k: v[1] for k, v in want.items()}
    optional = {k for k in want if k.endswith("decoder_layer.ln_1.bias") or k.endswith("decoder_layer.ln_2.bias")}
    missing = [k for k in want if k not in sd and k not in optional]
    return sd, {"origin": origin, "ignored": ignored, "missing": missing, "kinds": kinds}


def _torch_load(path: str):
    if path.endswith(".safetensors"):
        from safetensors.torch import load_file
        return load_file(path)
    try:                                                                     # checkpoints hold tensors and plain Python values only
        return torch.load(path, map_location="cpu", weights_only=True)
    except Exception as e:
        if os.environ.get("DEER_UNSAFE_PICKLE") == "1":                      # explicit opt-in: full unpickling executes code from the file
            return torch.load(path, map_location="cpu", weights_only=False)
        raise RuntimeError(f"{path}: not loadable with weights_only=True ({type(e).__name__}: {e}); if the file is trusted, set "
                           "DEER_UNSAFE_PICKLE=1 to allow full unpickling") from e


def load_checkpoint_files(deer_ckpt: str, openflamingo_ckpt: Optional[str] = None, clip_state: Optional[str] = None,
                          mpt_state: Optional[str] = None, max_layer: Optional[int] = None, trunk: Optional[DeerConfig] = None,
                          mpt_config: Optional[str] = None):
    """Read the files the reference's eval reads and return (cfg, state_dict, info).  ``clip_state`` / ``mpt_state``: state dicts
    of the open_clip model and of the HF MPT model saved to disk (the reference obtains them through open_clip / transformers
    at construction time, factory.py:109-136; neither package's hub access exists here).  ``mpt_config``: the HF repo's
    ``config.json`` (``mpt_dict[llm_name]["lang_encoder_path"]/config.json``): when given, attn_qk_ln / alibi_bias_max / sizes are READ
    from it (apply_hf_mpt_config) instead of taken from this package's defaults."""
    ck = _torch_load(deer_ckpt)
    name_args = args_from_checkpoint_name(deer_ckpt)
    ckpt_args = args_from_checkpoint_dict(ck if isinstance(ck, dict) else {}, name_args["llm_name"], max_layer)
    cfg = config_from_args(name_args, ckpt_args, trunk)
    if mpt_config is not None:
        cfg = apply_hf_mpt_config(cfg, mpt_config)
    sources = []
    if clip_state:
        sources.append(("open_clip", map_open_clip_keys(_torch_load(clip_state))))
    if mpt_state:
        sources.append(("hf_mpt", map_hf_mpt_keys(_torch_load(mpt_state))))
    if openflamingo_ckpt:
        sources.append(("openflamingo", _torch_load(openflamingo_ckpt)))
    try:
        deer_sd = ck["model_state_dict"]                                    # eval_calvin.py:572-575
    except (KeyError, TypeError):
        deer_sd = ck
    sources.append(("deer", deer_sd))
    sd, report = assemble_state_dict(cfg, sources)
    info = {"name_args": name_args, "ckpt_args": ckpt_args, "values": ck.get("values") if isinstance(ck, dict) else None, **report}
    return cfg, sd, info


def build_model_from_checkpoint(deer_ckpt: str, openflamingo_ckpt: Optional[str] = None, clip_state: Optional[str] = None,
                                mpt_state: Optional[str] = None, max_layer: Optional[int] = None, device="cuda", strict: bool = True,
                                trunk: Optional[DeerConfig] = None, precision: Optional[str] = None, n_envs: int = 1, mpt_config: Optional[str] = None):
    """-> (MPTFlamingo, info) ready for ``forward`` / ``ModelWrapper``; raises when tensors are missing (strict).  precision: None / "fp16" = the
    product arithmetic on fp16 operands (the README's `--precision fp32 --amp 1` evaluation: every Linear on fp16-rounded weights under
    autocast), "bf16" = a ``--precision bf16`` run, "fp32" keeps the checkpoint's f32 weights (f32 / hi+lo copies on the device) - the
    arithmetic of a reference run at ``--precision fp32`` without amp."""
    from .flamingo_mpt import MPTFlamingo
    cfg, sd, info = load_checkpoint_files(deer_ckpt, openflamingo_ckpt, clip_state, mpt_state, max_layer, trunk, mpt_config)
    if strict and info["missing"]:
        raise RuntimeError(f"{len(info['missing'])} tensors missing after loading all checkpoint files, e.g. {info['missing'][:4]}")
    model = MPTFlamingo(cfg, sd, window_size=info["name_args"]["window_size"], device=device, precision=precision, n_envs=n_envs)
    return model, info
