"""PyTorch custom operators over the native spine (``torch.ops.deer.*``; BASELINE north_star: "host code stays Python calling HIP
through PyTorch-ROCm custom ops").  These are the three coarse operators SURVEY.md §8b specifies; each is one call into
``libdeer_hip.so`` (include/deer_model.h) on the CURRENT PyTorch HIP stream, so they compose with torch streams / graphs:

  deer::vit_l14_encode(Tensor images, int model) -> Tensor tokens              open_clip ``visual(x)[1]`` (flamingo_mpt.py:556-583)
  deer::perceiver_resample(Tensor tokens, int model) -> Tensor media           PerceiverResampler.forward (helpers.py:107-132)
  deer::llm_early_exit(Tensor ids, Tensor? key_mask, Tensor media, int model, int exit_id, bool shadow)
                                        -> (Tensor ctl, Tensor hidden)        mosaic_gpt_3b.py:274-449 + value_net.py:277-297
                                                                               + action_head.py:499-611 (exit gate on the device)

``model`` is the address of a ``deer_model`` (``NativeModel.handle``): weights, workspace, LSTM state and the exit controller
configuration live in that object, like they live in the reference's stateful ``MPTFlamingo`` module.
There is no CPU implementation: the ops raise without a HIP device / without the library.
"""
from __future__ import annotations

import ctypes
from typing import Dict, Optional, Sequence, Tuple

import torch

from . import _abi as abi

_MODELS: Dict[int, "NativeModel"] = {}


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class NativeModel:
    """Owner of one ``deer_model``: creates it from a DeerConfig, allocates the weight arena + workspace as torch tensors and
    ingests a reference state dict.  No orchestration here - that is the spine's; ``DeerEngine`` adds graph scheduling on top."""

    def __init__(self, cfg, state_dict: Dict[str, torch.Tensor], device="cuda", n_envs: int = 1, max_text_len: int = 32,
                 precision: Optional[str] = None):
        if not torch.cuda.is_available():
            raise abi.DeerHipError("deer_vla_amd needs a HIP device (no CPU fallback)")
        self.lib = abi.lib()
        self.cfg, self.B = cfg, n_envs
        self.dev = torch.device(device)
        precision = abi.resolve_precision(precision)        # "fp16" (default) / "bf16": product arithmetic on fp16 / bf16 operands; "fp32": parity arithmetic
        self.max_T = min(max_text_len, abi.max_trunk_rows(cfg, precision) // n_envs)   # the same row budget as DeerEngine (512 rows in the product arithmetic)
        self._h = ctypes.c_void_p()
        self.precision = precision
        # format of the camera frames / media tokens the operators hand around
        self.img_dtype = abi.torch_dtype16(precision)
        self.media_dtype = torch.float16 if precision == "fp16" else torch.bfloat16
        cc = abi.config_to_c(cfg, n_envs, self.max_T, precision=precision)
        abi.check(self.lib.deer_model_create(ctypes.byref(cc), ctypes.byref(self._h)), "deer_model_create")
        with torch.cuda.device(self.dev):
            self.arena = torch.zeros(self.lib.deer_model_arena_bytes(self._h), dtype=torch.uint8, device=self.dev)
            self.workspace = torch.zeros(self.lib.deer_model_workspace_bytes(self._h), dtype=torch.uint8, device=self.dev)
            abi.check(self.lib.deer_model_bind(self._h, abi.ptr(self.arena), abi.ptr(self.workspace)), "deer_model_bind")
            for name, t in state_dict.items():
                if not self.lib.deer_model_knows_tensor(self._h, name.encode()):
                    continue
                is_bf = t.dtype == torch.bfloat16
                src = t.detach().to(device=self.dev, dtype=torch.bfloat16 if is_bf else torch.float32).contiguous()
                abi.check(self.lib.deer_model_load_tensor(self._h, name.encode(), abi.ptr(src), 1 if is_bf else 0, src.numel(), _stream()),
                          f"deer_model_load_tensor({name})")
                torch.cuda.current_stream().synchronize()
        buf = ctypes.create_string_buffer(4096)
        n = self.lib.deer_model_missing_tensors(self._h, buf, len(buf))
        if n:
            raise abi.DeerHipError(f"{n} required parameters missing, e.g. {buf.value.decode().split()[:4]}")
        self.buffer("thresholds").view(torch.float32).fill_(1e8)
        _MODELS[self.handle] = self

    @property
    def handle(self) -> int:
        return int(self._h.value)

    def buffer(self, name: str) -> torch.Tensor:
        off, nbytes = ctypes.c_long(), ctypes.c_long()
        abi.check(self.lib.deer_model_buffer(self._h, 1, name.encode(), ctypes.byref(off), ctypes.byref(nbytes)), f"buffer {name}")
        return self.workspace[off.value: off.value + nbytes.value]

    def configure_exit(self, exit_ids: Sequence[int], max_layer: int, thresholds: Optional[Sequence[float]] = None,
                       threshold_type: str = "L2", leq: bool = True):
        ids = (ctypes.c_int * len(exit_ids))(*exit_ids)
        abi.check(self.lib.deer_model_configure_exit(self._h, ids, len(exit_ids), max_layer, abi.THR_TYPES[threshold_type], 1 if leq else 0),
                  "deer_model_configure_exit")
        if thresholds is not None:
            t = torch.full((16,), 1e8, dtype=torch.float32)
            t[: len(thresholds)] = torch.tensor([float(v) for v in thresholds])
            self.buffer("thresholds").view(torch.float32).copy_(t)

    def reset(self):
        for n in ("h_state", "c_state", "ctl"):
            self.buffer(n).zero_()

    def close(self):
        if self._h.value:
            _MODELS.pop(self.handle, None)
            torch.cuda.synchronize(self.dev)
            self.lib.deer_model_destroy(self._h)
            self._h = ctypes.c_void_p()


def _model(handle: int) -> NativeModel:
    m = _MODELS.get(int(handle))
    if m is None:
        raise abi.DeerHipError(f"deer op called with an unknown model handle {handle:#x}")
    return m


@torch.library.custom_op("deer::vit_l14_encode", mutates_args=(), device_types="cuda")
def vit_l14_encode(images: torch.Tensor, model: int) -> torch.Tensor:
    """images (N,3,S,S) CLIP-normalised (any float dtype; converted to the tower's format) -> patch tokens (N,256,W) f32 (x[:,1:], no ln_post)."""
    m = _model(model)
    cfg = m.cfg
    img = images.to(m.img_dtype).contiguous()
    out = torch.empty(img.shape[0], cfg.n_patches, cfg.vit_width, dtype=torch.float32, device=img.device)
    abi.check(m.lib.deer_vit_l14_encode(m._h, abi.ptr(img), img.shape[0], abi.ptr(out), _stream()), "deer_vit_l14_encode")
    return out


@vit_l14_encode.register_fake
def _(images, model):
    m = _model(model)
    return images.new_empty((images.shape[0], m.cfg.n_patches, m.cfg.vit_width), dtype=torch.float32)


@torch.library.custom_op("deer::perceiver_resample", mutates_args=(), device_types="cuda")
def perceiver_resample(tokens: torch.Tensor, model: int) -> torch.Tensor:
    """patch tokens (N,256,W) f32 -> media tokens (N*64, W) in the tower's 16-bit format (fp16 / bf16), frame order (rgb, gripper per environment);
    (N/2*64, W) with fusion_mode="pre" (one set of latents per environment)."""
    m = _model(model)
    cfg = m.cfg
    tok = tokens.to(torch.float32).contiguous()
    out = torch.empty(tok.shape[0] // 2 * cfg.n_media, cfg.vit_width, dtype=m.media_dtype, device=tok.device)
    abi.check(m.lib.deer_perceiver_resample(m._h, abi.ptr(tok), tok.shape[0], abi.ptr(out), None, _stream()), "deer_perceiver_resample")
    return out


@perceiver_resample.register_fake
def _(tokens, model):
    m = _model(model)
    return tokens.new_empty((tokens.shape[0] // 2 * m.cfg.n_media, m.cfg.vit_width), dtype=m.media_dtype)


@torch.library.custom_op("deer::llm_early_exit", mutates_args=(), device_types="cuda")
def llm_early_exit(ids: torch.Tensor, key_mask: Optional[torch.Tensor], media: torch.Tensor, model: int, exit_id: int,
                   shadow: bool) -> Tuple[torch.Tensor, torch.Tensor]:
    """ids (n_envs,T) int64, key_mask (n_envs,T) or None, media (n_envs*128, W) in the tower's 16-bit format; exit_id >= 0 static, < 0 dynamic (the
    model's configured controller).  Returns (ctl int32 [n_envs,64]: exit layer, action, deltas - include/deer_hip.h -,
    hidden f32 [n_layers, n_envs*T, d]); the LSTM state of the head is carried inside the model."""
    m = _model(model)
    cfg = m.cfg
    ids_c = ids.to(torch.int64).contiguous()
    T = ids_c.reshape(m.B, -1).shape[1]
    km = key_mask.to(torch.uint8).contiguous() if key_mask is not None else None
    # fp32-activation models keep their media tokens in f32 inside the model (the bf16 tensor handed around is only a view of them)
    med = None if m.precision == "fp32" else media.to(m.media_dtype).contiguous()
    abi.check(m.lib.deer_llm_early_exit(m._h, abi.ptr(ids_c), abi.ptr(km), T, abi.ptr(med), exit_id, 1 if shadow else 0, None, None, _stream()),
              "deer_llm_early_exit")
    rows = min(m.B * m.max_T, abi.max_trunk_rows(cfg, m.precision))
    ctl = m.buffer("ctl").view(torch.int32).view(m.B, abi.CTL_WORDS).clone()
    hidden = m.buffer("hidden").view(torch.float32).view(cfg.n_layers, rows, cfg.d_model)[:, : m.B * T].clone()
    return ctl, hidden


@llm_early_exit.register_fake
def _(ids, key_mask, media, model, exit_id, shadow):
    m = _model(model)
    T = ids.reshape(m.B, -1).shape[1]
    return (ids.new_empty((m.B, abi.CTL_WORDS), dtype=torch.int32),
            media.new_empty((m.cfg.n_layers, m.B * T, m.cfg.d_model), dtype=torch.float32))


def decode_ctl(ctl: torch.Tensor):
    """control blocks -> list of dict(exit_layer, pose (6,), gripper, gripper_logit, deltas) per environment"""
    c = ctl.cpu()
    f = c.view(torch.float32)
    out = []
    for b in range(c.shape[0]):
        a = f[b, abi.CTL_OUT_ACTION: abi.CTL_OUT_ACTION + 8]
        out.append(dict(exit_layer=int(c[b, abi.CTL_EXIT_LAYER]), pose=a[:6].clone(), gripper=float(a[6]), gripper_logit=float(a[7]),
                        deltas=f[b, abi.CTL_DELTAS: abi.CTL_DELTAS + 16].clone()))
    return out
