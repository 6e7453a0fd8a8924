"""Host-side mirror of ``MPTFlamingo`` (robot_flamingo/models/flamingo_mpt.py:17-778) and of the multi-exit
``lang_encoder`` surface (mosaic_gpt_3b.py:274-449 + open_flamingo/src/flamingo_lm.py:131-247) over the HIP engine.

Drop-in contract: same constructor keywords that matter for inference, same ``forward`` signature,
same returned object (``.logits=(pose (1,1,6), gripper (1,1,1))``, ``.exit_layer``, ``.hidden_states`` tuple with
``len == exit_layer + 1``), same attributes the CALVIN harness reads through ``.module`` (eval_utils.py:192-480),
same state-dict key names (``load_state_dict(..., strict=False)`` accepts the OpenFlamingo ``.pt`` and the DeeR
``.pth`` ``model_state_dict`` with its ``module.`` prefix).

Two execution paths:
 * native controller (``deer_vla_amd.value_net.ExitController``) or static ``exit_id``: the whole control step is one
   HIP-graph replay; the exit criterion runs on the device (no host round trip per layer).
 * foreign controller (any callable ``ctl(all_hidden_states, b_idx) -> bool``, e.g. the reference's own class): the
   reference's host loop - one layer at a time, one sync per exit check - on the same kernels.
There is no CPU path: constructing the engine without a HIP device raises.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Any, Dict, Optional, Tuple

import torch
from torch import nn

from . import _abi as abi
from .action_head import DeterministicDecoder
from .config import DeerConfig
from .engine import DeerEngine
from .synthetic import param_shapes
from .value_net import ExitController


@dataclass
class CausalLMOutputWithPast:
    """mosaic_gpt_3b.py:27-61 (fields used on this path)."""
    loss: Optional[torch.Tensor] = None
    logits: Any = None
    past_key_values: Any = None
    hidden_states: Optional[Tuple[torch.Tensor, ...]] = None
    attentions: Any = None
    exit_layer: Optional[int] = None


class _LazyHidden:
    """``hidden_states`` of a step as a tuple-like object that copies the layer outputs off the engine's buffers on FIRST access (valid
    until the next step overwrites them): ``len`` is the reference's exit_layer + 1 (flamingo_mpt.py:458)."""

    def __init__(self, engine, exit_layer, T):
        self._e, self._n, self._T, self._t = engine, exit_layer + 1, T, None
        self._seq = engine._seq                     # the step these buffers belong to

    def _materialise(self):
        if self._t is None:
            if self._e._seq != self._seq:           # a later step has overwritten the engine's buffers: never hand out its data
                raise RuntimeError("hidden_states of a host_outputs step were first read after a later control step; read them before "
                                   "the next step or call the model with host_outputs=False (eager device copies)")
            hs = self._e.hidden[: self._n, : self._T].clone()
            self._t = tuple(hs[i].unsqueeze(0) for i in range(self._n))
        return self._t

    def __len__(self):
        return self._n

    def __getitem__(self, i):
        return self._materialise()[i]

    def __iter__(self):
        return iter(self._materialise())


class _Cfg:
    def __init__(self, n_layers, d_model):
        self.n_layers, self.d_model = n_layers, d_model


class _Layer:
    """Stand-in for ``FlamingoLayer`` (flamingo_lm.py:6-44): conditioning state only."""

    def __init__(self, has_xattn):
        self.gated_cross_attn_layer = object() if has_xattn else None
        self.vis_x = None
        self.media_locations = None
        self.use_cached_media = False

    def is_conditioned(self):
        return self.vis_x is not None and self.media_locations is not None

    def condition_vis_x(self, v):
        self.vis_x = v

    def condition_media_locations(self, m):
        self.media_locations = m

    def condition_use_cached_media(self, u):
        self.use_cached_media = u


class LangEncoder:
    """The ``lang_encoder`` object of the reference (MosaicGPT + FlamingoLMMixin) as seen from outside."""

    def __init__(self, model: "MPTFlamingo"):
        self._m = model
        cfg = model.cfg
        self.config = _Cfg(cfg.n_layers, cfg.d_model)
        self.media_token_id = cfg.media_token_id
        self.initialized_flamingo = True
        self._layers = [_Layer(cfg.has_xattn(i)) for i in range(cfg.n_layers)]
        self.gated_cross_attn_layers = [l.gated_cross_attn_layer for l in self._layers]
        self.old_decoder_blocks = list(range(cfg.n_layers))
        self.lm_head = nn.Identity()                            # flamingo_mpt.py:188

    def _get_decoder_layers(self):
        return self._layers

    def is_conditioned(self) -> bool:
        return all(l.is_conditioned() for l in self._layers)

    def clear_conditioned_layers(self):
        for l in self._layers:
            l.condition_vis_x(None)
            l.condition_media_locations(None)
            l.condition_use_cached_media(None)

    def get_input_embeddings(self):
        return self._m.engine.wte

    # ---- construction-time mixin API (flamingo_lm.py:136-202, factory.py:139-159); the native model is built finished ----
    def init_flamingo(self, media_token_id, lang_hidden_size=None, vis_hidden_size=None, cross_attn_every_n_layers=None,
                      gradient_checkpointing=False):
        """Already initialised by the factory; accepted for call-site compatibility (the structure is fixed by the config)."""
        cfg = self._m.cfg
        if media_token_id != cfg.media_token_id or (cross_attn_every_n_layers not in (None, cfg.cross_attn_every_n_layers)):
            raise NotImplementedError("the native model's media token / x-attn spacing come from its DeerConfig")
        self.initialized_flamingo = True

    def _set_decoder_layers(self, value):
        raise NotImplementedError("decoder layers are device-resident weights of the engine, not Python modules")

    def _delete_decoder_layers(self, indices):
        """flamingo_mpt.py:191-198 truncates the LLM to early_exit_layer+1 layers; the native config is built truncated."""
        if any(i < self._m.cfg.n_layers for i in indices):
            raise NotImplementedError("rebuild the model with a smaller early_exit_layer instead")

    def resize_token_embeddings(self, new_num_tokens=None):
        """factory.py:148: grows wte for the added <image>/<|endofchunk|>/<PAD> tokens; the native wte already has them."""
        if new_num_tokens is not None and new_num_tokens > self._m.cfg.vocab_size:
            raise NotImplementedError("vocabulary larger than the configured %d" % self._m.cfg.vocab_size)
        return self.get_input_embeddings()

    def forward(self, input_ids, attention_mask=None, past_key_values=None, prefix_mask=None, sequence_id=None,
                return_dict=None, output_attentions=None, output_hidden_states=None, use_cache=None,
                exit_controller=None, exit_id=None, all_hidden_states=None, eval_flop=False, eval_time=False):
        """mosaic_gpt_3b.py:274-449 on the already-encoded media (``MPTFlamingo.forward`` conditions the layers)."""
        assert exit_controller is None or exit_id is None, "Only one exit indicator can be sepcified!"
        if output_attentions:
            raise NotImplementedError("output_attentions is not implemented yet for MosaicGPT")
        return self._m._run_llm(input_ids, attention_mask, exit_controller, exit_id)

    __call__ = forward


class MPTFlamingo(nn.Module):
    def __init__(self, cfg: DeerConfig, state_dict: Optional[Dict[str, torch.Tensor]] = None, window_size: int = 12,
                 use_gripper: bool = True, fusion_mode: str = "post", device="cuda", n_envs: int = 1, precision: Optional[str] = None, **unused):
        super().__init__()
        if unused:
            raise TypeError(f"MPTFlamingo: keywords {sorted(unused)} are not implemented by deer_vla_amd (they would be silently ignored)")
        # Built: fusion_mode 'post' (flamingo_mpt.py:380-381, every released DeeR checkpoint) and 'pre' (round 6; :378-379,585-607: one
        # PerceiverResampler call over both cameras' patch tokens).  The reference's own forward does not run the others in step mode
        # (tests/golden/fusion_modes_reference.npz records it): use_gripper=False and 'two_way' reach _encode_vision_x, which reads an
        # undefined name (:541, NameError); 'vit_concat' reshapes the batch into windows (:755).
        if not use_gripper or fusion_mode not in ("post", "pre"):
            raise NotImplementedError("use_gripper=True with fusion_mode='post' (released DeeR checkpoints, flamingo_mpt.py:380-381) or 'pre' "
                                      "(:378-379) are built; the reference's own forward raises on use_gripper=False / 'two_way' (NameError, "
                                      "flamingo_mpt.py:541) and on 'vit_concat' in step mode (:755)")
        if fusion_mode != getattr(cfg, "fusion_mode", "post"):
            import dataclasses
            cfg = dataclasses.replace(cfg, fusion_mode=fusion_mode)
        if cfg.fusion_mode == "pre" and getattr(cfg, "sep_resampler", False):
            raise ValueError("fusion_mode='pre' has one PerceiverResampler for both cameras (flamingo_mpt.py:602): sep_resampler does not apply")
        self.cfg = cfg
        # "fp16" (default): the product arithmetic on IEEE fp16 operands = the reference's evaluation arithmetic (fp32 weights under fp16
        # autocast, eval_utils.py:333); "bf16": the same on bf16 operands (a --precision bf16 / amp_bf16 run); "fp32": fp32 activations
        # everywhere (parity arithmetic, ~4x slower)
        self.precision = abi.resolve_precision(precision)
        self._device = device
        self.n_envs = n_envs                                      # environments per control step ("one env batch per rank")
        self._sd: Dict[str, torch.Tensor] = dict(state_dict) if state_dict is not None else {}
        self._engine: Optional[DeerEngine] = None
        # attributes read by the CALVIN harness through `.module` (eval_utils.py:192-247,300-331,456,480)
        self.window_size = window_size
        self.use_gripper, self.fusion_mode = use_gripper, fusion_mode
        self.use_state, self.sep_lm_head, self.tcp_rel = bool(getattr(cfg, "use_state", False)), True, False
        self.replan, self.refresh, self.pad_length = -1, -1, -1
        self.act_step = int(getattr(cfg, "multi_step_action", 1))     # flamingo_mpt.py:94; read by ModelWrapper (eval_utils.py:218,456,466)
        self.decoder_type, self.head_type = "lstm", "deterministic"
        self.use_diff, self.use_hist, self.sep_resampler = False, False, bool(getattr(cfg, "sep_resampler", False))
        self.eoc_token_id, self.media_token_id = cfg.eoc_token_id, cfg.media_token_id
        self.vis_dim, self.lang_dim = cfg.vit_width, cfg.d_model
        self.early_exit_layer = cfg.early_exit_layer
        # per-layer heads lm_exits[k] / lm_head with their own LSTM histories produce the action of exit layer k (flamingo_mpt.py:253,
        # 450-457; eval_calvin.py:330,539); the exit decision stays with extra_exit, whose state nobody commits in this mode
        self.layerwise_exit_eval = bool(getattr(cfg, "layerwise_exit_eval", False))
        # host_outputs=True (set by rollout.ModelWrapper, which only reads the action and the exit layer): ``forward`` returns the action
        # as CPU tensors taken from the step's pinned verdict block and ``hidden_states`` as a lazy tuple that is copied off the engine's
        # buffers on first access - no device clone, no extra device read per step.  Default False: eager device tensors like the reference.
        self.host_outputs = False
        self.llm_inference_time = -1.0
        self.lm_exits = {i: None for i in range(cfg.exit_interval - 1, cfg.early_exit_layer, cfg.exit_interval)}
        self.lang_encoder = LangEncoder(self)
        self.extra_exit: Optional[DeterministicDecoder] = None
        self.lm_head = None
        self._ctl_sig = None

    @property
    def module(self):
        """DDP-style indirection expected by the harness (eval_utils.py:192-247); a property so that nn.Module does not
        register the model as its own child."""
        return self

    # ---- engine / weights ------------------------------------------------------------------------------------
    @property
    def engine(self) -> DeerEngine:
        if self._engine is None:
            missing = [k for k in param_shapes(self.cfg) if k not in self._sd]
            if missing:
                raise RuntimeError(f"{len(missing)} parameters missing before the first forward, e.g. {missing[:3]}")
            self._engine = DeerEngine(self.cfg, self._sd, device=self._device, n_envs=self.n_envs, precision=self.precision)
            self.extra_exit = DeterministicDecoder(self._engine, self.window_size)
            self.lm_head = self.extra_exit
        return self._engine

    def sibling(self) -> "MPTFlamingo":
        """A second model over the SAME device weights (own workspace, LSTM state, exit controller): a further env batch that can be
        stepped concurrently with this one from another host thread / stream (rollout.evaluate_policy_batched(groups=...))."""
        other = MPTFlamingo(self.cfg, None, window_size=self.window_size, use_gripper=self.use_gripper, fusion_mode=self.fusion_mode,
                            device=self._device, n_envs=self.n_envs, precision=self.precision)
        other._sd = self._sd
        other._engine = DeerEngine(self.cfg, None, device=self._device, n_envs=self.n_envs, precision=self.precision, weights_from=self.engine)
        other.extra_exit = DeterministicDecoder(other._engine, self.window_size)
        other.lm_head = other.extra_exit
        return other

    def state_dict(self, *a, **k):
        return dict(self._sd)

    def load_state_dict(self, state_dict, strict: bool = True):
        """Accepts reference key names, with or without the DDP ``module.`` prefix, plus the
        ``lang_encoder.gated_cross_attn_layers.N.*`` alias of the x-attn weights."""
        want = param_shapes(self.cfg)
        unexpected, loaded = [], set()
        for k, v in state_dict.items():
            k2 = k[len("module."):] if k.startswith("module.") else k
            if k2.startswith("lang_encoder.gated_cross_attn_layers."):
                rest = k2[len("lang_encoder.gated_cross_attn_layers."):]
                n, tail = rest.split(".", 1)
                k2 = f"lang_encoder.transformer.blocks.{n}.gated_cross_attn_layer.{tail}"
            if k2.startswith("lang_encoder.old_decoder_blocks."):
                rest = k2[len("lang_encoder.old_decoder_blocks."):]
                n, tail = rest.split(".", 1)
                k2 = f"lang_encoder.transformer.blocks.{n}.decoder_layer.{tail}"
            if k2 in want:
                if tuple(v.shape) != tuple(want[k2][0]):
                    if k2 == "lang_encoder.transformer.wte.weight" and v.shape[1] == want[k2][0][1]:
                        pass                                      # vocab may be larger (resize_token_embeddings)
                    else:
                        raise RuntimeError(f"size mismatch for {k2}: {tuple(v.shape)} vs {want[k2][0]}")
                self._sd[k2] = v.detach()
                loaded.add(k2)
            else:
                unexpected.append(k)
        missing = [k for k in want if k not in self._sd]
        if strict and (missing or unexpected):
            raise RuntimeError(f"missing keys {missing[:5]}..., unexpected keys {unexpected[:5]}...")
        self._engine = None                                       # weights changed: rebuild on next use
        return missing, unexpected

    def set_precision(self, precision: str):
        """"fp16" (default) / "bf16": the product arithmetic on fp16 / bf16 operands; "fp32": fp32 activations end to end
        (csrc/precise.hip).  Rebuilds the engine on a change (the weights are stored in the arithmetic's format)."""
        precision = abi.resolve_precision(precision)
        if precision != self.precision:
            self.precision = precision
            self._engine = None
        return self

    def to(self, *a, **k):
        return self

    # eval_calvin.py:559-564 casts the module by `--precision`: .bfloat16() for bf16 / amp_bf16, .half() for fp16, .float() otherwise (the
    # README's `--precision fp32 --amp 1`: fp32 weights whose Linears then run under fp16 autocast, eval_utils.py:333).  Here the casts choose
    # the 16-bit format of the product arithmetic the same way: bf16 operands for a bf16 run, fp16 operands for the fp16 / fp32 + amp runs.
    # An engine the caller asked to be "fp32" (parity arithmetic) keeps it.
    def float(self):
        return self if self.precision == "fp32" else self.set_precision("fp16")

    def half(self):
        return self if self.precision == "fp32" else self.set_precision("fp16")

    def bfloat16(self):
        return self if self.precision == "fp32" else self.set_precision("bf16")

    # ---- exits bookkeeping (flamingo_mpt.py:268-306) ----------------------------------------------------------
    def get_all_exit_idx(self):
        return list(self.lm_exits.keys()) + [self.lang_encoder.config.n_layers - 1]

    def get_exit_num(self):
        return len(self.get_all_exit_idx())

    def set_all_exit_window_size(self, new_window_size):
        old = self.extra_exit.window_size if self.extra_exit is not None else self.window_size
        if self.extra_exit is not None:
            self.extra_exit.window_size = new_window_size
        return old

    def clear_all_exit_memory(self):
        self.engine.reset()
        self.extra_exit.history_memory = []

    # ---- forward ---------------------------------------------------------------------------------------------------
    def _sync_controller(self, ctl: ExitController):
        e = self.engine
        sig = (tuple(ctl.exit_id_list), ctl.max_layer, ctl.steps_per_stage, ctl._version, ctl.leq)
        if sig != self._ctl_sig:
            e.configure_exit(ctl.exit_id_list, ctl.max_layer + 1, ctl.steps_per_stage)
            assert e.ctl_max_layer == ctl.max_layer
            assert ctl.thresholds is not None, "Please set thresholds before calling forward"
            e.set_thresholds(ctl.threshold_list())
            e.leq = 1 if ctl.leq else 0
            if ctl.value_net is not None:
                e.thr_type = abi.THR_TYPES[ctl.value_net.threshold_type]
            self._ctl_sig = sig
        e.cur_step = ctl.cur_step

    def _run_llm(self, input_ids, attention_mask, exit_controller, exit_id):
        """Slow host loop (foreign controllers, or a direct ``lang_encoder(...)`` call): vision must already be encoded."""
        e, cfg = self.engine, self.cfg
        ids = input_ids.reshape(-1)
        T = ids.numel()
        e.ids[:T].copy_(ids)
        e._ids_tag = None                 # the engine's "same instruction tensor as last step" shortcut no longer holds
        e._mask_tag = None
        e._shadow_on = False              # ctl is zeroed below (shadow flag included)
        use_mask = False
        if attention_mask is not None:
            m = attention_mask.reshape(-1).to(torch.uint8)
            use_mask = bool((m == 0).any())
            e.key_mask[:T].copy_(m)
        if exit_id is not None and exit_id < 0:
            exit_id += cfg.n_layers
        e.ctl.zero_()
        e.enqueue_embed(T)
        hidden = ()
        pending = None
        b = -1
        for b in range(cfg.n_layers):
            pending = e.enqueue_llm_layer(b, T, pending, use_mask, finalize=True, ctl=False)
            hidden = hidden + (e.hidden[b, :T].unsqueeze(0),)
            if exit_id is not None and exit_id == b:
                break
            if exit_controller is not None:
                torch.cuda.current_stream().synchronize()
                if exit_controller(hidden, b):
                    break
        return CausalLMOutputWithPast(logits=[1.0], hidden_states=hidden, exit_layer=b)

    def step_env_batch(self, vision_x, lang_x, attention_mask, vision_gripper, exit_controller=None, exit_id=None, ensemble=False,
                       env_steps=None):
        """One control step of ALL n_envs environments (north_star: one env batch per rank): vision_x / vision_gripper
        (n_envs, ..., 3, S, S), lang_x / attention_mask (n_envs, T) right-padded.  Returns (pose (n_envs, 6), gripper prob (n_envs,),
        exit layers list); ``ensemble``: the mean of every environment's last two exit-check actions instead (eval_utils.py:457-461).  Every environment exits at its own layer on the device; LSTM state per environment.
        ``env_steps``: every environment's step index inside its own sub-task - the per-rollout ``set_timestep(step)`` of the reference
        (eval_utils.py:662-663) for a controller with ``steps_per_stage`` > 1 (value_net.py:285-286); default: the controller's cur_step."""
        e = self.engine
        ctl = getattr(exit_controller, "module", exit_controller)
        if exit_id is None:
            assert isinstance(ctl, ExitController), "env batches take the native controller (device-side exit gate)"
            self._sync_controller(ctl)
        r = e.step(vision_x, vision_gripper, lang_x, attention_mask, exit_id=exit_id, env_steps=env_steps)
        r = r if isinstance(r, list) else [r]
        exits = [x["exit_layer"] for x in r]
        A = e.A
        if self.layerwise_exit_eval:                           # every environment's action from the head of ITS exit layer (flamingo_mpt.py:450-457)
            if ensemble:
                raise NotImplementedError("use_action_ensemble averages extra_exit's actions; layerwise_exit_eval acts with the per-layer heads")
            e.h_state.zero_()
            e.c_state.zero_()
            e._head_state_changed()
            use_mask = attention_mask is not None and bool((attention_mask == 0).any())
            rows = e.layerwise_actions(exits, lang_x.shape[-1], use_mask)
            return torch.stack([x[:6 * A] for x in rows]), torch.stack([x[6 * A:7 * A] for x in rows]).squeeze(-1) if A == 1 else \
                torch.stack([x[6 * A:7 * A] for x in rows]), exits
        pk, gk = ("ens_pose", "ens_gripper") if ensemble else ("pose", "gripper")    # ensemble: get_ensemble_action per environment
        grip = torch.tensor([x[gk] for x in r]) if A == 1 else torch.stack([x[gk] for x in r])
        return torch.stack([x[pk] for x in r]), grip, exits

    def _forward_window(self, vision_x, lang_x, attention_mask, vision_gripper, with_gripper_logits=False, generator=None, rand_layers=None):
        """Window mode (flamingo_mpt.py:463-517 as ``generate_action_values`` calls it, value_net.py:375-385): the batch rows are
        bs * window_size frames (one instruction per row); every layer's hidden state is returned, the history fed to
        ``extra_exit`` comes from a random exit layer per (window, time step) ("sampling strategy 1", :485-497) and the head runs the
        windows as sequences from a zero state (action_head.py:588-595).  Returns (output, [], extra_exit_output,
        rand_layer_feat, rand_layer_indices) like the reference with ``return_in_feat=True``.  The head pools over ALL T rows of a
        right-padded instruction like the reference's (no mask reaches ``DeterministicDecoder``, action_head.py:519-520;
        tests/golden/deer_window_padded.npz).  The frames run as batch rows through
        the env-batch engine (``DeerEngine.window_hidden_states``), the windows as the environments of the head evaluations."""
        e, cfg = self.engine, self.cfg
        Wn = self.window_size
        F = vision_x.shape[0]
        assert F % Wn == 0, f"window mode: {F} batch rows are not a multiple of window_size {Wn}"
        bs, S = F // Wn, cfg.image_size
        T = lang_x.shape[-1]
        hid = e.window_hidden_states(vision_x.reshape(F, 3, S, S).to(e.dev, e.img_dtype),
                                     vision_gripper.reshape(F, 3, S, S).to(e.dev, e.img_dtype), lang_x.reshape(F, T).to(e.dev),
                                     attention_mask.reshape(F, T).to(e.dev) if attention_mask is not None else None)     # (F, L, T, d)
        hidden = tuple(hid[:, l] for l in range(cfg.n_layers))
        exit_ids = self.get_all_exit_idx()
        if rand_layers is None:
            idx = torch.randint(0, len(exit_ids), (bs, Wn), generator=generator)             # :485
            rand_layers = torch.tensor([exit_ids[int(i)] for i in idx.reshape(-1)]).reshape(bs, Wn)
        else:                                             # the history layers of a recorded reference call (parity tests)
            rand_layers = torch.as_tensor(rand_layers).reshape(bs, Wn).long().cpu()
            assert all(int(v) in exit_ids for v in rand_layers.reshape(-1)), "rand_layers must be exit layers"
        rand_feat = hid[torch.arange(F, device=hid.device), rand_layers.reshape(-1).to(hid.device)]      # (F, T, d)
        # extra_exit on the random-layer features, windows as sequences from a zero LSTM state: groups of <= 8 windows per evaluation
        G = max(1, min(8, e.MAX_ROWS // T, bs))   # groups of <= 8 windows per head evaluation
        w = e.sibling(G)
        rows = []
        rf = rand_feat.view(bs, Wn, T, cfg.d_model)
        for b0 in range(0, bs, G):
            gidx = [min(b0 + i, bs - 1) for i in range(G)]
            w.h_state.zero_()
            w.c_state.zero_()
            acts = [w._head_eval(rf[gidx, t].reshape(G * T, cfg.d_model).contiguous(), commit=True) for t in range(Wn)]
            rows.append(torch.stack(acts, dim=1)[: min(G, bs - b0)])                        # (g, W, 8)
        a = torch.cat(rows, dim=0)                                                          # (bs, W, 8)
        A = e.A
        pose, grip, glog = a[..., :6 * A], a[..., 6 * A:7 * A], a[..., 7 * A:8 * A]
        extra = (pose, (grip, glog)) if with_gripper_logits else (pose, grip)
        out = CausalLMOutputWithPast(logits=extra, hidden_states=hidden, exit_layer=cfg.n_layers - 1)
        return out, [], extra, rand_feat, rand_layers.to(hid.device)

    def forward(self, vision_x: torch.Tensor, lang_x: torch.Tensor, attention_mask: torch.Tensor = None, labels=None,
                use_cached_vision_x: bool = False, clear_conditioned_layers: bool = True, past_key_values=None,
                use_cache: bool = False, vision_gripper=None, state_tensor=None, return_feature=False, policy_mask=None,
                act=None, deterministic=False, with_gripper_logits=False, exit_id=None, dynamic_early_exit=False,
                exit_controller=None, return_in_feat=False, return_aggregate_feature=False, only_extra_exit=False,
                eval_time=False, no_backbone_grad=False):
        assert (vision_x is not None) or use_cached_vision_x, "Must provide either vision_x or use_cached_vision_x to True."
        if use_cached_vision_x:
            raise NotImplementedError("use_cached_vision_x is not used on the DeeR robot path")
        if vision_gripper is None:
            raise ValueError("vision_gripper is required (flamingo_mpt.py:355 clones it unconditionally)")
        if vision_x.ndim != 6 or vision_x.shape[1] != 1:
            raise NotImplementedError("vision_x must be (B, 1, 1, 3, S, S)")
        assert vision_x.shape[2] == 1, "Only single frame supported"
        if self.use_state and exit_id is None and not (dynamic_early_exit and exit_controller is not None):
            raise NotImplementedError("window-mode calibration with use_state is not implemented")
        if exit_id is None and not (dynamic_early_exit and exit_controller is not None):
            # the all-exits branch (flamingo_mpt.py:463-517): inference-side use = window-mode calibration (value_net.py:375-385)
            if not (only_extra_exit and return_in_feat):
                raise NotImplementedError("the training losses of the all-exits branch are out of scope; the calibration call "
                                          "(only_extra_exit=True, return_in_feat=True, value_net.py:375-385) is supported")
            return self._forward_window(vision_x, lang_x, attention_mask, vision_gripper, with_gripper_logits)
        if vision_x.shape[0] != 1:
            raise NotImplementedError("step mode takes one frame pair (B=1); batches of frames go through the window-mode call")
        if self.use_state and exit_id is None:
            # the reference itself fails here: ActionValueNet.forward calls exit_head(feats[i]) without state_tensor (value_net.py:122-129)
            raise NotImplementedError("use_state + dynamic_early_exit: the reference raises TypeError (value_net.py:122-129); use a static exit_id")
        e, cfg = self.engine, self.cfg
        ctl = getattr(exit_controller, "module", exit_controller)
        native = isinstance(ctl, ExitController)
        t0 = None
        if eval_time:
            torch.cuda.synchronize()
            import time
            t0 = time.time()
            e._time_stages = True                                   # GPU time of the vision tower / trunk + exit checks of this step
        if exit_id is not None or native:
            if native and exit_id is None:
                self._sync_controller(ctl)
            r = e.step(vision_x, vision_gripper, lang_x, attention_mask, exit_id=exit_id, state=state_tensor if self.use_state else None)
            exit_layer = r["exit_layer"]
            if native and exit_id is None:
                ctl.cur_exit_id = int(e.ctl_host[abi.CTL_CUR_EXIT_ID])
                if ctl.value_net is not None:                        # ActionValueNet.get_ensemble_action (eval_utils.py:460)
                    ctl.value_net._ensemble = (r["ens_pose"], r["ens_gripper"], r["ens_count"])
        else:
            # foreign controller: the reference's host loop on our kernels
            T, use_mask = e.load_inputs(vision_x, vision_gripper, lang_x, attention_mask)
            e.enqueue_vision()
            out = self._run_llm(lang_x, attention_mask, exit_controller, None)
            exit_layer = out.exit_layer
            self.extra_exit(out.hidden_states[exit_layer], update_hidden_state=True)      # flamingo_mpt.py:459
            e.ctl_host.copy_(e.ctl)
        if eval_time:
            torch.cuda.synchronize()
            e._time_stages = False
            self.forward_time = time.time() - t0
            st = e.last_stage_ms
            # the reference times the lang_encoder call only (flamingo_mpt.py:386-417); the vision tower is reported beside it
            self.llm_inference_time = st["llm_and_exit_checks"] / 1e3 if st else self.forward_time
            self.vision_time = st["vision"] / 1e3 if st else float("nan")
        T = lang_x.reshape(-1).numel()
        A = e.A
        fast = self.host_outputs and (exit_id is not None or native)
        lw_row = None
        if self.layerwise_exit_eval:
            # flamingo_mpt.py:450-457: the action comes from the exit layer's OWN head on hidden_states[exit_layer] (own LSTM history,
            # update_hidden_state=True); extra_exit only decided where to stop - the reference never commits its state here (nobody calls
            # update_exit_hidden_state, value_net.py:88-90), and a static exit_id never touches it
            if not (exit_id is not None or native):
                raise NotImplementedError("layerwise_exit_eval with a foreign exit controller")
            e.h_state.zero_()
            e.c_state.zero_()
            e._head_state_changed()
            lw_row = e.layerwise_actions([exit_layer], T)[0]
        if lw_row is not None:
            a = lw_row
            hidden = _LazyHidden(e, exit_layer, T) if fast else None
        elif fast:
            if A > 1:
                a = torch.cat([r["pose"], r["gripper"], r["gripper_logit"]])
            else:
                a = torch.cat([r["pose"], torch.tensor([r["gripper"], r["gripper_logit"]], dtype=torch.float32)])   # from the pinned verdict block
            hidden = _LazyHidden(e, exit_layer, T)
        else:
            if (exit_id is not None or native) and A > 1:
                a = e.act_ext[0, 1, :8 * A].clone()
            elif exit_id is not None or native:
                a = e.ctl.view(torch.float32)[abi.CTL_OUT_ACTION: abi.CTL_OUT_ACTION + 8].clone()
            else:
                a = e.action_dbg[0].clone()
            # every layer's output up to the exit is real (the spine writes hidden[i] of a non-exit layer with the first row op
            # of layer i+1); ONE copy, so that the tuple survives the next step overwriting the engine's buffers
            hs = e.hidden[: exit_layer + 1, :T].clone()
            hidden = tuple(hs[i].unsqueeze(0) for i in range(exit_layer + 1))
        if hidden is None:
            hs = e.hidden[: exit_layer + 1, :T].clone()
            hidden = tuple(hs[i].unsqueeze(0) for i in range(exit_layer + 1))
        assert len(hidden) == exit_layer + 1                                                # flamingo_mpt.py:458
        pose, grip = a[:6 * A].view(1, 1, 6 * A), a[6 * A:7 * A].view(1, 1, A)               # action_head.py:472-473: A actions per call
        logits = (pose, (grip, a[7 * A:8 * A].view(1, 1, A))) if with_gripper_logits else (pose, grip)
        vis = e.vis_x_f32.view(1, 1, cfg.n_media, cfg.vit_width)
        media_locations = lang_x.reshape(1, -1) == self.media_token_id                      # flamingo_lm.py:211
        for l in self.lang_encoder._get_decoder_layers():
            l.condition_vis_x(vis)                                                          # flamingo_mpt.py:665-666
            l.condition_media_locations(media_locations)
        return CausalLMOutputWithPast(logits=logits, hidden_states=hidden, exit_layer=exit_layer)
