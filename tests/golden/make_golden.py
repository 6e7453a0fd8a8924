#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running the REFERENCE's own modules (imported
from /root/reference where they lie; see ref_import.py) on seeded inputs.  Run in the build
container only:  ``python tests/golden/make_golden.py``.

A fixture is data: inputs + expected outputs (+ the config/seed that regenerates the weights with
``deer_vla_amd.synthetic``).  Weights are NOT stored: every reference module is loaded with
``make_synthetic_state``-style tensors through ``load_state_dict`` - which at the same time pins this
repo's parameter inventory (names + shapes) against the reference's state-dict keys.

Cases (SURVEY.md §8c golden-vector plan):
  perceiver.npz        PerceiverResampler                      helpers.py:68-132
  xattn.npz            GatedCrossAttentionBlock (3 mask cases) helpers.py:136-279
  flamingo_layer.npz   FlamingoLayer ordering                  flamingo_lm.py:46-83
  head_ln.npz/head_plain.npz  DeterministicDecoder step sequence with update_hidden_state
                       interleaving + a 12-long window call    action_head.py:408-611
  controller_*.npz     ActionValueNet + ExitController traces  value_net.py:72-297
  thresholds.npz       ExitController.set_threshold solver     value_net.py:185-272
  mosaic_loop.npz      MosaicGPT.forward multi-exit loop       mosaic_gpt_3b.py:274-449 (blocks = stand-ins)
  deer_forward.npz     MPTFlamingo.forward, static exit_id and dynamic exit over several steps
                       flamingo_mpt.py:308-461 (config[0] of BASELINE.json: fixed exit, B=1, CPU)
  hf_mpt_block.npz     transformers MptBlock cross-check of the un-vendored block arithmetic
  hf_clip_vit.npz      transformers CLIPVisionModel cross-check of the un-vendored ViT arithmetic
"""
from __future__ import annotations

import json
import os
import sys
import types
import zlib

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

import ref_import  # noqa: E402

ref_import.install_stubs()

from deer_vla_amd.config import DeerConfig  # noqa: E402
from deer_vla_amd import synthetic as syn  # noqa: E402
from oracle import deer_oracle as orc  # noqa: E402  (only to host the ViT inside the reference's MPTFlamingo)

from open_flamingo.src.helpers import PerceiverResampler, GatedCrossAttentionBlock  # noqa: E402
from open_flamingo.src.flamingo_lm import FlamingoLayer, FlamingoLMMixin  # noqa: E402
from open_flamingo.src.utils import extend_instance  # noqa: E402
from robot_flamingo.models.action_head import DeterministicDecoder  # noqa: E402
from robot_flamingo.models.value_net import ActionValueNet, ExitController  # noqa: E402
from robot_flamingo.models.flamingo_mpt import MPTFlamingo  # noqa: E402

torch.set_grad_enabled(False)


def seeded(name: str, shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed((zlib.crc32(name.encode()) + 7919 * seed) & 0x7FFFFFFF)
    return torch.randn(shape, generator=g) * scale


def sub_state(sd, prefix):
    return {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}


def load_strict(module: nn.Module, sd):
    """strict load: the reference module's keys/shapes must equal this repo's inventory."""
    ref_keys = {k: tuple(v.shape) for k, v in module.state_dict().items()}
    my_keys = {k: tuple(v.shape) for k, v in sd.items()}
    assert ref_keys == my_keys, (sorted(set(ref_keys) ^ set(my_keys)),
                                 [(k, ref_keys[k], my_keys[k]) for k in ref_keys if k in my_keys and ref_keys[k] != my_keys[k]])
    module.load_state_dict(sd, strict=True)


def save(name, cfg: DeerConfig, seed: int, **arrays):
    out = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrays.items()}
    out["cfg_json"] = np.frombuffer(json.dumps(cfg.to_dict()).encode(), dtype=np.uint8)
    out["seed"] = np.asarray(seed)
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **out)
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB  keys={sorted(arrays)}")


# ------------------------------------------------------------------------------------------------
def small_cfg(**kw):
    base = dict(image_size=28, patch_size=14, vit_width=32, vit_layers=1, vit_heads=2, vit_mlp=64,
                perc_depth=2, perc_heads=2, perc_dim_head=16, perc_latents=8,
                d_model=32, n_heads=2, n_layers_total=6, vocab_size=100, media_token_id=98, eoc_token_id=97,
                xattn_heads=2, xattn_dim_head=16, early_exit_layer=5, head_hidden=16)
    base.update(kw)
    return DeerConfig(**base)


def gen_perceiver():
    cfg, seed = small_cfg(), 1
    sd = syn.make_synthetic_state(cfg, seed)
    m = PerceiverResampler(dim=cfg.vit_width, depth=cfg.perc_depth, dim_head=cfg.perc_dim_head,
                           heads=cfg.perc_heads, num_latents=cfg.perc_latents).eval()
    load_strict(m, sub_state(sd, "perceiver."))
    x = seeded("perc.x", (2, 1, 1, 10, cfg.vit_width))
    save("perceiver.npz", cfg, seed, x=x, out=m(x))


def gen_xattn():
    cfg, seed = small_cfg(), 2
    sd = syn.make_synthetic_state(cfg, seed)
    pfx = "lang_encoder.transformer.blocks.0.gated_cross_attn_layer."
    m = GatedCrossAttentionBlock(dim=cfg.d_model, dim_visual=cfg.vit_width, dim_head=cfg.xattn_dim_head,
                                 heads=cfg.xattn_heads).eval()
    s = sub_state(sd, pfx)
    s["attn_gate"] = torch.tensor([0.5])
    s["ff_gate"] = torch.tensor([-0.3])
    load_strict(m, s)
    x = seeded("xattn.x", (2, 6, cfg.d_model))
    media1 = seeded("xattn.media1", (2, 1, 8, cfg.vit_width))
    media2 = seeded("xattn.media2", (2, 2, 4, cfg.vit_width))
    loc_a = torch.zeros(2, 6, dtype=torch.bool)
    loc_a[:, 0] = True                                   # DeeR step mode: <image> is token 0
    loc_b = torch.zeros(2, 6, dtype=torch.bool)
    loc_b[0, 0] = True
    loc_b[1, 2] = True                                   # tokens 0,1 of row 1 precede any media -> zeroed rows
    loc_c = torch.zeros(2, 6, dtype=torch.bool)
    loc_c[:, 0] = True
    loc_c[:, 3] = True                                   # two media, only_attend_immediate_media
    save("xattn.npz", cfg, seed, x=x, media1=media1, media2=media2,
         loc_a=loc_a, loc_b=loc_b, loc_c=loc_c, attn_gate=0.5, ff_gate=-0.3,
         out_a=m(x, media1, media_locations=loc_a), out_b=m(x, media1, media_locations=loc_b),
         out_c=m(x, media2, media_locations=loc_c),
         out_cached=m(x, media1, media_locations=loc_a, use_cached_media=True))


class _ToyDecoder(nn.Module):
    """A decoder_layer stand-in for the ordering test: x -> tanh(x W^T) (returns a tuple like GPTBlock)."""

    def __init__(self, w):
        super().__init__()
        self.w = nn.Parameter(w)

    def forward(self, x, attention_mask=None, **kw):
        return torch.tanh(x @ self.w.t()), None


def gen_flamingo_layer():
    cfg, seed = small_cfg(), 3
    sd = syn.make_synthetic_state(cfg, seed)
    pfx = "lang_encoder.transformer.blocks.0.gated_cross_attn_layer."
    xa = GatedCrossAttentionBlock(dim=cfg.d_model, dim_visual=cfg.vit_width, dim_head=cfg.xattn_dim_head,
                                  heads=cfg.xattn_heads).eval()
    load_strict(xa, sub_state(sd, pfx))
    w = seeded("toy.w", (cfg.d_model, cfg.d_model), scale=cfg.d_model ** -0.5)
    layer = FlamingoLayer(xa, _ToyDecoder(w)).eval()
    x = seeded("fl.x", (1, 5, cfg.d_model))
    media = seeded("fl.media", (1, 1, 8, cfg.vit_width))
    loc = torch.zeros(1, 5, dtype=torch.bool)
    loc[:, 0] = True
    layer.condition_vis_x(media)
    layer.condition_media_locations(loc)
    layer.condition_use_cached_media(False)
    out = layer(x, attention_mask=None)[0]
    save("flamingo_layer.npz", cfg, seed, x=x, media=media, loc=loc, toy_w=w, out=out)


def build_ref_head(cfg, sd, prefix="extra_exit."):
    m = DeterministicDecoder(cfg.d_model, cfg.window_size, 0.0, 0.0, "layerwise", cfg.mlp_layernorm,
                             cfg.lstm_layernorm, cfg.mlp_num_hidden_layers, hidden_size=cfg.head_hidden,
                             lstm_num_layers=cfg.lstm_num_layers, pooling=cfg.pooling, use_state=cfg.use_state).eval()
    load_strict(m, sub_state(sd, prefix))
    return m


def gen_head(name, **kw):
    cfg, seed = small_cfg(**kw), 4
    sd = syn.make_synthetic_state(cfg, seed)
    m = build_ref_head(cfg, sd)
    m.window_size = 1                                   # ModelWrapper.step: set_all_exit_window_size(1)
    T = 7
    feats = seeded("head.feats", (6, 1, T, cfg.d_model))
    upd = [False, False, True, False, True, True]       # controller evals (False) vs committing eval (True)
    poses, grips, hs, cs = [], [], [], []
    for t in range(6):
        a, g = m(feats[t], update_hidden_state=upd[t])
        poses.append(a)
        grips.append(g)
        if m.hidden_state is None:
            hs.append(torch.zeros(cfg.lstm_num_layers, 1, cfg.head_hidden))
            cs.append(torch.zeros(cfg.lstm_num_layers, 1, cfg.head_hidden))
        else:
            hs.append(m.hidden_state[0].clone())
            cs.append(m.hidden_state[1].clone())
    # window mode: (bs*12, T, d) -> 12-step LSTM from h_0=None (action_head.py:588-595)
    m2 = build_ref_head(cfg, sd)
    m2.window_size = cfg.window_size
    wfeat = seeded("head.wfeat", (2 * cfg.window_size, T, cfg.d_model))
    wa, wg = m2(wfeat)
    m2.last_action = True
    wa_last, wg_last = m2(wfeat)
    _, (wgp, wgl) = m2(wfeat, with_gripper_logits=True)
    save(name, cfg, seed, feats=feats, upd=np.asarray(upd), pose=torch.stack(poses), grip=torch.stack(grips),
         h=torch.stack(hs), c=torch.stack(cs), wfeat=wfeat, wpose=wa, wgrip=wg, wpose_last=wa_last,
         wgrip_last=wg_last, wgrip_logits=wgl)


def gen_head_state(name="head_state.npz"):
    """DeterministicDecoder(use_state=True) (action_head.py:443-453,524-536): the robot state - arm pose robot_obs[:6] and the
    gripper opening robot_obs[-1] in {-1, +1} (eval_utils.py:324-332 hands over all 15 values) - is embedded and ADDED to the pooled
    feature; step mode with LSTM carry."""
    cfg, seed = small_cfg(use_state=True), 4
    sd = syn.make_synthetic_state(cfg, seed)
    m = build_ref_head(cfg, sd)
    m.window_size = 1
    T, n = 7, 6
    feats = seeded("head.feats", (n, 1, T, cfg.d_model))
    state = seeded("head.state", (n, 1, 1, 1, 15))
    state[..., -1] = torch.tensor([1.0, -1.0, -1.0, 1.0, 1.0, -1.0]).view(n, 1, 1, 1)
    upd = [False, True, True, False, True, True]
    poses, grips = [], []
    for t in range(n):
        a, g = m(feats[t], state_tensor=state[t], update_hidden_state=upd[t])
        poses.append(a)
        grips.append(g)
    save(name, cfg, seed, feats=feats, state=state, upd=np.asarray(upd), pose=torch.stack(poses), grip=torch.stack(grips))


def gen_deer_forward_variant(name, use_state=False, sep_resampler=False, layerwise=False, multi_step_action=1):
    """The reference's own MPTFlamingo.forward with ``use_state`` / ``sep_resampler`` (flamingo_mpt.py:132-136,656-659; the state
    reaches ONLY the action head on the post-fusion path: ``_encode_multi_vision_post_fusion`` is called without it, :381).  Static
    exits; with sep_resampler also a dynamic episode (with use_state the reference's ``ActionValueNet`` calls the head without a
    state tensor, value_net.py:122-129, and raises - dynamic exit does not exist for that variant).
    layerwise: ``multi_exit=True, layerwise_exit_eval=True`` (eval_calvin.py:530,539): the action of exit layer k comes from that layer's
    own head ``lm_exits[k]`` / ``lm_head`` (flamingo_mpt.py:450-457), each with its own LSTM history; the exit decision stays with
    ``extra_exit`` (whose hidden state nobody commits in this mode).  multi_step_action: ``6 A`` pose + ``A`` gripper outputs per
    head call (action_head.py:472-473)."""
    _dist_init()
    cfg, seed = llm_cfg(use_state=use_state, sep_resampler=sep_resampler, layerwise_exit_eval=layerwise, multi_step_action=multi_step_action), 7
    sd = syn.make_synthetic_state(cfg, seed, bf16_round=True)
    lm, mod = build_ref_lang_encoder(cfg, sd)
    extend_instance(lm, FlamingoLMMixin)
    lm.set_decoder_layers_attr_name("transformer.blocks")
    venc_mod = nn.Module()
    venc_mod.visual = _OracleVisual(cfg, sd)
    model = MPTFlamingo(venc_mod, lm, cfg.eoc_token_id, cfg.media_token_id, vis_dim=cfg.vit_width,
                        cross_attn_every_n_layers=cfg.cross_attn_every_n_layers, window_size=cfg.window_size,
                        use_gripper=True, fusion_mode="post", llm="mpt_dolly_3b", pooling="max", use_state=use_state,
                        sep_resampler=sep_resampler, early_exit_layer=cfg.early_exit_layer, multi_exit=bool(layerwise),
                        layerwise_exit_eval=bool(layerwise), multi_step_action=multi_step_action,
                        exit_interval=cfg.exit_interval, mlp_layernorm=True, lstm_layernorm=True, mlp_num_hidden_layers=2,
                        lstm_num_layers=4).eval()
    sd_model = {k: v for k, v in sd.items() if not k.startswith("vision_encoder.")}
    ref_keys = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    for k, v in sd_model.items():
        assert k in ref_keys and ref_keys[k] == tuple(v.shape), (k, tuple(v.shape), ref_keys.get(k))
    missing, unexpected = model.load_state_dict(sd_model, strict=False)
    assert not unexpected, unexpected
    if layerwise:                                            # every per-layer head's parameters are this repo's inventory (config.layerwise_heads)
        assert not any(k.startswith(("lm_exit_modules.", "lm_head.")) for k in missing), missing
        assert sorted(model.lm_exits.keys()) == [e for _, e in cfg.layerwise_heads()][:-1]
    if sep_resampler:
        assert not any(k.startswith("perceiver_gripper.") for k in missing), missing
    if use_state:                                            # state_fc (a media token on other fusion paths) is never applied here
        assert all(not k.startswith("extra_exit.embed") for k in missing), missing
    T, S, n_steps = 8, cfg.image_size, 6
    ids = torch.tensor([[cfg.media_token_id, 5, 17, 3, 42, 8, cfg.eoc_token_id, 0]])
    mask = torch.ones(1, T, dtype=torch.bool)
    rgb = seeded("deer.rgb", (n_steps, 1, 1, 1, 3, S, S))
    grip = seeded("deer.grip", (n_steps, 1, 1, 1, 3, S, S))
    state = seeded("deer.state", (n_steps, 1, 1, 1, 15))
    state[..., -1] = torch.tensor([1.0, -1.0, 1.0, 1.0, -1.0, -1.0]).view(n_steps, 1, 1, 1)
    model.set_all_exit_window_size(1)
    outs = {}
    for eid in (3, 4):                                       # static exits, LSTM carried over the steps
        model.clear_all_exit_memory()
        ps, gs = [], []
        for s_ in range(n_steps):
            o = model(vision_x=rgb[s_], lang_x=ids, attention_mask=mask, vision_gripper=grip[s_], state_tensor=state[s_],
                      return_feature=True, deterministic=True, exit_id=eid, dynamic_early_exit=False, exit_controller=None)
            ps.append(o.logits[0])
            gs.append(o.logits[1])
        outs[f"static{eid}_pose"], outs[f"static{eid}_grip"] = torch.stack(ps), torch.stack(gs)
    outs["vis_x"] = lm._get_decoder_layers()[0].vis_x
    if not use_state:
        exit_ids = model.get_all_exit_idx()
        model.clear_all_exit_memory()
        vn = _RecVN(exit_list=exit_ids, exit_head=model.extra_exit, interval=cfg.exit_interval, window_size=cfg.window_size,
                    threshold_type="L2")
        ctl = ExitController(vn, exit_id_list=exit_ids, steps_per_stage=1, leq=True, exit_dist="exp", max_layer=12)
        real = len([x for x in exit_ids if x <= ctl.max_layer])
        ctl._set_threshold_value([-1.0] * real)
        for s_ in range(n_steps):
            ctl.set_timestep(s_)
            model(vision_x=rgb[s_], lang_x=ids, attention_mask=mask, vision_gripper=grip[s_], state_tensor=state[s_], return_feature=True,
                  deterministic=True, exit_id=None, dynamic_early_exit=True, exit_controller=ctl)
        thr = [gap_threshold([v for (i, v) in vn.rec if i == e], 0.2, 0.8) for e in exit_ids[:real]]
        thr[-1] = 1e5
        model.clear_all_exit_memory()
        vn.reset_actions()
        vn.rec = []
        ctl._set_threshold_value(thr)
        ex, ps, gs = [], [], []
        for s_ in range(n_steps):
            ctl.set_timestep(s_)
            o = model(vision_x=rgb[s_], lang_x=ids, attention_mask=mask, vision_gripper=grip[s_], state_tensor=state[s_], return_feature=True,
                      deterministic=True, exit_id=None, dynamic_early_exit=True, exit_controller=ctl)
            ex.append(o.exit_layer)
            ps.append(o.logits[0])
            gs.append(o.logits[1])
        print(f"  {name} dyn: exits={ex}")
        outs.update(dyn_thr=np.asarray(thr), dyn_exit=np.asarray(ex), dyn_pose=torch.stack(ps), dyn_grip=torch.stack(gs), dyn_max_layer=12,
                    dyn_rec_layer=np.asarray([i for i, _ in vn.rec]), dyn_rec_delta=np.asarray([v for _, v in vn.rec]))
    else:                                                    # pin the reference's behaviour: dynamic exit + use_state raises
        vn = ActionValueNet(exit_list=model.get_all_exit_idx(), exit_head=model.extra_exit, interval=cfg.exit_interval,
                            window_size=cfg.window_size, threshold_type="L2")
        ctl = ExitController(vn, exit_id_list=model.get_all_exit_idx(), steps_per_stage=1, leq=True, exit_dist="exp", max_layer=12)
        ctl._set_threshold_value([1e5] * len([x for x in model.get_all_exit_idx() if x <= ctl.max_layer]))
        model.clear_all_exit_memory()
        ctl.set_timestep(0)
        try:
            model(vision_x=rgb[0], lang_x=ids, attention_mask=mask, vision_gripper=grip[0], state_tensor=state[0], return_feature=True,
                  deterministic=True, exit_id=None, dynamic_early_exit=True, exit_controller=ctl)
            outs["dynamic_raises"] = np.asarray(0)
        except TypeError as e:
            outs["dynamic_raises"] = np.asarray(1)
            print(f"  {name}: the reference's dynamic exit with use_state raises {type(e).__name__}: {e}")
    save(name, cfg, seed, ids=ids, mask=mask, rgb=rgb, grip=grip, state=state, bf16_round=1, **outs)


def gen_deer_forward_r6(name, cfg_kw=None, thr_types=("L2",), exit_id_list=None, n_steps=8):
    """Round 6: the reference's own MPTFlamingo.forward for the head / criterion variants its constructors take and that only the CPU
    oracle was pinned on so far: plain ``nn.LSTM`` + MLP heads without LayerNorm (action_head.py:72-79,86-116), ``pooling='avg'``
    (:480-483) with three hidden layers, and one dynamic-exit episode (LSTM carried, ModelWrapper.step protocol) per ``threshold_type`` of
    ``ActionValueNet.get_delta`` (value_net.py:105-117: mean / L2 / max / cosine).  ``exit_id_list``: the controller's exit list when it is
    not the model's own (consecutive exits 1, 2, 3, 4: ``ExitController`` takes any list, value_net.py:164-173).  Thresholds sit in the
    widest gap of each exit's never-exit deltas; recorded per type as ``<type>_*``."""
    _dist_init()
    cfg, seed = llm_cfg(**(cfg_kw or {})), 7
    sd = syn.make_synthetic_state(cfg, seed, bf16_round=True)
    lm, mod = build_ref_lang_encoder(cfg, sd)
    extend_instance(lm, FlamingoLMMixin)
    lm.set_decoder_layers_attr_name("transformer.blocks")
    venc_mod = nn.Module()
    venc_mod.visual = _OracleVisual(cfg, sd)
    model = MPTFlamingo(venc_mod, lm, cfg.eoc_token_id, cfg.media_token_id, vis_dim=cfg.vit_width,
                        cross_attn_every_n_layers=cfg.cross_attn_every_n_layers, window_size=cfg.window_size,
                        use_gripper=True, fusion_mode=cfg.fusion_mode, llm="mpt_dolly_3b", pooling=cfg.pooling,
                        early_exit_layer=cfg.early_exit_layer, multi_exit=False, exit_interval=cfg.exit_interval,
                        mlp_layernorm=cfg.mlp_layernorm, lstm_layernorm=cfg.lstm_layernorm,
                        mlp_num_hidden_layers=cfg.mlp_num_hidden_layers, lstm_num_layers=cfg.lstm_num_layers).eval()
    assert model.get_all_exit_idx() == cfg.exit_ids()
    sd_model = {k: v for k, v in sd.items() if not k.startswith("vision_encoder.")}
    ref_keys = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    for k, v in sd_model.items():
        assert k in ref_keys and ref_keys[k] == tuple(v.shape), (k, tuple(v.shape), ref_keys.get(k))
    missing, unexpected = model.load_state_dict(sd_model, strict=False)
    assert not unexpected, unexpected
    assert not any(k.startswith("extra_exit.") for k in missing), missing
    T, S = 8, cfg.image_size
    ids = torch.tensor([[cfg.media_token_id, 5, 17, 3, 42, 8, cfg.eoc_token_id, 0]])
    mask = torch.ones(1, T, dtype=torch.bool)
    rgb = seeded("deer.rgb", (n_steps, 1, 1, 1, 3, S, S))
    grip = seeded("deer.grip", (n_steps, 1, 1, 1, 3, S, S))
    state = torch.zeros(1, 1, 1, 15)
    model.set_all_exit_window_size(1)
    outs = {}
    for eid in (3, 4):                                       # static exits, LSTM carried over the steps
        model.clear_all_exit_memory()
        ps, gs = [], []
        for s_ in range(n_steps):
            o = model(vision_x=rgb[s_], lang_x=ids, attention_mask=mask, vision_gripper=grip[s_], state_tensor=state,
                      return_feature=True, deterministic=True, exit_id=eid, dynamic_early_exit=False, exit_controller=None)
            ps.append(o.logits[0])
            gs.append(o.logits[1])
        outs[f"static{eid}_pose"], outs[f"static{eid}_grip"] = torch.stack(ps), torch.stack(gs)
    exit_ids = list(exit_id_list) if exit_id_list is not None else model.get_all_exit_idx()
    for ttype in thr_types:
        model.clear_all_exit_memory()
        vn = _RecVN(exit_list=exit_ids, exit_head=model.extra_exit, interval=cfg.exit_interval, window_size=cfg.window_size,
                    threshold_type=ttype)
        ctl = ExitController(vn, exit_id_list=exit_ids, steps_per_stage=1, leq=True, exit_dist="exp", max_layer=12)
        real = len([x for x in exit_ids if x <= ctl.max_layer])
        ctl._set_threshold_value([-1.0] * real)              # never-exit pass (not recorded): the scale of every exit's deltas
        for s_ in range(n_steps):
            ctl.set_timestep(s_)
            model(vision_x=rgb[s_], lang_x=ids, attention_mask=mask, vision_gripper=grip[s_], state_tensor=state, return_feature=True,
                  deterministic=True, exit_id=None, dynamic_early_exit=True, exit_controller=ctl)
        thr = [gap_threshold([v for (i, v) in vn.rec if i == e], 0.2, 0.8) for e in exit_ids[:real]]
        thr[-1] = 1e5

        def episode(thr):
            model.clear_all_exit_memory()
            vn.reset_actions()
            vn.rec = []
            ctl._set_threshold_value(thr)
            ex, ps, gs = [], [], []
            for s_ in range(n_steps):
                ctl.set_timestep(s_)
                o = model(vision_x=rgb[s_], lang_x=ids, attention_mask=mask, vision_gripper=grip[s_], state_tensor=state, return_feature=True,
                          deterministic=True, exit_id=None, dynamic_early_exit=True, exit_controller=ctl)
                ex.append(o.exit_layer)
                ps.append(o.logits[0])
                gs.append(o.logits[1])
            thr_by = dict(zip(exit_ids, thr))
            marg = min([abs(v - thr_by[i]) / thr_by[i] for (i, v) in vn.rec if thr_by[i] < 1e4] or [1.0])
            return ex, ps, gs, list(vn.rec), marg

        # on-policy refinement (the deltas an exit sees depend on where the earlier steps exited: the LSTM carries the history): re-pick
        # every threshold in the widest gap of the deltas the policy visits, keep the set with the largest minimum margin
        best = None
        for _ in range(6):
            ex, ps, gs, rec, marg = episode(thr)
            if best is None or marg > best[0]:
                best = (marg, list(thr))
            if marg > 0.08:
                break
            new_thr = list(thr)
            for k, e in enumerate(exit_ids[:real - 1]):
                vals = [v for (i, v) in rec if i == e]
                if len(vals) >= 4:
                    new_thr[k] = gap_threshold(vals, 0.15, 0.85)
            if new_thr == thr:
                break
            thr = new_thr
        thr = best[1]
        ex, ps, gs, rec, marg = episode(thr)
        vn.rec = rec
        print(f"  {name} {ttype}: exits={ex} thr={np.round(thr, 5).tolist()} min margin {marg:.3f}")
        outs.update({f"{ttype}_thr": np.asarray(thr), f"{ttype}_exit": np.asarray(ex), f"{ttype}_pose": torch.stack(ps),
                     f"{ttype}_grip": torch.stack(gs), f"{ttype}_rec_layer": np.asarray([i for i, _ in vn.rec]),
                     f"{ttype}_rec_delta": np.asarray([v for _, v in vn.rec]), f"{ttype}_min_margin": marg})
    outs["vis_x"] = lm._get_decoder_layers()[0].vis_x            # the media tokens of the last step (b, T, n_media, D)
    save(name, cfg, seed, ids=ids, mask=mask, rgb=rgb, grip=grip, bf16_round=1, exit_ids=np.asarray(exit_ids),
         thr_types=np.frombuffer(",".join(thr_types).encode(), dtype=np.uint8), **outs)


def gen_exit_interval_1():
    """exit_interval = 1 makes layer 0 an exit (flamingo_mpt.py:239-250): the reference's dynamic exit then fails in
    ``ActionValueNet.forward`` (``assert i > 0, 'the first layer similarity is not implemented yet'``, value_net.py:119).  Pinned as data:
    the exception type; static exits work."""
    _dist_init()
    cfg, seed = llm_cfg(exit_interval=1), 7
    sd = syn.make_synthetic_state(cfg, seed, bf16_round=True)
    lm, mod = build_ref_lang_encoder(cfg, sd)
    extend_instance(lm, FlamingoLMMixin)
    lm.set_decoder_layers_attr_name("transformer.blocks")
    venc_mod = nn.Module()
    venc_mod.visual = _OracleVisual(cfg, sd)
    model = MPTFlamingo(venc_mod, lm, cfg.eoc_token_id, cfg.media_token_id, vis_dim=cfg.vit_width,
                        cross_attn_every_n_layers=cfg.cross_attn_every_n_layers, window_size=cfg.window_size,
                        use_gripper=True, fusion_mode="post", llm="mpt_dolly_3b", pooling="max",
                        early_exit_layer=cfg.early_exit_layer, multi_exit=False, exit_interval=1,
                        mlp_layernorm=True, lstm_layernorm=True, mlp_num_hidden_layers=2, lstm_num_layers=4).eval()
    assert model.get_all_exit_idx() == cfg.exit_ids() == [0, 1, 2, 3, 4]
    model.load_state_dict({k: v for k, v in sd.items() if not k.startswith("vision_encoder.")}, strict=False)
    S = cfg.image_size
    ids = torch.tensor([[cfg.media_token_id, 5, 17, 3, 42, 8, cfg.eoc_token_id, 0]])
    mask = torch.ones(1, 8, dtype=torch.bool)
    rgb, grip = seeded("deer.rgb", (1, 1, 1, 1, 3, S, S)), seeded("deer.grip", (1, 1, 1, 1, 3, S, S))
    model.set_all_exit_window_size(1)
    vn = ActionValueNet(exit_list=model.get_all_exit_idx(), exit_head=model.extra_exit, interval=1, window_size=cfg.window_size, threshold_type="L2")
    ctl = ExitController(vn, exit_id_list=model.get_all_exit_idx(), steps_per_stage=1, leq=True, exit_dist="exp", max_layer=12)
    ctl._set_threshold_value([1e5] * 5)
    ctl.set_timestep(0)
    try:
        model(vision_x=rgb[0], lang_x=ids, attention_mask=mask, vision_gripper=grip[0], state_tensor=torch.zeros(1, 1, 1, 15), return_feature=True,
              deterministic=True, exit_id=None, dynamic_early_exit=True, exit_controller=ctl)
        raised = ""
    except AssertionError as e:
        raised = f"AssertionError: {e}"
    model.clear_all_exit_memory()
    o = model(vision_x=rgb[0], lang_x=ids, attention_mask=mask, vision_gripper=grip[0], state_tensor=torch.zeros(1, 1, 1, 15), return_feature=True,
              deterministic=True, exit_id=0, dynamic_early_exit=False, exit_controller=None)
    print(f"  exit_interval=1: dynamic exit raises {raised!r}; static exit 0 pose {o.logits[0].reshape(-1)[:3].tolist()}")
    save("deer_forward_int1.npz", cfg, seed, ids=ids, mask=mask, rgb=rgb, grip=grip, bf16_round=1,
         dynamic_raises=np.frombuffer(raised.encode(), dtype=np.uint8), static0_pose=o.logits[0], static0_grip=o.logits[1])


def gen_window_padded(name="deer_window_padded.npz"):
    """Window-mode calibration call of the reference (``generate_action_values``, value_net.py:333-386 -> MPTFlamingo.forward's all-exits
    branch, flamingo_mpt.py:463-517) on a batch of windows whose instructions have DIFFERENT lengths, right-padded to the longest
    (data.py:905-919 ``padding="longest"``) with the attention mask the tokenizer returns: the pad rows run through the trunk as queries
    (the mask only removes them as attention keys), and ``DeterministicDecoder`` pools over ALL T rows, pad rows included
    (action_head.py:519-520: no mask reaches the head).  Records every layer's hidden states, the random history layers the reference
    drew, extra_exit's window outputs and ``ActionValueNet(mode='generate')`` deltas."""
    _dist_init()
    cfg, seed = llm_cfg(window_size=4), 7
    sd = syn.make_synthetic_state(cfg, seed, bf16_round=True)
    model, lm = build_ref_mptflamingo(cfg, sd)
    model.load_state_dict({k: v for k, v in sd.items() if not k.startswith("vision_encoder.")}, strict=False)
    W, bs, S = cfg.window_size, 3, cfg.image_size
    lens = [8, 5, 6]
    T = max(lens)
    body = [[5, 17, 3, 42, 8], [9, 33], [61, 7, 12]]
    ids = torch.full((bs, T), 96, dtype=torch.long)                      # 96: the <PAD> id of this toy vocabulary
    mask = torch.zeros(bs, T, dtype=torch.bool)
    for b in range(bs):
        row = [cfg.media_token_id] + body[b] + [cfg.eoc_token_id, 0]
        assert len(row) == lens[b]
        ids[b, :lens[b]] = torch.tensor(row)
        mask[b, :lens[b]] = True
    rgb = seeded("win.rgb", (bs, W, 3, S, S))
    grip = seeded("win.grip", (bs, W, 3, S, S))
    # generate_action_values' reshaping (value_net.py:333-372): (bs, W, ...) -> (bs*W, 1, 1, 3, S, S); ids / mask repeated per frame
    images = rgb.unsqueeze(2).unsqueeze(2).flatten(0, 1)
    gripper = grip.unsqueeze(2).unsqueeze(2).flatten(0, 1)
    input_ids = ids.unsqueeze(1).repeat(1, W, 1).flatten(0, 1)
    attention_mask = mask.unsqueeze(1).repeat(1, W, 1).flatten(0, 1)
    torch.manual_seed(1234)
    import random as _r
    _r.seed(1234)
    final_output, exit_outputs, extra, rand_feat, rand_idx = model(vision_x=images, lang_x=input_ids, attention_mask=attention_mask,
                                                                   vision_gripper=gripper, state_tensor=None, with_gripper_logits=True,
                                                                   return_in_feat=True, only_extra_exit=True)
    feats = final_output.hidden_states
    vn = ActionValueNet(exit_list=model.get_all_exit_idx(), exit_head=model.extra_exit, interval=cfg.exit_interval, window_size=W, threshold_type="L2")
    delta = vn(feats, mode="generate", rand_layer_feat=rand_feat)
    print(f"  {name}: hidden {len(feats)} x {tuple(feats[0].shape)}, rand layers {rand_idx.tolist()}, delta {tuple(delta.shape)}")
    save(name, cfg, seed, ids=ids, mask=mask, rgb=rgb, grip=grip, hidden=torch.stack(feats), rand_layers=rand_idx, extra_pose=extra[0],
         extra_grip=extra[1][0], extra_grip_logits=extra[1][1], delta=delta, lens=np.asarray(lens), bf16_round=1)


def gen_round6_variants():
    gen_deer_forward_r6("deer_forward_plain.npz", cfg_kw=dict(lstm_layernorm=False, mlp_layernorm=False), thr_types=("L2", "mean"))
    gen_deer_forward_r6("deer_forward_avg3.npz", cfg_kw=dict(pooling="avg", mlp_num_hidden_layers=3), thr_types=("L2", "cosine"))
    gen_deer_forward_r6("deer_forward_thr.npz", thr_types=("mean", "max", "cosine"))
    gen_deer_forward_r6("deer_forward_consec.npz", thr_types=("L2", "max"), exit_id_list=[1, 2, 3, 4])
    gen_exit_interval_1()
    gen_window_padded()
    gen_fusion_modes_in_reference()
    gen_deer_forward_r6("deer_forward_pre.npz", cfg_kw=dict(fusion_mode="pre"), thr_types=("L2", "max"))   # flamingo_mpt.py:585-607


def gen_fusion_modes_in_reference():
    """What the reference's OWN forward does for the vision paths this repo does not build (VERDICT r5 missing-5): recorded as data.
    ``use_gripper=False`` and ``fusion_mode='two_way'`` both go through ``_encode_vision_x`` (flamingo_mpt.py:375-376), whose body reads an
    undefined name (``if eval_flop:``, :541) - the reference raises NameError on the first step, there is nothing to be compatible with;
    ``'pre'`` (both cameras' ViT tokens through ONE Perceiver call, :585-607) runs; ``'vit_concat'`` needs window-sized batches (:755)."""
    _dist_init()
    cfg = llm_cfg()
    S = cfg.image_size
    ids = torch.tensor([[cfg.media_token_id, 5, 17, 3, 42, 8, cfg.eoc_token_id, 0]])
    mask = torch.ones(1, 8, dtype=torch.bool)
    rgb, grip = seeded("deer.rgb", (1, 1, 1, 3, S, S)), seeded("deer.grip", (1, 1, 1, 3, S, S))
    names, outcomes = [], []
    for use_gripper, fusion in ((False, "post"), (True, "two_way"), (True, "pre"), (True, "vit_concat"), (True, "post")):
        try:
            sd = syn.make_synthetic_state(cfg, 7, bf16_round=True)
            lm, mod = build_ref_lang_encoder(cfg, sd)
            extend_instance(lm, FlamingoLMMixin)
            lm.set_decoder_layers_attr_name("transformer.blocks")
            venc_mod = nn.Module()
            venc_mod.visual = _OracleVisual(cfg, sd)
            model = MPTFlamingo(venc_mod, lm, cfg.eoc_token_id, cfg.media_token_id, vis_dim=cfg.vit_width,
                                cross_attn_every_n_layers=cfg.cross_attn_every_n_layers, window_size=cfg.window_size,
                                use_gripper=use_gripper, fusion_mode=fusion, llm="mpt_dolly_3b", pooling=cfg.pooling,
                                early_exit_layer=cfg.early_exit_layer, multi_exit=False, exit_interval=cfg.exit_interval,
                                mlp_layernorm=cfg.mlp_layernorm, lstm_layernorm=cfg.lstm_layernorm,
                                mlp_num_hidden_layers=cfg.mlp_num_hidden_layers, lstm_num_layers=cfg.lstm_num_layers).eval()
            model.set_all_exit_window_size(1)
            with torch.no_grad():
                o = model(vision_x=rgb, lang_x=ids, attention_mask=mask, vision_gripper=grip, state_tensor=torch.zeros(1, 1, 1, 15),
                          return_feature=True, deterministic=True, exit_id=3, dynamic_early_exit=False, exit_controller=None)
            out = "ok:" + "x".join(str(v) for v in lm._get_decoder_layers()[0].vis_x.shape)
        except Exception as e:                              # noqa: BLE001 - the exception type IS the recorded result
            out = type(e).__name__ + ":" + str(e)[:80]
        names.append(f"use_gripper={use_gripper},fusion_mode={fusion}")
        outcomes.append(out)
        print("  reference forward,", names[-1], "->", out)
    enc = lambda xs: np.frombuffer("|".join(xs).encode(), dtype=np.uint8)
    save("fusion_modes_reference.npz", cfg, 7, modes=enc(names), outcomes=enc(outcomes))


def gen_round5_variants():
    """layerwise_exit_eval and multi_step_action (round 5), each from the reference's own MPTFlamingo.forward"""
    gen_deer_forward_variant("deer_forward_lw.npz", layerwise=True)
    gen_deer_forward_variant("deer_forward_ms2.npz", multi_step_action=2)
    gen_deer_forward_variant("deer_forward_lw_ms3.npz", layerwise=True, multi_step_action=3)


def _dist_init():
    import torch.distributed as dist
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29577")
        dist.init_process_group("gloo", rank=0, world_size=1)


def gap_threshold(vals, lo=0.25, hi=0.75):
    """A threshold in the middle of the widest gap between sorted deltas (within the central quantiles), so
    that exit decisions are robust to bf16-level noise in a device implementation."""
    v = np.sort(np.asarray(vals, dtype=np.float64))
    a, b = int(len(v) * lo), max(int(len(v) * hi), int(len(v) * lo) + 2)
    b = min(b, len(v))
    gaps = v[a + 1:b] - v[a:b - 1]
    i = int(np.argmax(gaps)) + a
    return float(0.5 * (v[i] + v[i + 1]))


class _RecVN(ActionValueNet):
    """ActionValueNet that remembers every (exit id, delta) it produced (fixture generation only)."""

    def forward(self, feats, i=None, mode="infer", rand_layer_feat=None):
        v = super().forward(feats, i, mode, rand_layer_feat)
        if not hasattr(self, "rec"):
            self.rec = []
        self.rec.append((i, float(v)))
        return v


def gen_controller(name, n_layers, max_layer, steps_per_stage, threshold_type="L2", n_steps=9, **kw):
    """Drive the reference ActionValueNet/ExitController exactly like the LLM loop does
    (mosaic_gpt_3b.py:397-443) + the committing head call (flamingo_mpt.py:459) on random features."""
    _dist_init()
    cfg, seed = small_cfg(early_exit_layer=n_layers - 1, **kw), 5
    sd = syn.make_synthetic_state(cfg, seed)
    head = build_ref_head(cfg, sd)
    head.window_size = 1
    exit_ids = cfg.exit_ids()
    vn = _RecVN(exit_list=exit_ids, exit_head=head, interval=cfg.exit_interval, window_size=1,
                threshold_type=threshold_type)
    ctl = ExitController(vn, exit_id_list=exit_ids, steps_per_stage=1, leq=True, exit_dist="exp",
                         max_layer=max_layer)
    T = 5
    feats = seeded("ctl.feats", (n_steps, n_layers, 1, T, cfg.d_model))
    # make later layers converge (smaller change between consecutive exits) like a trained model does
    for l in range(1, n_layers):
        feats[:, l] = feats[:, l - 1] + feats[:, l] * (0.6 ** l)
    real = len([x for x in exit_ids if x <= ctl.max_layer])
    # pass 1 (not recorded): thresholds=-1 => only the forced exit fires; collect every delta of the stateful run
    ctl._set_threshold_value([-1.0] * real)
    for s_ in range(n_steps):
        ctl.set_timestep(0)
        hidden = ()
        for b_ in range(n_layers):
            hidden = hidden + (feats[s_, b_],)
            if ctl(hidden, b_):
                break
        head(hidden[b_], update_hidden_state=True)
    deltas = [[v for (i, v) in vn.rec if i == e] for e in exit_ids[:real]]
    thresholds = [gap_threshold(d) for d in deltas]
    thresholds[-1] = 1e5                                 # README.md:142 style: last threshold is "always exit"
    # the run that is recorded
    head2 = build_ref_head(cfg, sd)
    head2.window_size = 1
    vn = _RecVN(exit_list=exit_ids, exit_head=head2, interval=cfg.exit_interval, window_size=1,
                threshold_type=threshold_type)
    ctl = ExitController(vn, exit_id_list=exit_ids, steps_per_stage=steps_per_stage, leq=True, exit_dist="exp",
                         max_layer=max_layer)
    ctl._set_threshold_value(thresholds)
    exit_layers, poses, grips, n_evals = [], [], [], []
    for s in range(n_steps):
        ctl.set_timestep(s)
        hidden = ()
        n0 = len(vn.action_list)
        for b in range(n_layers):
            hidden = hidden + (feats[s, b],)
            if ctl(hidden, b):
                break
        exit_layers.append(b)
        n_evals.append(len(vn.action_list) - n0)
        a, g = head2(hidden[b], update_hidden_state=True)   # flamingo_mpt.py:459
        poses.append(a)
        grips.append(g)
    print(f"  {name}: exits={exit_layers} thresholds={np.round(thresholds, 5).tolist()}")
    save(name, cfg, seed, feats=feats, thresholds=np.asarray(thresholds), exit_layers=np.asarray(exit_layers),
         pose=torch.stack(poses), grip=torch.stack(grips), n_evals=np.asarray(n_evals),
         rec_layer=np.asarray([i for i, _ in vn.rec]), rec_delta=np.asarray([v for _, v in vn.rec]),
         max_layer=max_layer, steps_per_stage=steps_per_stage,
         threshold_type=np.frombuffer(threshold_type.encode(), dtype=np.uint8), ctl_max_layer=ctl.max_layer)


def gen_valuenet_generate(name="valuenet_generate.npz", threshold_type="L2"):
    """Calibration ('generate') mode of the reference ActionValueNet (value_net.py:134-160): for the time steps in the second
    half of a window, every exit's action is predicted in WINDOW mode from [history features of random exit layers ; this
    exit's feature at the step]; deltas between consecutive exits (starting from layer 0's pseudo action)."""
    _dist_init()
    cfg, seed = small_cfg(early_exit_layer=7, window_size=8), 9
    sd = syn.make_synthetic_state(cfg, seed)
    head = build_ref_head(cfg, sd)
    ws, bs, T = cfg.window_size, 2, 5
    head.window_size = ws
    exit_ids = cfg.exit_ids()
    vn = ActionValueNet(exit_list=exit_ids, exit_head=head, interval=cfg.exit_interval, window_size=ws, threshold_type=threshold_type)
    n_layers = cfg.n_layers
    feats = seeded("gen.feats", (n_layers, bs * ws, T, cfg.d_model))
    for l in range(1, n_layers):
        feats[l] = feats[l - 1] + feats[l] * (0.6 ** l)
    g = torch.Generator().manual_seed(11)
    idx = torch.randint(0, len(exit_ids), (bs * ws,), generator=g)               # flamingo_mpt.py:486-490 (sampling strategy 1)
    rand_layers = torch.tensor([exit_ids[int(i)] for i in idx])
    rand_feat = torch.stack([feats[int(rand_layers[j]), j] for j in range(bs * ws)])
    with torch.no_grad():
        delta = vn(tuple(feats[l] for l in range(n_layers)), mode="generate", rand_layer_feat=rand_feat)
    print(f"  {name}: delta {tuple(delta.shape)} mean per exit {delta.mean(1).numpy().round(4).tolist()}")
    save(name, cfg, seed, feats=feats, rand_layers=rand_layers.numpy(), delta=delta, bs=bs,
         threshold_type=np.frombuffer(threshold_type.encode(), dtype=np.uint8))


def gen_thresholds():
    _dist_init()
    cfg = small_cfg(early_exit_layer=11)
    args = types.SimpleNamespace(rank=1)
    g = torch.Generator().manual_seed(11)
    values = torch.rand(6, 400, generator=g) * torch.tensor([0.05, 0.02, 0.02, 0.01, 0.01, 0.01]).view(6, 1)
    values[2, 10:20] = values[2, 5]                      # ties
    out = {"values": values}
    for model_name in ("mpt_dolly_3b", "mpt_9b"):
        for ratio in (0.8, 1.0, 1.5):
            for max_layer in (12, 8):
                ctl = ExitController(None, exit_id_list=cfg.exit_ids(), steps_per_stage=1, leq=True, exit_dist="exp",
                                     max_layer=max_layer)
                real = len([x for x in ctl.exit_id_list if x <= ctl.max_layer])
                ctl.set_threshold(args, None, None, ratio, model_name, values=values[:real].clone())
                T = torch.stack([torch.as_tensor(ctl.thresholds[i]) for i in ctl.exit_id_list[:real]])
                out[f"T_{model_name}_{ratio}_{max_layer}"] = T
    # round 6: the other exit distributions / the ">= threshold" criterion of the same solver (value_net.py:214-231,248-258); for 'gauss'
    # and 'gamma' ``exit_ratio`` is the centre / the shape parameter (eval_calvin.py passes it through unchanged)
    for dist, leq, ratio, max_layer in (("gamma", True, 2.0, 12), ("gamma", True, 1.5, 8), ("gauss", True, 2.0, 12), ("gauss", True, 0.5, 8),
                                        ("exp", False, 0.8, 12), ("gamma", False, 3.0, 12)):
        ctl = ExitController(None, exit_id_list=cfg.exit_ids(), steps_per_stage=1, leq=leq, exit_dist=dist, max_layer=max_layer)
        real = len([x for x in ctl.exit_id_list if x <= ctl.max_layer])
        ctl.set_threshold(args, None, None, ratio, "mpt_dolly_3b", values=values[:real].clone())
        out[f"D_{dist}_{int(leq)}_{ratio}_{max_layer}"] = torch.stack([torch.as_tensor(ctl.thresholds[i]) for i in ctl.exit_id_list[:real]])
    save("thresholds.npz", cfg, 0, **out)


def llm_cfg(**kw):
    base = dict(image_size=28, patch_size=14, vit_width=64, vit_layers=1, vit_heads=1, vit_mlp=128,
                d_model=64, n_heads=2, n_layers_total=6, vocab_size=100, media_token_id=98, eoc_token_id=97,
                early_exit_layer=4, head_hidden=1024)     # perceiver / x-attn / head dims = reference ctor defaults
    base.update(kw)
    return DeerConfig(**base)


def build_ref_lang_encoder(cfg, sd):
    mod = ref_import.load_mosaic_gpt()
    st = sys.modules["deer_mpt1b_pkg._standins"]
    hf = st.MosaicGPTConfig(d_model=cfg.d_model, n_heads=cfg.n_heads, n_layers=cfg.n_layers_total, max_seq_len=32,
                            vocab_size=cfg.vocab_size, attn_qk_ln=cfg.attn_qk_ln, alibi_bias_max=cfg.alibi_bias_max)
    lm = mod.MosaicGPT(hf).eval()
    return lm, mod


def gen_mosaic_loop():
    """The reference's own MosaicGPT.forward loop (exit_id / exit_controller / hidden_states semantics,
    ALiBi + key-padding bias plumbing) on stand-in blocks; no x-attn here."""
    cfg, seed = llm_cfg(cross_attn_every_n_layers=10 ** 6, early_exit_layer=5), 6
    sd = syn.make_synthetic_state(cfg, seed)
    lm, mod = build_ref_lang_encoder(cfg, sd)
    s = {k[len("lang_encoder."):].replace(".decoder_layer.", "."): v for k, v in sd.items()
         if k.startswith("lang_encoder.transformer.")}
    missing, unexpected = lm.load_state_dict(s, strict=False)
    assert not unexpected, unexpected
    assert all(k.startswith("transformer.ln_f") for k in missing), missing
    ids = torch.randint(0, 90, (2, 9), generator=torch.Generator().manual_seed(3))
    mask = torch.ones(2, 9, dtype=torch.bool)
    mask[1, 6:] = False                                  # right padding (data.py:914 padding="longest")
    o_full = lm(ids, attention_mask=mask, output_hidden_states=True, return_dict=True)
    o_e2 = lm(ids, attention_mask=mask, output_hidden_states=True, return_dict=True, exit_id=2)
    o_neg = lm(ids, attention_mask=mask, output_hidden_states=True, return_dict=True, exit_id=-2)
    calls = []

    def ctl(hidden, b):
        calls.append((len(hidden), b))
        return b == 3
    o_ctl = lm(ids, attention_mask=mask, output_hidden_states=True, return_dict=True, exit_controller=ctl)
    save("mosaic_loop.npz", cfg, seed, ids=ids, mask=mask,
         full=torch.stack(o_full.hidden_states), full_exit=o_full.exit_layer,
         e2=torch.stack(o_e2.hidden_states), e2_exit=o_e2.exit_layer,
         neg=torch.stack(o_neg.hidden_states), neg_exit=o_neg.exit_layer,
         ctl=torch.stack(o_ctl.hidden_states), ctl_exit=o_ctl.exit_layer, ctl_calls=np.asarray(calls))


def gen_mpt9b_loop():
    """The reference's own ``MPTModel.forward`` multi-exit loop of the 9B variant (modeling_gpt_9b.py:352-503: fp32 attn bias with
    the key-padding mask kept as ``attention_mask``, 3-tuple block returns, ``norm_f`` + one extra hidden state only when NO exit
    fires) on stand-in MPT-7B blocks; no x-attn here."""
    mod = ref_import.load_mpt_9b()
    st = sys.modules["deer_mpt7b_pkg._standins"]
    cfg = DeerConfig(image_size=28, patch_size=14, vit_width=64, vit_layers=1, vit_heads=1, vit_mlp=128, llm_name="mpt_9b",
                     d_model=128, n_heads=4, n_layers_total=8, vocab_size=100, media_token_id=98, eoc_token_id=97, attn_qk_ln=False,
                     cross_attn_every_n_layers=10 ** 6, early_exit_layer=5, head_hidden=1024)
    seed = 9
    sd = syn.make_synthetic_state(cfg, seed)
    hf = st.MPTConfig(d_model=cfg.d_model, n_heads=cfg.n_heads, n_layers=cfg.n_layers, max_seq_len=32, vocab_size=cfg.vocab_size)
    lm = mod.MPTModel(hf).eval()
    s = {k[len("lang_encoder.transformer."):].replace(".decoder_layer.", "."): v for k, v in sd.items()
         if k.startswith("lang_encoder.transformer.")}
    missing, unexpected = lm.load_state_dict(s, strict=False)
    assert not unexpected, unexpected
    assert all(k.startswith("norm_f") for k in missing), missing
    with torch.no_grad():
        lm.norm_f.weight.fill_(1.0)
    ids = torch.randint(0, 90, (2, 9), generator=torch.Generator().manual_seed(3))
    mask = torch.ones(2, 9, dtype=torch.bool)
    mask[1, 6:] = False
    o_full = lm(ids, attention_mask=mask, output_hidden_states=True, return_dict=True)
    o_e2 = lm(ids, attention_mask=mask, output_hidden_states=True, return_dict=True, exit_id=2)
    calls = []

    def ctl(hidden, b):
        calls.append((len(hidden), b))
        return b == 3
    o_ctl = lm(ids, attention_mask=mask, output_hidden_states=True, return_dict=True, exit_controller=ctl)
    save("mpt9b_loop.npz", cfg, seed, ids=ids, mask=mask,
         full=torch.stack(o_full.hidden_states), full_exit=o_full.exit_layer,
         e2=torch.stack(o_e2.hidden_states), e2_exit=o_e2.exit_layer,
         ctl=torch.stack(o_ctl.hidden_states), ctl_exit=o_ctl.exit_layer, ctl_calls=np.asarray(calls))
    print("mpt9b_loop.npz: full run", len(o_full.hidden_states), "hidden states (n_layers + norm_f), exit", o_full.exit_layer)


class _OracleVisual(nn.Module):
    """Hosts the (un-vendored) ViT inside the reference's MPTFlamingo: ``visual(x) -> (pooled, tokens)``."""

    def __init__(self, cfg, sd):
        super().__init__()
        self.cfg, self.sd = cfg, sd
        self.output_tokens = True

    def forward(self, x):
        t = orc.vit_visual_tokens(self.sd, self.cfg, x.float())
        return t[:, 0], t


def build_ref_mptflamingo(cfg, sd):
    """The reference's MPTFlamingo around the reference's MosaicGPT loop (stand-in blocks) and the oracle-hosted ViT."""
    lm, mod = build_ref_lang_encoder(cfg, sd)
    extend_instance(lm, FlamingoLMMixin)
    lm.set_decoder_layers_attr_name("transformer.blocks")
    venc_mod = nn.Module()
    venc_mod.visual = _OracleVisual(cfg, sd)
    model = MPTFlamingo(venc_mod, lm, cfg.eoc_token_id, cfg.media_token_id, vis_dim=cfg.vit_width,
                        cross_attn_every_n_layers=cfg.cross_attn_every_n_layers, window_size=cfg.window_size,
                        use_gripper=True, fusion_mode="post", llm="mpt_dolly_3b", pooling="max",
                        early_exit_layer=cfg.early_exit_layer, multi_exit=False, exit_interval=cfg.exit_interval,
                        mlp_layernorm=True, lstm_layernorm=True, mlp_num_hidden_layers=2, lstm_num_layers=4).eval()
    return model, lm


def gen_ckpt_meta():
    """tests/golden/ckpt_meta.json: the LAYOUT of the two checkpoint files the reference's eval loads
    (eval_calvin.py:541-543 OpenFlamingo ``.pt``, strict=False; :572-578 DeeR ``.pth`` dict -> ``model_state_dict`` through the
    DDP wrapper, strict=False) as the reference's own code produces it: key names + shapes of ``model.state_dict()``, the
    trainable-only subset ``get_checkpoint`` keeps (train_utils.py:631-638) under the factory's freeze policy
    (factory.py:203-235, train_params=-1, freeze_embed=False), the ``module.`` prefix DDP adds, and the scalar fields
    ``save_ckpt`` writes (train_utils.py:31-50).  Data only: no tensor values (weights are seeded, deer_vla_amd.synthetic)."""
    _dist_init()
    cfg, seed = llm_cfg(), 7
    sd = syn.make_synthetic_state(cfg, seed, bf16_round=True)
    model, lm = build_ref_mptflamingo(cfg, sd)
    # factory.py:203-235
    model.requires_grad_(False)
    model.lang_encoder.gated_cross_attn_layers.requires_grad_(True)
    model.perceiver.requires_grad_(True)
    # factory.py:227 `get_input_embeddings()`: resolved by transformers-4.x to transformer.wte (this container's 5.x needs the override)
    model.lang_encoder.transformer.wte.requires_grad_(True)
    model.lang_encoder.lm_head.requires_grad_(True)
    if model.sep_lm_head:
        model.lm_head.requires_grad_(True)
    if len(model.lm_exits) > 0:
        model.lm_exit_modules.requires_grad_(True)
    model.extra_exit.requires_grad_(True)
    full = model.state_dict()
    # train_utils.py:631-638 get_checkpoint on the DDP-wrapped model (keys carry the "module." prefix)
    trainable = {"module." + k: v for k, v in full.items()}
    for name, p_ in model.named_parameters():
        if not p_.requires_grad and "normalizer" not in name:
            del trainable["module." + name]
    # which state-dict keys are aliases of the same storage (x-attn layers are registered twice: flamingo_lm.py:160-176)
    by_ptr = {}
    for k, v in full.items():
        by_ptr.setdefault(v.data_ptr(), []).append(k)
    aliases = [ks for ks in by_ptr.values() if len(ks) > 1]
    meta = {
        "cfg": cfg.to_dict(), "seed": seed,
        "full_state_dict": {k: list(v.shape) for k, v in full.items()},
        "deer_model_state_dict": {k: list(v.shape) for k, v in trainable.items()},
        "alias_groups": aliases,
        # train_utils.py:31-50 (what eval_calvin.py:455-476 reads back), values of the released DeeR configuration
        "deer_ckpt_fields": {"epoch": 3, "head_type": "deterministic", "early_exit_layer": cfg.early_exit_layer, "multi_exit": True,
                             "share_exit": False, "exit_interval": cfg.exit_interval, "exit_dropout": 0.4, "lstm_dropout": 0.3,
                             "dropout_mode": "layerwise", "mlp_layernorm": True, "lstm_layernorm": True, "mlp_num_hidden_layers": 2,
                             "lstm_num_layers": 4, "pooling": "max", "precision": "fp32"},
    }
    with open(os.path.join(HERE, "ckpt_meta.json"), "w") as fh:
        json.dump(meta, fh, indent=0)
    print("ckpt_meta.json:", len(full), "state-dict keys,", len(trainable), "in the DeeR model_state_dict,", len(aliases), "alias groups")


def gen_metrics():
    """tests/golden/metrics.json: the reference's OWN metric functions (robot_flamingo/eval/eval_utils.py:47-118 ``merge_multi_list``,
    ``count_success``, ``count_exit_ratio``, ``print_and_save``) executed on seeded per-chain results.  The module itself cannot be
    imported (calvin_agent / hydra / calvin_env are absent), so the four function definitions are taken from the file where it
    lies with ``ast`` and compiled as they are.  The evaluation chains are the first 16 entries of the reference's data file
    ``eval_sequences.json`` with the matching entries of ``lang_annotation_cache.json`` (data files, kept as a fixture)."""
    import ast
    import contextlib
    import io
    from collections import Counter
    src = open("/root/reference/robot_flamingo/eval/eval_utils.py").read()
    tree = ast.parse(src)
    want = {"merge_multi_list", "count_success", "count_exit_ratio", "print_and_save"}
    mod = ast.Module(body=[n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in want], type_ignores=[])
    ns = {"Counter": Counter, "np": np}
    exec(compile(mod, "eval_utils.py[47-118]", "exec"), ns)
    seqs = json.load(open("/root/reference/eval_sequences.json"))[:16]
    ann_all = json.load(open("/root/reference/lang_annotation_cache.json"))
    tasks = sorted({t for _, chain in seqs for t in chain})
    ann = {t: ann_all[t] for t in tasks} if isinstance(ann_all, dict) else ann_all[:16]   # per-chain enriched instructions
    g = torch.Generator().manual_seed(11)
    n_layer = 12
    exits = [1, 3, 5, 7, 9, 11]
    per_chain = []
    for _, chain in seqs:
        n_ok = int(torch.randint(0, 6, (1,), generator=g))
        ok_exits, fail_exits, ok_steps, ok_llm, fail_llm = [], [], [], [], []
        for k in range(min(n_ok + 1, 5)):
            n_steps = int(torch.randint(20, 120, (1,), generator=g)) if k < n_ok else 360
            ex = [exits[int(i)] for i in torch.randint(0, 6, (n_steps,), generator=g)]
            tm = [round(float(t), 5) for t in (torch.rand(n_steps, generator=g) * 4e-3)]
            if k < n_ok:
                ok_exits.extend(ex); ok_steps.append(n_steps); ok_llm.extend(tm)
            else:
                fail_exits.extend(ex); fail_llm.extend(tm)
        per_chain.append((n_ok, ok_exits, fail_exits, ok_steps, ok_llm, fail_llm))
    res_list, success_exit_list, fail_exit_list, step_list, ok_llm_list, fail_llm_list = map(list, zip(*per_chain))
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):                      # eval_utils.py:575-577
        ret = ns["print_and_save"](res_list, ns["merge_multi_list"](success_exit_list), ns["merge_multi_list"](fail_exit_list),
                                   ns["merge_multi_list"](step_list), ns["merge_multi_list"](ok_llm_list), ns["merge_multi_list"](fail_llm_list),
                                   seqs, None, n_layer, 0)
    out = {"n_layer": n_layer, "sequences": seqs, "annotations": ann,
           "per_chain": [{"n_ok": a, "ok_exits": b, "fail_exits": c, "ok_steps": d, "ok_llm": e, "fail_llm": f} for a, b, c, d, e, f in per_chain],
           "count_success": ns["count_success"](res_list),
           "count_exit_ratio_success": ns["count_exit_ratio"](ns["merge_multi_list"](success_exit_list), n_layer),
           "print_and_save_return": [float(ret[0]), float(ret[1])],
           "print_and_save_stdout": buf.getvalue()}
    with open(os.path.join(HERE, "metrics.json"), "w") as fh:
        json.dump(out, fh)
    print("metrics.json: avg_seq_len %.3f avg exit %.3f, %d stdout lines" % (ret[0], ret[1], buf.getvalue().count("\n")))


def gen_deer_forward():
    """The reference's own MPTFlamingo.forward (+FlamingoLMMixin +MosaicGPT loop +Perceiver +x-attn
    +DeterministicDecoder +ExitController) end to end on CPU."""
    _dist_init()
    cfg, seed = llm_cfg(), 7
    # GEMM operands are bf16-representable (what the device engine keeps in HBM); the reference still runs fp32
    sd = syn.make_synthetic_state(cfg, seed, bf16_round=True)
    lm, mod = build_ref_lang_encoder(cfg, sd)
    extend_instance(lm, FlamingoLMMixin)
    lm.set_decoder_layers_attr_name("transformer.blocks")
    venc = types.SimpleNamespace(visual=_OracleVisual(cfg, sd))
    venc_mod = nn.Module()
    venc_mod.visual = venc.visual
    model = MPTFlamingo(venc_mod, lm, cfg.eoc_token_id, cfg.media_token_id, vis_dim=cfg.vit_width,
                        cross_attn_every_n_layers=cfg.cross_attn_every_n_layers, window_size=cfg.window_size,
                        use_gripper=True, fusion_mode="post", llm="mpt_dolly_3b", pooling="max",
                        early_exit_layer=cfg.early_exit_layer, multi_exit=False, exit_interval=cfg.exit_interval,
                        mlp_layernorm=True, lstm_layernorm=True, mlp_num_hidden_layers=2, lstm_num_layers=4).eval()
    assert model.get_all_exit_idx() == cfg.exit_ids(), (model.get_all_exit_idx(), cfg.exit_ids())
    ref_keys = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd_model = {k: v for k, v in sd.items() if not k.startswith("vision_encoder.")}   # ViT hosted by the oracle
    for k, v in sd_model.items():
        assert k in ref_keys and ref_keys[k] == tuple(v.shape), (k, tuple(v.shape), ref_keys.get(k))
    missing, unexpected = model.load_state_dict(sd_model, strict=False)
    assert not unexpected, unexpected
    # every tensor we did not provide must be an alias of one we did, an unused head, or the never-applied ln_f
    provided = {id(p) for k, p in model.state_dict(keep_vars=True).items() if k in sd}
    for k in missing:
        p = model.state_dict(keep_vars=True)[k]
        assert id(p) in provided or k.startswith("lm_head.") or "ln_f" in k, k

    T = 8
    ids = torch.tensor([[cfg.media_token_id, 5, 17, 3, 42, 8, cfg.eoc_token_id, 0]])
    mask = torch.ones(1, T, dtype=torch.bool)
    S = cfg.image_size
    n_steps = 8
    rgb = seeded("deer.rgb", (n_steps, 1, 1, 1, 3, S, S))
    grip = seeded("deer.grip", (n_steps, 1, 1, 1, 3, S, S))
    state = torch.zeros(1, 1, 1, 15)
    model.set_all_exit_window_size(1)

    # --- static exit (BASELINE config[0]) -------------------------------------------------------
    outs = {}
    for eid in (3, 4, -1):
        model.clear_all_exit_memory()
        o = model(vision_x=rgb[0], lang_x=ids, attention_mask=mask, vision_gripper=grip[0], state_tensor=state,
                  return_feature=True, deterministic=True, exit_id=eid, dynamic_early_exit=False, exit_controller=None)
        tag = f"static{eid}"
        outs[tag + "_pose"], outs[tag + "_grip"] = o.logits[0], o.logits[1]
        outs[tag + "_hidden"] = torch.stack(o.hidden_states)
        outs[tag + "_exit"] = o.exit_layer
    outs["vis_x"] = lm._get_decoder_layers()[0].vis_x

    # --- dynamic exit over n_steps control steps with LSTM carry (ModelWrapper.step protocol) ----
    exit_ids = model.get_all_exit_idx()
    for tag, max_layer in (("dyn", 12), ("dynS", 4)):
        model.clear_all_exit_memory()
        vn = _RecVN(exit_list=exit_ids, exit_head=model.extra_exit, interval=cfg.exit_interval,
                    window_size=cfg.window_size, threshold_type="L2")
        ctl = ExitController(vn, exit_id_list=exit_ids, steps_per_stage=1, leq=True, exit_dist="exp", max_layer=max_layer)
        real = len([x for x in exit_ids if x <= ctl.max_layer])
        ctl._set_threshold_value([-1.0] * real)          # never-exit pass (not recorded) to see the delta scale
        for s in range(n_steps):
            ctl.set_timestep(s)
            model(vision_x=rgb[s], lang_x=ids, attention_mask=mask, vision_gripper=grip[s], state_tensor=state,
                  return_feature=True, deterministic=True, exit_id=None, dynamic_early_exit=True, exit_controller=ctl)
        thr = [gap_threshold([v for (i, v) in vn.rec if i == e], 0.2, 0.8) for e in exit_ids[:real]]
        thr[-1] = 1e5
        model.clear_all_exit_memory()
        vn.reset_actions()
        vn.rec = []
        ctl._set_threshold_value(thr)
        ex, ps, gs, hid, eps_, egs, ecn = [], [], [], [], [], [], []
        for s in range(n_steps):
            ctl.set_timestep(s)
            vn.reset_actions()                           # the ensembling harness resets after every step (eval_utils.py:460-461);
                                                         # the exit decisions do not depend on it (action_list[-1] is only read
                                                         # after a check of the same step appended to it)
            o = model(vision_x=rgb[s], lang_x=ids, attention_mask=mask, vision_gripper=grip[s], state_tensor=state,
                      return_feature=True, deterministic=True, exit_id=None, dynamic_early_exit=True, exit_controller=ctl)
            ex.append(o.exit_layer)
            ps.append(o.logits[0])
            gs.append(o.logits[1])
            hid.append(o.hidden_states[o.exit_layer])
            ep, eg = vn.get_ensemble_action()            # value_net.py:92-95: mean over action_list[-2:]
            eps_.append(ep)
            egs.append(eg)
            ecn.append(min(len(vn.action_list), 2))
        print(f"  deer_forward {tag}: exits={ex} ensemble over {ecn} actions")
        outs[tag + "_ens_pose"] = torch.stack(eps_)
        outs[tag + "_ens_grip"] = torch.stack(egs)
        outs[tag + "_ens_count"] = np.asarray(ecn)
        outs[tag + "_thr"] = np.asarray(thr)
        outs[tag + "_exit"] = np.asarray(ex)
        outs[tag + "_pose"] = torch.stack(ps)
        outs[tag + "_grip"] = torch.stack(gs)
        outs[tag + "_hidden"] = torch.stack(hid)
        outs[tag + "_max_layer"] = max_layer
        outs[tag + "_rec_layer"] = np.asarray([i for i, _ in vn.rec])
        outs[tag + "_rec_delta"] = np.asarray([v for _, v in vn.rec])
    save("deer_forward.npz", cfg, seed, ids=ids, mask=mask, rgb=rgb, grip=grip, bf16_round=1, **outs)


def gen_hf_mpt_block():
    from transformers import MptConfig
    from transformers.models.mpt import modeling_mpt as mm
    cfg, seed = llm_cfg(llm_name="mpt_9b", attn_qk_ln=False, cross_attn_every_n_layers=10 ** 6, n_heads=4), 8
    sd = syn.make_synthetic_state(cfg, seed)
    hf = MptConfig(d_model=cfg.d_model, n_heads=cfg.n_heads, n_layers=1, expansion_ratio=cfg.mlp_ratio, max_seq_len=32,
                   vocab_size=cfg.vocab_size)
    blk = mm.MptBlock(hf).eval()
    load_strict(blk, sub_state(sd, "lang_encoder.transformer.blocks.0.decoder_layer."))
    S = 9
    x = seeded("hf.x", (2, S, cfg.d_model))
    alibi = mm.build_mpt_alibi_tensor(cfg.n_heads, 32)                       # (H,1,32)
    causal = torch.ones(S, S, dtype=torch.bool).triu(1).view(1, 1, S, S)
    out, _ = blk(x, position_bias=alibi, attention_mask=causal)
    save("hf_mpt_block.npz", cfg, seed, x=x, out=out)


def gen_hf_clip():
    from transformers import CLIPVisionConfig, CLIPVisionModel
    cfg, seed = small_cfg(image_size=42, vit_width=64, vit_layers=2, vit_heads=2, vit_mlp=128), 9
    sd = syn.make_synthetic_state(cfg, seed)
    hf = CLIPVisionConfig(hidden_size=cfg.vit_width, intermediate_size=cfg.vit_mlp, num_hidden_layers=cfg.vit_layers,
                          num_attention_heads=cfg.vit_heads, image_size=cfg.image_size, patch_size=cfg.patch_size,
                          hidden_act="quick_gelu", layer_norm_eps=1e-5)
    m = CLIPVisionModel(hf).eval()
    v = "vision_encoder.visual."
    W = cfg.vit_width
    t = {"embeddings.class_embedding": sd[v + "class_embedding"],
         "embeddings.patch_embedding.weight": sd[v + "conv1.weight"],
         "embeddings.position_embedding.weight": sd[v + "positional_embedding"],
         "pre_layrnorm.weight": sd[v + "ln_pre.weight"],
         "pre_layrnorm.bias": sd[v + "ln_pre.bias"]}
    for l in range(cfg.vit_layers):
        p, q = f"{v}transformer.resblocks.{l}.", f"encoder.layers.{l}."
        wq, wk, wv = sd[p + "attn.in_proj_weight"].chunk(3, dim=0)
        bq, bk, bv = sd[p + "attn.in_proj_bias"].chunk(3, dim=0)
        t.update({q + "self_attn.q_proj.weight": wq, q + "self_attn.k_proj.weight": wk, q + "self_attn.v_proj.weight": wv,
                  q + "self_attn.q_proj.bias": bq, q + "self_attn.k_proj.bias": bk, q + "self_attn.v_proj.bias": bv,
                  q + "self_attn.out_proj.weight": sd[p + "attn.out_proj.weight"],
                  q + "self_attn.out_proj.bias": sd[p + "attn.out_proj.bias"],
                  q + "layer_norm1.weight": sd[p + "ln_1.weight"], q + "layer_norm1.bias": sd[p + "ln_1.bias"],
                  q + "layer_norm2.weight": sd[p + "ln_2.weight"], q + "layer_norm2.bias": sd[p + "ln_2.bias"],
                  q + "mlp.fc1.weight": sd[p + "mlp.c_fc.weight"], q + "mlp.fc1.bias": sd[p + "mlp.c_fc.bias"],
                  q + "mlp.fc2.weight": sd[p + "mlp.c_proj.weight"], q + "mlp.fc2.bias": sd[p + "mlp.c_proj.bias"]})
    missing, unexpected = m.load_state_dict(t, strict=False)
    assert not unexpected, unexpected
    assert all("post_layernorm" in k or "position_ids" in k for k in missing), missing
    img = seeded("clip.img", (2, 3, cfg.image_size, cfg.image_size))
    out = m(pixel_values=img).last_hidden_state[:, 1:]                        # pre-post_layernorm patch tokens
    save("hf_clip_vit.npz", cfg, seed, img=img, out=out)


if __name__ == "__main__":
    if len(sys.argv) > 1:                                   # e.g. `make_golden.py gen_ckpt_meta`: regenerate one fixture
        for fn in sys.argv[1:]:
            globals()[fn]()
        sys.exit(0)
    gen_ckpt_meta()
    gen_metrics()
    gen_perceiver()
    gen_xattn()
    gen_flamingo_layer()
    gen_head("head_ln.npz")
    gen_head("head_plain.npz", lstm_layernorm=False, mlp_layernorm=False)
    gen_head("head_avg3.npz", pooling="avg", mlp_num_hidden_layers=3)
    gen_head_state()
    gen_controller("controller_b12.npz", n_layers=12, max_layer=12, steps_per_stage=1)
    gen_controller("controller_s4.npz", n_layers=5, max_layer=4, steps_per_stage=1)
    gen_controller("controller_sps3.npz", n_layers=12, max_layer=12, steps_per_stage=3)
    gen_controller("controller_max.npz", n_layers=12, max_layer=8, steps_per_stage=1, threshold_type="max")
    gen_valuenet_generate()
    gen_thresholds()
    gen_mosaic_loop()
    gen_mpt9b_loop()
    gen_deer_forward()
    gen_deer_forward_variant("deer_forward_state.npz", use_state=True)
    gen_deer_forward_variant("deer_forward_sep.npz", sep_resampler=True)
    gen_round5_variants()
    gen_round6_variants()
    gen_hf_mpt_block()
    gen_hf_clip()
