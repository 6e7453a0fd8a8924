"""CPU oracle for the DeeR-VLA early-exit forward path.  *** TEST INFRASTRUCTURE ONLY ***

Plain PyTorch fp32 on the host; a restatement of the reference's algorithm for one control step
(``ModelWrapper.step`` -> ``MPTFlamingo.forward``).  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module; the
product package ``deer_vla_amd`` never does (it fails loudly when the HIP library is missing).

Pinning status:
* vendored pieces (Perceiver, gated x-attn, FlamingoLayer ordering, DeterministicDecoder,
  ActionValueNet, ExitController incl. the threshold solver, MPTFlamingo.forward dispatch) are
  pinned against golden vectors produced by importing the reference's own modules
  (``tests/golden/make_golden.py`` -> ``tests/golden/*.npz``; ``tests/test_oracle_golden.py``).
* un-vendored pieces (MPT ``GPTBlock`` arithmetic, open_clip ViT-L/14) are NOT in /root/reference
  (HF ``mosaicml/mpt-1b-redpajama-200b-dolly`` remote code, no version pin, ``factory.py:15-24``;
  ``open-clip-torch==2.20.0``, ``requirements.txt:16``).  Their published algorithm is restated here
   and cross-checked against the independent ``transformers`` implementations
  (``MptBlock`` / ``CLIPVisionModel``) through fixtures; relative to the reference itself those two
  pieces are "parity unpinned".

Every function cites the reference file:line it follows (paths relative to /root/reference).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]


# ----------------------------------------------------------------------------------------------
# small helpers
# ----------------------------------------------------------------------------------------------
def _ln(x, w, b=None, eps: float = 1e-5):
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def _lin(x, w, b=None):
    return F.linear(x, w, b)


def _get(sd: SD, key: str) -> Optional[torch.Tensor]:
    return sd.get(key, None)


# ----------------------------------------------------------------------------------------------
# CLIP ViT-L/14 visual tower  (EXTERNAL open_clip; call site
# robot_flamingo/models/flamingo_mpt.py:580 ``self.vision_encoder.visual(vision_x)[1]`` with
# ``output_tokens=True`` set at robot_flamingo/models/factory.py:114)
# ----------------------------------------------------------------------------------------------
def vit_visual_tokens(sd: SD, cfg, img: torch.Tensor, prefix: str = "vision_encoder.visual.") -> torch.Tensor:
    """img (N,3,S,S) already CLIP-normalised -> patch tokens (N, n_patches, width); the cls token
    is dropped and ln_post / proj are NOT applied (they only touch the pooled token)."""
    W, H = cfg.vit_width, cfg.vit_heads
    x = F.conv2d(img, sd[prefix + "conv1.weight"], None, stride=cfg.patch_size)       # (N,W,g,g), no bias
    N = x.shape[0]
    x = x.reshape(N, W, -1).permute(0, 2, 1)                                           # (N,P,W)
    cls = sd[prefix + "class_embedding"].view(1, 1, W).expand(N, 1, W)
    x = torch.cat([cls, x], dim=1) + sd[prefix + "positional_embedding"].unsqueeze(0)
    x = _ln(x, sd[prefix + "ln_pre.weight"], sd[prefix + "ln_pre.bias"])
    hd = W // H
    for l in range(cfg.vit_layers):
        p = f"{prefix}transformer.resblocks.{l}."
        y = _ln(x, sd[p + "ln_1.weight"], sd[p + "ln_1.bias"])
        qkv = _lin(y, sd[p + "attn.in_proj_weight"], sd[p + "attn.in_proj_bias"])
        q, k, v = qkv.chunk(3, dim=-1)
        L = x.shape[1]
        q = q.view(N, L, H, hd).transpose(1, 2) * hd ** -0.5          # nn.MultiheadAttention scales q
        k = k.view(N, L, H, hd).transpose(1, 2)
        v = v.view(N, L, H, hd).transpose(1, 2)
        a = torch.softmax(q @ k.transpose(-1, -2), dim=-1) @ v
        a = a.transpose(1, 2).reshape(N, L, W)
        x = x + _lin(a, sd[p + "attn.out_proj.weight"], sd[p + "attn.out_proj.bias"])
        y = _ln(x, sd[p + "ln_2.weight"], sd[p + "ln_2.bias"])
        y = _lin(y, sd[p + "mlp.c_fc.weight"], sd[p + "mlp.c_fc.bias"])
        y = y * torch.sigmoid(1.702 * y)                               # QuickGELU ("openai" weights)
        x = x + _lin(y, sd[p + "mlp.c_proj.weight"], sd[p + "mlp.c_proj.bias"])
    return x[:, 1:, :]


# ----------------------------------------------------------------------------------------------
# Perceiver resampler  (open_flamingo/open_flamingo/src/helpers.py)
# ----------------------------------------------------------------------------------------------
def feed_forward(sd: SD, p: str, x: torch.Tensor) -> torch.Tensor:
    """helpers.py:15-22  LN -> Linear(no bias) -> GELU(exact) -> Linear(no bias)."""
    y = _ln(x, sd[p + "0.weight"], sd[p + "0.bias"])
    y = F.gelu(_lin(y, sd[p + "1.weight"]))
    return _lin(y, sd[p + "3.weight"])


def perceiver_attention(sd: SD, p: str, x: torch.Tensor, latents: torch.Tensor, heads: int, dim_head: int):
    """helpers.py:39-65.  x (b,T,n1,D) media, latents (b,T,n2,D)."""
    x = _ln(x, sd[p + "norm_media.weight"], sd[p + "norm_media.bias"])
    latents = _ln(latents, sd[p + "norm_latents.weight"], sd[p + "norm_latents.bias"])
    q = _lin(latents, sd[p + "to_q.weight"])
    kv_in = torch.cat((x, latents), dim=-2)                                # helpers.py:51
    k, v = _lin(kv_in, sd[p + "to_kv.weight"]).chunk(2, dim=-1)
    b, T = q.shape[:2]

    def split(t):                                                          # "b t n (h d) -> b h t n d"
        return t.view(b, T, t.shape[2], heads, dim_head).permute(0, 3, 1, 2, 4)

    q, k, v = split(q), split(k), split(v)
    q = q * dim_head ** -0.5                                               # helpers.py:54
    sim = q @ k.transpose(-1, -2)
    sim = sim - sim.amax(dim=-1, keepdim=True)                             # helpers.py:58
    attn = sim.softmax(dim=-1)
    out = attn @ v
    out = out.permute(0, 2, 3, 1, 4).reshape(b, T, -1, heads * dim_head)   # "b h t n d -> b t n (h d)"
    return _lin(out, sd[p + "to_out.weight"])


def perceiver_resampler(sd: SD, cfg, x: torch.Tensor, prefix: str = "perceiver.") -> torch.Tensor:
    """helpers.py:107-132.  x (b,T,F,v,D) -> (b,T,n_latents,D).  frame_embs / media_time_embs are
    None for DeeR (ctor defaults helpers.py:77-78; built at flamingo_mpt.py:97)."""
    b, T, Fr, v, D = x.shape
    x = x.reshape(b, T, Fr * v, D)
    latents = sd[prefix + "latents"].view(1, 1, -1, D).expand(b, T, -1, D)
    for l in range(cfg.perc_depth):
        latents = perceiver_attention(sd, f"{prefix}layers.{l}.0.", x, latents,
                                      cfg.perc_heads, cfg.perc_dim_head) + latents
        latents = feed_forward(sd, f"{prefix}layers.{l}.1.", latents) + latents
    return _ln(latents, sd[prefix + "norm.weight"], sd[prefix + "norm.bias"])


# ----------------------------------------------------------------------------------------------
# Gated cross attention  (helpers.py:136-279)
# ----------------------------------------------------------------------------------------------
def masked_cross_attention(sd: SD, p: str, x, media, media_locations=None, use_cached_media=False,
                           heads: int = 8, dim_head: int = 64, only_attend_immediate_media=True):
    """helpers.py:160-233.  x (B,T_txt,D), media (B,T_img,n,D_img), media_locations (B,T_txt) bool."""
    T_txt = x.shape[1]
    B, T_img, n = media.shape[:3]
    x = _ln(x, sd[p + "norm.weight"], sd[p + "norm.bias"])
    q = _lin(x, sd[p + "to_q.weight"])
    media = media.reshape(B, T_img * n, -1)
    k, v = _lin(media, sd[p + "to_kv.weight"]).chunk(2, dim=-1)

    def split(t):                                                          # "b n (h d) -> b h n d"
        return t.view(B, t.shape[1], heads, dim_head).transpose(1, 2)

    q, k, v = split(q), split(k), split(v)
    q = q * dim_head ** -0.5                                               # helpers.py:192
    sim = q @ k.transpose(-1, -2)
    text_time = None
    if media_locations is not None:
        media_time = torch.arange(T_img) + 1
        if use_cached_media:
            text_time = torch.count_nonzero(media_locations, dim=1).view(B, 1).expand(B, T_txt)
        else:
            text_time = media_locations.cumsum(dim=-1)                     # helpers.py:208
        mask_op = torch.eq if only_attend_immediate_media else torch.ge
        t2m = mask_op(text_time.view(B, 1, T_txt, 1),
                      media_time.repeat_interleave(n).view(1, 1, 1, T_img * n))
        sim = sim.masked_fill(~t2m, -torch.finfo(sim.dtype).max)           # helpers.py:218
    sim = sim - sim.amax(dim=-1, keepdim=True)
    attn = sim.softmax(dim=-1)
    if media_locations is not None and only_attend_immediate_media:
        no_media = (text_time == 0).view(B, 1, T_txt, 1)                   # helpers.py:223-229
        attn = attn.masked_fill(no_media, 0.0)
    out = (attn @ v).transpose(1, 2).reshape(B, T_txt, heads * dim_head)
    return _lin(out, sd[p + "to_out.weight"])


def gated_cross_attention_block(sd: SD, p: str, x, media, media_locations=None, use_cached_media=False,
                                heads: int = 8, dim_head: int = 64):
    """helpers.py:260-279."""
    x = masked_cross_attention(sd, p + "attn.", x, media, media_locations, use_cached_media,
                               heads, dim_head) * sd[p + "attn_gate"].tanh() + x
    x = feed_forward(sd, p + "ff.", x) * sd[p + "ff_gate"].tanh() + x
    return x


# ----------------------------------------------------------------------------------------------
# MPT decoder block  (EXTERNAL GPTBlock/MPTBlock; constructed at
# mosaic_gpt_3b.py:104-106, called at :413-417; 9B modeling_gpt_9b.py:274,465-468)
# ----------------------------------------------------------------------------------------------
def alibi_slopes(n_heads: int, alibi_bias_max: int = 8) -> torch.Tensor:
    """slope_h = 2^(-alibi_bias_max*(h+1)/H); H is a power of two for MPT-1B (16) and MPT-7B (32),
    so the "closest power of 2 + interleave" branch of later llm-foundry versions is not taken."""
    m = torch.arange(1, n_heads + 1, dtype=torch.float32) * (alibi_bias_max / n_heads)
    return 1.0 / torch.pow(2.0, m)


def mpt_attn_bias(cfg, S: int, attention_mask: Optional[torch.Tensor]) -> torch.Tensor:
    """mosaic_gpt_3b.py:158-219 (``_attn_bias``): ALiBi bias (1,H,1,S) sliced to the last S keys
    (:208) with the key-padding mask folded in as finfo.min (:215-217)."""
    j = torch.arange(1 - S, 1, dtype=torch.float32).view(1, 1, 1, S)      # -(S-1-j)
    bias = j * alibi_slopes(cfg.n_heads, cfg.alibi_bias_max).view(1, cfg.n_heads, 1, 1)
    if attention_mask is not None:
        B = attention_mask.shape[0]
        bias = bias.expand(B, -1, -1, -1).masked_fill(~attention_mask.bool().view(B, 1, 1, S),
                                                      torch.finfo(torch.float32).min)
    return bias


def mpt_block(sd: SD, p: str, cfg, x: torch.Tensor, attn_bias: torch.Tensor) -> torch.Tensor:
    """Pre-LN block, bias-free (``no_bias`` stripping at mosaic_gpt_3b.py:147-153), exact GELU."""
    B, S, d = x.shape
    H, hd = cfg.n_heads, cfg.head_dim
    nine_b = cfg.llm_name == "mpt_9b"
    ln1, ln2 = ("norm_1", "norm_2") if nine_b else ("ln_1", "ln_2")
    up, down = ("ffn.up_proj", "ffn.down_proj") if nine_b else ("mlp.mlp_up", "mlp.mlp_down")
    a = _ln(x, sd[p + ln1 + ".weight"], _get(sd, p + ln1 + ".bias"))
    qkv = _lin(a, sd[p + "attn.Wqkv.weight"])
    q, k, v = qkv.chunk(3, dim=-1)
    if cfg.attn_qk_ln:                                                     # LN over full d_model
        q = _ln(q, sd[p + "attn.q_ln.weight"], _get(sd, p + "attn.q_ln.bias"))
        k = _ln(k, sd[p + "attn.k_ln.weight"], _get(sd, p + "attn.k_ln.bias"))
    q = q.view(B, S, H, hd).transpose(1, 2)
    k = k.view(B, S, H, hd).transpose(1, 2)
    v = v.view(B, S, H, hd).transpose(1, 2)
    w = (q @ k.transpose(-1, -2)) * hd ** -0.5 + attn_bias
    causal = torch.ones(S, S, dtype=torch.bool).tril().logical_not()
    w = w.masked_fill(causal.view(1, 1, S, S), torch.finfo(w.dtype).min)
    o = (torch.softmax(w, dim=-1) @ v).transpose(1, 2).reshape(B, S, d)
    x = x + _lin(o, sd[p + "attn.out_proj.weight"])
    m = _ln(x, sd[p + ln2 + ".weight"], _get(sd, p + ln2 + ".bias"))
    x = x + _lin(F.gelu(_lin(m, sd[p + up + ".weight"])), sd[p + down + ".weight"])
    return x


# ----------------------------------------------------------------------------------------------
# Action head  (robot_flamingo/models/action_head.py)
# ----------------------------------------------------------------------------------------------
def lstm_cell(x, h, c, w_ih, w_hh, b_ih, b_hh):
    """torch.nn.LSTM single step, gate order i,f,g,o; biases b_ih + b_hh."""
    g = _lin(x, w_ih, b_ih) + _lin(h, w_hh, b_hh)
    i, f, gg, o = g.chunk(4, dim=-1)
    c2 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
    h2 = torch.sigmoid(o) * torch.tanh(c2)
    return h2, c2


def head_rnn_step(sd: SD, cfg, prefix: str, x: torch.Tensor, state):
    """One time step through the 4-layer (LayerNorm-)LSTM (action_head.py:15-64 / :72-79).
    x (B,in) ; state = (h (L,B,H), c (L,B,H)) or None.  Stored h is the raw LSTM h (pre-LN,
    action_head.py:55-58); the next layer sees LN(h)."""
    L, Hh = cfg.lstm_num_layers, cfg.head_hidden
    B = x.shape[0]
    if state is None:
        h0 = torch.zeros(L, B, Hh)
        c0 = torch.zeros(L, B, Hh)
    else:
        h0, c0 = state
    hs, cs = [], []
    for l in range(L):
        if cfg.lstm_layernorm:
            r, sfx = f"{prefix}rnn.layers.{3 * l}.", "_l0"
        else:
            r, sfx = f"{prefix}rnn.", f"_l{l}"
        h, c = lstm_cell(x, h0[l], c0[l], sd[r + "weight_ih" + sfx], sd[r + "weight_hh" + sfx],
                         sd[r + "bias_ih" + sfx], sd[r + "bias_hh" + sfx])
        hs.append(h)
        cs.append(c)
        x = h
        if cfg.lstm_layernorm:
            x = _ln(h, sd[f"{prefix}rnn.layers.{3 * l + 1}.weight"], sd[f"{prefix}rnn.layers.{3 * l + 1}.bias"])
    return x, (torch.stack(hs), torch.stack(cs))


def head_mlp(sd: SD, cfg, p: str, x: torch.Tensor, final: str, with_logits: bool = False):
    """MLPTanhHead / MLPSigmoidHead, dropout_mode='layerwise' (action_head.py:86-116,198-217,263-269):
    (Linear -> LN|Identity -> ReLU) * n_hidden -> Linear -> tanh|sigmoid.  Dropout is identity in eval."""
    n = cfg.mlp_num_hidden_layers
    for i in range(n):
        x = _lin(x, sd[f"{p}mlp.{1 + 4 * i}.weight"], sd[f"{p}mlp.{1 + 4 * i}.bias"])
        if cfg.mlp_layernorm:
            x = _ln(x, sd[f"{p}mlp.{2 + 4 * i}.weight"], sd[f"{p}mlp.{2 + 4 * i}.bias"])
        x = F.relu(x)
    logits = _lin(x, sd[f"{p}mlp.{1 + 4 * n}.weight"], sd[f"{p}mlp.{1 + 4 * n}.bias"])
    out = torch.tanh(logits) if final == "tanh" else torch.sigmoid(logits)
    return (out, logits) if with_logits else out


class OracleHead:
    """Stateful mirror of ``DeterministicDecoder`` (action_head.py:408-611) for the paths DeeR uses:
    max/avg pool over tokens (:519-520), step mode with ``update_hidden_state`` commit/stash
    semantics (:548-558) and window mode (:588-595)."""

    def __init__(self, sd: SD, cfg, prefix: str = "extra_exit."):
        self.sd, self.cfg, self.prefix = sd, cfg, prefix
        self.window_size = cfg.window_size
        self.hidden_state = None
        self.tmp_hidden_state = None
        self.history_memory: list = []
        self.last_action = False

    def clear_hidden_state(self):
        self.hidden_state = None

    def update_hidden_state(self):
        assert self.tmp_hidden_state is not None
        self.hidden_state = self.tmp_hidden_state
        self.tmp_hidden_state = None

    def __call__(self, input_feature, h_0=None, state_tensor=None, return_feature=False,
                 return_aggregate_feature=False, with_gripper_logits=False, layer_indices=None,
                 update_hidden_state=True):
        cfg = self.cfg
        if input_feature.dim() == 4:
            cur_ws = input_feature.shape[1]
            input_feature = input_feature.reshape(-1, *input_feature.shape[2:])
        else:
            cur_ws = self.window_size
        if input_feature.dim() == 3:                                        # (bs*seq, T, d) -> (bs*seq, d)
            input_feature = (input_feature.amax(dim=1) if cfg.pooling == "max"
                             else input_feature.mean(dim=1))
        input_feature = input_feature.reshape(-1, cur_ws, input_feature.shape[1])
        if getattr(cfg, "use_state", False):                                # action_head.py:524-536
            p = self.prefix
            arm = torch.relu(_lin(state_tensor[..., :6], self.sd[p + "embed_arm_state.0.weight"]) + self.sd[p + "embed_arm_state.0.bias"])
            arm = arm.view(-1, self.window_size, arm.shape[-1])
            gidx = ((state_tensor[..., -1] + 1.0) / 2).long()
            grp = torch.relu(self.sd[p + "embed_gripper_state.0.weight"][gidx])
            grp = grp.view(-1, self.window_size, grp.shape[-1])
            emb = _lin(torch.cat((arm, grp), dim=2), self.sd[p + "embed_state.weight"]) + self.sd[p + "embed_state.bias"]
            input_feature = input_feature + emb
        if input_feature.shape[1] == 1:                                     # step mode (:548)
            self.history_memory.append(input_feature)
            x, h_n = head_rnn_step(self.sd, cfg, self.prefix, input_feature[:, 0], self.hidden_state)
            if update_hidden_state:
                self.hidden_state = h_n
            else:
                self.tmp_hidden_state = h_n
            x = x.unsqueeze(1)
        else:                                                               # window mode (:588-595)
            state = h_0
            outs = []
            for t in range(input_feature.shape[1]):
                y, state = head_rnn_step(self.sd, cfg, self.prefix, input_feature[:, t], state)
                outs.append(y)
            self.hidden_state = state
            x = torch.stack(outs, dim=1)
            if self.last_action:
                x = x[:, -1].unsqueeze(1)
        actions = head_mlp(self.sd, cfg, self.prefix + "actions.", x, "tanh")
        gripper = head_mlp(self.sd, cfg, self.prefix + "gripper.", x, "sigmoid", with_gripper_logits)
        return actions, gripper


# ----------------------------------------------------------------------------------------------
# Exit criterion  (robot_flamingo/models/value_net.py)
# ----------------------------------------------------------------------------------------------
def get_delta(action1, action2, threshold_type: str = "L2"):
    """value_net.py:105-117."""
    delta = torch.abs(action1 - action2)
    if threshold_type == "mean":
        return delta.mean(-1)
    if threshold_type == "L2":
        return delta.pow(2).mean(-1).pow(0.5)
    if threshold_type == "max":
        return delta.max(-1)[0]
    if threshold_type == "cosine":
        a = F.normalize(action1, p=2.0, dim=-1, eps=1e-5)
        b = F.normalize(action2, p=2.0, dim=-1, eps=1e-5)
        return 1 - (a * b).sum(-1)
    raise NotImplementedError(threshold_type)


class OracleValueNet:
    """``ActionValueNet`` (value_net.py:72-160)."""

    def __init__(self, exit_list, exit_head: OracleHead, interval: int, window_size: int, threshold_type="L2"):
        self.exit_list = exit_list
        self.exit_head = exit_head
        self.interval = interval
        self.window_size = window_size
        self.threshold_type = threshold_type
        self.action_list: list = []

    def reset_actions(self):
        self.action_list = []

    def get_ensemble_action(self):
        assert len(self.action_list) > 0
        actions, grippers = zip(*self.action_list[-2:])
        return torch.stack(actions, dim=0).mean(0), torch.stack(grippers, dim=0).mean(0)

    def __call__(self, feats, i=None, mode="infer", rand_layer_feat=None):
        if mode == "infer":                                                 # value_net.py:120-133
            assert i > 0, "the first layer similarity is not implemented yet"
            if i - self.interval < 0:
                prev_action = self.exit_head(feats[i - 1], update_hidden_state=False)
            else:
                prev_action = self.action_list[-1]
            action = self.exit_head(feats[i], update_hidden_state=False)
            self.action_list.append(action)
            return get_delta(action[0], prev_action[0], self.threshold_type)
        # mode == 'generate'  (value_net.py:134-160): calibration deltas over a window batch
        assert 0 not in self.exit_list
        exits_action_list = []
        lang_len, d = feats[0].shape[1:]
        ws = self.window_size
        for seq_id in range(ws // 2 - 1, ws - 1):
            prev_time_feat = rand_layer_feat.reshape(-1, ws, lang_len, d)[:, :seq_id]
            exit_action = []
            for i in [0] + list(self.exit_list):
                last_time_feat = feats[i].reshape(-1, ws, lang_len, d)[:, seq_id:seq_id + 1]
                combined = torch.cat([prev_time_feat, last_time_feat], dim=1)
                self.exit_head.last_action = True
                action = self.exit_head(combined)
                self.exit_head.last_action = False
                exit_action.append(action[0].squeeze(1))
            exits_action_list.append(torch.stack(exit_action))
        exits_action_list = torch.stack(exits_action_list).permute(1, 2, 0, 3)
        prev_actions, last_actions = exits_action_list[:-1], exits_action_list[1:]
        return get_delta(prev_actions, last_actions, self.threshold_type).flatten(1, 2)


def solve_thresholds(values: torch.Tensor, real_num_exit: int, exit_ratio: float, exit_dist: str = "exp",
                     leq: bool = True, model_name: str = "mpt_dolly_3b") -> torch.Tensor:
    """The solver half of ``ExitController.set_threshold`` (value_net.py:203-260) on a gathered
    (n_stage, n_sample) matrix of deltas.  Returns T (real_num_exit,)."""
    n_stage, n_sample = values.shape
    _, sorted_idx = values.sort(dim=1, descending=not leq)
    filtered = torch.zeros(n_sample)
    T = torch.full((real_num_exit,), -1e8 if leq else 1e8)
    if exit_dist == "exp":
        probs = exit_ratio ** torch.arange(1, real_num_exit + 1)
    elif exit_dist == "gauss":
        probs = torch.tensor([math.exp(-(i - exit_ratio) ** 2 / 2.0) for i in range(real_num_exit)])
    elif exit_dist == "gamma":                                  # value_net.py:226-231: shape = exit_ratio, scale = 2
        import scipy.stats
        x = torch.arange(1, real_num_exit + 1, dtype=torch.float32)
        probs = torch.tensor([scipy.stats.gamma.pdf(val, exit_ratio, scale=2.0) for val in x], dtype=torch.float32)
    else:
        raise ValueError("Unsupported exit distribution")
    probs = probs.to(torch.float32) if probs.dtype != torch.float32 else probs
    if "mpt_9b" in model_name:
        probs[0] = 0
    probs = probs / probs.sum()
    for k in range(real_num_exit - 1):
        count = 0
        out_n = math.floor(n_sample * probs[k])
        for i in range(n_sample):
            ori_idx = sorted_idx[k][i]
            if filtered[ori_idx] == 0:
                count += 1
                if count == out_n:
                    T[k] = values[k][ori_idx]
                    break
        if leq:
            filtered.add_(values[k].le(T[k]).type_as(filtered))
        else:
            filtered.add_(values[k].ge(T[k]).type_as(filtered))
    T[real_num_exit - 1] = 1e8 if leq else -1e8
    return T


class OracleExitController:
    """``ExitController`` (value_net.py:163-297)."""

    def __init__(self, value_net: OracleValueNet, exit_id_list, steps_per_stage=1, exit_dist="exp",
                 leq=True, max_layer=12):
        self.value_net = value_net
        self.thresholds = None
        self.leq = leq
        self.exit_id_list = list(exit_id_list)
        self.num_exit = len(self.exit_id_list)
        self.steps_per_stage = steps_per_stage
        self.exit_dist = exit_dist
        self.max_layer = min(max_layer - 1, self.exit_id_list[-1])          # value_net.py:173
        self.cur_step = 0
        self.cur_exit_id = None

    @property
    def real_num_exit(self):
        return len([x for x in self.exit_id_list if x <= self.max_layer])

    def _set_threshold_value(self, thresholds):
        assert len(thresholds) == self.real_num_exit
        self.thresholds = {self.exit_id_list[i]: thresholds[i] for i in range(self.real_num_exit)}

    def set_threshold_from_values(self, values, exit_ratio, model_name="mpt_dolly_3b"):
        T = solve_thresholds(values, self.real_num_exit, exit_ratio, self.exit_dist, self.leq, model_name)
        self.thresholds = {self.exit_id_list[i]: T[i] for i in range(self.real_num_exit)}
        return T

    def set_timestep(self, t):
        self.cur_step = t

    def __call__(self, x, i: int) -> bool:
        """value_net.py:277-297."""
        assert self.thresholds is not None
        assert isinstance(i, int)
        if i not in self.exit_id_list:
            return False
        if self.cur_step % self.steps_per_stage != 0:
            return i >= self.cur_exit_id
        value = self.value_net(x, i)
        if i >= self.max_layer or bool(value <= self.thresholds[i]) is self.leq:
            self.cur_exit_id = i
            return True
        return False


# ----------------------------------------------------------------------------------------------
# Multi-exit LLM loop + full model
# ----------------------------------------------------------------------------------------------
def llm_forward(sd: SD, cfg, input_ids, attention_mask, vis_x, exit_controller=None, exit_id=None,
                prefix: str = "lang_encoder."):
    """``FlamingoLMMixin.forward`` (flamingo_lm.py:204-233) + ``MosaicGPT.forward``
    (mosaic_gpt_3b.py:274-449; 9B modeling_gpt_9b.py:352-503) + ``FlamingoLayer.forward``
    (flamingo_lm.py:46-83).  Returns (hidden_states tuple, exit_layer).
    hidden_states[i] is the OUTPUT of layer i (mosaic_gpt_3b.py:424-427); ln_f / vocab logits are
    never computed on this path (:431-449)."""
    assert exit_controller is None or exit_id is None, "Only one exit indicator can be sepcified!"
    n_layers = cfg.n_layers
    if exit_id is not None and exit_id < 0:
        exit_id += n_layers
    media_locations = input_ids == cfg.media_token_id                       # flamingo_lm.py:211
    x = F.embedding(input_ids, sd[prefix + "transformer.wte.weight"])       # ALiBi => no wpe (:341-343)
    S = input_ids.shape[1]
    attn_bias = mpt_attn_bias(cfg, S, attention_mask)
    hidden = ()
    b_idx = -1
    for b_idx in range(n_layers):
        blk = f"{prefix}transformer.blocks.{b_idx}."
        if cfg.has_xattn(b_idx):                                            # flamingo_lm.py:52-66
            x = gated_cross_attention_block(sd, blk + "gated_cross_attn_layer.", x, vis_x, media_locations,
                                            False, cfg.xattn_heads, cfg.xattn_dim_head)
        x = mpt_block(sd, blk + "decoder_layer.", cfg, x, attn_bias)
        hidden = hidden + (x,)
        if exit_id is not None and exit_id == b_idx:
            return hidden, b_idx
        if exit_controller is not None and exit_controller(hidden, b_idx):
            return hidden, b_idx
    return hidden, b_idx


class OracleDeer:
    """``MPTFlamingo`` inference path (flamingo_mpt.py:308-461) with use_gripper=True and fusion_mode='post' (the released DeeR configuration)
    or 'pre' (``cfg.fusion_mode``; flamingo_mpt.py:585-607)."""

    def __init__(self, sd: SD, cfg):
        self.sd, self.cfg = sd, cfg
        self.extra_exit = OracleHead(sd, cfg, "extra_exit.")
        self.window_size = cfg.window_size
        # layerwise_exit_eval (flamingo_mpt.py:236-244,253,450-457): one head per internal exit + lm_head for the last layer, each
        # with its OWN LSTM history (it only advances on the steps that exit at its layer)
        self.layerwise_exit_eval = bool(getattr(cfg, "layerwise_exit_eval", False))
        self.lm_exits = {}
        self.lm_head = None
        if self.layerwise_exit_eval:
            for prefix, layer in cfg.layerwise_heads():
                if prefix == "lm_head.":
                    self.lm_head = OracleHead(sd, cfg, prefix)
                else:
                    self.lm_exits[layer] = OracleHead(sd, cfg, prefix)

    def _heads(self):
        return [self.extra_exit] + list(self.lm_exits.values()) + ([self.lm_head] if self.lm_head is not None else [])

    def get_all_exit_idx(self):
        return self.cfg.exit_ids()

    def set_all_exit_window_size(self, ws):                                 # flamingo_mpt.py:275-285: every head
        old = self.extra_exit.window_size
        for h in self._heads():
            h.window_size = ws
        return old

    def clear_all_exit_memory(self):                                        # flamingo_mpt.py:287-301: every head
        for h in self._heads():
            h.hidden_state = None
            h.history_memory = []

    def encode_vision(self, vision_x, vision_gripper):
        """``_encode_multi_vision_post_fusion`` (flamingo_mpt.py:609-668)."""
        cfg = self.cfg

        def enc(v):                                                         # _encode_vision :556-583
            b, T, Fr = v.shape[:3]
            assert Fr == 1, "Only single frame supported"
            t = vit_visual_tokens(self.sd, cfg, v.reshape(b * T * Fr, *v.shape[3:]))
            return t.reshape(b, T, Fr, t.shape[1], t.shape[2])

        if getattr(cfg, "fusion_mode", "post") == "pre":                   # _encode_multi_vision_pre_fusion :585-607: ViT tokens of both cameras
            return perceiver_resampler(self.sd, cfg, torch.cat([enc(vision_x), enc(vision_gripper)], dim=3))   # cat along v -> ONE call, (b,T,n,D)
        rgb = perceiver_resampler(self.sd, cfg, enc(vision_x))
        grip = perceiver_resampler(self.sd, cfg, enc(vision_gripper),       # :656-659: own weights when sep_resampler
                                   "perceiver_gripper." if getattr(cfg, "sep_resampler", False) else "perceiver.")
        return torch.cat([rgb, grip], dim=2)                                # :661  (b,T,2n,D)

    # ---- trunk memo (test speed only) ---------------------------------------------------------------------------------------------
    # The media tokens and every layer's output are functions of (weights, frames, instruction) alone: no exit decision, threshold or
    # LSTM history enters them.  Tests that run SEVERAL oracle episodes over the same inputs (threshold probing: a never-exit pass,
    # on-policy refinements, the recorded pass) switch the memo on: the full-depth hidden states of an input are computed once and the
    # exit loop of ``llm_forward`` (controller called with the growing tuple after every layer, mosaic_gpt_3b.py:424-443) is replayed over
    # them - the same values the lazy loop produces, a few seconds per full-size forward saved.  Off by default.
    TRUNK_MEMO: Optional[dict] = None
    TRUNK_MEMO_MAX = 256

    def _memo_key(self, vision_x, vision_gripper, lang_x, attention_mask):
        import json
        if not hasattr(self, "_fp"):
            # fingerprint of the WHOLE state (two states that differ anywhere - e.g. synthetic.harden_state on top of a plain one - must not
            # share entries): every tensor contributes, small ones in full, large ones through a strided sample of 4096 elements
            acc = []
            for k in sorted(self.sd):
                t = self.sd[k].reshape(-1)
                if t.numel() > (1 << 16):
                    t = t[:: t.numel() // 4096]
                acc.append(float(t.double().sum()) + 3.0 * float(t.double().abs().sum()))
            self._fp = (len(acc), json.dumps(self.cfg.to_dict(), sort_keys=True, default=str), hash(tuple(acc)))
        f = lambda t: (tuple(t.shape), float(t.double().sum()), float(t.double().abs().sum()), float(t.reshape(-1)[:: max(1, t.numel() // 97)].double().sum()))
        return (self._fp, f(vision_x), f(vision_gripper), tuple(lang_x.reshape(-1).tolist()), tuple(attention_mask.reshape(-1).int().tolist()))

    def _forward_memo(self, vision_x, lang_x, attention_mask, vision_gripper, exit_id, exit_controller):
        memo = OracleDeer.TRUNK_MEMO
        key = self._memo_key(vision_x, vision_gripper, lang_x, attention_mask)
        if key not in memo:
            if len(memo) >= OracleDeer.TRUNK_MEMO_MAX:
                memo.pop(next(iter(memo)))
            vis_x = self.encode_vision(vision_x, vision_gripper)
            full, _ = llm_forward(self.sd, self.cfg, lang_x, attention_mask.bool(), vis_x, exit_id=self.cfg.n_layers - 1)
            memo[key] = (vis_x, full)
        vis_x, full = memo[key]
        assert exit_controller is None or exit_id is None, "Only one exit indicator can be sepcified!"
        if exit_id is not None and exit_id < 0:
            exit_id += self.cfg.n_layers
        hidden, b_idx = (), -1
        for b_idx in range(self.cfg.n_layers):                              # the loop of llm_forward over the stored layer outputs
            hidden = hidden + (full[b_idx],)
            if exit_id is not None and exit_id == b_idx:
                break
            if exit_controller is not None and exit_controller(hidden, b_idx):
                break
        return vis_x, hidden, b_idx

    def forward(self, vision_x, lang_x, attention_mask=None, vision_gripper=None, state_tensor=None,
                exit_id=None, dynamic_early_exit=False, exit_controller=None, vis_x=None):
        ctl = exit_controller if (dynamic_early_exit and exit_controller is not None) else None
        if OracleDeer.TRUNK_MEMO is not None and vis_x is None and (exit_id is not None or ctl is not None):
            vis_x, hidden, exit_layer = self._forward_memo(vision_x, lang_x, attention_mask, vision_gripper,
                                                           None if ctl is not None else exit_id, ctl)
        else:
            if vis_x is None:
                vis_x = self.encode_vision(vision_x, vision_gripper)
            hidden, exit_layer = llm_forward(self.sd, self.cfg, lang_x, attention_mask.bool(), vis_x, exit_controller=ctl,
                                             exit_id=None if ctl is not None else exit_id)
        if ctl is not None:
            exit_id = exit_layer                                            # :443-444
        if exit_id is not None:
            if exit_id < 0:
                exit_id += self.cfg.n_layers
            assert 0 <= exit_id < self.cfg.n_layers
            assert len(hidden) == exit_id + 1                               # :458
            head = self.extra_exit                                          # :450-457
            if self.layerwise_exit_eval:
                head = self.lm_head if exit_id == self.cfg.n_layers - 1 else self.lm_exits[exit_id]
            logits = head(hidden[exit_id], state_tensor=state_tensor)       # commits that head's LSTM state (:459)
            return {"logits": logits, "exit_layer": exit_layer, "hidden_states": hidden, "vis_x": vis_x}
        return {"logits": None, "exit_layer": exit_layer, "hidden_states": hidden, "vis_x": vis_x}


def postprocess_action(pose: torch.Tensor, gripper: torch.Tensor) -> torch.Tensor:
    """``ModelWrapper.step`` tail (robot_flamingo/eval/eval_utils.py:458-477): cat(pose, gripper>0.5),
    gripper -> {-1,+1}; returns (7,) float16-rounded values as float32."""
    action = torch.cat((pose, (gripper > 0.5).to(pose.dtype)), dim=2).squeeze(0)[-1]
    action = action.clone()
    action[-1] = (action[-1] - 0.5) * 2
    return action.to(torch.float16).to(torch.float32)
