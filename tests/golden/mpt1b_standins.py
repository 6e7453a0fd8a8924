"""Stand-ins for the HF remote-code modules that /root/reference/mosaic_gpt_3b.py imports relatively
(``.attention``, ``.gpt_blocks``, ``.configuration_mosaic_gpt``, ``.param_init_fns``,
``.low_precision_layernorm`` - mosaic_gpt_3b.py:18-23).  They belong to
``mosaicml/mpt-1b-redpajama-200b-dolly`` and are NOT vendored in the reference, so this file is this
repo's own restatement of their published behaviour, written as nn.Modules so
that the reference's *own* ``MosaicGPT.forward`` loop can run on top of them during fixture generation.
Fixture-generation infrastructure only.
"""
from __future__ import annotations

import torch
import torch.nn as nn
import torch.nn.functional as F
from transformers import PretrainedConfig


class MosaicGPTConfig(PretrainedConfig):
    model_type = "mosaic_gpt"

    def __init__(self, d_model=2048, n_heads=16, n_layers=24, mlp_ratio=4, max_seq_len=2048, vocab_size=50368,
                 attn_pdrop=0.0, resid_pdrop=0.0, emb_pdrop=0.0, attn_impl="torch", attn_qk_ln=True,
                 attn_clip_qkv=None, softmax_scale=None, prefix_lm=False, attn_uses_sequence_id=False,
                 alibi=True, alibi_bias_max=8, init_device="cpu", logit_scale=None, no_bias=True, verbose=0,
                 param_init_fn="noop_", embedding_fraction=1.0, low_precision_layernorm=False, use_cache=False,
                 **kwargs):
        self.d_model, self.n_heads, self.n_layers, self.mlp_ratio = d_model, n_heads, n_layers, mlp_ratio
        self.max_seq_len, self.vocab_size = max_seq_len, vocab_size
        self.attn_pdrop, self.resid_pdrop, self.emb_pdrop = attn_pdrop, resid_pdrop, emb_pdrop
        self.attn_impl, self.attn_qk_ln, self.attn_clip_qkv = attn_impl, attn_qk_ln, attn_clip_qkv
        self.softmax_scale, self.prefix_lm = softmax_scale, prefix_lm
        self.attn_uses_sequence_id = attn_uses_sequence_id
        self.alibi, self.alibi_bias_max = alibi, alibi_bias_max
        self.init_device, self.logit_scale, self.no_bias, self.verbose = init_device, logit_scale, no_bias, verbose
        self.param_init_fn, self.embedding_fraction = param_init_fn, embedding_fraction
        self.low_precision_layernorm = low_precision_layernorm
        self.use_cache = use_cache
        super().__init__(**kwargs)


def _noop(module=None, **kwargs):
    return None


MODEL_INIT_REGISTRY = {"noop_": _noop}


class LPLayerNorm(nn.LayerNorm):
    """"low precision" LayerNorm = the same arithmetic executed in the autocast dtype."""


def attn_bias_shape(attn_impl, n_heads, seq_len, alibi, prefix_lm, causal, use_sequence_id):
    if attn_impl == "flash":
        return None
    if alibi:
        if (prefix_lm or not causal) or use_sequence_id:
            return (1, n_heads, seq_len, seq_len)
        return (1, n_heads, 1, seq_len)
    if prefix_lm or use_sequence_id:
        return (1, 1, seq_len, seq_len)
    return None


def alibi_bias(n_heads, seq_len, full=False, alibi_bias_max=8, device=None, dtype=None):
    b = torch.arange(1 - seq_len, 1, dtype=dtype, device=device).view(1, 1, 1, seq_len)
    if full:
        b = b - torch.arange(1 - seq_len, 1, dtype=dtype, device=device).view(1, 1, seq_len, 1)
        b = b.abs().mul(-1)
    m = torch.arange(1, n_heads + 1, dtype=dtype, device=device).mul(alibi_bias_max / n_heads)
    return b * (1.0 / (2 ** m.view(1, n_heads, 1, 1)))


def attn_bias(attn_impl, attn_bias, n_heads, seq_len, causal=False, alibi=False, alibi_bias_max=8):
    if attn_impl == "flash":
        return None
    if alibi:
        attn_bias = attn_bias.add(alibi_bias(n_heads, seq_len, full=not causal, alibi_bias_max=alibi_bias_max,
                                             device=attn_bias.device, dtype=attn_bias.dtype))
    return attn_bias


class _Attn(nn.Module):
    def __init__(self, d_model, n_heads, attn_qk_ln, device=None):
        super().__init__()
        self.d_model, self.n_heads, self.attn_qk_ln = d_model, n_heads, attn_qk_ln
        self.Wqkv = nn.Linear(d_model, 3 * d_model, device=device)
        if attn_qk_ln:
            self.q_ln = nn.LayerNorm(d_model, device=device)
            self.k_ln = nn.LayerNorm(d_model, device=device)
        self.out_proj = nn.Linear(d_model, d_model, device=device)

    def forward(self, x, past_key_value=None, attn_bias=None, attention_mask=None, is_causal=True):
        B, S, d = x.shape
        H, hd = self.n_heads, d // self.n_heads
        q, k, v = self.Wqkv(x).chunk(3, dim=2)
        if self.attn_qk_ln:
            q, k = self.q_ln(q), self.k_ln(k)
        q = q.view(B, S, H, hd).transpose(1, 2)
        k = k.view(B, S, H, hd).transpose(1, 2)
        v = v.view(B, S, H, hd).transpose(1, 2)
        w = q.matmul(k.transpose(-1, -2)) * hd ** -0.5
        if attn_bias is not None:
            w = w + attn_bias
        min_val = torch.finfo(w.dtype).min
        if attention_mask is not None:
            w = w.masked_fill(~attention_mask.view(B, 1, 1, S), min_val)
        if is_causal:
            cm = torch.ones(S, S, dtype=torch.bool, device=x.device).tril().logical_not()
            w = w.masked_fill(cm.view(1, 1, S, S), min_val)
        o = torch.softmax(w, dim=-1).matmul(v).transpose(1, 2).reshape(B, S, d)
        return self.out_proj(o), None, past_key_value


class _MLP(nn.Module):
    def __init__(self, d_model, mlp_ratio, device=None):
        super().__init__()
        self.mlp_up = nn.Linear(d_model, mlp_ratio * d_model, device=device)
        self.mlp_act = nn.GELU(approximate="none")
        self.mlp_down = nn.Linear(mlp_ratio * d_model, d_model, device=device)

    def forward(self, x):
        return self.mlp_down(self.mlp_act(self.mlp_up(x)))


class GPTBlock(nn.Module):
    def __init__(self, d_model=None, n_heads=None, mlp_ratio=4, attn_qk_ln=True, device=None, **kwargs):
        super().__init__()
        self.ln_1 = nn.LayerNorm(d_model, device=device)
        self.attn = _Attn(d_model, n_heads, attn_qk_ln, device=device)
        self.ln_2 = nn.LayerNorm(d_model, device=device)
        self.mlp = _MLP(d_model, mlp_ratio, device=device)

    def forward(self, x, past_key_value=None, attn_bias=None, attention_mask=None, is_causal=True):
        a = self.ln_1(x)
        b, _, past_key_value = self.attn(a, past_key_value=past_key_value, attn_bias=attn_bias,
                                         attention_mask=attention_mask, is_causal=is_causal)
        x = x + b
        x = x + self.mlp(self.ln_2(x))
        return x, past_key_value
