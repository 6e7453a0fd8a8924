"""Stand-ins for the HF remote-code modules that /root/reference/modeling_gpt_9b.py imports relatively (``.attention``, ``.blocks``,
``.custom_embedding``, ``.fc``, ``.ffn``, ``.norm``, ``.configuration_mpt``, ``.adapt_tokenizer``, ``.hf_prefixlm_converter``,
``.meta_init_context``, ``.param_init_fns`` - modeling_gpt_9b.py:12-45).  They belong to ``mosaicml/mpt-7b`` and are NOT vendored in
the reference, so this file is this repo's own restatement of their published behaviour (bias-free pre-LN
block, fused Wqkv, ALiBi with slopes 2^(-8 i / H), exact GELU, no q/k LayerNorm), written as nn.Modules so that the reference's
*own* ``MPTModel.forward`` multi-exit loop (modeling_gpt_9b.py:352-503) can run on top of them during fixture generation.
Fixture-generation infrastructure only."""
from __future__ import annotations

import math

import torch
import torch.nn as nn
from transformers import PretrainedConfig


def is_flash_v1_installed():
    return False


def is_flash_v2_installed():
    return False


class MPTConfig(PretrainedConfig):
    model_type = "mpt"

    def __init__(self, d_model=4096, n_heads=32, n_layers=32, expansion_ratio=4, max_seq_len=2048, vocab_size=50368, resid_pdrop=0.0,
                 emb_pdrop=0.0, learned_pos_emb=False, attn_config=None, ffn_config=None, init_device="cpu", logit_scale=None,
                 no_bias=True, embedding_fraction=1.0, norm_type="low_precision_layernorm", use_cache=False, init_config=None,
                 fc_type="torch", tie_word_embeddings=True, **kwargs):
        self.d_model, self.n_heads, self.n_layers, self.expansion_ratio = d_model, n_heads, n_layers, expansion_ratio
        self.max_seq_len, self.vocab_size, self.resid_pdrop, self.emb_pdrop = max_seq_len, vocab_size, resid_pdrop, emb_pdrop
        self.learned_pos_emb = learned_pos_emb
        self.attn_config = attn_config or {"attn_type": "multihead_attention", "attn_pdrop": 0.0, "attn_impl": "torch", "qk_ln": False,
                                           "clip_qkv": None, "softmax_scale": None, "prefix_lm": False, "attn_uses_sequence_id": False,
                                           "alibi": True, "alibi_bias_max": 8, "rope": False}
        self.ffn_config = ffn_config or {"ffn_type": "mptmlp"}
        self.init_device, self.logit_scale, self.no_bias = init_device, logit_scale, no_bias
        self.embedding_fraction, self.norm_type, self.use_cache = embedding_fraction, norm_type, use_cache
        self.init_config = init_config or {"name": "noop_"}
        self.fc_type = fc_type
        super().__init__(tie_word_embeddings=tie_word_embeddings, **kwargs)

    def _validate_config(self):
        assert self.d_model % self.n_heads == 0


class SharedEmbedding(nn.Embedding):
    def forward(self, input, unembed: bool = False):
        if unembed:
            return torch.nn.functional.linear(input, self.weight)
        return super().forward(input)


class LPLayerNorm(nn.LayerNorm):
    """"low precision" LayerNorm = the same arithmetic executed in the autocast dtype."""


NORM_CLASS_REGISTRY = {"layernorm": nn.LayerNorm, "low_precision_layernorm": LPLayerNorm}
FC_CLASS_REGISTRY = {"torch": nn.Linear}


def gen_slopes(n_heads, alibi_bias_max=8, device=None, return_1d=False):
    _n = 2 ** math.ceil(math.log2(n_heads))
    m = torch.arange(1, _n + 1, dtype=torch.float32, device=device).mul(alibi_bias_max / _n)
    slopes = 1.0 / torch.pow(2, m)
    if _n != n_heads:
        slopes = torch.concat([slopes[1::2], slopes[::2]])[:n_heads]
    return slopes if return_1d else slopes.view(1, n_heads, 1, 1)


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


def build_alibi_bias(n_heads, seq_len, full=False, alibi_bias_max=8, device=None, dtype=None):
    b = torch.arange(1 - seq_len, 1, dtype=torch.int32, device=device).view(1, 1, 1, seq_len)
    if full:
        b = b - torch.arange(1 - seq_len, 1, dtype=torch.int32, device=device).view(1, 1, seq_len, 1)
        b = b.abs().mul(-1)
    return (b * gen_slopes(n_heads, alibi_bias_max, device=device)).to(dtype=dtype)


def build_attn_bias(attn_impl, attn_bias, n_heads, seq_len, causal=False, alibi=False, alibi_bias_max=8):
    if attn_impl == "flash":
        return None
    if alibi:
        attn_bias = attn_bias.add(build_alibi_bias(n_heads, seq_len, full=not causal, alibi_bias_max=alibi_bias_max,
                                                   device=attn_bias.device, dtype=attn_bias.dtype))
    return attn_bias


class _Attn(nn.Module):
    def __init__(self, d_model, n_heads, device=None):
        super().__init__()
        self.d_model, self.n_heads = d_model, n_heads
        self.Wqkv = nn.Linear(d_model, 3 * d_model, device=device)
        self.out_proj = nn.Linear(d_model, d_model, device=device)

    def forward(self, x, past_key_value=None, attn_bias=None, attention_mask=None, is_causal=True, needs_weights=False, **kw):
        B, S, d = x.shape
        H, hd = self.n_heads, d // self.n_heads
        q, k, v = self.Wqkv(x).chunk(3, dim=2)
        q = q.view(B, S, H, hd).transpose(1, 2)
        k = k.view(B, S, H, hd).transpose(1, 2)
        v = v.view(B, S, H, hd).transpose(1, 2)
        w = q.matmul(k.transpose(-1, -2)) * hd ** -0.5
        if attn_bias is not None:
            w = w + attn_bias[:, :, -S:, -S:] if attn_bias.size(-2) != 1 else w + attn_bias[:, :, :, -S:]
        min_val = torch.finfo(w.dtype).min
        if attention_mask is not None:
            w = w.masked_fill(~attention_mask.view(B, 1, 1, S), min_val)
        if is_causal:
            cm = torch.ones(S, S, dtype=torch.bool, device=x.device).tril().logical_not()
            w = w.masked_fill(cm.view(1, 1, S, S), min_val)
        p = torch.softmax(w, dim=-1)
        o = p.matmul(v).transpose(1, 2).reshape(B, S, d)
        return self.out_proj(o), (p if needs_weights else None), past_key_value


ATTN_CLASS_REGISTRY = {"multihead_attention": _Attn}


class MPTMLP(nn.Module):
    def __init__(self, d_model, expansion_ratio, device=None, **kw):
        super().__init__()
        self.up_proj = nn.Linear(d_model, expansion_ratio * d_model, device=device)
        self.act = nn.GELU(approximate="none")
        self.down_proj = nn.Linear(expansion_ratio * d_model, d_model, device=device)

    def forward(self, x):
        return self.down_proj(self.act(self.up_proj(x)))


FFN_CLASS_REGISTRY = {"mptmlp": MPTMLP}


def build_ffn(d_model, expansion_ratio, device=None, **kw):
    return MPTMLP(d_model, expansion_ratio, device=device)


class MPTBlock(nn.Module):
    def __init__(self, d_model=None, n_heads=None, expansion_ratio=4, norm_type="low_precision_layernorm", device=None, **kwargs):
        super().__init__()
        norm = NORM_CLASS_REGISTRY[norm_type.lower()]
        self.norm_1 = norm(d_model, device=device)
        self.attn = _Attn(d_model, n_heads, device=device)
        self.norm_2 = norm(d_model, device=device)
        self.ffn = MPTMLP(d_model, expansion_ratio, device=device)

    def forward(self, x, past_key_value=None, attn_bias=None, rotary_emb_w_meta_info=None, attention_mask=None, is_causal=True,
                output_attentions=False, alibi_slopes=None, flash_attn_padding_info=None):
        a = self.norm_1(x)
        b, w, past_key_value = self.attn(a, past_key_value=past_key_value, attn_bias=attn_bias, attention_mask=attention_mask,
                                         is_causal=is_causal, needs_weights=output_attentions)
        x = x + b
        x = x + self.ffn(self.norm_2(x))
        return x, w, past_key_value


def _noop(module=None, **kwargs):
    return None


MODEL_INIT_REGISTRY = {"noop_": _noop}
generic_param_init_fn_ = _noop


class AutoTokenizerForMOD:          # adapt_tokenizer.py: unused on this path
    pass


def adapt_tokenizer_for_denoising(tokenizer):
    return tokenizer


def add_bidirectional_mask_if_missing(batch):
    return batch


def convert_hf_causal_lm_to_prefix_lm(model):
    return model


class init_empty_weights:           # meta_init_context.py
    def __init__(self, include_buffers=False):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False
