"""Import the reference's OWN Python modules in the build container (fixture generation only).

Runs only where /root/reference exists (never on the GPU box, never from tests).  The reference
needs packages the image lacks; the pieces on the hot path never touch them, so they are replaced
by empty stand-in *modules* (SURVEY.md §8c):

* ``einops_exts.rearrange_many``  - 3-line shim over ``einops.rearrange``
* ``open_clip``                  - empty module (only imported by open_flamingo/__init__ -> factory)
* ``robot_flamingo.models.normalizer`` / ``trajectory_gpt2`` - imported at the top of
  action_head.py:8-9, unused by DeterministicDecoder
* ``fvcore.nn`` / ``thop``       - imported at flamingo_mpt.py:11-12, only used by eval_flop branches

``load_mosaic_gpt()`` additionally loads /root/reference/mosaic_gpt_3b.py *from where it lies* as a
module of a synthetic package whose sibling modules (``gpt_blocks``, ``attention``, ... - the HF
``mosaicml/mpt-1b-redpajama-200b-dolly`` remote code that is NOT vendored in the reference) are this
repo's restatement (``mpt1b_standins.py``).  That pins the reference's multi-exit loop / attn-bias
plumbing (mosaic_gpt_3b.py:274-449), not the un-vendored block arithmetic.
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

REF = "/root/reference"


def install_stubs():
    if not os.path.isdir(REF):
        raise RuntimeError("/root/reference is not available: goldens can only be (re)generated in the build container")
    for p in (REF, os.path.join(REF, "open_flamingo")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import einops

    if "einops_exts" not in sys.modules:
        m = types.ModuleType("einops_exts")
        m.rearrange_many = lambda tensors, pattern, **kw: [einops.rearrange(t, pattern, **kw) for t in tensors]
        sys.modules["einops_exts"] = m
    sys.modules.setdefault("open_clip", types.ModuleType("open_clip"))
    for n, attrs in (("robot_flamingo.models.normalizer", {"LinearNormalizer": object}),
                     ("robot_flamingo.models.trajectory_gpt2", {"get_gpt_model": None})):
        if n not in sys.modules:
            mm = types.ModuleType(n)
            for k, v in attrs.items():
                setattr(mm, k, v)
            sys.modules[n] = mm
    if "fvcore" not in sys.modules:
        fv, fvn = types.ModuleType("fvcore"), types.ModuleType("fvcore.nn")
        fvn.FlopCountAnalysis = None
        fv.nn = fvn
        sys.modules["fvcore"], sys.modules["fvcore.nn"] = fv, fvn
    if "thop" not in sys.modules:
        th = types.ModuleType("thop")
        th.profile = None
        sys.modules["thop"] = th


def load_mosaic_gpt():
    """Returns the module object of the reference's mosaic_gpt_3b.py, imported as
    ``deer_mpt1b_pkg.mosaic_gpt_3b`` with this repo's stand-ins as its sibling modules."""
    install_stubs()
    here = os.path.dirname(os.path.abspath(__file__))
    pkg_name = "deer_mpt1b_pkg"
    if pkg_name + ".mosaic_gpt_3b" in sys.modules:
        return sys.modules[pkg_name + ".mosaic_gpt_3b"]
    pkg = types.ModuleType(pkg_name)
    pkg.__path__ = []            # a namespace-like package; siblings registered explicitly below
    sys.modules[pkg_name] = pkg
    spec = importlib.util.spec_from_file_location(pkg_name + "._standins", os.path.join(here, "mpt1b_standins.py"))
    st = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = st
    spec.loader.exec_module(st)
    for sib in ("attention", "gpt_blocks", "configuration_mosaic_gpt", "param_init_fns", "low_precision_layernorm"):
        m = types.ModuleType(f"{pkg_name}.{sib}")
        for k in dir(st):
            if not k.startswith("__"):
                setattr(m, k, getattr(st, k))
        sys.modules[m.__name__] = m
    spec = importlib.util.spec_from_file_location(pkg_name + ".mosaic_gpt_3b", os.path.join(REF, "mosaic_gpt_3b.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = mod
    spec.loader.exec_module(mod)
    return mod


def load_mpt_9b():
    """Returns the module object of the reference's modeling_gpt_9b.py (the MPT-7B / OpenFlamingo-9B variant of the multi-exit
    loop, :352-503), imported as ``deer_mpt7b_pkg.modeling_gpt_9b`` with this repo's stand-ins (mpt7b_standins.py) as its sibling
    modules.  transformers-5.x no longer has the two llama rotary classes the file imports by name (unused here: rope is off for
    MPT-7B, ALiBi): they are aliased to the remaining one."""
    install_stubs()
    here = os.path.dirname(os.path.abspath(__file__))
    pkg_name = "deer_mpt7b_pkg"
    if pkg_name + ".modeling_gpt_9b" in sys.modules:
        return sys.modules[pkg_name + ".modeling_gpt_9b"]
    import transformers.models.llama.modeling_llama as ml
    for n in ("LlamaDynamicNTKScalingRotaryEmbedding", "LlamaLinearScalingRotaryEmbedding"):
        if not hasattr(ml, n):
            setattr(ml, n, ml.LlamaRotaryEmbedding)
    pkg = types.ModuleType(pkg_name)
    pkg.__path__ = []
    sys.modules[pkg_name] = pkg
    spec = importlib.util.spec_from_file_location(pkg_name + "._standins", os.path.join(here, "mpt7b_standins.py"))
    st = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = st
    spec.loader.exec_module(st)
    for sib in ("attention", "blocks", "custom_embedding", "fc", "ffn", "norm", "configuration_mpt", "adapt_tokenizer",
                "hf_prefixlm_converter", "meta_init_context", "param_init_fns"):
        m = types.ModuleType(f"{pkg_name}.{sib}")
        for k in dir(st):
            if not k.startswith("__"):
                setattr(m, k, getattr(st, k))
        sys.modules[m.__name__] = m
    spec = importlib.util.spec_from_file_location(pkg_name + ".modeling_gpt_9b", os.path.join(REF, "modeling_gpt_9b.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = mod
    spec.loader.exec_module(mod)
    return mod
