"""How far is the reference's OWN evaluation arithmetic from fp32, next to this engine's?  ("an fp16 activation
arithmetic, or a written decision that bf16 stands in for amp with the measured difference".)

The README's evaluation commands run ``--precision fp32 --amp 1`` (README.md:161-167; eval_utils.py:333:
``torch.cuda.amp.autocast(enabled=self.amp)``): f32 weights, every Linear / matmul in fp16 under autocast, LayerNorm / softmax in f32.
Up to round 5 the engine computed on bf16 operands whatever the harness's ``amp`` / ``cast_dtype`` said; this tool's result (the
bf16 rounding of the WEIGHTS is 2e-2 ... 3e-2 on the action, the reference's fp16 autocast 4e-3) is why round 6 made IEEE fp16 the
default operand format (``--parts`` splits the weight rounding by tower / trunk / head).

CPU-only (runs in the build container): the oracle's restatement of the step (oracle/deer_oracle.py, pinned to the reference) on
UNROUNDED f32 weights in four arithmetics - f32; torch.autocast(cpu, float16) = the reference's amp; torch.autocast(cpu, bfloat16) = the
reference's ``--precision amp_bf16``; f32 arithmetic on bf16-ROUNDED weights = what the engine's bf16 path shares with a
``model.bfloat16()`` reference run - over an episode with LSTM carry, static exit at the last layer.  Prints the largest action
difference to the f32 run per arithmetic.  The engine's own distance to the oracle on shared bf16 weights (<= 2.7e-3, GPU suite) adds to
the last column.  usage: amp_difference.py [tiny|mid] [n_steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deer_vla_amd import synthetic as syn
from deer_vla_amd.config import deer_tiny
from oracle import deer_oracle as orc

torch.set_grad_enabled(False)
which = sys.argv[1] if len(sys.argv) > 1 else "tiny"
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 8
from deer_vla_amd.config import deer_3b
cfg = deer_3b(max_layer=12) if which == "full" else deer_tiny() if which == "tiny" else deer_tiny(image_size=112, vit_width=256, vit_layers=6, vit_heads=4, vit_mlp=1024, perc_depth=3,
                                                     d_model=512, n_heads=4, n_layers_total=12, early_exit_layer=7, head_hidden=256)
sd = syn.make_synthetic_state(cfg, 3, bf16_round=False, std="0.02" if which == "full" else "fanin")
sd_bf = syn.round_state_to_bf16(cfg, sd)
inputs = [syn.synthetic_step_inputs(cfg, s) for s in range(n_steps)]
last = cfg.n_layers - 1


def episode(state, amp_dtype=None):
    model = orc.OracleDeer(state, cfg)
    model.set_all_exit_window_size(1)
    out = []
    for rgb, grip, ids, mask in inputs:
        if amp_dtype is None:
            o = model.forward(rgb, ids, mask, grip, exit_id=last)
        else:
            with torch.autocast("cpu", dtype=amp_dtype):
                o = model.forward(rgb, ids, mask, grip, exit_id=last)
        out.append(torch.cat([o["logits"][0].float().reshape(-1), o["logits"][1].float().reshape(-1)]))
    return torch.stack(out)


def rounded(parts, fmt=torch.bfloat16):
    """sd with the GEMM operands of the named parts ("tower": ViT + Perceiver + x-attn to_kv, "trunk": MPT blocks + x-attn q / out / ff + wte,
    "head": extra_exit) rounded to `fmt`"""
    kinds = {k: v[1] for k, v in syn.param_shapes(cfg).items()}
    out = {}
    for k, t in sd.items():
        part = "head" if k.startswith(("extra_exit.", "lm_exit_modules.", "lm_head.")) else \
            ("tower" if k.startswith(("vision_encoder.", "perceiver")) or k.endswith("attn.to_kv.weight") else "trunk")
        out[k] = t.to(fmt).to(torch.float32) if (kinds.get(k) in syn.BF16_KINDS and part in parts) else t.clone()
    return out


def mixed():
    """the engine's round-6 product arithmetic restated on the oracle: tower weights fp16 AND the tower's Linears under fp16 autocast is
    approximated by fp16-rounded tower weights in f32 arithmetic (the activation side is what the GPU suite measures: <= 1.3e-3 at full size);
    trunk + head weights bf16"""
    st = rounded({"trunk", "head"})
    t16 = rounded({"tower"}, torch.float16)
    for k in st:
        if k.startswith(("vision_encoder.", "perceiver")) or k.endswith("attn.to_kv.weight"):
            st[k] = t16[k]
    return st


ref = episode(sd)
if "--parts" in sys.argv:
    print(f"config {which}: which weights' bf16 rounding moves the action (f32 arithmetic, max |action - f32| over {n_steps} steps)")
    variants = (("tower only -> bf16", lambda: rounded({"tower"})), ("tower only -> fp16", lambda: rounded({"tower"}, torch.float16)),
                ("trunk only -> bf16", lambda: rounded({"trunk"})), ("trunk only -> fp16", lambda: rounded({"trunk"}, torch.float16)),
                ("head only -> bf16", lambda: rounded({"head"})), ("head only -> fp16", lambda: rounded({"head"}, torch.float16)),
                ("tower fp16 + trunk bf16 + head bf16", mixed),
                ("everything -> fp16 (an amp run's weights: autocast casts every Linear's weight to fp16)", lambda: rounded({"tower", "trunk", "head"}, torch.float16)),
                ("everything -> bf16 (the engine's weights up to round 5; a --precision bf16 run)", lambda: sd_bf))
    for name, mk in variants:
        st = mk()
        d = (episode(st) - ref).abs()
        del st
        print(f"  {name:90s} {float(d.max()):.3e}   mean {float(d.mean()):.3e}", flush=True)
    sys.exit(0)
rows = [("reference amp (f32 weights, fp16 autocast)", episode(sd, torch.float16)),
        ("reference amp_bf16 (f32 weights, bf16 autocast)", episode(sd, torch.bfloat16)),
        ("f32 arithmetic on bf16-rounded weights (the engine's bf16 path is within 2.7e-3 of this)", episode(sd_bf))]
print(f"config {which}: d_model {cfg.d_model}, ViT {cfg.vit_layers} x {cfg.vit_width}, {cfg.n_layers} LLM layers, {n_steps} steps, static exit {last}")
print(f"{'arithmetic':90s} max |action - f32|   mean")
for name, a in rows:
    d = (a - ref).abs()
    print(f"{name:90s} {float(d.max()):.3e}          {float(d.mean()):.3e}")
