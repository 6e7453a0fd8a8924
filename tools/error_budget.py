"""Where the bf16 path's distance to the fp32 oracle comes from (full-size 3B model, static exit at the last layer, first step of an
episode): the three coarse operators are fed either their own upstream result or the ORACLE's, so every stage's own contribution
to the action error shows up alone.  usage: error_budget.py [n_steps] [3b|tiny] [easy|hard] [text_len]
(hard: synthetic.harden_state - outlier channels, wide LayerNorm gains, x-attn gates near +-3, saturated LSTM biases: tests/test_hard_inputs.py)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deer_vla_amd import synthetic as syn, ops
from deer_vla_amd.config import deer_3b, deer_tiny
from oracle import deer_oracle as orc

torch.set_num_threads(32)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3
size = sys.argv[2] if len(sys.argv) > 2 else "3b"
hard = len(sys.argv) > 3 and sys.argv[3] == "hard"
TL = int(sys.argv[4]) if len(sys.argv) > 4 else 14
cfg = deer_3b(max_layer=12) if size == "3b" else deer_tiny()
sd = syn.make_synthetic_state(cfg, seed=0, std="0.02", bf16_round=True) if size == "3b" else syn.make_synthetic_state(cfg, 3, bf16_round=True)
if hard:
    sd = syn.harden_state(cfg, sd, seed=TL if size == "tiny" else 0)
print(f"config {size}, {'hard' if hard else 'easy'} weights, {TL} text tokens")
m = ops.NativeModel(cfg, sd)
S, E = cfg.image_size, cfg.n_layers - 1
rows = []
for s in range(N):
    rgb, grip, ids, mask = syn.synthetic_step_inputs(cfg, s, text_len=TL)
    rgb, grip = rgb.bfloat16().float(), grip.bfloat16().float()             # both arms see the same (bf16-exact) frames
    od = orc.OracleDeer(sd, cfg)
    od.set_all_exit_window_size(1)                                          # step mode (eval_utils.py:246)
    with torch.no_grad():
        t_rgb = orc.vit_visual_tokens(sd, cfg, rgb.reshape(1, 3, S, S))
        t_grip = orc.vit_visual_tokens(sd, cfg, grip.reshape(1, 3, S, S))
        tok_o = torch.cat([t_rgb, t_grip])                                    # (2, P, W)
        vis_o = od.encode_vision(rgb.reshape(1, 1, 1, 3, S, S), grip.reshape(1, 1, 1, 3, S, S))   # (1,1,128,W)
        ro = od.forward(None, ids, mask, exit_id=E, vis_x=vis_o)
    act_o = torch.cat([ro["logits"][0].reshape(-1), ro["logits"][1].reshape(-1)])[:7]
    images = torch.stack([rgb.reshape(3, S, S), grip.reshape(3, S, S)]).cuda()

    def llm(media):
        m.reset()
        ctl, hidden = torch.ops.deer.llm_early_exit(ids.cuda(), None, media, m.handle, E, False)
        torch.cuda.synchronize()
        r = ops.decode_ctl(ctl)[0]
        return torch.cat([r["pose"], torch.tensor([r["gripper"]])]), hidden

    tok_e = torch.ops.deer.vit_l14_encode(images, m.handle)
    med_ee = torch.ops.deer.perceiver_resample(tok_e, m.handle)               # engine ViT -> engine Perceiver
    med_oe = torch.ops.deer.perceiver_resample(tok_o.cuda(), m.handle)        # oracle ViT tokens -> engine Perceiver
    med_oo = vis_o.reshape(-1, cfg.vit_width).cuda().bfloat16()               # oracle media (rounded to bf16 once)
    a_full, _ = llm(med_ee)
    a_perc, _ = llm(med_oe)
    a_llm, h_llm = llm(med_oo)
    e = lambda a: float((a - act_o).abs().max())
    tok_err = float((tok_e.cpu() - tok_o).abs().max() / tok_o.abs().max())
    med_err = float((med_ee.float().cpu() - vis_o.reshape(-1, cfg.vit_width)).abs().max() / vis_o.abs().max())
    hid_err = float((h_llm[E, : ids.shape[1]].cpu() - ro["hidden_states"][E][0]).abs().max() / ro["hidden_states"][E].abs().max())
    rows.append((e(a_full), e(a_perc), e(a_llm), tok_err, med_err, hid_err))
    print(f"step {s}: action err  whole path {e(a_full):.2e} | oracle ViT tokens {e(a_perc):.2e} | oracle media (bf16) {e(a_llm):.2e} || "
          f"ViT tokens rel {tok_err:.2e}, media rel {med_err:.2e}, last hidden (oracle media) rel {hid_err:.2e}", flush=True)
m.close()
