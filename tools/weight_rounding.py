"""What the arena's bf16 weight rounding costs against a reference that keeps f32 weights (full-size 3B, synthetic N(0, 0.02^2) weights,
static exit at the last layer): the bf16 arithmetic vs the fp32 arithmetic (f32 / hi+lo weight copies).  usage: weight_rounding.py"""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests'))
import torch
from deer_vla_amd import synthetic as syn
from deer_vla_amd.config import deer_3b
from deer_vla_amd.engine import DeerEngine
from oracle import deer_oracle as orc
torch.set_num_threads(32)
cfg = deer_3b(max_layer=12)
sd = syn.make_synthetic_state(cfg, 0, std="0.02", bf16_round=False)      # genuine f32 weights: the bf16 arithmetic rounds GEMM operands, the fp32 one keeps them
for prec in ("fp32", "bf16"):
    eng = DeerEngine(cfg, sd, precision=prec)
    eng.configure_exit(cfg.exit_ids(), 12, 1)
    worst = 0.0
    for s in range(3):
        rgb, grip, ids, mask = syn.synthetic_step_inputs(cfg, s)
        od = orc.OracleDeer(sd, cfg); od.set_all_exit_window_size(1)
        with torch.no_grad():
            o = od.forward(rgb, ids, mask, grip, exit_id=11)
        a_o = torch.cat([o["logits"][0].reshape(-1), o["logits"][1].reshape(-1)])[:7]
        eng.reset()
        r = eng.step(rgb, grip, ids, mask, exit_id=11)
        a_e = torch.cat([r["pose"], torch.tensor([r["gripper"]])])
        worst = max(worst, float((a_e - a_o).abs().max()))
    print(prec, "arithmetic vs the oracle on UNROUNDED f32 weights: worst action err %.2e" % worst)
    del eng
