"""Does a dynamic run reproduce the exit distribution the on-policy calibration solved for? (bench.py calibration logic)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deer_vla_amd import synthetic as syn
from deer_vla_amd.config import deer_3b
from deer_vla_amd.engine import DeerEngine
from deer_vla_amd.value_net import ExitController

max_layer = int(sys.argv[1]) if len(sys.argv) > 1 else 12
cfg = deer_3b(max_layer=max_layer)
sd = syn.make_synthetic_state(cfg, seed=0, std="0.02", bf16_round=True)
eng = DeerEngine(cfg, sd)
ctl = ExitController(None, cfg.exit_ids(), steps_per_stage=1, max_layer=max_layer)
eng.configure_exit(ctl.exit_id_list, max_layer, 1)
POOL, N = 32, 128
frames = []
for s in range(POOL):
    p = syn.synthetic_step_inputs(cfg, s)
    frames.append((p[0].cuda().bfloat16(), p[1].cuda().bfloat16()))
ids = p[2].cuda()
real = ctl.real_num_exit
thr = [-1.0] * (real - 1) + [1e5]


def run(n, shadow):
    eng.reset()
    out, vals = [], []
    for i in range(n):
        r = eng.step(frames[i % POOL][0], frames[i % POOL][1], ids, None, shadow=shadow)
        out.append(r["exit_layer"])
        vals.append(r["deltas"][:real].clone())
    return out, torch.stack(vals, dim=1)


for it in range(int(os.environ.get('ITERS', '6'))):
    eng.set_thresholds(thr)
    ex_s, values = run(N, True)
    ctl.set_threshold_from_values(values, 0.8, cfg.llm_name)
    new = ctl.threshold_list()
    hs = {e: ex_s.count(e) for e in sorted(set(ex_s))}
    eng.set_thresholds(thr)
    ex_d, _ = run(N, False)
    hd = {e: ex_d.count(e) for e in sorted(set(ex_d))}
    print(f"iter {it}: thr {[round(t, 4) for t in thr[:-1]]}\n   shadow-commit hist {hs}\n   dynamic hist       {hd}   same per step: {ex_s == ex_d}")
    damp = float(os.environ.get('DAMP', '0'))
    thr = new if (it == 0 or damp == 0) else [((a * b) ** 0.5 if (a > 0 and b > 0 and b < 1e4) else b) for a, b in zip(thr, new)] if damp == 1 else [(0.7 * a + 0.3 * b) if b < 1e4 else b for a, b in zip(thr, new)]
