"""Full-size (MPT-1B / ViT-L/14) dynamic episode against the fp32 oracle: exit layers must match step for step, actions within
1e-2.  usage: long_episode_parity.py [steps]   (oracle ~1 s per step per pass on 32 threads; three passes)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
torch.set_num_threads(32)
from deer_vla_amd import synthetic as syn
from deer_vla_amd.config import deer_3b
from deer_vla_amd.engine import DeerEngine
import test_engine_parity as tp

n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
cfg = deer_3b(max_layer=12)
sd = syn.make_synthetic_state(cfg, 0, std="0.02", bf16_round=True)
inputs = tp.make_inputs(cfg, n)
t0 = time.time()
thr, margin = tp.probe_thresholds(cfg, sd, inputs, 12, iters=2)
ref, rec, _ = tp.oracle_episode(cfg, sd, inputs, thr, 12)
print(f"oracle passes {time.time() - t0:.0f}s; thresholds {[round(t, 4) for t in thr[:-1]]}; min relative margin {margin:.3f}")
eng = DeerEngine(cfg, sd)
eng.configure_exit(cfg.exit_ids(), 12, 1)
eng.set_thresholds(thr)
eng.reset()
bad, worst = 0, 0.0
exits = []
for s, (rgb, grip, ids, mask) in enumerate(inputs):
    r = eng.step(rgb, grip, ids, mask)
    exits.append(r["exit_layer"])
    e = float((r["pose"] - ref[s][1]).abs().max())
    worst = max(worst, e, abs(r["gripper"] - ref[s][2]))
    if r["exit_layer"] != ref[s][0]:
        bad += 1
        print("EXIT MISMATCH at step", s, r["exit_layer"], ref[s][0], r["deltas"][:6].tolist())
print(f"{n} steps: exit layers {sorted(set(exits))} hist {[exits.count(e) for e in sorted(set(exits))]}, mismatches {bad}, worst |action - oracle| {worst:.2e}")
sys.exit(1 if bad or worst > 1e-2 else 0)
