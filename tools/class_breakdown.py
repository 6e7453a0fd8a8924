"""Per-kernel-class breakdown of one FULL-DEPTH control step at n_envs = B (event brackets on the launch stream, the same
machinery as bench.py's roofline pass) + the same step as one graph.  usage: class_breakdown.py [B] [workload]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from deer_vla_amd import synthetic as syn
from deer_vla_amd.config import deer_3b, deer_9b
from deer_vla_amd.engine import DeerEngine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
wl = sys.argv[2] if len(sys.argv) > 2 else "deer_b"
cfg = deer_9b(max_layer=12) if wl == "deer_9b" else deer_3b(max_layer=12)
sd = syn.make_synthetic_state(cfg, seed=0, std="0.02", bf16_round=True)
eng = DeerEngine(cfg, sd, n_envs=B)
eng.configure_exit(cfg.exit_ids(), 12, 1)
dev = eng.dev
frames = []
for s in range(4):
    per_env = [syn.synthetic_step_inputs(cfg, s, rank=e, text_seed=7 + e) for e in range(B)]
    frames.append((torch.stack([p[0] for p in per_env]).to(dev, torch.bfloat16), torch.stack([p[1] for p in per_env]).to(dev, torch.bfloat16)))
ids = torch.cat([p[2] for p in per_env]).to(dev)
r = bench.measure_roofline(eng, cfg, frames, ids)
print(f"B={B} {wl}: brackets {r['gpu_us_per_full_depth_step']} us, one graph {r['graph_us_per_full_depth_step']} us per full-depth step")
print(f"{'class':28s} {'n':>4s} {'avg_us':>8s} {'share':>7s} {'TF/s':>8s} {'GB/s':>8s}")
tot = 0
for k, c in r["classes"].items():
    tot += c["launches_per_step"]
    print(f"{k:28s} {c['launches_per_step']:4d} {c['avg_us']:8.2f} {100 * c['share']:6.1f}% {c.get('TFLOP/s', 0):8.1f} {c.get('GB/s', 0):8.1f}")
print("launches per full-depth step:", tot)
