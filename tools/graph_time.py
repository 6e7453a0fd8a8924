"""Full-depth control step as ONE graph at n_envs = B: microseconds per step (quick A/B of kernel knobs via environment variables)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deer_vla_amd import synthetic as syn
from deer_vla_amd.config import deer_3b
from deer_vla_amd.engine import DeerEngine
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = deer_3b(max_layer=12)
sd = syn.make_synthetic_state(cfg, seed=0, std="0.02", bf16_round=True)
eng = DeerEngine(cfg, sd, n_envs=B)
eng.configure_exit(cfg.exit_ids(), 12, 1)
per_env = [syn.synthetic_step_inputs(cfg, 0, rank=e, text_seed=7 + e) for e in range(B)]
rgb = torch.stack([p[0] for p in per_env]).cuda().bfloat16(); grip = torch.stack([p[1] for p in per_env]).cuda().bfloat16()
ids = torch.cat([p[2] for p in per_env]).cuda()
for _ in range(3):
    eng.step(rgb, grip, ids, None, exit_id=11, sync=False)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    eng.step(rgb, grip, ids, None, exit_id=11, sync=False)
e1.record(); torch.cuda.synchronize()
print(f"B={B} knobs {dict((k, v) for k, v in os.environ.items() if k.startswith('DEER_'))}: {1e3 * e0.elapsed_time(e1) / 20:.1f} us per full-depth step")
