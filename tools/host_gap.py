"""Host-side share of a dynamic control step: time from step() entry to the first graph submission (the GPU is idle until then),
and from the verdict to step() returning.  usage: host_gap.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deer_vla_amd import synthetic as syn
from deer_vla_amd.config import deer_3b
from deer_vla_amd.engine import DeerEngine

cfg = deer_3b(max_layer=12)
sd = syn.make_synthetic_state(cfg, seed=0, std="0.02", bf16_round=True)
eng = DeerEngine(cfg, sd)
eng.configure_exit(cfg.exit_ids(), 12, 1)
eng.set_thresholds([1e5, 1e5, 1e5, 1e5, 1e5, 1e8])      # every step exits at the first check
rgb, grip, ids, mask = syn.synthetic_step_inputs(cfg, 0)
rgb, grip, ids = rgb.cuda().bfloat16(), grip.cuda().bfloat16(), ids.cuda()
for _ in range(5):
    eng.step(rgb, grip, ids, None)
marks = {}
orig = eng._replay_vision_chains
def patched(*a, **k):
    marks["submit"] = time.perf_counter()
    return orig(*a, **k)
eng._replay_vision_chains = patched
N = 200
pre = tot = 0.0
t_prev_end = None
gap = 0.0
torch.cuda.synchronize()
t_all = time.perf_counter()
for i in range(N):
    t0 = time.perf_counter()
    r = eng.step(rgb, grip, ids, None)
    t1 = time.perf_counter()
    pre += marks.get("submit", t0) - t0
    tot += t1 - t0
    if t_prev_end is not None:
        gap += t0 - t_prev_end
    t_prev_end = t1
t_all = time.perf_counter() - t_all
print(f"exit layer {r['exit_layer']}: step() {1e6 * tot / N:.1f} us, of which before the first graph submission {1e6 * pre / N:.1f} us; "
      f"loop overhead between steps {1e6 * gap / (N - 1):.1f} us; wall per step {1e6 * t_all / N:.1f} us")
