"""LSTM head with the recurrent half computed once per step (DEER_HEAD_PRE=1, default) against the fused form (DEER_HEAD_PRE=0): two sibling
engines on identical inputs, static and dynamic steps back to back; prints the largest difference of actions / LSTM state per leg."""
import os, sys
sys.path.insert(0, "/root/repo")
import torch
from deer_vla_amd import synthetic as syn
from deer_vla_amd.config import deer_3b
from deer_vla_amd.engine import DeerEngine
cfg = deer_3b(max_layer=12)
sd = syn.make_synthetic_state(cfg, 0, std="0.02", bf16_round=True)
os.environ["DEER_HEAD_PRE"] = "1"
a = DeerEngine(cfg, sd)
os.environ["DEER_HEAD_PRE"] = "0"
b = DeerEngine(cfg, None, weights_from=a)
frames = [syn.synthetic_step_inputs(cfg, s) for s in range(8)]
for e in (a, b):
    e.configure_exit(cfg.exit_ids(), 12, 1)
def leg(name, n, **kw):
    for e in (a, b): e.reset()
    worst = 0.0; first = None; exits = [[], []]
    for i in range(n):
        f = frames[i % 8]
        ra = a.step(f[0], f[1], f[2], None, **kw)
        rb = b.step(f[0], f[1], f[2], None, **kw)
        d = float((ra["pose"] - rb["pose"]).abs().max())
        exits[0].append(ra["exit_layer"]); exits[1].append(rb["exit_layer"])
        if first is None and (ra["exit_layer"] != rb["exit_layer"]): first = (i, ra["exit_layer"], rb["exit_layer"], d)
        worst = max(worst, d)
    torch.cuda.synchronize()
    ds = float((a.h_state - b.h_state).abs().max())
    print(name, "steps", n, "max |pose diff|", worst, "final |h_state diff|", ds, "first exit mismatch", first, "exits equal", exits[0] == exits[1], flush=True)
leg("static exit 5", 60, exit_id=5)
leg("static exit 11", 60, exit_id=11)
for e in (a, b): e.set_thresholds([-1.0, -1.0, 1e5, 1e5, 1e5, 1e5])
leg("dynamic forced exit at layer 5", 60)
for e in (a, b): e.set_thresholds([0.021694, 0.02259, 0.020928, 0.024579, 0.020147, 1e8])
leg("dynamic, calibrated thresholds", 200)
