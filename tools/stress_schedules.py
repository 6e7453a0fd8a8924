"""Race hunt: the pipelined schedule (pieces on two streams, host polling the verdict mirror) against the single-graph
schedule on identical inputs, step by step, for thousands of control steps with exits that keep changing.
usage: stress_schedules.py [tiny|full] [steps] [n_envs] [early_exit_layer (tiny only)] [reset period]
(early_exit_layer 4 gives exits {1,3,4}: after an exit at layer 3 a speculative head evaluation for layer 4 is still queued
when the host moves on - the case reset() must drain)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deer_vla_amd import synthetic as syn
from deer_vla_amd.config import deer_3b, deer_tiny
from deer_vla_amd.engine import DeerEngine

which = sys.argv[1] if len(sys.argv) > 1 else "tiny"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
EEL = int(sys.argv[4]) if len(sys.argv) > 4 else 5
RESET = int(sys.argv[5]) if len(sys.argv) > 5 else 211
cfg = deer_tiny(early_exit_layer=EEL) if which == "tiny" else deer_3b(max_layer=12)
sd = syn.make_synthetic_state(cfg, seed=1, std="fanin" if which == "tiny" else "0.02", bf16_round=True)
a = DeerEngine(cfg, sd, n_envs=B, segmented=True)
b = DeerEngine(cfg, sd, n_envs=B, segmented=False)
for e in (a, b):
    e.configure_exit(cfg.exit_ids(), 12, 1)
POOL = 16
frames = []
for s in range(POOL):
    per = [syn.synthetic_step_inputs(cfg, s, rank=e, text_seed=7 + e) for e in range(B)]
    frames.append((torch.stack([p[0] for p in per]).cuda().bfloat16(), torch.stack([p[1] for p in per]).cuda().bfloat16()))
ids = torch.cat([p[2] for p in per]).cuda()
# calibrate thresholds from a shadow pass so that exits are spread, then jitter them during the run
real = a.real_num_exit
a.set_thresholds([-1.0] * (real - 1) + [1e5])
vals = []
for s in range(32):
    r = a.step(frames[s % POOL][0], frames[s % POOL][1], ids, None, shadow=True)
    for re in (r if B > 1 else [r]):
        vals.append(re["deltas"][:real].clone())
med = torch.stack(vals).nan_to_num(0).median(0)[0]
g = torch.Generator().manual_seed(0)
hist = {}
t0 = time.time()
bad = 0
a.reset(); b.reset()
for s in range(steps):
    if s % 37 == 0:
        thr = [float(med[k]) * float(torch.empty(1).uniform_(0.3, 3.0, generator=g)) for k in range(real - 1)] + [1e5]
        a.set_thresholds(thr); b.set_thresholds(thr)
    if s % RESET == 0:
        a.reset(); b.reset()
    rgb, grip = frames[s % POOL]
    ra = a.step(rgb, grip, ids, None)
    rb = b.step(rgb, grip, ids, None)
    for x, y in zip(ra if B > 1 else [ra], rb if B > 1 else [rb]):
        hist[x["exit_layer"]] = hist.get(x["exit_layer"], 0) + 1
        same = x["exit_layer"] == y["exit_layer"] and torch.equal(x["pose"], y["pose"]) and x["gripper"] == y["gripper"] \
            and x["n_evals"] == y["n_evals"]
        if not same:
            bad += 1
            if bad < 5:
                print("MISMATCH step", s, x["exit_layer"], y["exit_layer"], x["pose"], y["pose"], x["n_evals"], y["n_evals"])
torch.cuda.synchronize()
assert torch.equal(a.h_state, b.h_state) and torch.equal(a.c_state, b.c_state)
print(f"{which} B={B}: {steps} steps in {time.time() - t0:.1f}s, exit histogram {dict(sorted(hist.items()))}, mismatches {bad}")
sys.exit(1 if bad else 0)
