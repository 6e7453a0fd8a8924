"""GPU-side timeline of the pieces of one dynamic control step (events after every piece)."""
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
exit_at = int(sys.argv[1]) if len(sys.argv) > 1 else 1       # index of the exit that fires
thr = [-1.0] * 6
thr[exit_at] = 1e5
thr[5] = 1e5
eng.set_thresholds(thr)
rgb, grip, ids, mask = syn.synthetic_step_inputs(cfg, 0)
rgb, grip, ids = rgb.cuda().bfloat16(), grip.cuda().bfloat16(), ids.cuda()
for _ in range(5):
    r = eng.step(rgb, grip, ids, None)
print("exit layer", r["exit_layer"])
acc = {}
N = 20
for _ in range(N):
    torch.cuda.synchronize()
    eng._trace = []
    t0 = time.perf_counter()
    eng.step(rgb, grip, ids, None)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    tr, eng._trace = eng._trace, None
    s_ev, s_t = tr[0][1], tr[0][2]
    for lab, ev, th in tr[1:]:
        a = acc.setdefault(lab, [0.0, 0.0])
        a[0] += s_ev.elapsed_time(ev) * 1e3 / N
        a[1] += (th - s_t) * 1e6 / N
    a = acc.setdefault("step() returned", [0.0, 0.0])
    a[1] += (t1 - s_t) * 1e6 / N
for lab, (g, h) in acc.items():
    print(f"{lab:18s} gpu done at {g:8.1f} us   host enqueued at {h:8.1f} us")
