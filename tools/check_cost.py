"""What the exit checks cost a one-environment step (value vs scripted): for every exit slot k the dynamic pipeline
with the verdict scripted to fire at slot k (the k checks before it run and decline) against the static-exit step of the same layer,
full 3B size, the product schedule, ms per step over `n` steps each (host reads the action after every step, like bench.py).
usage: python tools/check_cost.py [n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deer_vla_amd import synthetic as syn
from deer_vla_amd.config import deer_3b
from deer_vla_amd.engine import DeerEngine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
cfg = deer_3b(max_layer=12)
sd = syn.make_synthetic_state(cfg, seed=0, std="0.02", bf16_round=True)
eng = DeerEngine(cfg, sd)
exits = cfg.exit_ids()
eng.configure_exit(exits, 12, 1)
frames = [syn.synthetic_step_inputs(cfg, s) for s in range(8)]
ids = frames[0][2]
frames = [(f[0].cuda(), f[1].cuda()) for f in frames]
real = eng.real_num_exit


def timed(fn):
    for i in range(6):
        fn(i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        fn(i)
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


# CHECK_COST_PRIO=1: the caller's stream (vision chain 0, the trunk) is a HIGH-priority stream; the engine's side stream (chain 1, head
# evaluations) keeps the default priority
if os.environ.get("CHECK_COST_PRIO") == "1":
    _hp = torch.cuda.Stream(priority=-1)
    torch.cuda.set_stream(_hp)
    print("high-priority caller stream", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, "priority_range") else "")
print(f"{'exit layer':>10s} {'static ms':>10s} {'dynamic ms':>11s} {'difference us':>14s}   (checks that ran before the firing one)")
rows = []
for k, layer in enumerate(exits):
    thr = [-1.0] * k + [1e8] * (real - k)
    eng.reset()
    eng.set_thresholds(thr)
    dyn = timed(lambda i: eng.step(frames[i % 8][0], frames[i % 8][1], ids, None, use_graph=True))
    eng.reset()
    sta = timed(lambda i: eng.step(frames[i % 8][0], frames[i % 8][1], ids, None, exit_id=layer, use_graph=True))
    rows.append((layer, sta, dyn))
    print(f"{layer:10d} {sta:10.3f} {dyn:11.3f} {1e3 * (dyn - sta):14.1f}   ({k})", flush=True)
