import json, os, sys, time
sys.path.insert(0, "/root/repo")
import torch
from deer_vla_amd import synthetic as syn
from deer_vla_amd.config import deer_3b
from deer_vla_amd.engine import DeerEngine
cfg = deer_3b(max_layer=12)
sd = syn.make_synthetic_state(cfg, 0, std="0.02", bf16_round=True)
per = DeerEngine(cfg, sd)
per.set_persistent_layer(True)
frames = [syn.synthetic_step_inputs(cfg, s) for s in range(4)]
shown = [0]
def dump(e):
    tr = per.persistent_layer_trace(clear=True).numpy().astype("int64")
    for L in range(e + 1):
        clk = (tr[L, :, 0] & 0xffffffff) | (tr[L, :, 1] << 32)
        row = []
        for ep in range(1, 13):
            row.append((round((clk[ep] - clk[ep - 1]) / 100.0, 1), int(tr[L, ep, 2]), int(tr[L, ep, 3])))
        print("   layer", L, row, flush=True)
def static_leg(tag, mask_none, n=5):
    for e in (1, 5, 11):
        per.reset()
        ts = []
        for i in range(n):
            f = frames[i % 4]
            per.persistent_layer_trace(clear=True)
            t0 = time.perf_counter()
            per.step(f[0], f[1], f[2], None if mask_none else f[3], exit_id=e)
            torch.cuda.synchronize()
            ts.append(round((time.perf_counter() - t0) * 1e3, 1))
            if ts[-1] > 100 and shown[0] < 4 and i >= 1:
                shown[0] += 1
                print(tag, "exit", e, "step", i, "ms", ts[-1]); dump(e)
            elif i == n - 1 and e == 1 and tag.startswith("A"):
                print("normal step:"); dump(e)
        print(tag, "exit", e, "ms/step", ts, per.persistent_layer_error_detail(), flush=True)
static_leg("A static, mask given", False)
static_leg("B static, mask None", True)
per.configure_exit(cfg.exit_ids(), 12, 1)
per.set_thresholds([0.02] * 5 + [1e8])
per.set_persistent_layer(True)
per.reset()
for i in range(20):
    f = frames[i % 4]
    r = per.step(f[0], f[1], f[2], None)
torch.cuda.synchronize()
static_leg("D static after dynamic, mask None", True)
static_leg("E static after dynamic, mask given", False)
