"""N1 measurement: the trunk layer of a one-environment step as ONE persistent launch (csrc/persistent_layer.hip) against
the twelve-launch layer, full size.  Per static exit id: ms per step (host-read action, two-chain vision + one trunk graph), the slope in
us per layer, whether the actions are bit-identical, and the persistent kernel's barrier error word after every leg.
usage: persistent_layer_check.py [n_steps]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deer_vla_amd import synthetic as syn
from deer_vla_amd.config import deer_3b
from deer_vla_amd.engine import DeerEngine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
cfg = deer_3b(max_layer=12)
sd = syn.make_synthetic_state(cfg, 0, std="0.02", bf16_round=True)
ref = DeerEngine(cfg, sd)
per = DeerEngine(cfg, None, weights_from=ref)
per.set_persistent_layer(True)
frames = [syn.synthetic_step_inputs(cfg, s) for s in range(4)]
out = {"steps_per_leg": n, "by_exit": {}}
for e in cfg.exit_ids():
    row = {}
    acts = []
    for name, eng in (("twelve_launch", ref), ("persistent", per)):
        eng.reset()
        for _ in range(3):
            eng.step(*frames[0], exit_id=e)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(n):
            r = eng.step(*frames[i % 4], exit_id=e)
        torch.cuda.synchronize()
        row[name + "_ms"] = round((time.perf_counter() - t0) / n * 1e3, 4)
        acts.append(r["pose"].clone())
    row["bit_identical"] = bool(torch.equal(acts[0], acts[1]))
    row["barrier_error_word"] = per.persistent_layer_error()
    out["by_exit"][str(e)] = row
    print(e, row, flush=True)
ex = cfg.exit_ids()
for name in ("twelve_launch", "persistent"):
    a, b = out["by_exit"][str(ex[0])][name + "_ms"], out["by_exit"][str(ex[-1])][name + "_ms"]
    out[name + "_us_per_layer"] = round((b - a) / (ex[-1] - ex[0]) * 1e3, 1)
print(json.dumps(out))
