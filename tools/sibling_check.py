"""Race hunt for env batches in flight: two sibling engines (one weight arena, two workspaces / streams / host threads) run 16 dynamic
control steps of 8 environments each CONCURRENTLY, three times, and every environment-step is compared bit for bit with the same engine
stepped alone (the full-size form of tests/test_batch_parity.py, with the list of mismatches).  usage: sibling_check.py"""
import json, os, sys, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from deer_vla_amd import synthetic as syn
from deer_vla_amd.config import DeerConfig
from deer_vla_amd.engine import DeerEngine
z = np.load("tests/golden/episode_batch8.npz")
cfg = DeerConfig(**json.loads(bytes(z["cfg_json"]).decode()))
sd = syn.make_synthetic_state(cfg, int(z["seed"]), std="0.02", bf16_round=True)
B, n = 8, 16
eng = DeerEngine(cfg, sd, n_envs=B)
sib = DeerEngine(cfg, None, n_envs=B, weights_from=eng)
thr = [float(t) for t in z["thr"]]
engines, offsets = [eng, sib], [0, 100]
for e in engines:
    e.configure_exit(cfg.exit_ids(), int(z["max_layer"]), 1); e.set_thresholds(thr)
def inputs(s, off):
    per = [syn.synthetic_step_inputs(cfg, s + off, rank=e, text_seed=7 + e) for e in range(B)]
    return (torch.stack([p[0] for p in per]).cuda().bfloat16(), torch.stack([p[1] for p in per]).cuda().bfloat16(), torch.cat([p[2] for p in per]).cuda())
fr = [[inputs(s, off) for s in range(n)] for off in offsets]
def episode(e, frames, out):
    e.reset(); ids = frames[0][2]
    for rgb, grip, _ in frames:
        r = e.step(rgb, grip, ids, None)
        out.append([(x["exit_layer"], x["pose"].clone(), x["deltas"].clone()) for x in r])
alone = [[], []]
for k in range(2): episode(engines[k], fr[k], alone[k])
torch.cuda.synchronize()
for trial in range(3):
    tog = [[], []]
    def run(k, st):
        with torch.cuda.stream(st): episode(engines[k], fr[k], tog[k])
    th = [threading.Thread(target=run, args=(k, torch.cuda.Stream())) for k in range(2)]
    [t.start() for t in th]; [t.join() for t in th]; torch.cuda.synchronize()
    bad = []
    for k in range(2):
        for s in range(n):
            for e in range(B):
                a, b = alone[k][s][e], tog[k][s][e]
                if a[0] != b[0] or not torch.equal(a[1], b[1]):
                    bad.append((k, s, e, a[0], b[0], float((a[1] - b[1]).abs().max()), [round(float(v), 5) for v in a[2][:3]], [round(float(v), 5) for v in b[2][:3]]))
    print("trial", trial, "mismatches", len(bad), bad[:6])
