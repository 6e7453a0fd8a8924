"""OpenFlamingo-9B / MPT-7B (BASELINE configs[4]) at FULL size on the engine: a parity step against the fp32 oracle and the step
latency by exit.  usage: run_9b.py [--no-oracle]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deer_vla_amd import synthetic as syn
from deer_vla_amd.config import deer_9b
from deer_vla_amd.engine import DeerEngine

cfg = deer_9b(max_layer=12)
t0 = time.time()
sd = syn.make_synthetic_state(cfg, seed=0, std="0.02", bf16_round=True)
print(f"9B state: {sum(v.numel() for v in sd.values()) / 1e9:.2f} G params in {time.time() - t0:.0f}s; layers {cfg.n_layers}, exits {cfg.exit_ids()}")
eng = DeerEngine(cfg, sd)
print(f"engine weights {eng.weight_bytes() / 1e9:.2f} GB")
eng.configure_exit(cfg.exit_ids(), 12, 1)
real = eng.real_num_exit
rgb, grip, ids, mask = syn.synthetic_step_inputs(cfg, 0)
if "--no-oracle" not in sys.argv:
    from oracle import deer_oracle as orc
    torch.set_num_threads(32)
    model = orc.OracleDeer(sd, cfg)
    model.set_all_exit_window_size(1)
    for eid in (3, cfg.n_layers - 1):
        t0 = time.time()
        o = model.forward(rgb, ids, mask, grip, exit_id=eid)
        model.clear_all_exit_memory()
        eng.reset()
        r = eng.step(rgb, grip, ids, mask, exit_id=eid)
        err = float((r["pose"] - o["logits"][0].reshape(-1)).abs().max())
        print(f"static exit {eid}: max|pose - oracle| = {err:.2e}  gripper {abs(r['gripper'] - float(o['logits'][1])):.2e}  (oracle {time.time() - t0:.1f}s)")
        assert err < 1e-2
rgb, grip, ids = rgb.cuda().bfloat16(), grip.cuda().bfloat16(), ids.cuda()
for k in range(real):
    thr = [-1.0] * real
    thr[k] = 1e5
    thr[-1] = 1e5
    eng.set_thresholds(thr)
    for _ in range(4):
        r = eng.step(rgb, grip, ids, None)
    ts = []
    for _ in range(30):
        t0 = time.perf_counter()
        r = eng.step(rgb, grip, ids, None)
        ts.append(1e3 * (time.perf_counter() - t0))
    ts.sort()
    print(f"dynamic exit at layer {r['exit_layer']:2d}: median {ts[15]:.3f} ms  ({1e3 / ts[15]:.0f} steps/s)")
