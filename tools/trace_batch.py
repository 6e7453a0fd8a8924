"""GPU-side timeline of the pieces of dynamic control steps of an ENV BATCH (events after every piece), mid-episode with the
bench's calibrated thresholds.  usage: trace_batch.py [B] [n_steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deer_vla_amd import synthetic as syn
from deer_vla_amd.config import deer_3b
from deer_vla_amd.engine import DeerEngine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 12
cfg = deer_3b(max_layer=12)
sd = syn.make_synthetic_state(cfg, seed=0, std="0.02", bf16_round=True)
eng = DeerEngine(cfg, sd, n_envs=B)
eng.configure_exit(cfg.exit_ids(), 12, 1)
eng.set_thresholds([0.0218, 0.0227, 0.0209, 0.0247, 0.0201, 1e8])
dev = eng.dev
frames = []
for s in range(32):
    per_env = [syn.synthetic_step_inputs(cfg, s, rank=e, text_seed=7 + e) for e in range(B)]
    frames.append((torch.stack([p[0] for p in per_env]).to(dev, torch.bfloat16), torch.stack([p[1] for p in per_env]).to(dev, torch.bfloat16)))
ids = torch.cat([p[2] for p in per_env]).to(dev)
eng.reset()
for i in range(150):
    r = eng.step(frames[i % 32][0], frames[i % 32][1], ids, None)
tot = 0.0
for i in range(150, 150 + N):
    torch.cuda.synchronize()
    eng._trace = []
    t0 = time.perf_counter()
    r = eng.step(frames[i % 32][0], frames[i % 32][1], ids, None)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    tr, eng._trace = eng._trace, None
    s_ev, s_t = tr[0][1], tr[0][2]
    ex = [x["exit_layer"] for x in (r if B > 1 else [r])]
    line = " ".join(f"{lab.split()[0]}:{s_ev.elapsed_time(ev) * 1e3:.0f}/{(th - s_t) * 1e6:.0f}" for lab, ev, th in tr[1:])
    print(f"step {i}: exits {ex} host {1e6 * (t1 - t0):.0f} us | piece:gpu_done/host_enq  {line}")
    tot += t1 - t0
print(f"mean step {1e6 * tot / N:.0f} us -> {B * N / tot:.1f} env-steps/s")
