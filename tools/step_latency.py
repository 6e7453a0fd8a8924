"""Wall-clock distribution of DeerEngine.step() (dynamic, pipelined) for a forced exit index, next to the STATIC step that exits at the
same layer (no exit checks; two-chain vision + one trunk graph) - the difference is what the exit checks of the dynamic pipeline cost."""
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
rgb, grip, ids, mask = syn.synthetic_step_inputs(cfg, 0)
rgb, grip, ids = rgb.cuda().bfloat16(), grip.cuda().bfloat16(), ids.cuda()
for exit_at in range(6):
    thr = [-1.0] * 6
    thr[exit_at] = 1e5
    thr[5] = 1e5
    eng.set_thresholds(thr)
    for _ in range(5):
        r = eng.step(rgb, grip, ids, None)
    ts = []
    for _ in range(40):
        t0 = time.perf_counter()
        r = eng.step(rgb, grip, ids, None)
        ts.append(1e3 * (time.perf_counter() - t0))
    ts.sort()
    layer = r["exit_layer"]
    for _ in range(3):
        eng.step(rgb, grip, ids, None, exit_id=layer)
    st = []
    for _ in range(40):
        t0 = time.perf_counter()
        eng.step(rgb, grip, ids, None, exit_id=layer)
        st.append(1e3 * (time.perf_counter() - t0))
    st.sort()
    print(f"exit index {exit_at} (layer {layer:2d}): dynamic min {ts[0]:.3f}  median {ts[len(ts)//2]:.3f}  p90 {ts[int(.9*len(ts))]:.3f}  max {ts[-1]:.3f} ms"
          f"   static median {st[len(st)//2]:.3f} ms   dynamic - static {1e3 * (ts[len(ts)//2] - st[len(st)//2]):.0f} us")
