"""Host-side share of a dynamic control step: the native step driver (csrc/step_driver.hip: deer_step_plan_run) blocks from the first
graph submission to the verdict; whatever step() spends OUTSIDE it (draining the side streams, the D2D input staging, step bookkeeping,
reading the result) is time the GPU waits for the host (the staging copies excepted).  Full 3B size, every step exits at the first check.
usage: python tools/host_gap.py"""
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
eng.set_thresholds([1e5, 1e5, 1e5, 1e5, 1e5, 1e8])      # every step exits at the first check
rgb, grip, ids, mask = syn.synthetic_step_inputs(cfg, 0)
rgb, grip, ids = rgb.cuda().bfloat16(), grip.cuda().bfloat16(), ids.cuda()
for _ in range(5):
    eng.step(rgb, grip, ids, None)
acc = {"native": 0.0, "load": 0.0, "read": 0.0}
orig_run = eng.lib.deer_step_plan_run
def run(*a):
    t = time.perf_counter()
    r = orig_run(*a)
    acc["native"] += time.perf_counter() - t
    return r
class LibProxy:
    def __init__(self, lib): self._lib = lib
    def __getattr__(self, n): return run if n == "deer_step_plan_run" else getattr(self._lib, n)
eng.lib = LibProxy(eng.lib)
orig_load, orig_read = eng.load_inputs, eng.read_result
def load(*a, **k):
    t = time.perf_counter(); r = orig_load(*a, **k); acc["load"] += time.perf_counter() - t; return r
def read(*a, **k):
    t = time.perf_counter(); r = orig_read(*a, **k); acc["read"] += time.perf_counter() - t; return r
eng.load_inputs, eng.read_result = load, read
N = 300
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(N):
    eng.step(rgb, grip, ids, None)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / N * 1e6
nat, ld, rd = (acc[k] / N * 1e6 for k in ("native", "load", "read"))
print(f"wall per step {wall:.1f} us; inside the native driver (first submission .. verdict) {nat:.1f} us; outside it {wall - nat:.1f} us "
      f"(input staging calls {ld:.1f} us, result read {rd:.1f} us, the rest {wall - nat - ld - rd:.1f} us)")
