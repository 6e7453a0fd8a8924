"""Do two env batches overlap usefully on one GPU?  Two engines of B/2 environments over the same weight arena, full-depth static
steps as one graph each: (a) both on one stream, (b) on two streams (the GPU interleaves one batch's MFMA-bound vision tower with
the other's HBM-bound trunk), against (c) one engine of B environments.  usage: overlap_two_batches.py [B]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deer_vla_amd import synthetic as syn
from deer_vla_amd.config import deer_3b
from deer_vla_amd.engine import DeerEngine

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = deer_3b(max_layer=12)
sd = syn.make_synthetic_state(cfg, seed=0, std="0.02", bf16_round=True)


def inputs(n, dev):
    per = [syn.synthetic_step_inputs(cfg, 0, rank=e, text_seed=7 + e) for e in range(n)]
    return (torch.stack([p[0] for p in per]).to(dev, torch.bfloat16), torch.stack([p[1] for p in per]).to(dev, torch.bfloat16),
            torch.cat([p[2] for p in per]).to(dev))


def run(engs, streams, iters=30):
    for e, st in zip(engs, streams):
        with torch.cuda.stream(st):
            for _ in range(3):
                e.step(*e._inp, None, exit_id=11, sync=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        for e, st in zip(engs, streams):
            with torch.cuda.stream(st):
                e.step(*e._inp, None, exit_id=11, sync=False)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters


big = DeerEngine(cfg, sd, n_envs=min(B, 8), segmented=False)
big.configure_exit(cfg.exit_ids(), 12, 1)
big._inp = inputs(min(B, 8), big.dev)
h1 = DeerEngine(cfg, None, n_envs=B // 2, segmented=False, weights_from=big)
h2 = DeerEngine(cfg, None, n_envs=B // 2, segmented=False, weights_from=big)
for h in (h1, h2):
    h.configure_exit(cfg.exit_ids(), 12, 1)
    h._inp = inputs(B // 2, h.dev)
s0 = torch.cuda.current_stream()
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
t_big = run([big], [s1])
t_seq = run([h1, h2], [s1, s1])
t_par = run([h1, h2], [s1, s2])
print(f"B={B} full-depth static steps: one engine of {min(B, 8)}: {1e3 * t_big:.2f} ms ({min(B, 8) / t_big:.0f} env-steps/s) | two of {B // 2} on one stream: "
      f"{1e3 * t_seq:.2f} ms ({B / t_seq:.0f}) | two of {B // 2} on two streams: {1e3 * t_par:.2f} ms ({B / t_par:.0f})")
