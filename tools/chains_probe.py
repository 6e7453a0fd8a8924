"""How much does the vision tower gain from MORE concurrent chains at equal work?  Static exit-1 step of an env batch (2 or 4
environments = 4 or 8 camera frames) with DEER_CHAINS = 1, 2, 4, (8): the chains differ only in how many launches are in flight together
(frames per launch x concurrent launches = constant).  Decides whether splitting a one-environment frame into row-range sub-chains
(row-independent launches of a ViT block on 128 + 129 rows, joined at the attention) can pay."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deer_vla_amd import synthetic as syn
from deer_vla_amd.config import deer_3b
from deer_vla_amd.engine import DeerEngine

cfg = deer_3b(max_layer=12)
sd = syn.make_synthetic_state(cfg, seed=0, std="0.02", bf16_round=True)
base = None
for B in (int(a) for a in (sys.argv[1:] or ["2", "4", "1"])):
    for nch in (1, 2, 4, 8):
        if nch > 2 * B:
            continue
        os.environ["DEER_CHAINS"] = str(nch)
        eng = DeerEngine(cfg, sd if base is None else None, n_envs=B, weights_from=base)
        base = base or eng
        eng.configure_exit(cfg.exit_ids(), 12, 1)
        per_env = [syn.synthetic_step_inputs(cfg, 0, rank=e, text_seed=7 + e) for e in range(B)]
        rgb = torch.stack([p[0] for p in per_env]).to(eng.dev, eng.img_dtype)
        grip = torch.stack([p[1] for p in per_env]).to(eng.dev, eng.img_dtype)
        ids = torch.cat([p[2] for p in per_env]).to(eng.dev)
        for _ in range(5):
            eng.step(rgb, grip, ids, None, exit_id=1)
        ts = []
        for _ in range(60):
            t0 = time.perf_counter()
            eng.step(rgb, grip, ids, None, exit_id=1)
            ts.append(1e3 * (time.perf_counter() - t0))
        ts.sort()
        print(f"envs {B} ({2 * B} frames)  chains {eng.n_chains}: static exit-1 step  min {ts[0]:.3f}  median {ts[len(ts) // 2]:.3f} ms", flush=True)
        del eng
