#!/usr/bin/env python3
"""Round 6: the vision tower's 16-bit format - bf16 (a `--precision bf16` reference run) against IEEE fp16 (the
reference's evaluation arithmetic: fp32 weights under fp16 autocast, eval_utils.py:333) - on the weights of tests/test_hard_inputs.py
(planted outlier channels, LayerNorm gains over two decades) and on the plain seeded ones: worst action error against the f32 CPU oracle,
media-token error, exit flips inside / outside the knife-edge band.  Usage (GPU box): python tools/tower_format.py [--full-steps 24]"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from deer_vla_amd import synthetic as syn  # noqa: E402
from deer_vla_amd.config import deer_tiny, deer_3b  # noqa: E402
from deer_vla_amd.engine import DeerEngine  # noqa: E402
from oracle import deer_oracle as orc  # noqa: E402

orc.OracleDeer.TRUNK_MEMO = {}
BAND = 1e-2


def gap_threshold(vals, lo=0.2, hi=0.8):
    import numpy as np
    v = np.sort(np.asarray(vals, dtype=np.float64))
    a = int(len(v) * lo)
    b = min(max(int(len(v) * hi), a + 2), len(v))
    gaps = v[a + 1:b] - v[a:b - 1]
    i = int(np.argmax(gaps)) + a
    return float(0.5 * (v[i] + v[i + 1]))


class RecVN(orc.OracleValueNet):
    def __call__(self, feats, i=None, mode="infer", rand_layer_feat=None):
        v = super().__call__(feats, i, mode, rand_layer_feat)
        self.rec = getattr(self, "rec", [])
        self.rec.append((i, float(v)))
        return v


def oracle_episode(cfg, sd, inputs, thr):
    model = orc.OracleDeer(sd, cfg)
    model.set_all_exit_window_size(1)
    vn = RecVN(cfg.exit_ids(), model.extra_exit, cfg.exit_interval, 1, "L2")
    vn.rec = []
    ctl = orc.OracleExitController(vn, cfg.exit_ids(), steps_per_stage=1, max_layer=12)
    ctl._set_threshold_value(thr)
    tb = dict(zip(cfg.exit_ids(), thr))
    out = []
    for s, (rgb, grip, ids, mask) in enumerate(inputs):
        ctl.set_timestep(s)
        n0 = len(vn.rec)
        o = model.forward(rgb, ids, mask, grip, dynamic_early_exit=True, exit_controller=ctl)
        m = [abs(v - tb[i]) / tb[i] for (i, v) in vn.rec[n0:] if tb[i] < 1e4]
        out.append((o["exit_layer"], o["logits"][0].reshape(-1), float(o["logits"][1]), min(m) if m else float("inf"), o["vis_x"]))
    return out, vn.rec


def episode(cfg, sd, inputs, towers=("bf16", "fp16")):
    _, rec = oracle_episode(cfg, sd, inputs, [-1.0] * len(cfg.exit_ids()))
    thr = [gap_threshold([v for (i, v) in rec if i == e]) for e in cfg.exit_ids()]
    thr[-1] = 1e5
    ref, _ = oracle_episode(cfg, sd, inputs, thr)
    res = {}
    for tower in towers:
        eng = DeerEngine(cfg, sd, max_text_len=32, precision=tower)
        eng.configure_exit(cfg.exit_ids(), 12, 1)
        eng.set_thresholds(thr)
        eng.reset()
        worst, media, flips_in, flips_out, compared = 0.0, 0.0, 0, 0, 0
        for s, (rgb, grip, ids, mask) in enumerate(inputs):
            r = eng.step(rgb, grip, ids, mask, use_graph=(s >= 2))
            ex, pose, g, margin, vis = ref[s]
            torch.cuda.synchronize()
            v = eng.vis_x_f32.cpu()
            media = max(media, float((v - vis.reshape(v.shape)).norm() / vis.norm()))
            if r["exit_layer"] != ex:
                if margin <= BAND:
                    flips_in += 1
                else:
                    flips_out += 1
                break
            worst = max(worst, float((r["pose"] - pose).abs().max()), abs(r["gripper"] - g))
            compared += 1
        res[tower] = dict(worst_action_err=worst, media_rel_err=media, steps_compared=compared, knife_edge_flips=flips_in, flips_outside_band=flips_out)
        del eng
    return res


def static_steps(cfg, sd, n_steps, towers=("bf16", "fp16")):
    model = orc.OracleDeer(sd, cfg)
    model.set_all_exit_window_size(1)
    exits = [11, 5, 1]
    refs = []
    for s in range(n_steps):
        rgb, grip, ids, mask = syn.synthetic_step_inputs(cfg, s, text_len=20 if s % 3 == 1 else 14)
        o = model.forward(rgb, ids, mask, grip, exit_id=exits[s % 3])
        refs.append((rgb, grip, ids, mask, exits[s % 3], o["logits"][0].reshape(-1), float(o["logits"][1]), o["vis_x"]))
    res = {}
    for tower in towers:
        eng = DeerEngine(cfg, sd, precision=tower)
        eng.reset()
        worst, media = 0.0, 0.0
        for rgb, grip, ids, mask, eid, pose, g, vis in refs:
            r = eng.step(rgb, grip, ids, mask, exit_id=eid, use_graph=False)
            torch.cuda.synchronize()
            assert bool(torch.isfinite(eng.vx).all()) and bool(torch.isfinite(eng.vis_x_f32).all())
            v = eng.vis_x_f32.cpu()
            media = max(media, float((v - vis.reshape(v.shape)).norm() / vis.norm()))
            worst = max(worst, float((r["pose"] - pose).abs().max()), abs(r["gripper"] - g))
        res[tower] = dict(worst_action_err=worst, media_rel_err=media, steps=n_steps)
        del eng
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full-steps", type=int, default=6)
    ap.add_argument("--skip-full", action="store_true")
    a = ap.parse_args()
    out = {}
    cfg = deer_tiny()
    for T in (9, 14, 32):
        sd = syn.harden_state(cfg, syn.make_synthetic_state(cfg, 3), seed=T)
        out[f"tiny_hard_T{T}"] = episode(cfg, sd, [syn.synthetic_step_inputs(cfg, s, text_len=T) for s in range(12)])
        print(f"tiny_hard_T{T}", json.dumps(out[f"tiny_hard_T{T}"]), flush=True)
    sd = syn.make_synthetic_state(cfg, 3, bf16_round=True)
    out["tiny_easy"] = episode(cfg, sd, [syn.synthetic_step_inputs(cfg, s, text_len=11) for s in range(24)])
    print("tiny_easy", json.dumps(out["tiny_easy"]), flush=True)
    if not a.skip_full:
        cfg = deer_3b(max_layer=12)
        base = syn.make_synthetic_state(cfg, 0, std="0.02", bf16_round=True)
        out["full_3b_easy"] = static_steps(cfg, base, 3)
        print("full_3b_easy", json.dumps(out["full_3b_easy"]), flush=True)
        sd = syn.harden_state(cfg, base, seed=0)
        out["full_3b_hard"] = static_steps(cfg, sd, a.full_steps)
        print("full_3b_hard", json.dumps(out["full_3b_hard"]), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "tower_format.json"), "w") as f:
        json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
