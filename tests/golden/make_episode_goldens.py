#!/usr/bin/env python3
"""Full-size 360-step oracle episodes with thresholds from the REAL solver (SURVEY §8d parity gate
"exit_layer sequence identical over the 360-step episode").  Run offline in the build container (CPU, ~15 min on 8 cores):

    python tests/golden/make_episode_goldens.py [n_steps]

Output: tests/golden/episode_full.npz - data only (thresholds, per-step exit layer / action / deltas of the fp32 oracle);
weights and frames are regenerated from seeds by ``deer_vla_amd.synthetic`` on the test side.

Cases (BASELINE.json configs):
  b08   MPT-1B DeeR-B max_layer=12, exit_ratio 0.8   (configs[2])
  b10   MPT-1B DeeR-B max_layer=12, exit_ratio 1.0   (configs[3])
  s08   MPT-1B DeeR-S max_layer=4,  exit_ratio 0.8   (configs[1])

How the traces are made (all arithmetic is the oracle's, oracle/deer_oracle.py):
  1. trunk pass: for every step s the vision tower + ALL LLM layers (``llm_forward(exit_id=last)``); the hidden states do
     not depend on the exit policy or the LSTM history, so they are computed once and shared by every case (DeeR-S builds
     layers 0..4 of the same seeded weights: names and seeds are per tensor, deer_vla_amd/synthetic.py).
  2. calibration values: the delta of every exit at every step (``OracleValueNet`` / ``get_delta``) while the LSTM history
     follows a seeded RANDOM exit layer per step - the reference's calibration protocol (random-exit-layer history,
     flamingo_mpt.py:485-497 / value_net.py:134-160) in step mode.  ``solve_thresholds`` (value_net.py:203-260 restated)
     turns the (n_exit, n_steps) matrix into thresholds for the case's exit_ratio.
  3. episode: ``OracleExitController`` is driven through the cached hidden-state tuples exactly as the layer loop does
     (mosaic_gpt_3b.py:438-443), the committing head call follows (flamingo_mpt.py:459); equality with
     ``OracleDeer.forward(dynamic_early_exit=True)`` is asserted on the first steps.
"""
from __future__ import annotations

import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from deer_vla_amd import synthetic as syn  # noqa: E402
from deer_vla_amd.config import deer_3b  # noqa: E402
from oracle import deer_oracle as orc  # noqa: E402

torch.set_grad_enabled(False)
SEED, STD = 0, "0.02"
CALIB_SEED = 4242


class RecVN(orc.OracleValueNet):
    """records (exit layer, delta) of every evaluation"""

    def __call__(self, feats, i=None, mode="infer", rand_layer_feat=None):
        v = super().__call__(feats, i, mode, rand_layer_feat)
        self.rec.append((i, float(v)))
        return v


def trunk_pass(cfg, sd, n_steps):
    model = orc.OracleDeer(sd, cfg)
    hid = []
    t0 = time.time()
    for s in range(n_steps):
        rgb, grip, ids, mask = syn.synthetic_step_inputs(cfg, s)
        vis = model.encode_vision(rgb, grip)
        h, _ = orc.llm_forward(sd, cfg, ids, mask.bool(), vis, exit_id=cfg.n_layers - 1)
        hid.append(tuple(x.clone() for x in h))
        if s % 20 == 0:
            print(f"  trunk step {s}/{n_steps}  {time.time() - t0:.0f}s", flush=True)
    return hid


def calibration_values(cfg, sd, hid, exit_ids, real):
    """(real, n_steps) deltas, LSTM history following a random exit layer (seeded)."""
    head = orc.OracleHead(sd, cfg, "extra_exit.")
    head.window_size = 1
    g = torch.Generator().manual_seed(CALIB_SEED)
    vals = np.zeros((real, len(hid)), np.float64)
    for s, h in enumerate(hid):
        prev = head(h[exit_ids[0] - 1], update_hidden_state=False)          # value_net.py:122-125 (i - interval < 0)
        for k, e in enumerate(exit_ids[:real]):
            a = head(h[e], update_hidden_state=False)
            vals[k, s] = float(orc.get_delta(a[0], prev[0], "L2"))
            prev = a
        r = exit_ids[int(torch.randint(0, real, (1,), generator=g))]
        head(h[r], update_hidden_state=True)
    return vals


def episode(cfg, sd, hid, thresholds, max_layer, check_forward=3):
    model = orc.OracleDeer(sd, cfg)
    model.set_all_exit_window_size(1)
    exit_ids = cfg.exit_ids()
    vn = RecVN(exit_ids, model.extra_exit, cfg.exit_interval, 1, "L2")
    vn.rec = []
    ctl = orc.OracleExitController(vn, exit_ids, max_layer=max_layer)
    ctl._set_threshold_value(list(thresholds))
    n = len(hid)
    ex = np.zeros(n, np.int32)
    act = np.zeros((n, 8), np.float32)
    deltas = np.full((n, len(exit_ids)), np.nan, np.float32)
    # the same model/controller classes through the ordinary forward, for the equality check on the first steps
    model2 = orc.OracleDeer(sd, cfg)
    model2.set_all_exit_window_size(1)
    vn2 = orc.OracleValueNet(exit_ids, model2.extra_exit, cfg.exit_interval, 1, "L2")
    ctl2 = orc.OracleExitController(vn2, exit_ids, max_layer=max_layer)
    ctl2._set_threshold_value(list(thresholds))
    for s, h in enumerate(hid):
        ctl.set_timestep(s)
        vn.rec = []
        e = -1
        for b in range(cfg.n_layers):                                       # mosaic_gpt_3b.py:397-443
            if ctl(h[:b + 1], b):
                e = b
                break
        assert e >= 0
        pose, grip = model.extra_exit(h[e], with_gripper_logits=False)      # committing call (flamingo_mpt.py:459)
        ex[s] = e
        act[s, :6] = pose.reshape(-1).numpy()
        act[s, 6] = float(grip)
        for (i, v) in vn.rec:
            deltas[s, exit_ids.index(i)] = v
        if s < check_forward:
            rgb, grp, ids, mask = syn.synthetic_step_inputs(cfg, s)
            ctl2.set_timestep(s)
            o = model2.forward(rgb, ids, mask, grp, dynamic_early_exit=True, exit_controller=ctl2)
            assert o["exit_layer"] == e, (s, o["exit_layer"], e)
            assert float((o["logits"][0].reshape(-1) - pose.reshape(-1)).abs().max()) < 1e-6
    return ex, act, deltas


def main():
    n_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 360
    out_path = sys.argv[2] if len(sys.argv) > 2 else os.path.join(HERE, "episode_full.npz")
    torch.set_num_threads(os.cpu_count() or 8)
    cfg_b = deer_3b(max_layer=12)
    sd_b = syn.make_synthetic_state(cfg_b, SEED, std=STD, bf16_round=True)
    print("trunk pass (oracle, all layers)...", flush=True)
    hid = trunk_pass(cfg_b, sd_b, n_steps)
    out = dict(n_steps=np.int32(n_steps), seed=np.int32(SEED), calib_seed=np.int32(CALIB_SEED))
    cases = [("b08", 12, 0.8), ("b10", 12, 1.0), ("s08", 4, 0.8)]
    for tag, max_layer, ratio in cases:
        cfg = deer_3b(max_layer=max_layer)
        sd = sd_b if max_layer == 12 else syn.make_synthetic_state(cfg, SEED, std=STD, bf16_round=True)
        h_case = [h[:cfg.n_layers] for h in hid]
        exit_ids = cfg.exit_ids()
        ctl0 = orc.OracleExitController(None, exit_ids, max_layer=max_layer)
        real = ctl0.real_num_exit
        vals = calibration_values(cfg, sd, h_case, exit_ids, real)
        T = orc.solve_thresholds(torch.from_numpy(vals).float(), real, ratio)
        thr = [float(t) for t in T]
        ex, act, deltas = episode(cfg, sd, h_case, thr, max_layer)
        thr_row = np.array([thr[exit_ids.index(e)] if exit_ids.index(e) < real else np.inf for e in exit_ids], np.float64)
        with np.errstate(invalid="ignore"):
            rel = np.abs(deltas - thr_row[None].astype(np.float32)) / np.abs(thr_row[None].astype(np.float32))
        rel[:, np.abs(thr_row) > 1e4] = np.inf                                # forced / disabled exits are never knife-edge
        margin = np.nanmin(np.where(np.isnan(rel), np.inf, rel), axis=1)
        hist = {int(e): int((ex == e).sum()) for e in sorted(set(ex.tolist()))}
        print(f"{tag}: thresholds {np.round(thr[:-1], 5).tolist()}  exits {hist}  avg exit layer+1 {float((ex + 1).mean()):.2f}  "
              f"steps with margin < 1e-2: {int((margin < 1e-2).sum())}", flush=True)
        out[tag + "_cfg_json"] = np.frombuffer(json.dumps(cfg.to_dict()).encode(), dtype=np.uint8)
        out[tag + "_max_layer"] = np.int32(max_layer)
        out[tag + "_ratio"] = np.float32(ratio)
        out[tag + "_thr"] = np.array(thr, np.float64)
        out[tag + "_calib_values"] = vals.astype(np.float32)
        out[tag + "_exit"] = ex
        out[tag + "_action"] = act
        out[tag + "_deltas"] = deltas
        out[tag + "_margin"] = margin.astype(np.float32)
    np.savez_compressed(out_path, **out)
    print("wrote", out_path, os.path.getsize(out_path), "bytes")


if __name__ == "__main__":
    main()
