#!/usr/bin/env python3
"""Full-size oracle traces for an ENV BATCH: 8 independent environments x N steps of DeeR-B max_layer=12 exit_ratio 0.8 at
3B size (the env-batch / multi-engine throughput figures are quoted on this workload).  Run offline in the
build container (CPU, ~10 min on 8 cores):

    python tests/golden/make_batch_goldens.py [n_steps] [n_envs]

Environment e is what ``bench.py::run_workload`` and ``eval_utils.py:523-527`` give a rank's e-th chain: frames seeded with
``rank=e``, instruction ``text_seed=7+e``, its own LSTM / controller state.  Thresholds: the committed ``b08`` solution of
tests/golden/episode_full.npz (REAL solver on the reference's calibration protocol).  Every environment is an INDEPENDENT
single-environment oracle run (the reference has no env batch: value_net.py:293 needs a 1-element value) - the env batch of the
engine must reproduce each of them.

Output: tests/golden/episode_batch8.npz - data only (per env / step: exit layer, action, deltas of the checks the oracle ran,
knife-edge margin); weights and frames are regenerated from seeds on the test side."""
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
sys.path.insert(0, HERE)

from deer_vla_amd import synthetic as syn  # noqa: E402
from deer_vla_amd.config import deer_3b  # noqa: E402
from oracle import deer_oracle as orc  # noqa: E402
import make_episode_goldens as meg  # noqa: E402

torch.set_grad_enabled(False)


def trunk_pass_env(cfg, sd, n_steps, e):
    model = orc.OracleDeer(sd, cfg)
    hid = []
    for s in range(n_steps):
        rgb, grip, ids, mask = syn.synthetic_step_inputs(cfg, s, rank=e, text_seed=7 + e)
        vis = model.encode_vision(rgb, grip)
        h, _ = orc.llm_forward(sd, cfg, ids, mask.bool(), vis, exit_id=cfg.n_layers - 1)
        hid.append(tuple(x.clone() for x in h))
    return hid


def main():
    n_steps = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    n_envs = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    out_path = os.path.join(HERE, "episode_batch8.npz")
    torch.set_num_threads(os.cpu_count() or 8)
    z = np.load(os.path.join(HERE, "episode_full.npz"))
    thr = [float(t) for t in z["b08_thr"]]
    max_layer = 12
    cfg = deer_3b(max_layer=max_layer)
    sd = syn.make_synthetic_state(cfg, meg.SEED, std=meg.STD, bf16_round=True)
    exit_ids = cfg.exit_ids()
    real = orc.OracleExitController(None, exit_ids, max_layer=max_layer).real_num_exit
    thr_row = np.array([thr[k] if k < real else np.inf for k in range(len(exit_ids))], np.float64)
    ex_all = np.zeros((n_envs, n_steps), np.int32)
    act_all = np.zeros((n_envs, n_steps, 8), np.float32)
    del_all = np.zeros((n_envs, n_steps, len(exit_ids)), np.float32)
    mar_all = np.zeros((n_envs, n_steps), np.float32)
    t0 = time.time()
    for e in range(n_envs):
        hid = trunk_pass_env(cfg, sd, n_steps, e)
        # the forward-equality self check of meg.episode() feeds rank-0 inputs: only valid for environment 0
        ex, act, deltas = meg.episode(cfg, sd, hid, thr, max_layer, check_forward=2 if e == 0 else 0)
        with np.errstate(invalid="ignore"):
            rel = np.abs(deltas - thr_row[None].astype(np.float32)) / np.abs(thr_row[None].astype(np.float32))
        rel[:, np.abs(thr_row) > 1e4] = np.inf
        margin = np.nanmin(np.where(np.isnan(rel), np.inf, rel), axis=1)
        ex_all[e], act_all[e], del_all[e], mar_all[e] = ex, act, deltas, margin
        hist = {int(k): int((ex == k).sum()) for k in sorted(set(ex.tolist()))}
        print(f"env {e}: exits {hist}  knife-edge steps {int((margin < 1e-2).sum())}  {time.time() - t0:.0f}s", flush=True)
    np.savez_compressed(out_path, n_steps=np.int32(n_steps), n_envs=np.int32(n_envs), seed=np.int32(meg.SEED),
                        cfg_json=np.frombuffer(json.dumps(cfg.to_dict()).encode(), dtype=np.uint8), max_layer=np.int32(max_layer),
                        thr=np.array(thr, np.float64), exit=ex_all, action=act_all, deltas=del_all, margin=mar_all)
    print("wrote", out_path, os.path.getsize(out_path), "bytes")


if __name__ == "__main__":
    main()
