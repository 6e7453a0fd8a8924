"""Full-size parity of the paths the env-batch throughput figures are quoted on:

(a) ``n_envs=8`` DYNAMIC episode at 3B size (DeeR-B max_layer=12, exit_ratio 0.8 thresholds) - every environment against its OWN
    committed fp32-oracle trace (tests/golden/episode_batch8.npz, made by tests/golden/make_batch_goldens.py: 8 independent
    single-environment oracle runs, inputs ``rank=e, text_seed=7+e`` as bench.py::run_workload / eval_utils.py:523-527 feed them).
    Gate as tests/test_episode_parity.py: actions within 1e-2 at every env-step; exit layer identical wherever the oracle's decision
    is not knife-edge (margin > 1e-2); a flipped knife-edge env-step is re-aligned on the oracle's exit (that environment's LSTM state
    is taken from a static replay of the step at the oracle's exit layer), so every later env-step stays a like-for-like comparison.
(b) two sibling engines (one weight arena, two workspaces / streams / host threads - the ``batched_groups`` bench leg and
    ``evaluate_policy_batched(groups=...)``) stepped CONCURRENTLY give bit-identical results to each engine stepped alone."""
import json
import os
import threading

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from deer_vla_amd import synthetic as syn  # noqa: E402,F401
from golden_util import full_size_state  # noqa: E402
from deer_vla_amd.config import DeerConfig  # noqa: E402
from deer_vla_amd.engine import DeerEngine  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden", "episode_batch8.npz")
ACTION_TOL = 1e-2
BAND = 1e-2


@pytest.fixture(scope="module")
def setup():
    assert os.path.exists(GOLD), "tests/golden/episode_batch8.npz missing (run tests/golden/make_batch_goldens.py)"
    z = np.load(GOLD)
    cfg = DeerConfig(**json.loads(bytes(z["cfg_json"]).decode()))
    sd = full_size_state(cfg, int(z["seed"]), std="0.02", bf16_round=True)
    B = int(z["n_envs"])
    eng = DeerEngine(cfg, sd, n_envs=B)
    return z, cfg, eng, B


def batch_inputs(cfg, B, s, dev, step_offset=0):
    # environments beyond the eight committed traces replay them (environment e sees the inputs of trace e % 8)
    per_env = [syn.synthetic_step_inputs(cfg, s + step_offset, rank=e % 8, text_seed=7 + e % 8) for e in range(B)]
    rgb = torch.stack([p[0] for p in per_env]).to(dev, torch.bfloat16)
    grip = torch.stack([p[1] for p in per_env]).to(dev, torch.bfloat16)
    ids = torch.cat([p[2] for p in per_env]).to(dev)
    return rgb, grip, ids


@pytest.mark.parametrize("n_envs", [8, 16])
def test_eight_environment_dynamic_episode_matches_each_environments_oracle_trace(setup, n_envs):
    """n_envs = 16 (round 5): one engine carries 16 environments - 32 camera frames per vision chain (two rounds of
    frame tiles per GEMM launch), 224 trunk rows (two row blocks), head evaluations in two halves of eight.  Environments e and e + 8
    replay trace e of the committed eight; 24 steps."""
    z, cfg, eng, B = setup
    n = int(z["n_steps"])
    if n_envs != B:
        eng = DeerEngine(cfg, None, n_envs=n_envs, weights_from=eng)
        B, n = n_envs, 24
    thr = [float(t) for t in z["thr"]]
    ref_exit, ref_act, margin = (np.concatenate([z[k]] * (B // 8)) for k in ("exit", "action", "margin"))
    eng.configure_exit(cfg.exit_ids(), int(z["max_layer"]), 1)
    eng.set_thresholds(thr)
    eng.reset()
    flips, outside, worst, compared = [], [], 0.0, 0
    hist = {}
    ids_keep = None
    for s in range(n):
        rgb, grip, ids = batch_inputs(cfg, B, s, eng.dev)
        ids_keep = ids if ids_keep is None else ids_keep          # one instruction tensor for the whole episode (upload once)
        h0, c0 = eng.h_state.clone(), eng.c_state.clone()
        out = eng.step(rgb, grip, ids_keep, None)
        wrong = [e for e in range(B) if out[e]["exit_layer"] != int(ref_exit[e, s])]
        for e in wrong:
            rec = dict(env=e, step=s, engine=out[e]["exit_layer"], oracle=int(ref_exit[e, s]), margin=float(margin[e, s]))
            (flips if margin[e, s] <= BAND else outside).append(rec)
        if wrong:                                                  # re-align the flipped environments on the oracle's trajectory
            torch.cuda.synchronize()
            h_dyn, c_dyn = eng.h_state.clone(), eng.c_state.clone()
            for e in wrong:
                eng.h_state.copy_(h0)
                eng.c_state.copy_(c0)
                rep = eng.step(rgb, grip, ids_keep, None, exit_id=int(ref_exit[e, s]))
                torch.cuda.synchronize()
                h_dyn[:, e], c_dyn[:, e] = eng.h_state[:, e], eng.c_state[:, e]
                out[e] = rep[e]
            eng.h_state.copy_(h_dyn)
            eng.c_state.copy_(c_dyn)
        for e in range(B):
            err = max(float((out[e]["pose"] - torch.from_numpy(ref_act[e, s, :6])).abs().max()), abs(out[e]["gripper"] - float(ref_act[e, s, 6])))
            worst = max(worst, err)
            compared += 1
            hist[out[e]["exit_layer"]] = hist.get(out[e]["exit_layer"], 0) + 1
    knife = int((margin[:, :n] <= BAND).sum())
    rep = dict(env_steps=compared, exit_hist={int(k): v for k, v in sorted(hist.items())}, knife_edge_env_steps=knife,
               knife_edge_flips=flips, mismatches_outside_band=outside, worst_action_err=worst)
    out_dir = os.path.join(os.path.dirname(HERE), "gpurun_out")
    if os.path.isdir(out_dir):
        with open(os.path.join(out_dir, "batch_parity_report.json" if B == 8 else f"batch{B}_parity_report.json"), "w") as fh:
            json.dump(rep, fh, indent=1)
    print(f"\n[env batch of {B}, {n} steps, full size] {compared} env-steps, exits {rep['exit_hist']}, knife-edge env-steps {knife} "
          f"(engine decided differently on {len(flips)}), outside the band {len(outside)}, worst |action - oracle| {worst:.2e}")
    assert compared == B * n
    assert not outside, outside
    assert worst < ACTION_TOL, worst
    assert len(hist) > 2, hist                                      # the environments really exit at different layers


def test_two_sibling_engines_stepped_concurrently_equal_each_engine_alone(setup):
    z, cfg, eng, B = setup
    thr = [float(t) for t in z["thr"]]
    sib = DeerEngine(cfg, None, n_envs=B, weights_from=eng)
    assert sib.arena.data_ptr() == eng.arena.data_ptr()
    engines, offsets, n = [eng, sib], [0, 100], 16
    for e in engines:
        e.configure_exit(cfg.exit_ids(), int(z["max_layer"]), 1)
        e.set_thresholds(thr)
    inputs = [[batch_inputs(cfg, B, s, eng.dev, off) for s in range(n)] for off in offsets]

    def episode(e, frames, out):
        e.reset()
        ids = frames[0][2]
        for rgb, grip, _ in frames:
            r = e.step(rgb, grip, ids, None)
            out.append([(x["exit_layer"], x["pose"].clone(), x["gripper"]) for x in r])

    alone = [[], []]
    for k in range(2):
        episode(engines[k], inputs[k], alone[k])
    torch.cuda.synchronize()
    together, errors = [[], []], []

    def run(k, stream):
        try:
            with torch.cuda.stream(stream):
                episode(engines[k], inputs[k], together[k])
        except Exception as ex:                                     # surfaced below
            errors.append(ex)

    threads = [threading.Thread(target=run, args=(k, torch.cuda.Stream())) for k in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    assert not errors, errors
    n_exits = set()
    for k in range(2):
        assert len(together[k]) == n
        for s in range(n):
            for e in range(B):
                a, b = alone[k][s][e], together[k][s][e]
                assert a[0] == b[0], (k, s, e, a[0], b[0])
                assert torch.equal(a[1], b[1]) and a[2] == b[2], (k, s, e)
                n_exits.add(a[0])
    assert len(n_exits) > 1


def test_full_size_env_batch_padded_to_32_tokens_matches_the_unpadded_batch(setup):
    """VERDICT r3 item 3b at 3B size: the same 8 environments with their 14-token instructions right-padded to the reference's
    max_length = 32 (256 trunk rows: two row blocks of the hi/lo-plane GEMM, two MFMA row tiles per environment in both attention
    kernels, padded keys masked, pad rows out of the head's token pool) against the unpadded batch (112 rows - the path held against
    the oracle above).  Padding changes nothing but a constant ALiBi shift per row: actions agree to 1e-4, exit layers exactly."""
    z, cfg, eng, B = setup
    eng.configure_exit(cfg.exit_ids(), int(z["max_layer"]), 1)
    eng.set_thresholds([float(t) for t in z["thr"]])
    rgb, grip, ids = batch_inputs(cfg, B, 3, eng.dev)
    T = ids.shape[1]
    ids32 = torch.full((B, 32), 1, dtype=ids.dtype, device=ids.device)
    ids32[:, :T] = ids
    mask = torch.zeros(B, 32, dtype=torch.bool, device=ids.device)
    mask[:, :T] = True
    for exit_id in (11, None):
        outs = []
        for i, m in ((ids, None), (ids32, mask)):
            eng.reset()
            r = None
            for s in range(2):                                    # second step: LSTM carry included
                rgb_s, grip_s, _ = batch_inputs(cfg, B, 3 + s, eng.dev)
                r = eng.step(rgb_s, grip_s, i, m, exit_id=exit_id)
            outs.append(r)
        for e in range(B):
            assert outs[0][e]["exit_layer"] == outs[1][e]["exit_layer"], (exit_id, e)
            assert float((outs[0][e]["pose"] - outs[1][e]["pose"]).abs().max()) < 1e-4, (exit_id, e)
            assert abs(outs[0][e]["gripper"] - outs[1][e]["gripper"]) < 1e-4


def test_full_size_compaction_is_bit_identical_to_the_uncompacted_batch(setup):
    """SURVEY 8(f).4 at 3B size: the 8-environment dynamic episode above runs WITH compaction of exited environments (the default); the
    same episode on a sibling engine with compaction off must give bit-identical exit layers, actions and LSTM states."""
    z, cfg, eng, B = setup
    off = DeerEngine(cfg, None, n_envs=B, weights_from=eng)
    off.set_compaction(False)
    thr = [float(t) for t in z["thr"]]
    for e_ in (eng, off):
        e_.configure_exit(cfg.exit_ids(), int(z["max_layer"]), 1)
        e_.set_thresholds(thr)
        e_.reset()
    layers = set()
    ids_keep = None
    for s in range(10):
        rgb, grip, ids = batch_inputs(cfg, B, s, eng.dev)
        ids_keep = ids if ids_keep is None else ids_keep
        ra = eng.step(rgb, grip, ids_keep, None)
        rb = off.step(rgb, grip, ids_keep, None)
        for e in range(B):
            assert ra[e]["exit_layer"] == rb[e]["exit_layer"] and torch.equal(ra[e]["pose"], rb[e]["pose"]) and ra[e]["gripper"] == rb[e]["gripper"], (s, e)
            layers.add(ra[e]["exit_layer"])
        torch.cuda.synchronize()
        assert torch.equal(eng.h_state, off.h_state) and torch.equal(eng.c_state, off.c_state)
    assert len(layers) > 2, layers


_FRAME4_PROBE = r'''
import json, os, sys
import numpy as np, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], "tests"))
from deer_vla_amd import synthetic as syn
from deer_vla_amd.config import deer_3b
from deer_vla_amd.engine import DeerEngine
from golden_util import full_size_state
cfg = deer_3b(max_layer=12)
base = DeerEngine(cfg, full_size_state(cfg, 0, std="0.02", bf16_round=True), n_envs=8)
out = []
for B in (8, 16):          # 16 frames: one round of frame tiles per GEMM launch; 32 frames: two rounds
    eng = base if B == 8 else DeerEngine(cfg, None, n_envs=B, weights_from=base)
    for s in range(2):
        pe = [syn.synthetic_step_inputs(cfg, s, rank=e % 8, text_seed=7 + e % 8) for e in range(B)]
        rgb = torch.stack([p[0] for p in pe]).to(eng.dev, eng.img_dtype); grip = torch.stack([p[1] for p in pe]).to(eng.dev, eng.img_dtype)
        r = eng.step(rgb, grip, torch.cat([p[2] for p in pe]).to(eng.dev), None, exit_id=3)
        out.append(np.stack([np.concatenate([np.asarray(x["pose"], np.float32).reshape(-1), [np.float32(x["gripper"])]]) for x in r]))
        out.append(eng.vis_x_f32.float().cpu().numpy().copy())
np.savez(sys.argv[2], *out)
'''


def test_four_wave_frame_tiles_selected_by_env_knob_leave_every_engine_output_unchanged(tmp_path):
    """DEER_GEMM_FRAME4=1 swaps the 16-wave frame tiles of the env-batch vision tower for the four-wave ones (csrc/gemm_bigm.hip:
    gemm_frame4_kernel).  The knob is read once per process: two fresh processes run the same full-size static steps at 8 and 16
    environments; actions and media tokens must agree bit for bit."""
    import subprocess
    import sys
    root = os.path.dirname(HERE)
    res = []
    for v in ("0", "1"):
        path = str(tmp_path / f"frame4_{v}.npz")
        env = dict(os.environ, DEER_GEMM_FRAME4=v)
        subprocess.run([sys.executable, "-c", _FRAME4_PROBE, root, path], check=True, env=env, timeout=600)
        z = np.load(path)
        res.append([z[k] for k in z.files])
    for a, b in zip(*res):
        assert np.isfinite(a).all() and np.array_equal(a, b)
