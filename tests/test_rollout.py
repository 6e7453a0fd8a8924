"""Closed-loop rollout harness (deer_vla_amd/rollout.py) on the MI355X: ModelWrapper protocol, sub-task / chain drivers,
metric reduction.  Reference: robot_flamingo/eval/eval_utils.py:152-700."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deer_vla_amd import rollout as ro  # noqa: E402
from deer_vla_amd import synthetic as syn  # noqa: E402
from deer_vla_amd.config import deer_tiny  # noqa: E402


def test_preprocess_text_and_counts_cpu():
    from deer_vla_amd.factory import SyntheticTokenizer, ClipImageProcessor
    cfg = deer_tiny()
    tok = SyntheticTokenizer(cfg)
    ids, mask = ro.preprocess_text_calvin(["push the red block  "], tok)
    assert ids.shape == mask.shape and ids.shape[0] == 1
    assert int(ids[0, 0]) == cfg.media_token_id and int(ids[0, -2]) == cfg.eoc_token_id and int(ids[0, -1]) == tok.eos_token_id
    img = ro.preprocess_image([np.ones((200, 200, 3), dtype=np.uint8)], ClipImageProcessor(cfg.image_size))
    assert img.shape == (1, 3, cfg.image_size, cfg.image_size)
    assert ro.count_success([5, 0, 3, 1]) == [0.75, 0.5, 0.5, 0.25, 0.25]
    assert ro.count_exit_ratio([1, 1, 3], 4) == [0.0, 2 / 3, 0.0, 1 / 3]
    env = ro.SyntheticEnv(seed=3)
    o0 = env.get_obs()
    o1, _, _, info = env.step(np.zeros(7, dtype=np.float16))
    assert info == 1 and not np.array_equal(o0["rgb_obs"]["rgb_static"], o1["rgb_obs"]["rgb_static"])
    assert ro.steps_task_checker(2)(0, 2, "x") and not ro.steps_task_checker(2)(0, 1, "x")


@pytest.fixture(scope="module")
def harness():
    from deer_vla_amd.factory import create_model_and_transforms
    from deer_vla_amd.value_net import ExitController, ActionValueNet
    cfg = deer_tiny()
    sd = syn.make_synthetic_state(cfg, seed=5, std="fanin")
    model, image_processor, tokenizer = create_model_and_transforms(
        "ViT-L-14", "openai", "", "", cross_attn_every_n_layers=1, window_size=12, use_gripper=True, fusion_mode="post",
        llm_name="mpt_dolly_3b", state_dict=sd, cfg=cfg)
    vn = ActionValueNet(model.get_all_exit_idx(), model.extra_exit, cfg.exit_interval, 12, "L2")
    ctl = ExitController(vn, model.get_all_exit_idx(), steps_per_stage=1, leq=True, max_layer=cfg.early_exit_layer + 1)
    ctl._set_threshold_value([0.02] * (ctl.real_num_exit - 1) + [1e5])
    return cfg, model, image_processor, tokenizer, ctl


@pytest.mark.gpu
def test_model_wrapper_step_protocol(harness):
    cfg, model, image_processor, tokenizer, ctl = harness
    w = ro.ModelWrapper(model, tokenizer, image_processor, torch.float32, use_diff=False, amp=True, exit_id=None, early_exit=True,
                        exit_controller=ctl, multi_execution=1)
    env = ro.SyntheticEnv(seed=1)
    obs = env.get_obs()
    w.reset()
    acts, exits = [], []
    for t in range(6):
        ctl.module.set_timestep(t)
        a = w.step(obs, "lift the blue block")
        assert a.shape == (1, 7) and a.dtype == np.float16
        assert a[0, 6] in (-1.0, 1.0) and np.all(np.abs(a[0, :6]) <= 1.0)          # tanh pose, gripper in {-1, +1}
        assert w.current_exit_layer in model.get_all_exit_idx()
        acts.append(a[0].copy())
        exits.append(w.current_exit_layer)
        obs, _, _, _ = env.step(a[0])
    assert w.step(obs, "lift the blue block", get_action=False) is None
    # the same observations through a fresh wrapper reproduce the episode (reset clears the LSTM / controller state)
    env.reset()
    env.shift = 0
    obs = env.get_obs()
    w.reset()
    for t in range(6):
        ctl.module.set_timestep(t)
        a = w.step(obs, "lift the blue block")
        assert np.array_equal(a[0], acts[t]) and w.current_exit_layer == exits[t]
        obs, _, _, _ = env.step(a[0])
    # a static exit_id wrapper always reports that layer
    ws = ro.ModelWrapper(model, tokenizer, image_processor, torch.float32, exit_id=model.get_all_exit_idx()[0], early_exit=False)
    ws.step(obs, "lift the blue block")
    assert ws.current_exit_layer == model.get_all_exit_idx()[0]


@pytest.mark.gpu
def test_model_wrapper_action_ensemble_over_the_last_two_exits(harness):
    """``use_action_ensemble`` (eval_utils.py:457-461): the executed action is the mean of the last two exit-check actions of the step
    (``ActionValueNet.get_ensemble_action``), the gripper decided on the mean probability; the list is reset after every step.  The
    wrapper's output is compared with the engine's own per-exit actions (the device-side ensemble against the reference is pinned in
    tests/test_dropin_surface.py), and the same frames without ensembling exit at the same layers."""
    cfg, model, image_processor, tokenizer, ctl = harness
    with pytest.raises(ValueError):                                   # the ensemble lives in the exit controller's value net
        ro.ModelWrapper(model, tokenizer, image_processor, torch.float32, exit_id=1, early_exit=False, use_action_ensemble=True)
    plain = ro.ModelWrapper(model, tokenizer, image_processor, torch.float32, early_exit=True, exit_controller=ctl)
    ens = ro.ModelWrapper(model, tokenizer, image_processor, torch.float32, early_exit=True, exit_controller=ctl, use_action_ensemble=True)
    frames = []
    env = ro.SyntheticEnv(seed=3)
    obs = env.get_obs()
    for t in range(6):
        frames.append(obs)
        obs, _, _, _ = env.step(np.zeros(7, dtype=np.float16))
    runs = {}
    for name, w in (("plain", plain), ("ens", ens)):
        w.reset()
        rec = []
        for t, ob in enumerate(frames):
            ctl.module.set_timestep(t)
            a = w.step(ob, "push the red block")
            r = model.engine.read_result()
            rec.append((a[0].astype(np.float32), w.current_exit_layer, r["pose"].numpy(), r["gripper"], r["ens_pose"].numpy(), r["ens_gripper"],
                        r["ens_count"], r["n_evals"]))
        runs[name] = rec
    two = 0
    for (ap, lp, pose, gprob, _, _, _, _), (ae, le, _, _, epose, egrip, ecount, n_evals) in zip(runs["plain"], runs["ens"]):
        assert lp == le                                                # the exit decision does not depend on the ensembling
        assert np.allclose(ap[:6], pose.astype(np.float16).astype(np.float32)) and ap[6] == (1.0 if gprob > 0.5 else -1.0)
        assert np.allclose(ae[:6], epose.astype(np.float16).astype(np.float32)) and ae[6] == (1.0 if egrip > 0.5 else -1.0)
        assert ecount == min(2, n_evals - 1)                          # head evaluations of the step = pseudo action + the exit checks
        two += ecount == 2
    assert two > 0                                                     # some step exited at a later check: a real two-action mean


@pytest.mark.gpu
def test_env_batch_action_ensemble_equals_single_environment_wrappers():
    """``BatchedModelWrapper(use_action_ensemble=True)``: every slot's action equals what a single-environment ensembling wrapper
    returns for the same frames (the device forms the mean per environment)."""
    from deer_vla_amd.factory import create_model_and_transforms
    from deer_vla_amd.value_net import ExitController, ActionValueNet
    cfg = deer_tiny()
    sd = syn.make_synthetic_state(cfg, seed=5, std="fanin")
    B, n = 3, 5

    def build(n_envs):
        model, proc, tok = create_model_and_transforms("ViT-L-14", "openai", "", "", cross_attn_every_n_layers=1, window_size=12, use_gripper=True,
                                                       fusion_mode="post", llm_name="mpt_dolly_3b", state_dict=sd, cfg=cfg, n_envs=n_envs)
        vn = ActionValueNet(model.get_all_exit_idx(), model.extra_exit, cfg.exit_interval, 12, "L2")
        ctl = ExitController(vn, model.get_all_exit_idx(), steps_per_stage=1, leq=True, max_layer=cfg.early_exit_layer + 1)
        ctl._set_threshold_value([0.02] * (ctl.real_num_exit - 1) + [1e5])
        return model, proc, tok, ctl
    frames = []
    for e in range(B):
        env = ro.SyntheticEnv(seed=10 + e)
        obs, fl = env.get_obs(), []
        for t in range(n):
            fl.append(obs)
            obs, _, _, _ = env.step(np.zeros(7, dtype=np.float16))
        frames.append(fl)
    goals = ["lift the blue block"] * B
    model1, proc, tok, ctl1 = build(1)
    single = []
    for e in range(B):
        w = ro.ModelWrapper(model1, tok, proc, torch.float32, early_exit=True, exit_controller=ctl1, use_action_ensemble=True)
        w.reset()
        acts = []
        for t in range(n):
            ctl1.module.set_timestep(t)
            acts.append((w.step(frames[e][t], goals[e])[0].copy(), w.current_exit_layer))
        single.append(acts)
    modelB, procB, tokB, ctlB = build(B)
    wb = ro.BatchedModelWrapper(modelB, tokB, procB, torch.float32, exit_controller=ctlB, use_action_ensemble=True)
    for e in range(B):
        wb.reset_env(e)
    for t in range(n):
        a = wb.step([frames[e][t] for e in range(B)], goals)
        for e in range(B):
            assert wb.current_exit_layers[e] == single[e][t][1], (t, e)
            got, want = np.asarray(a[e], dtype=np.float32).reshape(-1), single[e][t][0].astype(np.float32)
            assert np.abs(got[:6] - want[:6]).max() < 2e-3 and got[6] == want[6], (t, e, got, want)


@pytest.mark.gpu
def test_chain_evaluation_and_metrics(harness):
    cfg, model, image_processor, tokenizer, ctl = harness
    w = ro.ModelWrapper(model, tokenizer, image_processor, torch.float32, early_exit=True, exit_controller=ctl)
    env = ro.SyntheticEnv(seed=2)
    ann = {"a": ["open the drawer"], "b": ["turn on the light"], "c": ["push the block left"]}
    seqs = [(None, ["a", "b", "c"]), (None, ["b", "a"]), (None, ["c"]), (None, ["a", "c", "b"])]
    ok = ro.steps_task_checker(4)                      # every sub-task "succeeds" after 4 steps
    out = ro.evaluate_policy_ddp(w, env, seqs, ann, ok, ep_len=6)
    assert out["n_chains"] == 4 and out["avg_seq_len"] == (3 + 2 + 1 + 3) / 4
    assert out["n_steps"] == out["n_steps_all"] == 4 * 9 and sum(out["exit_hist"]) == 36
    assert out["chain_sr"][:3] == [1.0, 0.75, 0.5] and out["steps_per_s"] > 0
    never = ro.steps_task_checker(10 ** 6)
    out = ro.evaluate_policy_ddp(w, env, seqs[:2], ann, never, ep_len=5)
    assert out["avg_seq_len"] == 0.0 and out["n_steps"] == 0 and out["n_steps_all"] == 2 * 5


@pytest.mark.gpu
def test_env_batch_per_rank_evaluation_matches_sequential_bookkeeping():
    """One env batch per rank through the public surface: create_model_and_transforms(n_envs=3) -> BatchedModelWrapper ->
    evaluate_policy_batched gives the same chain / step bookkeeping as the one-environment harness on the same chains."""
    from deer_vla_amd import synthetic as syn
    from deer_vla_amd.config import deer_tiny
    from deer_vla_amd.factory import create_model_and_transforms
    from deer_vla_amd.value_net import ActionValueNet, ExitController
    cfg = deer_tiny()
    sd = syn.make_synthetic_state(cfg, 3, bf16_round=True)
    ann = {"a": ["open the drawer"], "b": ["turn on the light bulb now"], "c": ["push the block left"]}
    seqs = [(None, ["a", "b", "c"]), (None, ["b", "a"]), (None, ["c"]), (None, ["a", "c", "b"]), (None, ["c", "b"])]
    outs = []
    for B in (1, 3):
        model, proc, tok = create_model_and_transforms("ViT-L-14", "openai", "", "", window_size=12, use_gripper=True, fusion_mode="post",
                                                       llm_name="mpt_dolly_3b", state_dict=sd, cfg=cfg, n_envs=B)
        vn = ActionValueNet(model.get_all_exit_idx(), None, cfg.exit_interval, 12, "L2")
        ctl = ExitController(vn, model.get_all_exit_idx(), max_layer=cfg.early_exit_layer + 1)
        ctl._set_threshold_value([0.02] * (ctl.real_num_exit - 1) + [1e5])
        ok = ro.steps_task_checker(4)
        if B == 1:
            w = ro.ModelWrapper(model, tok, proc, torch.float32, early_exit=True, exit_controller=ctl)
            outs.append(ro.evaluate_policy_ddp(w, ro.SyntheticEnv(seed=2), seqs, ann, ok, ep_len=6))
        else:
            w = ro.BatchedModelWrapper(model, tok, proc, torch.float32, exit_controller=ctl)
            outs.append(ro.evaluate_policy_batched(w, [ro.SyntheticEnv(seed=2 + b) for b in range(B)], seqs, ann, ok, ep_len=6))
    a, b = outs
    assert a["n_chains"] == b["n_chains"] == 5 and a["avg_seq_len"] == b["avg_seq_len"] == (3 + 2 + 1 + 3 + 2) / 5
    assert a["n_steps"] == b["n_steps"] == 11 * 4 and b["n_steps_all"] == 44 and b["envs_per_rank"] == 3
    assert a["chain_sr"] == b["chain_sr"] and sum(b["exit_hist"]) == 44
    assert all(b["exit_hist"][i] == 0 for i in range(cfg.n_layers) if i not in cfg.exit_ids())


@pytest.mark.gpu
def test_two_env_batches_in_flight_give_the_same_bookkeeping():
    """evaluate_policy_batched(groups=...): a second env batch (MPTFlamingo.sibling(): same device weights, own state) runs in its
    own host thread / stream and draws chains from the same queue - chain and step bookkeeping must equal the one-batch run."""
    from deer_vla_amd import synthetic as syn
    from deer_vla_amd.config import deer_tiny
    from deer_vla_amd.factory import create_model_and_transforms
    from deer_vla_amd.value_net import ActionValueNet, ExitController
    cfg = deer_tiny()
    sd = syn.make_synthetic_state(cfg, 3, bf16_round=True)
    ann = {"a": ["open the drawer"], "b": ["turn on the light bulb now"], "c": ["push the block left"]}
    seqs = [(None, ["a", "b", "c"]), (None, ["b", "a"]), (None, ["c"]), (None, ["a", "c", "b"]), (None, ["c", "b"]), (None, ["b"]), (None, ["a", "a"])]
    B = 2
    model, proc, tok = create_model_and_transforms("ViT-L-14", "openai", "", "", window_size=12, use_gripper=True, fusion_mode="post",
                                                   llm_name="mpt_dolly_3b", state_dict=sd, cfg=cfg, n_envs=B)
    ok = ro.steps_task_checker(4)

    def wrapper(m):
        vn = ActionValueNet(m.get_all_exit_idx(), None, cfg.exit_interval, 12, "L2")
        ctl = ExitController(vn, m.get_all_exit_idx(), max_layer=cfg.early_exit_layer + 1)
        ctl._set_threshold_value([0.02] * (ctl.real_num_exit - 1) + [1e5])
        return ro.BatchedModelWrapper(m, tok, proc, torch.float32, exit_controller=ctl)

    one = ro.evaluate_policy_batched(wrapper(model), [ro.SyntheticEnv(seed=2 + b) for b in range(B)], seqs, ann, ok, ep_len=6)
    second = model.sibling()
    assert second.engine.arena.data_ptr() == model.engine.arena.data_ptr()           # the same device weights
    two = ro.evaluate_policy_batched(wrapper(model), [ro.SyntheticEnv(seed=2 + b) for b in range(B)], seqs, ann, ok, ep_len=6,
                                     groups=[(wrapper(second), [ro.SyntheticEnv(seed=20 + b) for b in range(B)])])
    assert one["n_chains"] == two["n_chains"] == 7 and one["avg_seq_len"] == two["avg_seq_len"]
    assert one["n_steps"] == two["n_steps"] == two["n_steps_all"] == 14 * 4 and two["envs_per_rank"] == 2 * B
    assert one["chain_sr"] == two["chain_sr"] and sum(two["exit_hist"]) == 14 * 4


class _SpeedEnv(ro.SyntheticEnv):
    """an environment whose reset state changes the outcome: reset(speed=k) makes every step count k towards the task oracle"""

    def __init__(self, seed=0):
        super().__init__(seed)
        self.speed = 1

    def reset(self, speed=1, **kwargs):
        super().reset()
        self.speed = speed

    def step(self, action):
        obs, r, d, _ = super().step(action)
        self.t += self.speed - 1
        return self.get_obs(), r, d, self.get_info()


@pytest.mark.gpu
def test_env_batch_resets_every_chain_to_its_initial_state():
    """the env-batch evaluator used to drop each chain's initial_state (bare ``env.reset()``); the reference resets the
    environment to the chain's initial condition (eval_utils.py:587-588).  With an environment whose reset state decides how many
    steps a sub-task takes, the batched run must give the step counts of the one-environment harness."""
    from deer_vla_amd import synthetic as syn
    from deer_vla_amd.config import deer_tiny
    from deer_vla_amd.factory import create_model_and_transforms
    from deer_vla_amd.value_net import ActionValueNet, ExitController
    cfg = deer_tiny()
    sd = syn.make_synthetic_state(cfg, 3, bf16_round=True)
    ann = {"a": ["open the drawer"], "b": ["turn on the light bulb now"]}
    seqs = [({"speed": 2}, ["a", "b"]), (None, ["b", "a"]), ({"speed": 4}, ["a"]), ({"speed": 2}, ["b", "b", "a"])]
    ok = ro.steps_task_checker(4)
    outs = []
    for B in (1, 2):
        model, proc, tok = create_model_and_transforms("ViT-L-14", "openai", "", "", window_size=12, use_gripper=True, fusion_mode="post",
                                                       llm_name="mpt_dolly_3b", state_dict=sd, cfg=cfg, n_envs=B)
        vn = ActionValueNet(model.get_all_exit_idx(), None, cfg.exit_interval, 12, "L2")
        ctl = ExitController(vn, model.get_all_exit_idx(), max_layer=cfg.early_exit_layer + 1)
        ctl._set_threshold_value([0.02] * (ctl.real_num_exit - 1) + [1e5])
        if B == 1:
            w = ro.ModelWrapper(model, tok, proc, torch.float32, early_exit=True, exit_controller=ctl)
            outs.append(ro.evaluate_policy_ddp(w, _SpeedEnv(2), seqs, ann, ok, ep_len=6))
        else:
            w = ro.BatchedModelWrapper(model, tok, proc, torch.float32, exit_controller=ctl)
            outs.append(ro.evaluate_policy_batched(w, [_SpeedEnv(2 + b) for b in range(B)], seqs, ann, ok, ep_len=6))
    a, b = outs
    want = 2 * 2 + 2 * 4 + 1 + 3 * 2                 # steps per sub-task: ceil(4 / speed)
    assert a["n_steps"] == want and b["n_steps"] == want, (a["n_steps"], b["n_steps"], want)
    assert a["avg_seq_len"] == b["avg_seq_len"] == 8 / 4


@pytest.mark.gpu
def test_env_batch_takes_the_references_long_instructions_and_rejects_what_does_not_fit():
    """8 environments x the reference's max_length = 32 tokens (data.py:905-919) = 256 trunk rows fit the bf16 engine, so
    NO instruction of the reference's evaluation annotations is refused at 8 environments per rank: every distinct instruction of
    lang_annotation_cache.json (first line, as the harness feeds it: eval_utils.py:638-644) with more than 10 words (7.9 % of the file, up to
    14 words = 17 tokens with "<image>", "<|endofchunk|>" and eos even at one token per word; tests/golden/long_instructions.json, made by
    tests/golden/make_long_instructions.py) passes the up-front check, and a batch holding the longest ones rolls out.  What cannot fit
    still fails BEFORE the run starts with a clear error: the fp32 arithmetic keeps 128 rows (16 tokens at 8
    environments); a controller with steps_per_stage > 1 rolls out (per-environment stage hold; parity: tests/test_engine_parity.py)."""
    import json
    import os
    from deer_vla_amd import synthetic as syn
    from deer_vla_amd.config import deer_tiny
    from deer_vla_amd.factory import create_model_and_transforms
    from deer_vla_amd.value_net import ActionValueNet, ExitController
    cfg = deer_tiny()
    sd = syn.make_synthetic_state(cfg, 3, bf16_round=True)
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "long_instructions.json")))
    longs = gold["long_instructions"]
    assert len(longs) >= 100 and len(longs[0].split()) == 14 and abs(gold["share_over_10_words"] - 0.0794) < 1e-3

    def build(precision):
        model, proc, tok = create_model_and_transforms("ViT-L-14", "openai", "", "", window_size=12, use_gripper=True, fusion_mode="post",
                                                       llm_name="mpt_dolly_3b", state_dict=sd, cfg=cfg, n_envs=8, precision=precision)
        vn = ActionValueNet(model.get_all_exit_idx(), None, cfg.exit_interval, 12, "L2")
        ctl = ExitController(vn, model.get_all_exit_idx(), max_layer=cfg.early_exit_layer + 1)
        ctl._set_threshold_value([0.02] * (ctl.real_num_exit - 1) + [1e5])
        return model, proc, tok, vn, ctl

    model, proc, tok, vn, ctl = build("fp16")
    w = ro.BatchedModelWrapper(model, tok, proc, torch.float32, exit_controller=ctl)
    assert model.engine.max_T == 32
    w.check_instructions(longs)                                    # every long instruction of the reference's file fits
    ann = {"a": ["open the drawer"], "b": [longs[0]], "c": [longs[1]]}
    out = ro.evaluate_policy_batched(w, [ro.SyntheticEnv(seed=b) for b in range(8)], [(None, ["a", "b"]), (None, ["c", "a"])] * 4, ann,
                                     ro.steps_task_checker(2), ep_len=3)
    assert out["n_chains"] == 8 and out["n_steps"] > 0
    # steps_per_stage > 1 (eval_calvin.py:340): the hold is per environment on the device, the harness hands every slot's own step index
    ctl3 = ExitController(vn, model.get_all_exit_idx(), steps_per_stage=3, max_layer=cfg.early_exit_layer + 1)
    ctl3._set_threshold_value([0.02] * (ctl3.real_num_exit - 1) + [1e5])
    w3 = ro.BatchedModelWrapper(model, tok, proc, torch.float32, exit_controller=ctl3)
    out3 = ro.evaluate_policy_batched(w3, [ro.SyntheticEnv(seed=b) for b in range(8)], [(None, ["a", "b"]), (None, ["c", "a"])] * 4, ann,
                                      ro.steps_task_checker(4), ep_len=5)
    assert out3["n_chains"] == 8 and out3["n_steps"] > 0
    model32, proc32, tok32, _, ctl32 = build("fp32")
    w32 = ro.BatchedModelWrapper(model32, tok32, proc32, torch.float32, exit_controller=ctl32)
    with pytest.raises(ValueError, match="n_envs"):
        ro.evaluate_policy_batched(w32, [ro.SyntheticEnv(seed=b) for b in range(8)], [(None, ["a", "b"])], ann, ro.steps_task_checker(2), ep_len=3)
