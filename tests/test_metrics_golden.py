"""Harness metrics pinned to the reference: tests/golden/metrics.json holds the output of the reference's OWN
``count_success`` / ``count_exit_ratio`` / ``print_and_save`` (robot_flamingo/eval/eval_utils.py:47-118, executed by
make_golden.py::gen_metrics) on seeded per-chain results over the first 16 chains of the reference's ``eval_sequences.json``."""
import contextlib
import io
import json
import os

import pytest
import torch

from deer_vla_amd import distributed as dd
from deer_vla_amd import rollout as ro

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def gold():
    return json.load(open(os.path.join(HERE, "golden", "metrics.json")))


def lists(gold):
    pc = gold["per_chain"]
    return ([c["n_ok"] for c in pc], [c["ok_exits"] for c in pc], [c["fail_exits"] for c in pc], [c["ok_steps"] for c in pc],
            [c["ok_llm"] for c in pc], [c["fail_llm"] for c in pc])


def test_count_success_and_exit_ratio_match_the_reference(gold):
    res, ok_ex, _, _, _, _ = lists(gold)
    assert ro.count_success(res) == gold["count_success"]
    assert ro.count_exit_ratio(ro.merge_multi_list(ok_ex), gold["n_layer"]) == gold["count_exit_ratio_success"]


def test_print_and_save_report_matches_the_reference_line_by_line(gold):
    res, ok_ex, fail_ex, steps, ok_llm, fail_llm = lists(gold)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        ret = ro.print_and_save(res, ro.merge_multi_list(ok_ex), ro.merge_multi_list(fail_ex), ro.merge_multi_list(steps),
                                ro.merge_multi_list(ok_llm), ro.merge_multi_list(fail_llm), gold["sequences"], None, gold["n_layer"], 0)
    assert [float(ret[0]), float(ret[1])] == gold["print_and_save_return"]
    assert buf.getvalue().splitlines() == gold["print_and_save_stdout"].splitlines()


def test_packed_all_reduce_metrics_equal_the_reference_report(gold):
    """the ONE packed all-reduce that replaces gather_object (distributed.py) carries the same numbers"""
    res, ok_ex, _, _, _, _ = lists(gold)
    m = dd.reduce_metrics(dd.pack_metrics(res, ro.merge_multi_list(ok_ex), gold["n_layer"]))
    assert abs(m["avg_seq_len"] - gold["print_and_save_return"][0]) < 1e-12
    assert abs(m["avg_exit"] - gold["print_and_save_return"][1]) < 1e-9
    assert m["chain_sr"] == gold["count_success"]
    n = sum(m["exit_hist"])
    assert [h / n for h in m["exit_hist"]] == gold["count_exit_ratio_success"]


@pytest.mark.gpu
def test_rollout_driven_by_the_reference_sequence_and_annotation_files(gold):
    """evaluate_policy_ddp over chains / enriched instructions taken from the reference's JSON files (initial states are the
    simulator's business: the synthetic environment ignores them), exit bookkeeping and report included."""
    from deer_vla_amd import synthetic as syn
    from deer_vla_amd.config import deer_tiny
    from deer_vla_amd.factory import create_model_and_transforms
    from deer_vla_amd.value_net import ActionValueNet, ExitController
    cfg = deer_tiny()
    sd = syn.make_synthetic_state(cfg, 3, bf16_round=True)
    model, image_processor, tokenizer = create_model_and_transforms(
        "ViT-L-14", "openai", "", "", cross_attn_every_n_layers=1, window_size=12, use_gripper=True, fusion_mode="post",
        llm_name="mpt_dolly_3b", state_dict=sd, cfg=cfg)
    vn = ActionValueNet(model.get_all_exit_idx(), model.extra_exit, cfg.exit_interval, 12, "L2")
    ctl = ExitController(vn, model.get_all_exit_idx(), steps_per_stage=1, leq=True, max_layer=cfg.early_exit_layer + 1)
    ctl._set_threshold_value([0.02] * (ctl.real_num_exit - 1) + [1e5])
    w = ro.ModelWrapper(model, tokenizer, image_processor, torch.float32, early_exit=True, exit_controller=ctl)
    seqs = [(None, chain) for _, chain in gold["sequences"][:3]]
    env = ro.SyntheticEnv(0)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        out = ro.evaluate_policy_ddp(w, env, seqs, gold["annotations"], ro.steps_task_checker(4), ep_len=6, report=True)
    assert out["n_chains"] == 3 and out["avg_seq_len"] == 5.0 and out["n_steps"] == 3 * 5 * 4
    assert "Average successful sequence length: 5.0" in buf.getvalue()
    assert sum(out["exit_hist"]) == out["n_steps"] and all(out["exit_hist"][i] == 0 for i in range(cfg.n_layers) if i not in cfg.exit_ids())
