"""GPU parity of the head / exit-criterion VARIANTS the reference supports and the device code claims, driven through
the ENGINE (C ABI) on goldens the reference's own modules produced (tests/golden/make_golden.py, round 6):

 * plain ``nn.LSTM`` + MLP heads without LayerNorm (action_head.py:72-79,86-116), ``pooling='avg'`` + three hidden layers
   (:480-483): head_plain.npz / head_avg3.npz (DeterministicDecoder alone) and deer_forward_plain.npz / deer_forward_avg3.npz
   (MPTFlamingo.forward, static and dynamic exits);
 * ``threshold_type`` mean / max / cosine (value_net.py:105-117) with the DEVICE-side criterion deciding: deer_forward_thr.npz;
 * a controller over CONSECUTIVE exit layers at one environment: deer_forward_consec.npz;
 * ``exit_interval=1`` (layer 0 an exit): the reference's dynamic exit asserts (value_net.py:119) - the engine refuses likewise;
 * the window-mode calibration call on right-padded instructions of different lengths (value_net.py:333-386): deer_window_padded.npz.
Tolerances: actions within 1e-2 (bf16 arithmetic, north_star), exit layers exact (every fixture's decisions keep >= 5 % margin)."""
import dataclasses

import pytest
import torch

pytestmark = pytest.mark.gpu

from golden_util import load, s2str  # noqa: E402
from deer_vla_amd import synthetic as syn  # noqa: E402
from deer_vla_amd.engine import DeerEngine  # noqa: E402
from deer_vla_amd.action_head import DeterministicDecoder  # noqa: E402

ACTION_TOL = 1e-2


def _head_only_cfg(cfg):
    """The head goldens' toy config (d_model 32, hidden 16) with tower dims the engine can be built with: the head's parameters are
    seeded by NAME (synthetic.make_synthetic_state), so ``extra_exit.*`` is the tensor set the reference module was loaded with."""
    return dataclasses.replace(cfg, vit_width=128, vit_heads=2, vit_layers=1, vit_mlp=256, image_size=28, perc_heads=2, perc_dim_head=64,
                               perc_latents=8, perc_depth=1, xattn_heads=2, xattn_dim_head=64, n_layers_total=4, early_exit_layer=1)


@pytest.mark.parametrize("precision,tol", [("fp32", 2e-5), ("fp16", 2e-3), ("bf16", 2e-3)])
@pytest.mark.parametrize("name", ["head_ln.npz", "head_plain.npz", "head_avg3.npz"])
def test_engine_head_matches_reference_deterministic_decoder(name, precision, tol):
    """DeterministicDecoder step sequence with the commit / stash protocol (action_head.py:548-558) and the 12-step window call
    (:588-595) on the engine's head kernels.  fp32 arithmetic (f32 head weights): against the REFERENCE module's recorded outputs, 2e-5.
    Product arithmetic (fp16 or bf16 head weights, f32 activations): the fixture's weights are not representable in 16 bits, so the
    expected values come from the oracle head (pinned on the same fixture by tests/test_oracle_golden.py) run on the ROUNDED weights
    the engine holds - what remains is summation order."""
    from oracle import deer_oracle as orc
    cfg0, seed, g = load(name)
    cfg = _head_only_cfg(cfg0)
    sd = syn.make_synthetic_state(cfg, seed)
    ref_sd = syn.make_synthetic_state(cfg0, seed)
    for k, v in ref_sd.items():                                   # the head tensors are the ones the reference module was loaded with
        if k.startswith("extra_exit."):
            assert torch.equal(v, sd[k]), k
    W = cfg0.window_size
    exp = {k: g[k] for k in ("pose", "grip", "h", "c", "wpose", "wgrip", "wgrip_logits")}
    if precision != "fp32":
        rsd = syn.round_state_to_bf16(cfg0, ref_sd) if precision == "bf16" else syn.round_state_to_fp16(cfg0, ref_sd)
        o = orc.OracleHead(rsd, cfg0)
        o.window_size = 1
        ps, gs, hs, cs = [], [], [], []
        for t in range(g["feats"].shape[0]):
            a, gr = o(g["feats"][t], update_hidden_state=bool(g["upd"][t]))
            ps.append(a)
            gs.append(gr)
            z = torch.zeros(cfg0.lstm_num_layers, 1, cfg0.head_hidden)
            hs.append(z if o.hidden_state is None else o.hidden_state[0].clone())
            cs.append(z if o.hidden_state is None else o.hidden_state[1].clone())
        o2 = orc.OracleHead(rsd, cfg0)
        o2.window_size = W
        wa, (wg, wl) = o2(g["wfeat"], with_gripper_logits=True)
        o2.last_action = True
        _, (_, wl_last) = o2(g["wfeat"], with_gripper_logits=True)
        exp = dict(pose=torch.stack(ps), grip=torch.stack(gs), h=torch.stack(hs), c=torch.stack(cs), wpose=wa, wgrip=wg, wgrip_logits=wl_last)
    eng = DeerEngine(cfg, sd, precision=precision)
    head = DeterministicDecoder(eng, window_size=1)
    head.clear_hidden_state()
    for t in range(g["feats"].shape[0]):
        a, gr = head(g["feats"][t], update_hidden_state=bool(g["upd"][t]))
        assert float((a.cpu() - exp["pose"][t]).abs().max()) < tol, (t, a, exp["pose"][t])
        assert float((gr.cpu() - exp["grip"][t]).abs().max()) < tol, t
        hs = head.hidden_state
        if hs is None:
            assert float(exp["h"][t].abs().max()) == 0.0
        else:
            assert float((hs[0].cpu() - exp["h"][t]).abs().max()) < tol and float((hs[1].cpu() - exp["c"][t]).abs().max()) < 2 * tol, t
    # window mode: the two windows are the environments of one head evaluation, the LSTM runs the 12 steps from a zero state
    wf = g["wfeat"].view(2, W, -1, cfg.d_model)
    w = eng.sibling(2)
    w.h_state.zero_()
    w.c_state.zero_()
    w._head_state_changed()
    T = wf.shape[2]
    rows = [w._head_eval(wf[:, t].reshape(2 * T, cfg.d_model).contiguous().cuda(), commit=True).cpu() for t in range(W)]
    out = torch.stack(rows, dim=1)                                # (2, W, 8)
    assert float((out[..., :6] - exp["wpose"]).abs().max()) < tol
    assert float((out[..., 6:7] - exp["wgrip"]).abs().max()) < tol
    assert float((out[:, -1:, 7:8] - exp["wgrip_logits"]).abs().max()) < 4 * tol


R6 = [("deer_forward_plain.npz", "fp16"), ("deer_forward_avg3.npz", "fp16"), ("deer_forward_thr.npz", "fp16"), ("deer_forward_consec.npz", "fp16"),
      ("deer_forward_plain.npz", "bf16"), ("deer_forward_thr.npz", "bf16"),
      ("deer_forward_plain.npz", "fp32"), ("deer_forward_avg3.npz", "fp32"), ("deer_forward_thr.npz", "fp32"),
      # fusion_mode="pre" (flamingo_mpt.py:585-607): both cameras' patch tokens through ONE PerceiverResampler call, 64 media tokens
      ("deer_forward_pre.npz", "fp16"), ("deer_forward_pre.npz", "bf16"), ("deer_forward_pre.npz", "fp32")]


@pytest.mark.parametrize("name,precision", R6)
def test_engine_matches_reference_forward_for_head_and_criterion_variants(name, precision):
    """Static exits and one dynamic episode per threshold type of the reference's own MPTFlamingo.forward; the exit decision is taken by
    the device-side criterion (csrc/head.hip) with the fixture's thresholds."""
    cfg, seed, g = load(name)
    sd = syn.make_synthetic_state(cfg, seed, bf16_round=True)
    eng = DeerEngine(cfg, sd, precision=precision)
    tol = ACTION_TOL if precision != "fp32" else 1e-3
    ids, mask, rgb, grip = g["ids"].long(), g["mask"].bool(), g["rgb"], g["grip"]
    for eid in (3, 4):
        eng.reset()
        for s in range(rgb.shape[0]):
            r = eng.step(rgb[s], grip[s], ids, mask, exit_id=eid, use_graph=(s > 1))
            assert r["exit_layer"] == eid
            assert float((r["pose"] - g[f"static{eid}_pose"][s].reshape(-1)).abs().max()) < tol, (eid, s)
            assert abs(r["gripper"] - float(g[f"static{eid}_grip"][s])) < tol, (eid, s)
    if "vis_x" in g:                                         # the media tokens of the last step against the reference's
        ref_vis = g["vis_x"].reshape(-1, cfg.vit_width)
        vis = eng.vis_x_f32.cpu()
        assert vis.shape == ref_vis.shape == (cfg.n_media, cfg.vit_width)
        assert float((vis - ref_vis).abs().max() / ref_vis.abs().max()) < (2e-2 if precision != "fp32" else 1e-4)
    exit_ids = [int(v) for v in g["exit_ids"]]
    for ttype in s2str(g["thr_types"]).split(","):
        e2 = DeerEngine(cfg, None, precision=precision, weights_from=eng, threshold_type=ttype)
        e2.configure_exit(exit_ids, 12, 1)
        e2.set_thresholds([float(t) for t in g[ttype + "_thr"]])
        e2.reset()
        ref_deltas = list(zip(g[ttype + "_rec_layer"].tolist(), g[ttype + "_rec_delta"].tolist()))
        k = 0
        for s in range(rgb.shape[0]):
            r = e2.step(rgb[s], grip[s], ids, mask, use_graph=(s > 1))
            assert r["exit_layer"] == int(g[ttype + "_exit"][s]), (ttype, s, r["exit_layer"], g[ttype + "_exit"].tolist())
            assert float((r["pose"] - g[ttype + "_pose"][s].reshape(-1)).abs().max()) < tol, (ttype, s)
            assert abs(r["gripper"] - float(g[ttype + "_grip"][s])) < tol, (ttype, s)
            # the deltas the device criterion computed, exit by exit, against the ones the reference's ActionValueNet recorded
            for slot, e in enumerate(exit_ids):
                if e > r["exit_layer"]:
                    break
                layer, d_ref = ref_deltas[k]
                assert int(layer) == e
                d = float(r["deltas"][slot])
                assert abs(d - d_ref) < max(0.06 * abs(d_ref), 3e-4 if precision != "fp32" else 2e-5), (ttype, s, e, d, d_ref)
                k += 1
        assert k == len(ref_deltas)


def test_exit_interval_one_dynamic_exit_is_refused_like_the_reference():
    cfg, seed, g = load("deer_forward_int1.npz")
    sd = syn.make_synthetic_state(cfg, seed, bf16_round=True)
    eng = DeerEngine(cfg, sd)
    ids, mask, rgb, grip = g["ids"].long(), g["mask"].bool(), g["rgb"], g["grip"]
    assert s2str(g["dynamic_raises"]).startswith("AssertionError")
    with pytest.raises(NotImplementedError, match="first layer"):
        eng.step(rgb[0], grip[0], ids, mask)
    eng.reset()
    r = eng.step(rgb[0], grip[0], ids, mask, exit_id=0, use_graph=False)
    assert r["exit_layer"] == 0
    assert float((r["pose"] - g["static0_pose"].reshape(-1)).abs().max()) < ACTION_TOL
    assert abs(r["gripper"] - float(g["static0_grip"])) < ACTION_TOL


@pytest.mark.parametrize("precision", ["fp16", "bf16", "fp32"])
def test_window_mode_on_right_padded_instructions_matches_reference(precision):
    """``generate_action_values`` on a batch of windows with instructions of different lengths (value_net.py:333-386): every layer's hidden
    state at EVERY row (pad rows are queries like any other), extra_exit over the reference's random history layers, and the calibration
    deltas - the head pools over all T rows, pad rows included, as the reference does (action_head.py:519-520)."""
    from deer_vla_amd.flamingo_mpt import MPTFlamingo
    cfg, seed, g = load("deer_window_padded.npz")
    sd = syn.make_synthetic_state(cfg, seed, bf16_round=True)
    W = cfg.window_size
    model = MPTFlamingo(cfg, sd, window_size=W, precision=precision)
    ids, mask, rgb, grip = g["ids"].long(), g["mask"].bool(), g["rgb"], g["grip"]
    bs, T = ids.shape
    S = cfg.image_size
    vx = rgb.reshape(bs * W, 1, 1, 3, S, S)
    vg = grip.reshape(bs * W, 1, 1, 3, S, S)
    input_ids = ids.unsqueeze(1).repeat(1, W, 1).flatten(0, 1)
    attention_mask = mask.unsqueeze(1).repeat(1, W, 1).flatten(0, 1)
    out, exit_outputs, extra, rand_feat, rand_idx = model._forward_window(vx, input_ids, attention_mask, vg, with_gripper_logits=True,
                                                                         rand_layers=g["rand_layers"])
    hid = torch.stack(out.hidden_states).cpu()                   # (L, bs*W, T, d)
    ref = g["hidden"]
    rel = 2e-2 if precision != "fp32" else 1e-4
    assert float((hid - ref).norm() / ref.norm()) < rel
    pad = ~attention_mask                                         # the pad rows on their own (they must be real, not zeros / garbage)
    assert float((hid[:, pad] - ref[:, pad]).norm() / ref[:, pad].norm()) < rel
    assert torch.equal(rand_idx.cpu(), g["rand_layers"])
    tol = ACTION_TOL if precision != "fp32" else 1e-3
    assert float((extra[0].cpu() - g["extra_pose"]).abs().max()) < tol
    assert float((extra[1][0].cpu() - g["extra_grip"]).abs().max()) < tol
    eng = model.engine
    eng.configure_exit(cfg.exit_ids(), 12, 1)
    vals = eng.generate_values(hid.permute(1, 0, 2, 3).reshape(bs, W, cfg.n_layers, T, cfg.d_model).to(eng.dev), g["rand_layers"], "L2")
    assert vals.shape == g["delta"].shape
    assert float((vals - g["delta"]).abs().max()) < (5e-3 if precision != "fp32" else 1e-4), (vals, g["delta"])


def test_layerwise_exit_eval_env_batch_with_staggered_exits_matches_independent_oracle_runs():
    """``layerwise_exit_eval`` in an ENV BATCH on dynamic steps - the per-layer heads read hidden_states[layer] in
    environment order after the step, so the trunk must not have compacted exited environments away (csrc/model.hip: compact_on).  Every
    environment of the batch against its own single-environment oracle run (own per-layer LSTM histories), exits at different layers."""
    from test_engine_parity import probe_thresholds, oracle_episode_margins, BAND
    from deer_vla_amd.config import deer_tiny
    from deer_vla_amd.flamingo_mpt import MPTFlamingo
    from deer_vla_amd.value_net import ActionValueNet, ExitController
    cfg = deer_tiny(layerwise_exit_eval=True)
    sd = syn.make_synthetic_state(cfg, 3, bf16_round=True)
    B, n_steps = 4, 10
    env_inputs = [[syn.synthetic_step_inputs(cfg, s, rank=e, text_len=11, text_seed=7 + e) for s in range(n_steps)] for e in range(B)]
    thr, _ = probe_thresholds(cfg, sd, env_inputs[0], 12, 1)
    refs = [oracle_episode_margins(cfg, sd, env_inputs[e], thr, 12, 1) for e in range(B)]
    model = MPTFlamingo(cfg, sd, n_envs=B)
    vn = ActionValueNet(model.get_all_exit_idx(), None, cfg.exit_interval, cfg.window_size, "L2")
    ctl = ExitController(vn, model.get_all_exit_idx(), steps_per_stage=1, leq=True, exit_dist="exp", max_layer=12)
    ctl._set_threshold_value(thr)
    model.clear_all_exit_memory()
    ids = torch.cat([env_inputs[e][0][2] for e in range(B)]).cuda()
    mask = torch.ones_like(ids, dtype=torch.bool)
    alive, seen, compared = [True] * B, set(), 0
    for s in range(n_steps):
        ctl.set_timestep(s)
        rgb = torch.stack([env_inputs[e][s][0] for e in range(B)]).cuda()
        grip = torch.stack([env_inputs[e][s][1] for e in range(B)]).cuda()
        pose, gr, exits = model.step_env_batch(rgb, ids, mask, grip, exit_controller=ctl)
        for e in range(B):
            if not alive[e]:
                continue
            ex, p_ref, g_ref, margin = refs[e][s]
            if exits[e] != ex:
                assert margin <= BAND, ("exit mismatch outside the knife-edge band", e, s, exits[e], ex, margin)
                alive[e] = False
                continue
            assert float((pose[e].cpu() - p_ref).abs().max()) < ACTION_TOL, (e, s, exits[e])
            assert abs(float(gr[e]) - g_ref) < ACTION_TOL, (e, s)
            seen.add(ex)
            compared += 1
    assert compared >= B * n_steps - 6 and len(seen) > 1, (compared, seen)
