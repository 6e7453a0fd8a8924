"""CPU tests of the host-side logic of the product package (no GPU, no HIP calls)."""
import torch

from golden_util import load
from deer_vla_amd import value_net as vn
from deer_vla_amd.config import deer_3b, deer_9b, deer_tiny, DeerConfig
from deer_vla_amd import synthetic as syn


def test_product_threshold_solver_matches_reference_goldens():
    cfg, seed, g = load("thresholds.npz")
    values = g["values"]
    for key in [k for k in g if k.startswith("T_")]:
        model_name, ratio, max_layer = key[2:].rsplit("_", 2)
        ctl = vn.ExitController(None, cfg.exit_ids(), max_layer=int(max_layer))
        T = vn.solve_thresholds(values[: ctl.real_num_exit].clone(), ctl.real_num_exit, float(ratio), "exp", True, model_name)
        assert torch.equal(T, g[key].float()), key
        ctl.set_threshold_from_values(values, float(ratio), model_name)
        assert ctl.threshold_list() == [float(x) for x in g[key]]


def test_exit_structure_of_baseline_configs():
    """SURVEY §8a 'Max-layer semantics' / Appendix C."""
    b = deer_3b(max_layer=12)
    assert b.n_layers == 12 and b.exit_ids() == [1, 3, 5, 7, 9, 11]
    c = vn.ExitController(None, b.exit_ids(), max_layer=12)
    assert c.max_layer == 11 and c.real_num_exit == 6
    s = deer_3b(max_layer=4)
    assert s.n_layers == 5 and s.exit_ids() == [1, 3, 4]
    c = vn.ExitController(None, s.exit_ids(), max_layer=4)
    assert c.max_layer == 3 and c.real_num_exit == 2                      # <=4 layers ever run
    n = deer_9b(max_layer=12)
    assert n.n_layers == 13 and [i for i in range(13) if n.has_xattn(i)] == [3, 7, 11]
    c = vn.ExitController(None, n.exit_ids(), max_layer=12)
    assert c.max_layer == 11


def test_parameter_inventory_sizes_match_survey():
    """SURVEY §8a parameter counts: ViT 303 M, Perceiver 63 M, x-attn 36.7 M/layer, MPT block 50.3 M/layer, head 41 M."""
    cfg = deer_3b(12)
    P = syn.param_shapes(cfg)

    def count(prefix):
        n = 0
        for k, (shape, _) in P.items():
            if k.startswith(prefix):
                m = 1
                for s in shape:
                    m *= s
                n += m
        return n
    assert abs(count("vision_encoder.") / 1e6 - 303.2) < 1.0
    assert abs(count("perceiver.") / 1e6 - 63.0) < 0.5
    assert abs(count("lang_encoder.transformer.blocks.0.gated_cross_attn_layer.") / 1e6 - 36.7) < 0.1
    assert abs(count("lang_encoder.transformer.blocks.0.decoder_layer.") / 1e6 - 50.3) < 0.1
    assert abs(count("extra_exit.") / 1e6 - 40.95) < 0.1


def test_synthetic_state_is_order_independent_and_bf16_roundable():
    cfg = deer_tiny()
    a = syn.make_synthetic_state(cfg, 5)
    b = syn.make_synthetic_state(cfg, 5)
    assert all(torch.equal(a[k], b[k]) for k in a)
    r = syn.round_state_to_bf16(cfg, a)
    k = "perceiver.layers.0.0.to_q.weight"
    assert torch.equal(r[k], a[k].bfloat16().float()) and not torch.equal(r[k], a[k])
    assert torch.equal(r["perceiver.norm.weight"], a["perceiver.norm.weight"])      # LN params stay fp32
    c = syn.make_synthetic_state(cfg, 5, bf16_round=True)
    assert all(torch.equal(c[x], r[x]) for x in r)
    rgb, grip, ids, mask = syn.synthetic_step_inputs(cfg, 3)
    assert ids.shape == (1, 14) and int(ids[0, 0]) == cfg.media_token_id and int(ids[0, -2]) == cfg.eoc_token_id
    assert rgb.shape == (1, 1, 1, 3, cfg.image_size, cfg.image_size)


def test_factory_surface_and_state_dict_contract_on_cpu():
    """create_model_and_transforms keeps the reference signature/return triple; load_state_dict understands the DDP
    `module.` prefix and the gated_cross_attn_layers alias; forward without a GPU fails loudly."""
    import inspect
    import numpy as np
    import pytest
    from deer_vla_amd import factory, _abi
    from deer_vla_amd.flamingo_mpt import MPTFlamingo
    sig = inspect.signature(factory.create_model_and_transforms)
    for name in ("clip_vision_encoder_path", "clip_vision_encoder_pretrained", "lang_encoder_path", "tokenizer_path",
                 "cross_attn_every_n_layers", "use_local_files", "decoder_layers_attr_name", "window_size", "use_gripper",
                 "fusion_mode", "llm_name", "pooling", "decoder_type", "head_type"):
        assert name in sig.parameters, name
    cfg = deer_tiny()
    sd = syn.make_synthetic_state(cfg, 2)
    model, image_processor, tok = factory.create_model_and_transforms(
        "ViT-L-14", "openai", "", "", window_size=12, use_gripper=True, fusion_mode="post", llm_name="mpt_dolly_3b",
        state_dict=sd, cfg=cfg)
    assert model.module is model and model.get_all_exit_idx() == cfg.exit_ids()
    assert model.lang_encoder.config.n_layers == cfg.n_layers and model.lang_encoder.config.d_model == cfg.d_model
    # construction-time mixin API of the reference (flamingo_lm.py:136-202, factory.py:139-159) is accepted on the finished model
    le = model.lang_encoder
    le.init_flamingo(media_token_id=cfg.media_token_id, lang_hidden_size=cfg.d_model, vis_hidden_size=cfg.vit_width,
                     cross_attn_every_n_layers=cfg.cross_attn_every_n_layers, gradient_checkpointing=False)
    le._delete_decoder_layers(list(range(cfg.n_layers, cfg.n_layers + 4)))
    assert len(le._get_decoder_layers()) == cfg.n_layers == len(le.gated_cross_attn_layers) == len(le.old_decoder_blocks)
    assert le.initialized_flamingo and not le.is_conditioned()
    with pytest.raises(NotImplementedError):
        le._delete_decoder_layers([0])
    with pytest.raises(NotImplementedError):
        le.init_flamingo(media_token_id=cfg.media_token_id + 1)
    # DeeR ckpt style keys
    k = "lang_encoder.transformer.blocks.1.gated_cross_attn_layer.attn.to_q.weight"
    new = torch.full_like(sd[k], 0.25)
    missing, unexpected = model.load_state_dict({"module.lang_encoder.gated_cross_attn_layers.1.attn.to_q.weight": new,
                                                 "module.something.else": torch.zeros(1)}, strict=False)
    assert not missing and unexpected == ["module.something.else"]
    assert torch.equal(model.state_dict()[k], new)
    with pytest.raises(RuntimeError):
        model.load_state_dict({k: torch.zeros(3, 3)}, strict=False)
    # preprocessing (data.py:898-919): image -> (3,224,224) CLIP-normalised; text -> "<image>{instr}<|endofchunk|>{eos}"
    x = image_processor(np.full((200, 200, 3), 128, dtype=np.uint8))
    assert x.shape == (3, cfg.image_size, cfg.image_size) or x.shape == (3, 224, 224)
    t = tok([f"<image>push the block<|endofchunk|>{tok.eos_token}"], max_length=32, padding="longest", truncation="only_first",
            return_tensors="pt")
    ids = t["input_ids"]
    assert int(ids[0, 0]) == cfg.media_token_id and int(ids[0, -2]) == cfg.eoc_token_id and int(t["attention_mask"].sum()) == ids.shape[1]
    if not torch.cuda.is_available():
        with pytest.raises(_abi.DeerHipError):
            model(torch.zeros(1, 1, 1, 3, 56, 56), ids, t["attention_mask"], vision_gripper=torch.zeros(1, 1, 1, 3, 56, 56), exit_id=1)


def test_factory_fails_loudly_on_keywords_it_does_not_implement():
    """keywords of the reference factory (factory.py:53-91) that change the model's arithmetic and are not built
    here raise before anything is constructed (no GPU needed) instead of being accepted and ignored; keywords the reference's
    MPTFlamingo never reads (no_image_patch, global_latent: flamingo_mpt.py:55-56 only) and training-only ones stay accepted."""
    import pytest
    from deer_vla_amd.factory import create_model_and_transforms
    base = dict(clip_vision_encoder_path="ViT-L-14", clip_vision_encoder_pretrained="openai", lang_encoder_path="", tokenizer_path="",
                use_gripper=True, fusion_mode="post", llm_name="mpt_dolly_3b", device="cpu")
    for kw in (dict(multi_step_action=3), dict(last_action=True), dict(fwd_pred=True), dict(fwd_pred_hand=True), dict(residual=True),
               dict(pad_length=12), dict(refresh=2), dict(return_feature=True), dict(layerwise_exit_eval=True), dict(use_hist=True),
               dict(use_diff=True), dict(share_exit=True), dict(hidden_size=512), dict(multi_exit=False), dict(decoder_type="gpt"),
               dict(head_type="diffusion"), dict(llm_name="llama_9b"), dict(clip_vision_encoder_path="ViT-B-32")):
        with pytest.raises(NotImplementedError):
            create_model_and_transforms(**{**base, **kw})
