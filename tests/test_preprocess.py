"""On-GPU camera-frame preprocessing (csrc/preprocess.hip) against the host path the reference uses (PIL bicubic resize + center crop +
ToTensor + Normalize: robot_flamingo/data/data.py:898-902 with open_clip's eval transform).  One uint8 step is 0.0145 after
normalisation, so the resampler has to agree with PIL to the LAST BIT of the uint8 image for the 1e-2 gate to hold."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from deer_vla_amd.factory import ClipImageProcessor, GpuImageProcessor  # noqa: E402


@pytest.mark.parametrize("hw", [(200, 200), (84, 84), (120, 90), (90, 120), (300, 260), (256, 341), (224, 224)])
def test_gpu_preprocessing_matches_pil_bit_for_bit(hw):
    H, W = hw
    rng = np.random.default_rng(H * 1000 + W)
    frames = rng.integers(0, 256, size=(3, H, W, 3), dtype=np.uint8)
    frames[1, :, :, :] = (np.add.outer(np.arange(H), np.arange(W))[:, :, None] * np.array([1, 2, 3]) % 256).astype(np.uint8)   # smooth ramp
    host, gpu = ClipImageProcessor(224), GpuImageProcessor(224)
    ref = torch.stack([host(f) for f in frames])
    out = gpu(frames, out_f32=True).cpu()
    assert out.shape == ref.shape == (3, 3, 224, 224)
    assert float((out - ref).abs().max()) == 0.0, float((out - ref).abs().max())
    h16 = gpu(frames)                                                  # the engine's input format (fp16 by default): every pixel level distinct
    assert h16.dtype == torch.float16 and torch.equal(h16.cpu(), ref.to(torch.float16))
    bf = GpuImageProcessor(224, dtype=torch.bfloat16)(frames)
    assert bf.dtype == torch.bfloat16 and torch.equal(bf.cpu(), ref.to(torch.bfloat16))


def test_model_wrapper_takes_raw_frames_through_the_gpu_processor():
    """ModelWrapper with the on-device processor == ModelWrapper with the PIL processor (same actions, same exit layers)."""
    from deer_vla_amd import rollout as ro
    from deer_vla_amd import synthetic as syn
    from deer_vla_amd.config import deer_tiny
    from deer_vla_amd.factory import create_model_and_transforms
    cfg = deer_tiny()
    sd = syn.make_synthetic_state(cfg, 3, bf16_round=True)
    model, proc, tok = create_model_and_transforms("ViT-L-14", "openai", "", "", window_size=12, use_gripper=True, fusion_mode="post",
                                                   llm_name="mpt_dolly_3b", state_dict=sd, cfg=cfg)
    acts = []
    for p in (proc, GpuImageProcessor(cfg.image_size)):
        w = ro.ModelWrapper(model, tok, p, torch.float32, exit_id=3)      # cast_dtype of the README's `--precision fp32 --amp 1` evaluation (frames stay f32 / fp16)
        env = ro.SyntheticEnv(seed=5)
        obs = env.get_obs()
        a = []
        for _ in range(4):
            a.append(w.step(obs, "open the drawer")[0].copy())
            obs, _, _, _ = env.step(a[-1])
        acts.append(np.stack(a))
    assert np.array_equal(acts[0], acts[1])
