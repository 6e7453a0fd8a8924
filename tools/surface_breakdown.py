"""Where the drop-in surface's time goes beyond DeerEngine.step: wall-clock per piece of ModelWrapper.step (tiny sync'd sections; the
sum is an upper bound because syncs remove overlap).  usage: surface_breakdown.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from deer_vla_amd import synthetic as syn, rollout as ro
from deer_vla_amd.config import deer_3b
from deer_vla_amd.factory import create_model_and_transforms, GpuImageProcessor
from deer_vla_amd.value_net import ActionValueNet, ExitController

cfg = deer_3b(max_layer=12)
sd = syn.make_synthetic_state(cfg, 0, std="0.02", bf16_round=True)
model, _, tok = create_model_and_transforms("ViT-L-14", "openai", "", "", window_size=12, use_gripper=True, fusion_mode="post",
                                            llm_name="mpt_dolly_3b", state_dict=sd, cfg=cfg)
proc = GpuImageProcessor(cfg.image_size)
vn = ActionValueNet(model.get_all_exit_idx(), None, cfg.exit_interval, 12, "L2")
ctl = ExitController(vn, model.get_all_exit_idx(), max_layer=12)
ctl._set_threshold_value([0.02] * (ctl.real_num_exit - 1) + [1e5])
w = ro.ModelWrapper(model, tok, proc, torch.bfloat16, early_exit=True, exit_controller=ctl)
rng = np.random.default_rng(5)
obs = [{"rgb_obs": {"rgb_static": rng.integers(0, 255, (200, 200, 3), dtype=np.uint8), "rgb_gripper": rng.integers(0, 255, (84, 84, 3), dtype=np.uint8)},
        "robot_obs": np.zeros(15, np.float32)} for _ in range(8)]
goal = "lift the red block from the sliding cabinet"
for i in range(30):
    ctl.set_timestep(i); w.step(obs[i % 8], goal)
torch.cuda.synchronize()
N = 100
def T(fn):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(N): fn(i)
    torch.cuda.synchronize(); return 1e6 * (time.perf_counter() - t0) / N
t_step = T(lambda i: (ctl.set_timestep(i), w.step(obs[i % 8], goal)))
t_pre_s = T(lambda i: w.image_process_fn([obs[i % 8]["rgb_obs"]["rgb_static"]]))
t_pre_g = T(lambda i: w.image_process_fn([obs[i % 8]["rgb_obs"]["rgb_gripper"]]))
img = w.image_process_fn([obs[0]["rgb_obs"]["rgb_static"]]).unsqueeze(1).unsqueeze(1)
grp = w.image_process_fn([obs[0]["rgb_obs"]["rgb_gripper"]]).unsqueeze(1).unsqueeze(1)
ids, mask = ro.preprocess_text_calvin([goal], tok); ids, mask = ids.cuda(), mask.cuda()
t_fwd = T(lambda i: (ctl.set_timestep(i), model(vision_x=img, lang_x=ids, attention_mask=mask, vision_gripper=grp, dynamic_early_exit=True, exit_controller=ctl)))
e = model.engine
model._sync_controller(ctl)
t_eng = T(lambda i: e.step(img, grp, ids, None))
print(f"ModelWrapper.step {t_step:.0f} us | preprocess static {t_pre_s:.0f} us, gripper {t_pre_g:.0f} us | MPTFlamingo.forward {t_fwd:.0f} us | engine.step {t_eng:.0f} us")
