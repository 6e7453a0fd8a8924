import sys, os
sys.path.insert(0, os.getcwd())
import torch
from deer_vla_amd import ops, synthetic as syn
from deer_vla_amd.config import deer_tiny, deer_3b
from oracle import deer_oracle as orc
for cfg, kw in ((deer_tiny(), dict(seed=3)), (deer_3b(max_layer=12), dict(seed=0, std="0.02"))):
    sd = syn.make_synthetic_state(cfg, bf16_round=False, **kw)
    m = ops.NativeModel(cfg, sd, precision="fp32")
    rgb, grip, ids, mask = syn.synthetic_step_inputs(cfg, 0)
    S = cfg.image_size
    images = torch.stack([rgb.reshape(3, S, S), grip.reshape(3, S, S)]).cuda()
    tok = torch.ops.deer.vit_l14_encode(images, m.handle)
    torch.ops.deer.perceiver_resample(tok, m.handle)
    torch.cuda.synchronize()
    media = m.buffer("vis_x_f32").view(torch.float32).view(-1, cfg.vit_width).cpu()
    with torch.no_grad():
        t_o = torch.cat([orc.vit_visual_tokens(sd, cfg, rgb.reshape(1, 3, S, S)), orc.vit_visual_tokens(sd, cfg, grip.reshape(1, 3, S, S))])
        vis_o = orc.OracleDeer(sd, cfg).encode_vision(rgb, grip).reshape(-1, cfg.vit_width)
    print("tokens rel %.2e media rel %.2e" % (float((tok.cpu() - t_o).abs().max() / t_o.abs().max()), float((media - vis_o).abs().max() / vis_o.abs().max())))
    m.close()
