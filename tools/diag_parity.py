"""Diagnostic (GPU box): per-stage relative error of the HIP engine vs the CPU oracle."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deer_vla_amd import synthetic as syn
from deer_vla_amd.config import deer_tiny, deer_3b
from deer_vla_amd.engine import DeerEngine
from oracle import deer_oracle as orc

def rel(a, b): return float((a.double()-b.double()).norm()/b.double().norm())

def run(cfg, seed, std, exit_id):
    sd = syn.make_synthetic_state(cfg, seed, std=std, bf16_round=True)
    eng = DeerEngine(cfg, sd)
    rgb, grip, ids, mask = syn.synthetic_step_inputs(cfg, 0)
    m = orc.OracleDeer(sd, cfg); m.set_all_exit_window_size(1)
    # oracle with bf16-rounded inputs (the engine's img buffer is bf16)
    o = m.forward(rgb.bfloat16().float(), ids, mask, grip.bfloat16().float(), exit_id=exit_id)
    r = eng.step(rgb, grip, ids, mask, exit_id=exit_id, use_graph=False)
    T = ids.shape[1]
    print("vis_x rel", rel(eng.vis_x_f32.cpu(), o["vis_x"].reshape(cfg.n_media, cfg.vit_width)))
    for i in range(exit_id+1):
        print(f"hidden[{i}] rel", rel(eng.hidden[i,:T].cpu(), o["hidden_states"][i][0]))
    print("pose err", (r["pose"]-o["logits"][0].reshape(-1)).abs().tolist(), "grip err", abs(r["gripper"]-float(o["logits"][1])))
    # head alone on the ORACLE's hidden state
    eng.reset()
    eng.hidden[exit_id,:T].copy_(o["hidden_states"][exit_id][0])
    from deer_vla_amd import _abi as abi
    eng.ctl.zero_()
    eng.enqueue_head(exit_id, T, abi.KIND_COMMIT, use_ctl=False)
    torch.cuda.synchronize()
    a = eng.ctl.cpu().view(torch.float32)[abi.CTL_OUT_ACTION:abi.CTL_OUT_ACTION+7]
    ref = torch.cat([o["logits"][0].reshape(-1), o["logits"][1].reshape(-1)])
    print("head-only err", (a-ref).abs().max().item())
    # LLM alone from the ORACLE's vis_x
    eng.reset()
    eng.vis_x.copy_(o["vis_x"].reshape(cfg.n_media, cfg.vit_width))
    eng._gemm(eng.vis_x, eng.wkv_all, eng.kv_all, cfg.n_media, eng.n_xattn*2*eng.xinner, cfg.vit_width, abi.EPI_BF16)
    eng.load_inputs(rgb, grip, ids, mask)
    eng.enqueue_llm_static(T, False, exit_id)
    torch.cuda.synchronize()
    for i in range(exit_id+1):
        print(f"LLM-only hidden[{i}] rel", rel(eng.hidden[i,:T].cpu(), o["hidden_states"][i][0]))

if __name__ == "__main__":
    which = sys.argv[1] if len(sys.argv) > 1 else "tiny"
    if which == "tiny":
        run(deer_tiny(), 3, "fanin", 5)
    elif which == "full_fanin":                             # full size with fan-in scaled weights: O(1) activations everywhere
        run(deer_3b(12), 0, "fanin", 11)
    else:
        run(deer_3b(12), 0, "0.02", 11)
