"""Phase time stamps inside the frame-tile GEMMs of an eight-environment step (csrc/gemm_bigm.hip: gemm_frame_kernel, workgroup 0 of the
last ViT block's launches; 16 frames = 4112 rows), `make -C deer_vla_amd/csrc ktrace` build (see tools/ktrace_trunk.py).
usage: python tools/ktrace_frame.py [steps] [n_envs]"""
import ctypes, os, sys
os.environ.setdefault("DEER_HIP_LIB", "libdeer_hip_ktrace.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deer_vla_amd import synthetic as syn, _abi as abi
from deer_vla_amd.config import deer_3b
from deer_vla_amd.engine import DeerEngine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
B = int(sys.argv[2]) if len(sys.argv) > 2 else 8
cfg = deer_3b(max_layer=12)
sd = syn.make_synthetic_state(cfg, seed=0, std="0.02", bf16_round=True)
eng = DeerEngine(cfg, sd, n_envs=B)
lib = ctypes.CDLL(abi.LIB_PATH)
buf = torch.zeros(64, dtype=torch.int64, device="cuda")
fn = lib.deer_ktrace_set_frame
fn.argtypes = [ctypes.c_void_p]
assert fn(ctypes.c_void_p(buf.data_ptr())) == 0
P = ["start", "ring prologue issued", "first barrier (first K-step landed)", "ninth barrier (8 K-steps of 32)", "K loop done", "C tile staged in LDS, barrier", "C tile stored"]
acc = torch.zeros(64, dtype=torch.float64)
for s in range(n + 2):
    ins = [syn.synthetic_step_inputs(cfg, 100 * e + s) for e in range(B)]
    rgb = torch.cat([i[0] for i in ins]); grip = torch.cat([i[1] for i in ins]); ids = torch.cat([i[2] for i in ins]); mask = torch.cat([i[3] for i in ins])
    eng.step(rgb, grip, ids, mask, exit_id=1, use_graph=True)
    torch.cuda.synchronize()
    t = buf.cpu().double()
    if s >= 2:
        for base in (0, 8, 16):
            acc[base:base + 7] += (t[base:base + 7] - t[base]) / 100.0
acc /= n
for base, title in ((0, "in_proj  (N 3072, K 1024)"), (8, "c_fc     (N 4096, K 1024)"), (16, "c_proj   (N 1024, K 4096 as two halves)")):
    print(title)
    prev = 0.0
    for i, nm in enumerate(P):
        print(f"   {nm:42s} at {acc[base + i]:6.2f} us   (+{acc[base + i] - prev:5.2f})")
        prev = float(acc[base + i])
