"""Phase time stamps inside the four ViT projections of a one-environment step (csrc/gemm_tiled.hip: gemm_tiled_ring_kernel, workgroup
(0, 0, 0) of the LAST ViT block's launches), `make -C deer_vla_amd/csrc ktrace` build (see tools/ktrace_trunk.py).  Full 3B size, the product
schedule (two per-frame chains, M = 257).  usage: python tools/ktrace_gemm.py [steps]"""
import ctypes, os, sys
os.environ.setdefault("DEER_HIP_LIB", "libdeer_hip_ktrace.so")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deer_vla_amd import synthetic as syn, _abi as abi
from deer_vla_amd.config import deer_3b
from deer_vla_amd.engine import DeerEngine

n = int(sys.argv[1]) if len(sys.argv) > 1 else 12
cfg = deer_3b(max_layer=12)
sd = syn.make_synthetic_state(cfg, seed=0, std="0.02", bf16_round=True)
eng = DeerEngine(cfg, sd)
lib = ctypes.CDLL(abi.LIB_PATH)
buf = torch.zeros(64, dtype=torch.int64, device="cuda")
fn = lib.deer_ktrace_set_gemm
fn.argtypes = [ctypes.c_void_p]
assert fn(ctypes.c_void_p(buf.data_ptr())) == 0
P = ["start", "ring prologue issued", "first barrier (first K-steps landed)", "second barrier (one loop body)", "K loop done", "epilogue stored"]
acc = torch.zeros(64, dtype=torch.float64)
for s in range(n + 2):
    rgb, grip, ids, mask = syn.synthetic_step_inputs(cfg, s)
    eng.step(rgb, grip, ids, mask, exit_id=1, use_graph=True)
    torch.cuda.synchronize()
    t = buf.cpu().double()
    if s >= 2:
        for base in (0, 8, 16, 24):
            acc[base:base + 6] += (t[base:base + 6] - t[base]) / 100.0
acc /= n
for base, title in ((0, "in_proj  (M 257, N 3072, K 1024)"), (8, "out_proj (N 1024, K 1024, split-K)"), (16, "c_fc     (N 4096, K 1024)"), (24, "c_proj   (N 1024, K 4096, split-K)")):
    print(title)
    prev = 0.0
    for i, nm in enumerate(P):
        print(f"   {nm:42s} at {acc[base + i]:6.2f} us   (+{acc[base + i] - prev:5.2f})")
        prev = float(acc[base + i])
