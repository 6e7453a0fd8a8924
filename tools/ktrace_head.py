"""Phase time stamps inside the head evaluation of a one-environment step (csrc/head.hip: head_lstm_layer_kernel - the LAST of the four
layers - and head_final_kernel), `make -C deer_vla_amd/csrc ktrace` build (see tools/ktrace_trunk.py).  Full 3B size; dynamic steps replayed
as the engine's graph pieces, thresholds at -1 (no check fires: every check does all of its work), stamps of the last evaluation of a step
averaged over the steps.  usage: python tools/ktrace_head.py [steps]"""
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
eng.configure_exit(cfg.exit_ids(), 12, 1)
eng.set_thresholds([-1.0] * 5 + [1e5])
eng.reset()
lib = ctypes.CDLL(abi.LIB_PATH)
buf = torch.zeros(64, dtype=torch.int64, device="cuda")
fn = lib.deer_ktrace_set_head
fn.argtypes = [ctypes.c_void_p]
assert fn(ctypes.c_void_p(buf.data_ptr())) == 0
F = ["start (skip test done)", "control block read, output rows requested", "two LayerNorms staged, barrier", "7 dot products, barrier",
     "tanh / sigmoid, action_dbg", "delta, decision", "commit: control block, host mirror, flags", "check_done (host release)", "barrier",
     "LSTM state committed"]
L = ["start", "weights requested (4 + 2 k-steps)", "activations staged (LayerNorm), barrier", "dot products (weights arrived)", "gates, state stored"]
acc = torch.zeros(64, dtype=torch.float64)
for s in range(n + 2):
    rgb, grip, ids, mask = syn.synthetic_step_inputs(cfg, s)
    eng.step(rgb, grip, ids, mask, use_graph=True)
    torch.cuda.synchronize()
    t = buf.cpu().double()
    if s >= 2:
        acc[:10] += (t[:10] - t[0]) / 100.0
        acc[16:21] += (t[16:21] - t[16]) / 100.0
acc /= n
for base, names, title in ((16, L, "head_lstm_layer_kernel (layer 3 of the forced last check)"), (0, F, "head_final_kernel (forced exit at layer 11: commit path)")):
    print(title)
    prev = 0.0
    for i, nm in enumerate(names):
        print(f"   {nm:52s} at {acc[base + i]:6.2f} us   (+{acc[base + i] - prev:5.2f})")
        prev = float(acc[base + i])
