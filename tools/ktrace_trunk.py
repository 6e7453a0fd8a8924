"""Where the time goes INSIDE the two row kernels of a one-environment trunk layer: deer_xattn_fused and
deer_trunk_mpt_attn with phase time stamps (csrc/common.h: KT; `make -C deer_vla_amd/csrc ktrace` builds lib/libdeer_hip_ktrace.so - the
product library compiles the stamps to nothing).  Full 3B size, full-depth steps replayed as the engine's graph pieces; the stamps of the
LAST layer of each step (thread 0 of the first workgroup; 100 MHz counter = 10 ns) are averaged over the steps.
usage: python tools/ktrace_trunk.py [steps]"""
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
for name in ("xattn", "mpt_attn"):
    fn = getattr(lib, "deer_ktrace_set_" + name)
    fn.argtypes = [ctypes.c_void_p]
    assert fn(ctypes.c_void_p(buf.data_ptr())) == 0
NAMES = {0: "start", 1: "requests issued (Wo, K/V)", 2: "q projection done (Wq arrived, 64 MFMAs)", 3: "partials + K/V in LDS, barrier",
         4: "q reduced, barrier", 5: "attention done (wave 0)", 6: "barrier", 7: "output projection stored",
         8: "start", 9: "q/k LayerNorm moments, barrier", 10: "q k v in LDS, barrier", 11: "scores, barrier", 12: "softmax, barrier", 13: "P V stored"}
acc = torch.zeros(64, dtype=torch.float64)
for s in range(n + 2):
    rgb, grip, ids, mask = syn.synthetic_step_inputs(cfg, s)
    eng.step(rgb, grip, ids, mask, exit_id=11, use_graph=True)
    torch.cuda.synchronize()
    t = buf.cpu().double()
    if s >= 2:
        acc[:8] += (t[:8] - t[0]) / 100.0
        acc[8:14] += (t[8:14] - t[8]) / 100.0
acc /= n
for base, title in ((0, "deer_xattn_fused (packed, one environment)"), (8, "deer_trunk_mpt_attn")):
    print(title)
    prev = 0.0
    for i in range(base, base + (8 if base == 0 else 6)):
        print(f"   {NAMES[i]:48s} at {acc[i]:6.2f} us   (+{acc[i] - prev:5.2f})")
        prev = float(acc[i])
