"""What per-head split-K slabs would cost the ViT (attention + out_proj as ONE kernel writing 16 per-head partial
products of [rows, 1024] f32, reduced by deer_resadd_ln).  Measures the CONSUMER side only: deer_resadd_ln at s_in = 1 (today) against
s_in = 16 at 257 / 514 / 4112 rows, graph replay, 24 different slab buffers (cold, like the step).  The fused kernel would have to win
back this difference plus the 16 x larger write of its own epilogue before the saved launch counts."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deer_vla_amd import _abi as abi

lib = abi.lib()
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
W, NCOPY = 1024, 24
for M in (257, 514, 4112):
    x = torch.randn(M, W, device="cuda")
    g, b, bias = torch.ones(W, device="cuda"), torch.zeros(W, device="cuda"), torch.randn(W, device="cuda")
    out = torch.empty(M, W, device="cuda", dtype=torch.bfloat16)
    line = f"rows {M:5d}:"
    for S in (1, 2, 4, 8, 16):
        slabs = [torch.randn(S, M, W, device="cuda") for _ in range(NCOPY if M < 4000 else 6)]

        def run():
            for s in slabs:
                abi.check(lib.deer_resadd_ln(abi.ptr(x), abi.ptr(s), S, M * W, None, abi.ptr(bias), abi.ptr(g), abi.ptr(b), abi.ptr(out), None, None,
                                             M, W, 1e-5, None, st()), "resadd")
        run()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            run()
        gr.replay()
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); gr.replay(); e1.record()
            torch.cuda.synchronize()
            ts.append(1e3 * e0.elapsed_time(e1) / len(slabs))
        line += f"  s_in={S:2d} {sorted(ts)[2]:6.2f} us"
    print(line, flush=True)
