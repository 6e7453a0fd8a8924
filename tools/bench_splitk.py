"""Split-K / tile sweep for the two N = 1024 projections of the ViT at env-batch row counts (out_proj K = 1024, c_proj K = 4096):
graph replay over 24 cold weight copies, f32 slab output (what deer_resadd_ln consumes).  usage: bench_splitk.py M"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deer_vla_amd import _abi as abi

lib = abi.lib()
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 2056
NCOPY = 24


def timed(fn):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay()
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        ts.append(1e3 * e0.elapsed_time(e1) / NCOPY)
    return sorted(ts)[len(ts) // 2]


for name, N, K in (("out_proj", 1024, 1024), ("c_proj", 1024, 4096)):
    A = torch.randn(M, K, device="cuda").bfloat16()
    Ws = [torch.randn(N, K, device="cuda").bfloat16() * K ** -0.5 for _ in range(NCOPY)]
    slab = torch.zeros(8 * M * N, device="cuda")
    flops = 2.0 * M * N * K
    line = f"{name} M={M} N={N} K={K}:"
    for S in (1, 2, 4):
        for tile in (0, 17, 8, 39, 4, 5):
            def ours():
                for w in Ws:
                    lib.deer_gemm_bf16_nt_splitk(abi.ptr(A), K, abi.ptr(w), K, abi.ptr(slab), M, N, K, S, tile, None, st())
            if lib.deer_gemm_bf16_nt_splitk(abi.ptr(A), K, abi.ptr(Ws[0]), K, abi.ptr(slab), M, N, K, S, tile, None, st()) != 0:
                continue
            t = timed(ours)
            line += f" S{S}/t{tile} {t:5.1f}us"
    print(line, flush=True)
