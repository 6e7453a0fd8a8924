"""GPU microbenchmark of deer_gemm_skinny over the LLM shapes (cold weights: rotating copies > L2+MALL)."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deer_vla_amd import _abi as abi

lib = abi.lib()
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
SHAPES = [("xa q", 512, 2048), ("xa out", 2048, 512), ("xa ff1", 8192, 2048), ("xa ff2", 2048, 8192), ("qkv", 6144, 2048),
          ("out", 2048, 2048), ("up", 8192, 2048), ("down", 2048, 8192)]
import sys as _s
T = int(_s.argv[1]) if len(_s.argv) > 1 else 14
MP = abi.skinny_mpad(T)
print('rows', T)
for name, N, K in SHAPES:
    ncopy = max(4, int(600e6 / (N * K * 2)))
    ncopy = min(ncopy, 64)
    hot = int(os.environ.get("SKINNY_HOT", "0"))          # > 0: the launches cycle through only `hot` weight copies, which then stay in the
                                                          # Infinity Cache (256 MB) - what prefetching the next layer's weights would buy
    A = torch.randn(T, K, device="cuda")
    Ws = []
    for _ in range(ncopy):
        w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
        wp = torch.empty_like(w)
        lib.deer_pack_weight_mfma16(abi.ptr(w), abi.ptr(wp), N, K, st())
        Ws.append(wp)
    torch.cuda.synchronize()
    line = f"{name:7s} N={N:5d} K={K:5d} {N*K*2/1e6:5.1f}MB |"
    S0 = lib.deer_skinny_splitk(T, N, K)
    for S in sorted(set([1, 2, 4, 8, 16, 32, S0])):
        if K % (S * 32) or (K // S) < 128:
            continue
        part = torch.zeros(S, MP, N, device="cuda")
        def run(w):
            return lib.deer_gemm_skinny(abi.ptr(A), K, None, 0, 0, abi.A_F32, abi.ptr(w), abi.ptr(part), T, N, K, S, None, st())
        if run(Ws[0]) != 0:
            continue
        seq = [Ws[i % hot] for i in range(ncopy)] if hot else Ws
        for w in seq: run(w)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):                      # graph replay: no host launch cost in the measurement
            for w in seq: run(w)
        g.replay()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / ncopy
        line += f" S{S}{'*' if S == S0 else ''}:{us:6.1f}us {N*K*2/us/1e6:5.2f}TB/s"
    print(line, flush=True)
