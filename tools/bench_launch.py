"""Measures the dependent-kernel boundary cost under graph replay on this box (trivial kernels, and early-exit-skipped GEMMs)."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deer_vla_amd import _abi as abi
lib = abi.lib()
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
ctl = torch.zeros(64, dtype=torch.int32, device="cuda")

def timeit(fn, n=200, label=""):
    fn(); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(n): fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    print(f"{label:50s} {1e3*e0.elapsed_time(e1)/n:7.2f} us/launch")

timeit(lambda: lib.deer_ctl_begin_step(abi.ptr(ctl), None, 1, st()), label="trivial kernel (1 block x 64 threads)")
x = torch.zeros(14, 2048, device="cuda"); o = torch.zeros(14, 2048, device="cuda")
g_ = torch.ones(2048, device="cuda")
timeit(lambda: lib.deer_resadd_ln(abi.ptr(x), None, 0, 0, None, None, abi.ptr(g_), None, None, abi.ptr(o), None, 14, 2048, 1e-5, None, st()), label="resadd_ln 14x2048 (no slabs)")
A = torch.randn(14, 2048, device="cuda"); W = torch.randn(512, 2048, device="cuda").bfloat16(); Wp = torch.empty_like(W)
lib.deer_pack_weight_mfma16(abi.ptr(W), abi.ptr(Wp), 512, 2048, st())
part = torch.zeros(8, 16, 512, device="cuda")
timeit(lambda: lib.deer_gemm_skinny(abi.ptr(A), 2048, None, 0, 0, 3, abi.ptr(Wp), abi.ptr(part), 14, 512, 2048, 8, None, st()), label="skinny 512x2048 (2 MB, L2-warm, same W)")
ctl[7] = 1
timeit(lambda: lib.deer_gemm_skinny(abi.ptr(A), 2048, None, 0, 0, 3, abi.ptr(Wp), abi.ptr(part), 14, 512, 2048, 8, abi.ptr(ctl), st()), label="skinny 512x2048 skipped by exit flag (64 blocks)")
W2 = torch.randn(8192, 2048, device="cuda").bfloat16(); part2 = torch.zeros(4, 16, 8192, device="cuda")
timeit(lambda: lib.deer_gemm_skinny(abi.ptr(A), 2048, None, 0, 0, 3, abi.ptr(W2), abi.ptr(part2), 14, 8192, 2048, 4, abi.ptr(ctl), st()), label="skinny 8192x2048 skipped by exit flag (512 blocks)")
Ab = torch.randn(128, 1024, device="cuda").bfloat16(); Wb = torch.randn(512, 1024, device="cuda").bfloat16(); C = torch.zeros(128, 512, device="cuda", dtype=torch.bfloat16)
timeit(lambda: lib.deer_gemm_bf16_nt(abi.ptr(Ab), 1024, 0, abi.ptr(Wb), 1024, None, abi.ptr(C), 512, 0, 128, 512, 1024, 1, 0, None, 4, None, st()), label="tiled 128x512x1024 t4 (L2-warm, same W)")
Ab = torch.randn(128, 64, device="cuda").bfloat16(); Wb = torch.randn(512, 64, device="cuda").bfloat16()
timeit(lambda: lib.deer_gemm_bf16_nt(abi.ptr(Ab), 64, 0, abi.ptr(Wb), 64, None, abi.ptr(C), 512, 0, 128, 512, 64, 1, 0, None, 4, None, st()), label="tiled 128x512x64 t4 (1 K-step)")
