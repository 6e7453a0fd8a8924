#!/usr/bin/env python3
"""Does v_mfma_f32_16x16x32_f16 keep SUBNORMAL fp16 inputs on gfx950?  (The trunk's activation lo plane a - fp16(a) is subnormal in fp16
for |a| < 0.125.)  A = c (constant), W = 1: C = K * c through deer_gemm_f16_nt (f32 accumulate, f32 out)."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deer_vla_amd import _abi as abi
lib = abi.lib()
st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
M, N, K = 64, 64, 64
for c in (1e-3, 6.2e-5, 3e-5, 1e-6, 6e-8):
    A = torch.full((M, K), c, device="cuda").to(torch.float16)
    W = torch.ones(N, K, device="cuda", dtype=torch.float16)
    C = torch.zeros(M, N, device="cuda")
    abi.check(lib.deer_gemm_f16_nt(abi.ptr(A), K, 0, abi.ptr(W), K, None, abi.ptr(C), N, 0, M, N, K, 1, abi.EPI_F32, None, 1, None, st), "gemm")
    torch.cuda.synchronize()
    print(f"a = {c:g} (fp16 {float(A[0,0]):.3e}, subnormal {abs(float(A[0,0])) < 6.1e-5}): C = {float(C[0,0]):.6e}, expected {K * float(A[0,0]):.6e}")
    # subnormal WEIGHT side too
    C.zero_()
    abi.check(lib.deer_gemm_f16_nt(abi.ptr(W[:M]), K, 0, abi.ptr(A[:N]), K, None, abi.ptr(C), N, 0, M, N, K, 1, abi.EPI_F32, None, 1, None, st), "gemm")
    torch.cuda.synchronize()
    print(f"   as the weight operand: C = {float(C[0,0]):.6e}")
