"""Run one deer_gemm_bf16_nt shape / tile repeatedly (target of rocprofv3 --pmc passes).  usage: gemm_one.py M N K tile [reps]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deer_vla_amd import _abi as abi
lib = abi.lib()
M, N, K, tile = (int(v) for v in sys.argv[1:5])
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 20
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
A = torch.randn(M, K, device="cuda").bfloat16()
Ws = [torch.randn(N, K, device="cuda").bfloat16() * K ** -0.5 for _ in range(8)]
C = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
for r in range(reps):
    lib.deer_gemm_bf16_nt(abi.ptr(A), K, 0, abi.ptr(Ws[r % 8]), K, None, abi.ptr(C), N, 0, M, N, K, 1, abi.EPI_BF16, None, tile, None, st())
torch.cuda.synchronize()
