"""Env-batch ViT residual projections (out_proj 1024x1024, c_proj 1024x4096 at M = 257 x frames): the shipped pair
  split-K GEMM into f32 slabs  +  deer_resadd_ln (slab sum + bias + residual + LayerNorm)
against
  one full-K GEMM whose epilogue adds into the f32 residual stream (DEER_EPI_RESADD_F32)  +  a LayerNorm-only pass,
per candidate tile, on cold weights (8 weight copies), graph-replayed pairs.  usage: bench_resadd_direct.py [frames=16]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deer_vla_amd import _abi as abi
lib = abi.lib()
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 16
M, N = 257 * frames, 1024
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
dev = "cuda"
x = torch.randn(M, N, device=dev)
gamma, beta, bias = torch.randn(N, device=dev), torch.randn(N, device=dev), torch.randn(N, device=dev)
out_bf = torch.zeros(M, N, device=dev, dtype=torch.bfloat16)
NW = 8


def time_graph(fn, reps=6):
    fn(0); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for r in range(NW):
            fn(r)
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * NW) * 1e3


for name, K, splitk, stile in (("out_proj", 1024, 1, 0), ("c_proj", 4096, 2, 0)):
    A = torch.randn(M, K, device=dev).bfloat16()
    Ws = [(torch.randn(N, K, device=dev) * K ** -0.5).bfloat16() for _ in range(NW)]
    slab = torch.zeros(max(splitk, 1), M, N, device=dev)

    def shipped(r):
        abi.check(lib.deer_gemm_bf16_nt_splitk(abi.ptr(A), K, abi.ptr(Ws[r]), K, abi.ptr(slab), M, N, K, splitk, stile, None, st()), "splitk")
        abi.check(lib.deer_resadd_ln(abi.ptr(x), abi.ptr(slab), splitk, M * N, None, abi.ptr(bias), abi.ptr(gamma), abi.ptr(beta), abi.ptr(out_bf), None, None,
                                     M, N, 1e-5, None, st()), "resadd_ln")

    def gemm_only(r):
        abi.check(lib.deer_gemm_bf16_nt_splitk(abi.ptr(A), K, abi.ptr(Ws[r]), K, abi.ptr(slab), M, N, K, splitk, stile, None, st()), "splitk")

    t_pair, t_gemm = time_graph(shipped), time_graph(gemm_only)
    print(f"{name} M={M} K={K}: shipped split-K x{splitk} GEMM {t_gemm:.1f} us, GEMM + resadd_ln {t_pair:.1f} us", flush=True)
    for tile in (0, 3, 66, 67, 68, 69, 71):
        def direct(r, tile=tile):
            abi.check(lib.deer_gemm_bf16_nt(abi.ptr(A), K, 0, abi.ptr(Ws[r]), K, abi.ptr(bias), abi.ptr(x), N, 0, M, N, K, 1, abi.EPI_RESADD_F32, None, tile, None, st()),
                      "gemm resadd")
            abi.check(lib.deer_resadd_ln(abi.ptr(x), None, 0, 0, None, None, abi.ptr(gamma), abi.ptr(beta), abi.ptr(out_bf), None, None, M, N, 1e-5, None, st()), "ln")

        def direct_gemm(r, tile=tile):
            abi.check(lib.deer_gemm_bf16_nt(abi.ptr(A), K, 0, abi.ptr(Ws[r]), K, abi.ptr(bias), abi.ptr(x), N, 0, M, N, K, 1, abi.EPI_RESADD_F32, None, tile, None, st()),
                      "gemm resadd")
        try:
            tg = time_graph(direct_gemm)
            tp = time_graph(direct)
            print(f"   direct tile {tile:3d}: GEMM {tg:.1f} us, GEMM + LN-only pass {tp:.1f} us", flush=True)
        except Exception as e:
            print(f"   direct tile {tile}: {e}", flush=True)
        x.normal_()
