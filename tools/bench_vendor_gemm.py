"""Same-box CEILING measurement (measurement only - nothing here is on the product path): hipBLASLt through torch.matmul
on the exact ViT / Perceiver GEMM shapes of one control step, under graph replay with cold weights (24 weight copies per
shape, like tools/bench_gemm.py), next to this repo's deer_gemm_bf16_nt on the same buffers.  VERDICT r1 item 2(a): the
"per-CU fill roofline" explanation of the small-M GEMMs has to be checked against what a tuned vendor kernel reaches.

usage: bench_vendor_gemm.py [rows-multiplier | absolute M]   ->  one line per shape: vendor us / TF/s, ours us / TF/s"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deer_vla_amd import _abi as abi

lib = abi.lib()
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
MB = int(sys.argv[1]) if len(sys.argv) > 1 else 1
TILES = tuple(int(t) for t in sys.argv[2].split(",")) if len(sys.argv) > 2 else (0,)
SHAPES = [("vit qkv", 514, 3072, 1024), ("vit out", 514, 1024, 1024), ("vit fc1", 514, 4096, 1024), ("vit fc2", 514, 1024, 4096),
          ("perc kv", 512, 1024, 1024), ("perc ff1", 128, 4096, 1024), ("perc ff2", 128, 1024, 4096), ("media kv", 128, 12288, 1024)]
NCOPY = 24
EPI = int(os.environ.get("EPI", abi.EPI_BF16))     # 2 = bias + QuickGELU (the c_fc epilogue)


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


for name, M, N, K in SHAPES:
    M = MB if MB > 16 else M * MB
    A = torch.randn(M, K, device="cuda").bfloat16()
    Ws = [torch.randn(N, K, device="cuda").bfloat16() * K ** -0.5 for _ in range(NCOPY)]
    C = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    flops = 2.0 * M * N * K
    bias_t = torch.randn(N, device="cuda")
    bias = abi.ptr(bias_t) if EPI != abi.EPI_BF16 else None

    def vendor():
        for w in Ws:
            torch.matmul(A, w.t(), out=C)

    line = f"{name:9s} M={M:5d} N={N:5d} K={K:4d} | hipBLASLt {0:6.1f}"
    v = timed(vendor)
    line = f"{name:9s} M={M:5d} N={N:5d} K={K:4d} | hipBLASLt {v:6.1f} us {flops / v / 1e6:6.0f} TF/s |"
    for tile in TILES:
        def ours():
            for w in Ws:
                lib.deer_gemm_bf16_nt(abi.ptr(A), K, 0, abi.ptr(w), K, bias, abi.ptr(C), N, 0, M, N, K, 1, EPI, None, tile,
                                      None, st())
        rc = lib.deer_gemm_bf16_nt(abi.ptr(A), K, 0, abi.ptr(Ws[0]), K, bias, abi.ptr(C), N, 0, M, N, K, 1, EPI, None, tile,
                                   None, st())
        if rc != 0:
            line += f" t{tile}: n/a |"
            continue
        o = timed(ours)
        line += f" deer t{tile} {o:6.1f} us {flops / o / 1e6:6.0f} TF/s |"
    if os.environ.get("SPLITK"):                      # the projections that close a residual branch run through the split-K entry (f32 slabs)
        S = int(os.environ["SPLITK"]) if K >= 4096 else 1
        slab = torch.zeros(S, M, N, device="cuda")
        for tile in TILES:
            def ours_sk():
                for w in Ws:
                    lib.deer_gemm_bf16_nt_splitk(abi.ptr(A), K, abi.ptr(w), K, abi.ptr(slab), M, N, K, S, tile, None, st())
            if lib.deer_gemm_bf16_nt_splitk(abi.ptr(A), K, abi.ptr(Ws[0]), K, abi.ptr(slab), M, N, K, S, tile, None, st()) != 0:
                continue
            o = timed(ours_sk)
            line += f" slabs{S} t{tile} {o:6.1f} |"
    print(line, flush=True)
