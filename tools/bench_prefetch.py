"""Would prefetching the NEXT GEMM's weights (HBM -> Infinity Cache) shorten the one-environment vision-tower GEMMs?  At one
environment the step uses ~8 % of the HBM bandwidth and every ViT weight is cold (600 MB per step through a 256 MB memory-side
cache).  Upper bound of what a prefetcher could give: the same GEMMs over weight copies that cycle through COLD_MB megabytes -
400 (default): every copy comes from HBM like in the step; 96: the copies stay resident in the Infinity Cache.
Measured (profiles/r03_o_gemm_weights_cold_vs_cache_resident.txt): 0 - 1.8 us per launch (in_proj at 514 rows 11.0 -> 9.3, c_fc at
257 rows 10.0 -> 8.5, the others within 0.3 us) - and a fork / join edge per GEMM in a HIP graph costs +11 us per iteration (a side-stream
read of the next copy behind an event: 11.0 -> 21.4 us), so a paced prefetch stream was not built.
usage: COLD_MB=400|96 bench_prefetch.py [rows=514]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deer_vla_amd import _abi as abi

lib = abi.lib()
M = int(sys.argv[1]) if len(sys.argv) > 1 else 514
COLD_MB = float(os.environ.get("COLD_MB", "400"))
SHAPES = [("vit qkv", 3072, 1024), ("vit out", 1024, 1024), ("vit fc1", 4096, 1024), ("vit fc2", 1024, 4096)]
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)

for name, N, K in SHAPES:
    ncopy = max(4, int(COLD_MB * 1e6 / (N * K * 2)) + 1)
    A = torch.randn(M, K, device="cuda").bfloat16()
    Ws = [torch.randn(N, K, device="cuda").bfloat16() * K ** -0.5 for _ in range(ncopy)]
    C = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)

    def run():
        for w in Ws:
            abi.check(lib.deer_gemm_bf16_nt(abi.ptr(A), K, 0, abi.ptr(w), K, None, abi.ptr(C), N, 0, M, N, K, 1, abi.EPI_BF16, None, 0, None, st()), "gemm")
    run()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        run()
    g.replay(); torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        ts.append(1e3 * e0.elapsed_time(e1) / ncopy)
    print(f"{name:8s} M={M} N={N} K={K}  {ncopy} weight copies cycling through {ncopy * N * K * 2 / 1e6:.0f} MB: {sorted(ts)[3]:6.2f} us per launch", flush=True)
