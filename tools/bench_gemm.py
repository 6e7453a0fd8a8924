"""GPU microbenchmark of deer_gemm_bf16_nt over the ViT / Perceiver shapes of one control step (cold weights:
each repetition uses a different weight copy so that W comes from HBM like in the real step)."""
import ctypes, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deer_vla_amd import _abi as abi

lib = abi.lib()
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
import sys as _s
MB = int(_s.argv[1]) if len(_s.argv) > 1 else 1
SHAPES = [("vit qkv", 514, 3072, 1024), ("vit out", 514, 1024, 1024), ("vit fc1", 514, 4096, 1024), ("vit fc2", 514, 1024, 4096),
          ("patch", 512, 1024, 640), ("perc kv", 640, 1024, 1024), ("perc q", 128, 512, 1024), ("perc ff1", 128, 4096, 1024),
          ("perc ff2", 128, 1024, 4096), ("media kv", 128, 12288, 1024)]
NCOPY = 24
# EPI=qgelu|f32: the epilogue of the launches (default: plain 16-bit); BIAS=1 adds the bias vector; FMT=f16: fp16 operands
EPI = {"bf16": abi.EPI_BF16, "qgelu": abi.EPI_QGELU_BF16, "f32": abi.EPI_F32}[os.environ.get("EPI", "bf16")]
USE_BIAS = os.environ.get("BIAS", "0") == "1"
F16 = os.environ.get("FMT", "bf16") == "f16"
gemm = lib.deer_gemm_f16_nt if F16 else lib.deer_gemm_bf16_nt
tdt = torch.float16 if F16 else torch.bfloat16
TILES = tuple(int(t) for t in _s.argv[2].split(',')) if len(_s.argv) > 2 else (4, 5, 7, 8, 10, 0)
for name, M, N, K in SHAPES:
    M = MB if MB > 16 else M * MB                      # argv[1] > 16: absolute row count (257 = one camera frame)
    PAD = int(os.environ.get("PAD", "0"))                 # extra bf16 elements per row (row pitch K + PAD): L2 channel spread
    LD = K + PAD
    A = torch.randn(M, LD, device="cuda").to(tdt)
    Ws = [(torch.randn(N, LD, device="cuda") * K ** -0.5).to(tdt) for _ in range(NCOPY)]
    bias = torch.randn(N, device="cuda") * 0.1 if USE_BIAS else None
    C = torch.zeros(M, N, device="cuda", dtype=torch.float32 if EPI == abi.EPI_F32 else tdt)
    ref = (A[:, :K].float() @ Ws[0][:, :K].float().t())
    line = f"{name:9s} M={M:4d} N={N:5d} K={K:4d} |"
    for tile in TILES:
        rc = gemm(abi.ptr(A), LD, 0, abi.ptr(Ws[0]), LD, abi.ptr(bias), abi.ptr(C), N, 0, M, N, K, 1, EPI, None, tile, None, st())
        if rc != 0:
            line += f" t{tile}:  n/a "
            continue
        torch.cuda.synchronize()
        err = float((C.float() - ref).norm() / ref.norm())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):                      # graph replay: no host launch cost in the measurement
            for w in Ws:
                gemm(abi.ptr(A), LD, 0, abi.ptr(w), LD, abi.ptr(bias), abi.ptr(C), N, 0, M, N, K, 1, EPI, None, tile, None, st())
        g.replay()
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / NCOPY
        line += f" t{tile}:{us:6.1f}us" + ("" if err < 5e-3 else f"(ERR {err:.1e})")
    print(line + f" | {2.0 * M * N * K / 1e6:8.0f} MF", flush=True)
