"""deer_gemm_skinny (f32 activation, split in the kernel) vs deer_gemm_skinny_hl (pre-split planes, LDS-DMA ring) over the trunk
shapes at env-batch row counts; cold weights (rotating copies > L2 + MALL), graph replay.  usage: bench_skinny_hl.py [rows ...]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deer_vla_amd import _abi as abi

lib = abi.lib()
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
SHAPES = [("xa ff1", 8192, 2048), ("xa ff2", 2048, 8192), ("qkv", 6144, 2048), ("out", 2048, 2048), ("up", 8192, 2048), ("down", 2048, 8192),
          ("7b qkv", 12288, 4096), ("7b down", 4096, 16384)]
rows = [int(a) for a in sys.argv[1:]] or [56, 84, 112, 128]


def timed(fn, seq):
    for w in seq: fn(w)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for w in seq: fn(w)
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / len(seq)


for T in rows:
    MP = abi.skinny_mpad(T)
    print("rows", T)
    for name, N, K in SHAPES:
        ncopy = min(64, max(4, int(600e6 / (N * K * 2))))
        A = torch.randn(T, K, device="cuda")
        hi = A.bfloat16(); lo = (A - hi.float()).bfloat16()
        Ws = []
        for _ in range(ncopy):
            w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
            wp = torch.empty_like(w)
            lib.deer_pack_weight_mfma16(abi.ptr(w), abi.ptr(wp), N, K, st())
            Ws.append(wp)
        torch.cuda.synchronize()
        S0 = lib.deer_skinny_splitk(T, N, K)
        part = torch.zeros(S0, MP, N, device="cuda")
        t_old = timed(lambda w: lib.deer_gemm_skinny(abi.ptr(A), K, None, 0, 0, abi.A_F32, abi.ptr(w), abi.ptr(part), T, N, K, S0, None, st()), Ws)
        line = f"{name:8s} N={N:5d} K={K:5d} {N*K*2/1e6:6.1f}MB | old S{S0}: {t_old:6.1f}us |"
        S1 = lib.deer_skinny_hl_splitk(T, N, K)
        for S in sorted(set([S1, max(1, S1 // 2), S1 * 2])):
            if K % (S * 64) or K // S < 128: continue
            part2 = torch.zeros(S, MP, N, device="cuda")
            if lib.deer_gemm_skinny_hl(abi.ptr(hi), abi.ptr(lo), K, abi.ptr(Ws[0]), abi.ptr(part2), T, N, K, S, None, st()) != 0: continue
            t = timed(lambda w: lib.deer_gemm_skinny_hl(abi.ptr(hi), abi.ptr(lo), K, abi.ptr(w), abi.ptr(part2), T, N, K, S, None, st()), Ws)
            line += f" hl S{S}{'*' if S == S1 else ''}: {t:6.1f}us {N*K*2/t/1e6:5.2f}TB/s |"
        print(line, flush=True)
    # the GELU split pre-pass of the down-projection (reads the up-projection's slabs once)
    for S in (4, 8):
        slab = torch.randn(S, MP, 8192, device="cuda")
        h = torch.zeros(MP, 8192, device="cuda", dtype=torch.bfloat16); l = torch.zeros_like(h)
        t = timed(lambda w: lib.deer_slab_gelu_split(abi.ptr(slab), S, MP * 8192, 1, abi.ptr(h), abi.ptr(l), T, 8192, None, st()), list(range(16)))
        print(f"gelu_split rows {T} C 8192 S{S}: {t:5.1f}us")
