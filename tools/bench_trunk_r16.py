"""The one-environment trunk kernels (csrc/trunk_r16.hip) per shape: cold weights (rotating copies > L2 + MALL), graph replay of back-to-back
launches (so the figure includes one kernel boundary, like a launch in the step).  usage: bench_trunk_r16.py [T]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deer_vla_amd import _abi as abi

lib = abi.lib()
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
T = int(sys.argv[1]) if len(sys.argv) > 1 else 14
d = 2048


def timed(fn, seq):
    for w in seq: fn(w)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for w in seq: fn(w)
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ts = []
    for _ in range(5):
        e0.record(); g.replay(); e1.record()
        torch.cuda.synchronize()
        ts.append(1e3 * e0.elapsed_time(e1) / len(seq))
    return sorted(ts)[len(ts) // 2]


def packed(N, K, n):
    out = []
    for _ in range(n):
        w = (torch.randn(N, K, device="cuda") * K ** -0.5).bfloat16()
        wp = torch.empty_like(w)
        lib.deer_pack_weight_mfma16(abi.ptr(w), abi.ptr(wp), N, K, st())
        out.append(wp)
    torch.cuda.synchronize()
    return out


A = torch.randn(16, d, device="cuda")
hi_rm = A.bfloat16(); lo_rm = (A - hi_rm.float()).bfloat16()
pk = lambda p: p.view(16, d // 32, 4, 8).permute(1, 2, 0, 3).contiguous().view(-1)
ph, pl = pk(hi_rm), pk(lo_rm)
print(f"T = {T}")
for name, N, epi in (("to_q-sized", 512, 0), ("ff.1 / mlp_up (GELU planes)", 8192, 1), ("Wqkv (+moments)", 6144, 2)):
    Ws = packed(N, d, min(64, max(4, int(600e6 / (N * d * 2)))))
    out = torch.zeros(16, N, device="cuda")
    hi = torch.zeros(16, N, device="cuda", dtype=torch.bfloat16); lo = torch.zeros_like(hi)
    stats = torch.zeros(N // 32, 16, 2, device="cuda")
    f = lambda w: lib.deer_trunk_wide_gemm(abi.ptr(ph), abi.ptr(pl), abi.ptr(w), N, d, epi, abi.ptr(out), abi.ptr(hi), abi.ptr(lo), N, abi.ptr(stats), T, None, st())
    assert f(Ws[0]) == 0
    t = timed(f, Ws)
    print(f"wide_gemm {name:30s} N={N:5d} K={d:5d} {N*d*2/1e6:6.1f} MB  {t:6.2f} us  {N*d*2/t/1e6:5.2f} TB/s", flush=True)
    S = lib.deer_skinny_hl_splitk(T, N, d)
    part = torch.zeros(S, 16, N, device="cuda")
    f = lambda w: lib.deer_gemm_skinny_hl(abi.ptr(hi_rm), abi.ptr(lo_rm), d, abi.ptr(w), abi.ptr(part), T, N, d, S, None, st())
    print(f"   (deer_gemm_skinny_hl, S = {S} slabs: {timed(f, Ws):6.2f} us)", flush=True)
x = torch.randn(16, d, device="cuda")
slab = torch.randn(16, 16, d, device="cuda") * 0.3
gamma, beta = torch.ones(d, device="cuda"), torch.zeros(d, device="cuda")
for S in (4, 8, 16):
    for nm, fn in (("split ", lib.deer_resadd_ln_split), ("packed", lib.deer_resadd_ln_packed)):
        f = lambda w: fn(abi.ptr(x), abi.ptr(slab), S, 16 * d, None, None, abi.ptr(gamma), abi.ptr(beta), abi.ptr(ph), abi.ptr(pl), None, None, T, d, 1e-5, None, st())
        print(f"resadd_ln_{nm} S={S:2d}: {timed(f, list(range(32))):6.2f} us")
qkv = torch.randn(16, 3 * d, device="cuda")
stats = torch.zeros(3 * d // 32, 16, 2, device="cuda")
g1 = torch.ones(d, device="cuda")
hi = torch.zeros(16, d, device="cuda", dtype=torch.bfloat16); lo = torch.zeros_like(hi)
f = lambda w: lib.deer_trunk_mpt_attn(abi.ptr(qkv), abi.ptr(stats), d, 16, abi.ptr(g1), abi.ptr(g1), 1e-5, None, 8.0, abi.ptr(hi), abi.ptr(lo), d, T, None, st())
assert f(0) == 0
print(f"mpt_attn (q/k LN from moments + 16 heads)                         {timed(f, list(range(32))):6.2f} us")
