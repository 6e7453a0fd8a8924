"""deer_resadd_ln at env-batch vision sizes: one row per workgroup vs 2 / 4 rows (deer_resadd_ln_multirow), graph-replayed, slabs rewritten by
nobody (MALL-warm like behind a GEMM).  usage: bench_resadd_rows.py [frames=16]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deer_vla_amd import _abi as abi
lib = abi.lib()
frames = int(sys.argv[1]) if len(sys.argv) > 1 else 16
T, d = 257 * frames, 1024
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
x = torch.randn(T, d, device="cuda")
g, b, bias = (torch.randn(d, device="cuda") for _ in range(3))
out = torch.zeros(T, d, device="cuda", dtype=torch.bfloat16)
outf = torch.zeros(T, d, device="cuda")
NB = 6
slabs = [torch.randn(2, T, d, device="cuda") * 0.01 for _ in range(NB)]


def timeit(fn, reps=10):
    fn(0); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for r in range(NB):
            fn(r)
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        gr.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * NB) * 1e3


for s_in in (0, 1, 2):
    mb = T * d * (4 * (2 + s_in) + 2) / 1e6 if s_in else T * d * 6 / 1e6
    row = []
    def one(r):
        abi.check(lib.deer_resadd_ln(abi.ptr(x), abi.ptr(slabs[r]) if s_in else None, s_in, T * d, None, abi.ptr(bias), abi.ptr(g), abi.ptr(b), abi.ptr(out),
                                     abi.ptr(outf), None, T, d, 1e-5, None, st()), "one")
    t1 = timeit(one)
    row.append(f"one row (+f32 out) {t1:.1f} us")
    for R in (2, 4):
        def multi(r, R=R):
            abi.check(lib.deer_resadd_ln_multirow(abi.ptr(x), abi.ptr(slabs[r]) if s_in else None, s_in, T * d, None, abi.ptr(bias), abi.ptr(g), abi.ptr(b),
                                                  abi.ptr(out), T, d, 1e-5, R, st()), "multi")
        t = timeit(multi)
        row.append(f"R={R} {t:.1f} us ({mb / t:.2f} TB/s)")
    print(f"T={T} slabs={s_in} ({mb:.0f} MB): " + "  ".join(row), flush=True)
    x.normal_()
