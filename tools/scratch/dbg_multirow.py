import ctypes, os, sys, math
sys.path.insert(0, "/root/repo")
import torch
from deer_vla_amd import _abi as abi
lib = abi.lib()
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
T, d = 4112, 1024
torch.manual_seed(0)
x0 = torch.randn(T, d, device="cuda")
bias, g, be = torch.randn(d, device="cuda"), torch.randn(d, device="cuda"), torch.randn(d, device="cuda")
for s_in in (0, 1, 2):
    slab = torch.randn(max(s_in, 1), T, d, device="cuda")
    sp = abi.ptr(slab) if s_in else None
    for R in (2, 4):
        xa, xb = x0.clone(), x0.clone()
        oa, ob = (torch.zeros(T, d, device="cuda", dtype=torch.bfloat16) for _ in range(2))
        of = torch.zeros(T, d, device="cuda")
        abi.check(lib.deer_resadd_ln_multirow(abi.ptr(xb), sp, s_in, T * d, None, abi.ptr(bias), abi.ptr(g), abi.ptr(be), abi.ptr(ob), T, d, 1e-5, R, st()), "m")
        abi.check(lib.deer_resadd_ln(abi.ptr(xa), sp, s_in, T * d, None, abi.ptr(bias), abi.ptr(g), abi.ptr(be), abi.ptr(oa), abi.ptr(of), None, T, d, 1e-5, None, st()), "o")
        torch.cuda.synchronize()
        dx = (xa - xb).abs()
        do = (oa.float() - ob.float()).abs()
        print("s_in", s_in, "R", R, "x equal", torch.equal(xa, xb), "max dx", float(dx.max()), "n", int((dx > 0).sum()), "| out equal", torch.equal(oa, ob), "max", float(do.max()), "n", int((do > 0).sum()),
              "rows differing", int((do > 0).any(1).sum()))
