"""What does a HIP-event bracket add to a kernel of KNOWN duration?  (calibration of bench.py's per-launch timing)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deer_vla_amd import _abi as abi
lib = abi.lib()
st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
torch.zeros(1, device="cuda")
for us in (0, 5, 10, 20, 50):
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(64)]
    torch.cuda.synchronize()
    lib.deer_spin_us(3000, st())            # host runs ahead
    for a, b in evs:
        a.record()
        if us:
            lib.deer_spin_us(us, st())
        b.record()
    torch.cuda.synchronize()
    d = sorted(1e3 * a.elapsed_time(b) for a, b in evs)
    print(f"spin {us:3d} us: bracket median {d[32]:.2f} us  (min {d[0]:.2f}, p90 {d[57]:.2f})  -> overhead {d[32] - us:.2f} us")
