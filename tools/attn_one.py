"""ViT self-attention launch (deer_attn_mfma_hd64) on its own: timing under graph replay and a target for rocprofv3 --pmc passes.
The q | k | v buffer is what the ViT's in_proj writes: bf16 [frames * 257, 3 * 1024], 16 heads x 64.
usage: attn_one.py [frames] [reps] [check]     (check: compare with an fp32 torch reference)"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deer_vla_amd import _abi as abi

lib = abi.lib()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
check = len(sys.argv) > 3
H, tok, W = 16, 257, 1024
g = torch.Generator(device="cuda").manual_seed(0)
bufs = [(torch.randn(N * tok, 3 * W, device="cuda", generator=g) * 1.5).bfloat16() for _ in range(6)]   # rotating: each launch reads a cold buffer
out = torch.zeros(N * tok, W, device="cuda", dtype=torch.bfloat16)


def launch(qkv, s=None):
    s = s or ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = lib.deer_attn_mfma_hd64(abi.ptr(qkv), abi.ptr(qkv, W * 2), abi.ptr(qkv, 2 * W * 2), abi.ptr(out), N, H, tok, tok, 3 * W, 3 * W, 3 * W, W,
                                 tok * 3 * W, tok * 3 * W, tok * 3 * W, tok * W, 0.125, s)
    assert rc == 0, rc


launch(bufs[0])
torch.cuda.synchronize()
if check:
    q, k, v = (bufs[0].float().view(N, tok, 3, H, 64)[:, :, i].permute(0, 2, 1, 3) for i in range(3))
    ref = torch.softmax(q @ k.transpose(-1, -2) * 0.125, dim=-1) @ v
    ref = ref.permute(0, 2, 1, 3).reshape(N * tok, W)
    err = (out.float() - ref).abs().max().item()
    print(f"max |out - fp32 reference| = {err:.3e} (bf16 P and output rounding: ~1e-2 at these magnitudes)")
    assert err < 4e-2, err
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=st):
        for r in range(reps):
            launch(bufs[r % len(bufs)])
    for _ in range(3):
        gr.replay()
    st.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(10):
        gr.replay()
    e1.record(st)
    st.synchronize()
us = e0.elapsed_time(e1) * 1e3 / (10 * reps)
flop = 4.0 * N * H * tok * tok * 64
print(f"attention {N} frames: {us:.2f} us per launch, {flop / us / 1e6:.0f} TFLOP/s, q|k|v + o bytes {N * tok * 4 * W * 2 / 1e6:.1f} MB -> {N * tok * 4 * W * 2 / us / 1e6:.2f} TB/s")
