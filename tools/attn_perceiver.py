"""Perceiver attention launch (deer_attn_f16_hd64_2seg / deer_attn_mfma_hd64_2seg: 64 latents over [256 media tokens ; 64 latents], 8 heads x 64)
on its own under graph replay.  DEER_ATTN_VIT=0 selects the old kernel (attn_mfma_kernel, transposing V stores), default the two-segment
instantiation of attn_vit_kernel (round 6).  usage: [DEER_ATTN_VIT=0] attn_perceiver.py [frames=2] [f16|bf16]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deer_vla_amd import _abi as abi

lib = abi.lib()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 2
f16 = (sys.argv[2] if len(sys.argv) > 2 else "f16") == "f16"
tdt = torch.float16 if f16 else torch.bfloat16
fn = lib.deer_attn_f16_hd64_2seg if f16 else lib.deer_attn_mfma_hd64_2seg
H, nl, P, inner = 8, 64, 256, 512
g = torch.Generator(device="cuda").manual_seed(0)
qkvs = [torch.randn(N, nl, 3 * inner, device="cuda", generator=g).to(tdt) for _ in range(6)]
mkvs = [torch.randn(N, P, 2 * inner, device="cuda", generator=g).to(tdt) for _ in range(6)]
out = torch.zeros(N, nl, inner, device="cuda", dtype=tdt)


def launch(i):
    qkv, mkv = qkvs[i % 6], mkvs[i % 6]
    s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = fn(abi.ptr(qkv), abi.ptr(mkv), abi.ptr(mkv, inner * 2), abi.ptr(qkv, inner * 2), abi.ptr(qkv, 2 * inner * 2), abi.ptr(out), N, H, nl, P, nl,
            3 * inner, 2 * inner, 3 * inner, inner, nl * 3 * inner, P * 2 * inner, nl * 3 * inner, nl * inner, 0.125, s)
    assert rc == 0, rc


launch(0)
torch.cuda.synchronize()
q = qkvs[0][..., :inner].float().view(N, nl, H, 64).transpose(1, 2)
k = torch.cat([mkvs[0][..., :inner], qkvs[0][..., inner:2 * inner]], 1).float().view(N, P + nl, H, 64).transpose(1, 2)
v = torch.cat([mkvs[0][..., inner:], qkvs[0][..., 2 * inner:]], 1).float().view(N, P + nl, H, 64).transpose(1, 2)
ref = (torch.softmax(q @ k.transpose(-1, -2) * 0.125, -1) @ v).transpose(1, 2).reshape(N, nl, inner)
err = (out.float() - ref).abs().max().item()
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr, stream=st):
        for r in range(48):
            launch(r)
    gr.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(st)
    for _ in range(5):
        gr.replay()
    e1.record(st)
torch.cuda.synchronize()
print(f"frames {N} {'f16' if f16 else 'bf16'} DEER_ATTN_VIT={os.environ.get('DEER_ATTN_VIT', '1')}: {1e3 * e0.elapsed_time(e1) / (5 * 48):.2f} us per launch, max |out - fp32 reference| {err:.2e}")
