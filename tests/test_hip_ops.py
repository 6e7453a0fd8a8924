"""Per-kernel parity tests on a real MI355X, calling libdeer_hip.so THROUGH ITS C ABI (ctypes, raw device
pointers) and comparing with plain fp32 torch math on the same inputs.  Tolerances are written in each test:
integer/index results exact; fp32-accumulated bf16 GEMMs ~1e-5 relative (inputs are bf16-exact in both arms);
kernels that round an intermediate to bf16 (softmax probabilities, bf16 outputs) ~1e-2."""
import ctypes
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

from deer_vla_amd import _abi as abi  # noqa: E402


def st():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def dev(t, dt=None):
    return t.to("cuda", dtype=dt).contiguous() if dt is not None else t.to("cuda").contiguous()


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g) * scale


def rel_err(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


def _fmt(dt):
    """(torch dtype, suffix of the C entry points) of a 16-bit operand format"""
    return (torch.float16, "f16") if dt == "f16" else (torch.bfloat16, "bf16")


@pytest.fixture(scope="module")
def lib():
    l = abi.lib()
    assert l.deer_hip_arch() == b"gfx950"
    return l


# ------------------------------------------------------------------------------------------ skinny GEMM
@pytest.mark.parametrize("M", [1, 14, 16, 17, 32, 33, 56, 64, 70, 84, 98, 112, 128])
@pytest.mark.parametrize("N,K", [(512, 2048), (2048, 512), (6144, 2048), (2048, 8192), (64, 32), (48, 96)])
def test_gemm_skinny_packed(lib, M, N, K):
    A = dev(rnd(M, K, seed=1), torch.bfloat16)
    W = dev(rnd(N, K, seed=2, scale=K ** -0.5), torch.bfloat16)
    Wp = torch.empty_like(W)
    abi.check(lib.deer_pack_weight_mfma16(abi.ptr(W), abi.ptr(Wp), N, K, st()), "pack")
    S = lib.deer_skinny_splitk(M, N, K)
    assert S >= 1 and K % (S * 32) == 0
    mpad = abi.skinny_mpad(M)
    part = torch.full((S, mpad, N), float("nan"), device="cuda")
    abi.check(lib.deer_gemm_skinny(abi.ptr(A), K, None, 0, 0, abi.A_BF16, abi.ptr(Wp), abi.ptr(part), M, N, K, S, None, st()), "skinny")
    torch.cuda.synchronize()
    out = part.sum(0)[:M]
    ref = A.float() @ W.float().t()
    assert rel_err(out, ref) < 2e-5
    assert torch.isfinite(part).all()                      # padded rows are written as zeros, never NaN
    if mpad > M:
        assert float(part[:, M:].abs().max()) == 0.0


@pytest.mark.parametrize("M,N,K", [(14, 2048, 2048), (32, 512, 2048), (3, 64, 96), (56, 2048, 2048), (42, 8192, 2048), (112, 2048, 8192), (112, 6144, 2048), (98, 512, 2048), (14, 2048, 8192)])
def test_gemm_skinny_f32_split_activation(lib, M, N, K):
    """f32 activations enter the MFMA as bf16 hi + lo: result ~fp32-accurate w.r.t. the bf16 weights."""
    A = dev(rnd(M, K, seed=41))
    W = dev(rnd(N, K, seed=42, scale=K ** -0.5), torch.bfloat16)
    Wp = torch.empty_like(W)
    abi.check(lib.deer_pack_weight_mfma16(abi.ptr(W), abi.ptr(Wp), N, K, st()), "pack")
    S = lib.deer_skinny_splitk(M, N, K)
    mpad = abi.skinny_mpad(M)
    part = torch.zeros(S, mpad, N, device="cuda")
    abi.check(lib.deer_gemm_skinny(abi.ptr(A), K, None, 0, 0, abi.A_F32, abi.ptr(Wp), abi.ptr(part), M, N, K, S, None, st()), "skinny")
    torch.cuda.synchronize()
    ref = (A.double() @ W.double().t()).float()
    assert rel_err(part.sum(0)[:M], ref) < 3e-5              # vs 2e-3 if A were rounded to a single bf16


def test_gemm_skinny_slab_gelu_input_and_determinism(lib):
    M, K, N = 14, 8192, 2048
    s_in = 4
    slab = dev(rnd(s_in, 16, K, seed=3))
    W = dev(rnd(N, K, seed=4, scale=K ** -0.5), torch.bfloat16)
    Wp = torch.empty_like(W)
    abi.check(lib.deer_pack_weight_mfma16(abi.ptr(W), abi.ptr(Wp), N, K, st()), "pack")
    S = lib.deer_skinny_splitk(M, N, K)
    outs = []
    for _ in range(2):
        part = torch.zeros(S, 16, N, device="cuda")
        abi.check(lib.deer_gemm_skinny(None, 0, abi.ptr(slab), s_in, 16 * K, abi.A_SLABS_GELU, abi.ptr(Wp), abi.ptr(part), M, N, K, S,
                                       None, st()), "skinny")
        torch.cuda.synchronize()
        outs.append(part.sum(0)[:M].clone())
    a = torch.nn.functional.gelu(slab.sum(0)[:M])
    ref = (a.double() @ W.double().t()).float()
    assert rel_err(outs[0], ref) < 3e-5                      # gelu(sum) enters as bf16 hi+lo
    assert torch.equal(outs[0], outs[1])                     # bit-reproducible (no atomics)


def test_skinny_respects_exit_flag(lib):
    ctl = torch.zeros(abi.CTL_WORDS, dtype=torch.int32, device="cuda")
    ctl[abi.CTL_ALL_EXITED] = 1                               # every environment of the batch has exited
    A = dev(rnd(4, 64), torch.bfloat16)
    W = dev(rnd(32, 64), torch.bfloat16)
    Wp = torch.empty_like(W)
    abi.check(lib.deer_pack_weight_mfma16(abi.ptr(W), abi.ptr(Wp), 32, 64, st()), "pack")
    part = torch.full((1, 16, 32), 7.0, device="cuda")
    abi.check(lib.deer_gemm_skinny(abi.ptr(A), 64, None, 0, 0, abi.A_BF16, abi.ptr(Wp), abi.ptr(part), 4, 32, 64, 1, abi.ptr(ctl), st()), "skinny")
    torch.cuda.synchronize()
    assert float(part.min()) == 7.0                          # kernel returned at entry


def test_abi_rejects_bad_shapes(lib):
    # every operand valid except the property under test (the old "M > 64" line passed because A was NULL)
    A = torch.zeros(130, 64, device="cuda", dtype=torch.bfloat16)
    Wp = torch.zeros(64, 64, device="cuda", dtype=torch.bfloat16)
    part = torch.zeros(1, 144, 64, device="cuda")
    ok = lambda M, N, K, S: lib.deer_gemm_skinny(abi.ptr(A), 64, None, 0, 0, abi.A_BF16, abi.ptr(Wp), abi.ptr(part), M, N, K, S, None, st())
    assert ok(128, 64, 64, 1) == 0                                                                        # 128 rows = 8 envs x 16 tokens: the cap
    assert ok(129, 64, 64, 1) == 1                                                                        # M > 128
    assert ok(4, 60, 64, 1) == 1                                                                          # N % 16
    assert ok(4, 64, 48, 1) == 1                                                                          # K % 32
    assert ok(4, 64, 64, 4) == 1                                                                          # K % (splitk * 32)
    assert lib.deer_gemm_skinny(None, 0, None, 0, 0, 0, None, None, 4, 64, 64, 1, None, st()) == 1        # A == NULL
    torch.cuda.synchronize()
    assert lib.deer_gemm_skinny(None, 0, None, 0, 0, 0, None, None, 4, 60, 64, 1, None, st()) == 1       # N % 16
    assert lib.deer_gemm_bf16_nt(None, 8, 0, None, 8, None, None, 8, 0, 4, 16, 12, 1, 0, None, 0, None, st()) == 1   # K % 8
    q = torch.zeros(4, 64, device="cuda", dtype=torch.bfloat16)
    kv = torch.zeros(640, 64, device="cuda", dtype=torch.bfloat16)
    attn = lambda n, Q=q: lib.deer_attn_mfma_hd64(abi.ptr(Q) if Q is not None else None, abi.ptr(kv), abi.ptr(kv), abi.ptr(q), 1, 1, 4, n, 64, 64, 64, 64,
                                                  0, 0, 0, 0, 1.0, st())
    assert attn(576) == 0                                                                                 # 36 key tiles: the cap (pre fusion)
    assert attn(577) == 1                                                                                 # kv_len
    assert attn(400, None) == 1                                                                           # Q == NULL
    torch.cuda.synchronize()


@pytest.mark.parametrize("M", [70, 96, 112, 128])
@pytest.mark.parametrize("N,K,mode", [(6144, 2048, "f32"), (2048, 2048, "f32"), (8192, 2048, "f32"), (2048, 8192, "slabs_gelu"),
                                      (16384, 4096, "f32"), (4096, 16384, "slabs_gelu")])
def test_gemm_skinny_env_batch_rows_trunk_shapes(lib, M, N, K, mode):
    """The row tiles an env batch of 5-8 environments uses (MT = 5..8: 70-128 rows) on every projection shape of the MPT-1B / MPT-7B
    trunk, in the two activation modes the spine feeds them with (f32 rows as bf16 hi+lo; GELU of the up-projection's slabs), against
    fp64 torch math on the same bf16 weights."""
    W = dev(rnd(N, K, seed=52, scale=K ** -0.5), torch.bfloat16)
    Wp = torch.empty_like(W)
    abi.check(lib.deer_pack_weight_mfma16(abi.ptr(W), abi.ptr(Wp), N, K, st()), "pack")
    S = lib.deer_skinny_splitk(M, N, K)
    mpad = abi.skinny_mpad(M)
    part = torch.full((S, mpad, N), float("nan"), device="cuda")
    if mode == "f32":
        A = dev(rnd(M, K, seed=51))
        abi.check(lib.deer_gemm_skinny(abi.ptr(A), K, None, 0, 0, abi.A_F32, abi.ptr(Wp), abi.ptr(part), M, N, K, S, None, st()), "skinny")
        a = A.double()
    else:
        s_in = 3
        slab = dev(rnd(s_in, mpad, K, seed=53))
        abi.check(lib.deer_gemm_skinny(None, 0, abi.ptr(slab), s_in, mpad * K, abi.A_SLABS_GELU, abi.ptr(Wp), abi.ptr(part), M, N, K, S, None,
                                       st()), "skinny")
        a = torch.nn.functional.gelu(slab.sum(0)[:M]).double()
    torch.cuda.synchronize()
    ref = (a @ W.double().t()).float()
    assert torch.isfinite(part).all()
    assert rel_err(part.sum(0)[:M], ref) < 3e-5
    if mpad > M:
        assert float(part[:, M:].abs().max()) == 0.0


def _split_hl(a, dt=torch.bfloat16):
    hi = a.to(dt)
    lo = (a - hi.float()).to(dt)
    return hi.contiguous(), lo.contiguous()


SKHL = ([("bf16", M, N, K) for M in (1, 14, 33, 50, 56, 70, 84, 96, 112, 128)
         for (N, K) in ((6144, 2048), (2048, 2048), (8192, 2048), (2048, 8192), (16384, 4096), (4096, 16384), (512, 2048), (80, 128), (2048, 512))] +
        [("f16", M, N, K) for M in (1, 14, 56, 112, 128) for (N, K) in ((6144, 2048), (2048, 8192), (16384, 4096), (80, 128))])


@pytest.mark.parametrize("dt,M,N,K", SKHL)
def test_gemm_skinny_hl_env_batch_kernel(lib, dt, M, N, K):
    """deer_gemm_skinny_hl (LDS-DMA ring, pre-split hi/lo activation planes, 128-column workgroups) on every trunk projection shape
    of MPT-1B / MPT-7B (+ ragged N, short K) against fp64 torch math on the same 16-bit weights (bf16, and the fp16 twin of round 6);
    bit-reproducible; padded rows zero."""
    tdt, sfx = _fmt(dt)
    a = dev(rnd(M, K, seed=81))
    hi, lo = _split_hl(a, tdt)
    W = dev(rnd(N, K, seed=82, scale=K ** -0.5), tdt)
    Wp = torch.empty_like(W)
    abi.check(lib.deer_pack_weight_mfma16(abi.ptr(W), abi.ptr(Wp), N, K, st()), "pack")
    S = lib.deer_skinny_hl_splitk(M, N, K)
    assert S >= 1 and K % (S * 64) == 0
    mpad = abi.skinny_mpad(M)
    outs = []
    for _ in range(2):
        part = torch.full((S, mpad, N), float("nan"), device="cuda")
        abi.check(getattr(lib, "deer_gemm_skinny_hl" + ("_f16" if dt == "f16" else ""))(abi.ptr(hi), abi.ptr(lo), K, abi.ptr(Wp), abi.ptr(part), M, N, K, S, None, st()), "skinny_hl")
        torch.cuda.synchronize()
        outs.append(part)
    part = outs[0]
    assert torch.isfinite(part).all()
    ref = (a.double() @ W.double().t()).float()
    assert rel_err(part.sum(0)[:M], ref) < 3e-5
    assert torch.equal(outs[0], outs[1])
    if mpad > M:
        assert float(part[:, M:].abs().max()) == 0.0


def test_skinny_hl_producers_and_exit_flag(lib):
    """deer_slab_gelu_split == split(gelu(sum slabs)); deer_resadd_ln_split's planes == split of its own f32 LayerNorm output;
    deer_gemm_skinny_hl returns at entry once ALL_EXITED is set."""
    rows, C, s_in = 112, 8192, 4
    slab = dev(rnd(s_in, 112, C, seed=83))
    hi = torch.zeros(rows, C, device="cuda", dtype=torch.bfloat16)
    lo = torch.zeros_like(hi)
    abi.check(lib.deer_slab_gelu_split(abi.ptr(slab), s_in, 112 * C, 1, abi.ptr(hi), abi.ptr(lo), rows, C, None, st()), "gelu_split")
    torch.cuda.synchronize()
    a = torch.nn.functional.gelu(slab.sum(0))
    rh, rl = _split_hl(a)
    assert float(((hi.float() + lo.float()) - a).abs().max()) < 1e-4 * float(a.abs().max())
    assert (hi != rh).float().mean() < 1e-3 and rel_err(lo.float(), rl.float()) < 2e-2     # erf ulps may flip a rounding
    d, T = 2048, 112
    x = dev(rnd(T, d, seed=84))
    g, b = dev(1 + 0.1 * rnd(d, seed=85)), dev(0.1 * rnd(d, seed=86))
    y32 = torch.zeros(T, d, device="cuda")
    yh = torch.zeros(T, d, device="cuda", dtype=torch.bfloat16)
    yl = torch.zeros_like(yh)
    abi.check(lib.deer_resadd_ln_split(abi.ptr(x), None, 0, 0, None, None, abi.ptr(g), abi.ptr(b), abi.ptr(yh), abi.ptr(yl), abi.ptr(y32), None,
                                       T, d, 1e-5, None, st()), "resadd_ln_split")
    torch.cuda.synchronize()
    rh, rl = _split_hl(y32)
    assert torch.equal(yh, rh) and torch.equal(yl, rl)
    assert rel_err(y32, torch.nn.functional.layer_norm(x, (d,), g, b, 1e-5)) < 1e-5
    ctl = torch.zeros(abi.CTL_WORDS, dtype=torch.int32, device="cuda")
    ctl[abi.CTL_ALL_EXITED] = 1
    W = dev(rnd(128, d, seed=87), torch.bfloat16)
    Wp = torch.empty_like(W)
    abi.check(lib.deer_pack_weight_mfma16(abi.ptr(W), abi.ptr(Wp), 128, d, st()), "pack")
    part = torch.full((1, 112, 128), 7.0, device="cuda")
    abi.check(lib.deer_gemm_skinny_hl(abi.ptr(yh), abi.ptr(yl), d, abi.ptr(Wp), abi.ptr(part), T, 128, d, 1, abi.ptr(ctl), st()), "skinny_hl")
    torch.cuda.synchronize()
    assert float(part.min()) == 7.0


# ------------------------------------------------------------------------------------------- tiled GEMM
BIG_TILES = [17, 39, 45, 51, 52, 53, 54, 55, 56, 57, 58, 59, 60, 61, 62, 63, 64, 65, 66, 67, 68, 69, 70, 71, 72, 73, 74, 75, 76, 77, 78, 79]
# every tile on the 16-frame batch (whole frames) and on a small ragged M; the selector (tile 0) and the tiles it picks on the 8 / 12-frame
# batches and on a power-of-two M; the fp16 instantiation (round 6: the vision tower's fp16 arithmetic) on the selector and on one tile of
# every kernel of csrc/gemm_bigm.hip + the 128x128 ring of csrc/gemm_tiled.hip
BIG_CASES = ([("bf16", t, M) for t in BIG_TILES for M in (4112, 300)] + [("bf16", 0, M) for M in (2056, 3084, 4112, 4096, 300)] +
             [("bf16", t, M) for t in (17, 39, 45, 63, 64, 67) for M in (2056, 3084)] +
             [("f16", t, M) for t in (0, 17, 39, 51, 57, 63, 64, 67, 72, 74, 75) for M in (4112, 300)] + [("f16", 0, M) for M in (2056, 3084)] +
             [(dt, t, M) for dt in ("bf16", "f16") for t in (76, 77, 78) for M in (3084, 8224)] + [("f16", t, 4112) for t in (76, 77, 78, 79)])


@pytest.mark.parametrize("dt,tile,M", BIG_CASES)
@pytest.mark.parametrize("N,K,epi", [(3072, 1024, "bf16"), (4096, 1024, "qgelu"), (1024, 1024, "f32"), (1024, 4096, "f32")])
def test_gemm_tiled_big_m_tiles(lib, dt, tile, M, N, K, epi):
    """The tiles `gemm_dispatch` auto-selects for an env batch / calibration window (M = 257 x 8 / 12 / 16 frames: 17 = 128x128 / 16
    waves, 39 = 192x128, 45 = 128x192) on the four ViT-L projection shapes, called directly AND through the selector (tile 0), against
    fp32 torch math; the ragged last row block (M % 128 = 8, 12, 16) and every epilogue the tower uses."""
    if tile in (64, 65, 75) and N % 192:
        pytest.skip("192-column frame tiles")
    if tile in (74, 75) and (epi == "f32" or M % 257):
        pytest.skip("frame8 tiles (eight waves, csrc/gemm_bigm.hip: gemm_frame8_kernel): whole camera frames, bf16 epilogues only")
    if tile in (76, 77, 78, 79) and (M % 257 or (tile == 77 and N % 192)):
        pytest.skip("frame4 tiles (four waves, csrc/gemm_bigm.hip: gemm_frame4_kernel): whole camera frames")
    tdt, sfx = _fmt(dt)
    A = dev(rnd(M, K, seed=61), tdt)
    W = dev(rnd(N, K, seed=62, scale=K ** -0.5), tdt)
    bias = dev(rnd(N, seed=63, scale=0.1))
    ref = A.float() @ W.float().t() + bias
    e = {"bf16": abi.EPI_BF16, "qgelu": abi.EPI_QGELU_BF16, "f32": abi.EPI_F32}[epi]
    C = torch.full((M + 8, N), float("nan"), device="cuda", dtype=torch.float32 if epi == "f32" else tdt)
    abi.check(getattr(lib, f"deer_gemm_{sfx}_nt")(abi.ptr(A), K, 0, abi.ptr(W), K, abi.ptr(bias), abi.ptr(C), N, 0, M, N, K, 1, e, None, tile, None, st()), "gemm")
    torch.cuda.synchronize()
    assert torch.isnan(C[M:].float()).all()                                  # nothing written past row M
    tol16 = 4e-3 if dt == "bf16" else 5e-4                                   # one rounding of the output: 2^-9 (bf16) / 2^-12 (fp16) relative
    if epi == "f32":
        assert rel_err(C[:M], ref) < 2e-5
    elif epi == "bf16":
        assert rel_err(C[:M].float(), ref) < tol16
    else:
        assert rel_err(C[:M].float(), ref * torch.sigmoid(1.702 * ref)) < (tol16 if dt == "bf16" else 1e-3)   # quick_gelu_bf: v_exp + v_rcp


@pytest.mark.parametrize("dt", ["bf16", "f16"])
@pytest.mark.parametrize("t4,t16,N,K,epi,batch", [(76, 63, 4096, 1024, "qgelu", 1), (77, 64, 3072, 1024, "bf16", 1), (78, 67, 1024, 2048, "f32", 2), (79, 67, 1024, 2048, "f32", 2),
                                                  (76, 63, 1024, 1024, "bf16out", 1), (78, 67, 1024, 1024, "bf16", 1), (76, 63, 1024, 1024, "f32", 1), (77, 64, 3072, 1024, "f32", 1)])
@pytest.mark.parametrize("M", [4112, 3084, 8224])
def test_gemm_frame4_bit_identical_to_the_16_wave_frame_tile(lib, dt, t4, t16, N, K, epi, batch, M):
    """The four-wave frame tile (round 6: asm MFMAs with the accumulators tied in AGPRs, fragments of the next K-step read under the MFMAs
    of the current one) walks K in the same order per output element as the 16-wave tile it replaces in the selector: identical bits for
    every epilogue it takes (16-bit, QuickGELU, bf16-out, f32 / f32 slabs of K halves through blockIdx.z; bias where the 16-bit forms have
    one), at 12 / 16 / 32 frames."""
    tdt, sfx = _fmt(dt)
    A = dev(rnd(M, K * batch, seed=71), tdt)
    W = dev(rnd(N, K * batch, seed=72, scale=(K * batch) ** -0.5), tdt)
    bias = None if epi == "f32" else dev(rnd(N, seed=73, scale=0.1))
    e = {"bf16": abi.EPI_BF16, "qgelu": abi.EPI_QGELU_BF16, "f32": abi.EPI_F32, "bf16out": abi.EPI_BF16OUT}[epi]
    odt = torch.float32 if epi == "f32" else (torch.bfloat16 if epi == "bf16out" else tdt)
    out = []
    for tile in (t4, t16):
        C = torch.full((batch, M + 8, N), float("nan"), device="cuda", dtype=odt)
        abi.check(getattr(lib, f"deer_gemm_{sfx}_nt_wbatch")(abi.ptr(A), K * batch, K, abi.ptr(W), K * batch, K, abi.ptr(bias), abi.ptr(C), N, (M + 8) * N, M, N, K,
                                                             batch, e, tile, None, st()), "gemm")
        torch.cuda.synchronize()
        assert torch.isnan(C[:, M:].float()).all() and torch.isfinite(C[:, :M].float()).all()
        out.append(C[:, :M].clone())
    assert torch.equal(out[0], out[1])
    ref = sum(A[:, z * K:(z + 1) * K].float() @ W[:, z * K:(z + 1) * K].float().t() for z in range(batch))
    if epi == "f32":
        assert rel_err(out[0].sum(0), ref) < 2e-5


def test_gemm_frame4_respects_exit_flag_and_refuses_what_it_does_not_take(lib):
    M, N, K = 4112, 1024, 1024
    A = dev(rnd(M, K, seed=74), torch.bfloat16)
    W = dev(rnd(N, K, seed=75, scale=K ** -0.5), torch.bfloat16)
    C = torch.full((M, N), 7.0, device="cuda", dtype=torch.bfloat16)
    ctl = torch.zeros(abi.CTL_WORDS, dtype=torch.int32, device="cuda")
    ctl[abi.CTL_ALL_EXITED] = 1
    call = lambda m, k, e, t, c=None: lib.deer_gemm_bf16_nt(abi.ptr(A), K, 0, abi.ptr(W), K, None, abi.ptr(C), N, 0, m, N, k, 1, e, None, t, abi.ptr(c), st())
    assert call(M, K, abi.EPI_BF16, 76, ctl) == 0
    torch.cuda.synchronize()
    assert float(C.float().min()) == 7.0 and float(C.float().max()) == 7.0          # every environment has exited: nothing runs
    assert call(M, K, abi.EPI_GELU_BF16, 76) == 1                                       # erf GELU: not a frame-shaped GEMM's epilogue
    assert call(M - 1, K, abi.EPI_BF16, 76) == 1                                        # whole frames only
    assert call(M, 32 * 31, abi.EPI_BF16, 76) == 1                                      # K-steps in pairs


@pytest.mark.parametrize("M", [2056, 4112])
def test_gemm_tiled_big_m_splitk_halves(lib, M):
    """c_proj of an env batch runs as two K halves from 2048 rows (model.hip::pick_split): slab[0] + slab[1] == the full product."""
    N, K = 1024, 4096
    A = dev(rnd(M, K, seed=64), torch.bfloat16)
    W = dev(rnd(N, K, seed=65, scale=K ** -0.5), torch.bfloat16)
    slab = torch.zeros(2, M, N, device="cuda")
    abi.check(lib.deer_gemm_bf16_nt_splitk(abi.ptr(A), K, abi.ptr(W), K, abi.ptr(slab), M, N, K, 2, 0, None, st()), "splitk")
    torch.cuda.synchronize()
    assert rel_err(slab.sum(0), A.float() @ W.float().t()) < 2e-5


@pytest.mark.parametrize("dt,tile", [("bf16", t) for t in (1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 0)] + [("f16", t) for t in (1, 4, 8, 16, 0)])
@pytest.mark.parametrize("M,N,K", [(514, 3072, 1024), (514, 1024, 4096), (128, 1024, 512), (37, 128, 640), (640, 1024, 1024)])
def test_gemm_tiled_epilogues(lib, dt, tile, M, N, K):
    """every epilogue of the small-M kernels in both 16-bit formats; fp16 family: DEER_EPI_BF16OUT stores bf16 (media K/V for the trunk)"""
    tdt, sfx = _fmt(dt)
    tol16 = 4e-3 if dt == "bf16" else 5e-4
    A = dev(rnd(M, K, seed=5), tdt)
    W = dev(rnd(N, K, seed=6, scale=K ** -0.5), tdt)
    bias = dev(rnd(N, seed=7, scale=0.1))
    ref = A.float() @ W.float().t() + bias
    fn = getattr(lib, f"deer_gemm_{sfx}_nt")

    def run(epi, C, gate=None, b=bias):
        abi.check(fn(abi.ptr(A), K, 0, abi.ptr(W), K, abi.ptr(b), abi.ptr(C), N, 0, M, N, K, 1, epi, abi.ptr(gate), tile, None, st()), "gemm")
        torch.cuda.synchronize()

    C = torch.zeros(M, N, device="cuda")
    run(abi.EPI_F32, C)
    assert rel_err(C, ref) < 2e-5
    Cb = torch.zeros(M, N, device="cuda", dtype=tdt)
    run(abi.EPI_BF16, Cb)
    assert rel_err(Cb.float(), ref) < tol16                  # one rounding of the output
    run(abi.EPI_QGELU_BF16, Cb)
    assert rel_err(Cb.float(), ref * torch.sigmoid(1.702 * ref)) < max(tol16, 1e-3)
    run(abi.EPI_GELU_BF16, Cb)
    assert rel_err(Cb.float(), torch.nn.functional.gelu(ref)) < tol16
    Cx = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    run(abi.EPI_BF16OUT, Cx)                                 # bf16 store whatever the operand format
    assert rel_err(Cx.float(), ref) < 4e-3 and torch.equal(Cx, ref_round_bf16(C))
    R0 = dev(rnd(M, N, seed=8))
    R = R0.clone()
    gate = torch.tensor([0.5], device="cuda")
    run(abi.EPI_RESADD_F32, R, gate)
    assert rel_err(R, R0 + math.tanh(0.5) * ref) < 2e-5
    R = R0.clone()
    run(abi.EPI_RESADD_F32, R, None, None)
    assert rel_err(R, R0 + (ref - bias)) < 2e-5


def ref_round_bf16(c_f32):
    """the bf16 rounding of the kernel's own f32 result (same accumulation order): EPI_BF16OUT must equal it bit for bit"""
    return c_f32.to(torch.bfloat16)


def test_gemm_tiled_batched_strided(lib):
    B, M, N, K, rows = 2, 64, 512, 128, 80
    A = dev(rnd(B, rows, K, seed=9), torch.bfloat16)         # use rows 16..79 of each batch
    W = dev(rnd(N, K, seed=10, scale=K ** -0.5), torch.bfloat16)
    C = torch.zeros(B, M, N, device="cuda", dtype=torch.bfloat16)
    abi.check(lib.deer_gemm_bf16_nt(abi.ptr(A, 16 * K * 2), K, rows * K, abi.ptr(W), K, None, abi.ptr(C), N, M * N, M, N, K, B, abi.EPI_BF16,
                                    None, 0, None, st()), "gemm")
    torch.cuda.synchronize()
    ref = A[:, 16:].float() @ W.float().t()
    assert rel_err(C.float(), ref) < 4e-3


# -------------------------------------------------------------------------------------------- attention
@pytest.mark.parametrize("B,H,q_len,kv_len", [(2, 16, 257, 257), (8, 16, 257, 257), (3, 16, 200, 270), (2, 8, 64, 320), (2, 2, 17, 17), (1, 1, 5, 68),
                                              (1, 3, 100, 33)])
@pytest.mark.parametrize("dt", ["bf16", "f16"])
def test_attn_mfma(lib, dt, B, H, q_len, kv_len):
    """(2, 16, 257, 257) and (8, 16, 257, 257): the ViT shapes - round 5's attn_vit_kernel (V row-major in LDS + transpose reads), one query tile
    per wave and (>= 128 (head, image) pairs) the looping form; (3, 16, 200, 270): the same kernel with ragged query / key counts."""
    hd = 64
    tdt, _ = _fmt(dt)
    q = dev(rnd(B, q_len, H * hd, seed=11), tdt)
    k = dev(rnd(B, kv_len, H * hd, seed=12), tdt)
    v = dev(rnd(B, kv_len, H * hd, seed=13), tdt)
    o = torch.zeros(B, q_len, H * hd, device="cuda", dtype=tdt)
    scale = hd ** -0.5
    abi.check((lib.deer_attn_f16_hd64 if dt == "f16" else lib.deer_attn_mfma_hd64)(abi.ptr(q), abi.ptr(k), abi.ptr(v), abi.ptr(o), B, H, q_len, kv_len, H * hd, H * hd, H * hd, H * hd,
                                      q_len * H * hd, kv_len * H * hd, kv_len * H * hd, q_len * H * hd, scale, st()), "attn")
    torch.cuda.synchronize()
    qf = q.float().view(B, q_len, H, hd).transpose(1, 2)
    kf = k.float().view(B, kv_len, H, hd).transpose(1, 2)
    vf = v.float().view(B, kv_len, H, hd).transpose(1, 2)
    ref = (torch.softmax(qf @ kf.transpose(-1, -2) * scale, -1) @ vf).transpose(1, 2).reshape(B, q_len, H * hd)
    assert rel_err(o.float(), ref) < (8e-3 if dt == "bf16" else 1e-3)   # P and O are rounded to the 16-bit format
    assert float((o.float() - ref).abs().max()) < (3e-2 if dt == "bf16" else 4e-3)


@pytest.mark.parametrize("B,H,q_len,kv1,kv2,dt", [(2, 8, 64, 256, 64, "bf16"), (2, 8, 64, 256, 64, "f16"), (16, 8, 64, 256, 64, "f16"), (3, 8, 50, 256, 60, "bf16"), (1, 2, 16, 16, 16, "bf16"), (2, 2, 64, 17, 5, "bf16"), (1, 1, 5, 0, 33, "bf16"),
                                                  # long key sequences (pre fusion: 2 x 256 patch tokens + 64 latents; the 36-tile instantiation)
                                                  (2, 8, 64, 512, 64, "bf16"), (2, 8, 64, 512, 64, "f16"), (1, 2, 37, 300, 41, "f16"), (1, 1, 64, 321, 0 + 64, "bf16")])
def test_attn_mfma_two_segments(lib, B, H, q_len, kv1, kv2, dt):
    """Perceiver attention over [media K/V ; latent K/V] without the concat; q|k|v of the latents share one buffer.  64 latents over
    256 + 64 keys (20 key tiles) runs the two-segment instantiation of attn_vit_kernel (round 6), ragged forms of it included."""
    hd = 64
    inner = H * hd
    tdt = torch.float16 if dt == "f16" else torch.bfloat16
    qkv = dev(rnd(B, q_len, 3 * inner, seed=71), tdt)           # kv2 == q_len in the Perceiver; general here
    kv2buf = qkv if kv2 == q_len else dev(rnd(B, kv2, 3 * inner, seed=72), tdt)
    mkv = dev(rnd(B, max(kv1, 1), 2 * inner, seed=73), tdt)
    o = torch.zeros(B, q_len, inner, device="cuda", dtype=tdt)
    scale = hd ** -0.5
    abi.check((lib.deer_attn_f16_hd64_2seg if dt == "f16" else lib.deer_attn_mfma_hd64_2seg)(abi.ptr(qkv), abi.ptr(mkv), abi.ptr(mkv, inner * 2), abi.ptr(kv2buf, inner * 2),
                                           abi.ptr(kv2buf, 2 * inner * 2), abi.ptr(o), B, H, q_len, kv1, kv2, 3 * inner, 2 * inner,
                                           3 * inner, inner, q_len * 3 * inner, max(kv1, 1) * 2 * inner, kv2 * 3 * inner, q_len * inner,
                                           scale, st()), "attn 2seg")
    torch.cuda.synchronize()
    qf = qkv[..., :inner].float().view(B, q_len, H, hd).transpose(1, 2)
    k = torch.cat([mkv[:, :kv1, :inner], kv2buf[..., inner:2 * inner]], 1).float()
    v = torch.cat([mkv[:, :kv1, inner:], kv2buf[..., 2 * inner:]], 1).float()
    kf = k.view(B, kv1 + kv2, H, hd).transpose(1, 2)
    vf = v.view(B, kv1 + kv2, H, hd).transpose(1, 2)
    ref = (torch.softmax(qf @ kf.transpose(-1, -2) * scale, -1) @ vf).transpose(1, 2).reshape(B, q_len, inner)
    assert rel_err(o.float(), ref) < (8e-3 if dt == "bf16" else 1e-3)
    assert float((o.float() - ref).abs().max()) < (3e-2 if dt == "bf16" else 4e-3)


@pytest.mark.parametrize("dt", ["bf16", "f16"])
def test_attn_two_segments_perceiver_kernel_is_bit_identical_to_the_general_kernel(lib, dt):
    """64 queries over 256 + 64 keys run the two-segment instantiation of attn_vit_kernel (round 6); 80 queries run the general kernel
    (attn_mfma_kernel, eight waves).  Queries are independent, so the first 64 rows of the two launches must agree bit for bit: the new
    instantiation keeps the old arithmetic (scores scaled first, __expf of the difference)."""
    B, H, hd, kv1, kv2 = 2, 8, 64, 256, 64
    inner = H * hd
    tdt = torch.float16 if dt == "f16" else torch.bfloat16
    fn = lib.deer_attn_f16_hd64_2seg if dt == "f16" else lib.deer_attn_mfma_hd64_2seg
    q = dev(rnd(B, 80, inner, seed=91), tdt)
    lkv = dev(rnd(B, kv2, 3 * inner, seed=92), tdt)
    mkv = dev(rnd(B, kv1, 2 * inner, seed=93), tdt)
    outs = []
    for q_len in (64, 80):
        o = torch.zeros(B, 80, inner, device="cuda", dtype=tdt)
        abi.check(fn(abi.ptr(q), abi.ptr(mkv), abi.ptr(mkv, inner * 2), abi.ptr(lkv, inner * 2), abi.ptr(lkv, 2 * inner * 2), abi.ptr(o), B, H, q_len, kv1, kv2,
                     inner, 2 * inner, 3 * inner, inner, 80 * inner, kv1 * 2 * inner, kv2 * 3 * inner, 80 * inner, hd ** -0.5, st()), "attn 2seg")
        torch.cuda.synchronize()
        outs.append(o[:, :64].clone())
    assert float(outs[0].float().abs().max()) > 0 and torch.equal(outs[0], outs[1])


def test_xattn_small(lib):
    T, n_kv, heads, inner, ldkv = 14, 128, 8, 512, 3 * 1024
    s_in = 3
    qs = dev(rnd(s_in, 16, inner, seed=14))
    kv = dev(rnd(n_kv, ldkv, seed=15), torch.bfloat16)
    tt = torch.tensor([1] * T, dtype=torch.int32, device="cuda")
    tt[3] = 0                                                # a token without preceding media -> zero row
    out = torch.zeros(T, inner, device="cuda", dtype=torch.bfloat16)
    off = 1024                                               # second layer's slice
    abi.check(lib.deer_xattn_small(abi.ptr(qs), s_in, 16 * inner, inner, abi.ptr(kv, off * 2), ldkv, inner, abi.ptr(tt), 128, abi.ptr(out),
                                   0, inner, T, n_kv, heads, 1, 64 ** -0.5, None, st()), "xattn")
    outf = torch.zeros(T, inner, device="cuda")
    abi.check(lib.deer_xattn_small(abi.ptr(qs), s_in, 16 * inner, inner, abi.ptr(kv, off * 2), ldkv, inner, abi.ptr(tt), 128, abi.ptr(outf),
                                   1, inner, T, n_kv, heads, 1, 64 ** -0.5, None, st()), "xattn")
    torch.cuda.synchronize()
    q = qs.sum(0)[:T].view(T, heads, 64).transpose(0, 1) * 64 ** -0.5
    k = kv[:, off:off + inner].float().view(n_kv, heads, 64).transpose(0, 1)
    v = kv[:, off + inner:off + 2 * inner].float().view(n_kv, heads, 64).transpose(0, 1)
    a = torch.softmax(q @ k.transpose(-1, -2), -1)
    a[:, 3] = 0
    ref = (a @ v).transpose(0, 1).reshape(T, inner)
    assert rel_err(out.float(), ref) < 4e-3
    assert rel_err(outf, ref) < 2e-5
    assert float(out[3].abs().max()) == 0.0 and float(outf[3].abs().max()) == 0.0


@pytest.mark.parametrize("qk_ln", [True, False])
@pytest.mark.parametrize("T,d,H", [(14, 2048, 16), (9, 256, 2), (32, 64, 2)])
def test_mpt_attn_small(lib, qk_ln, T, d, H):
    s_in = 2
    mpad = 16 if T <= 16 else 32
    slab = dev(rnd(s_in, mpad, 3 * d, seed=16))
    qw, kw = dev(1 + 0.1 * rnd(d, seed=17)), dev(1 + 0.1 * rnd(d, seed=18))
    mask = torch.ones(T, dtype=torch.uint8, device="cuda")
    mask[T - 2:] = 0                                         # right padding
    out = torch.zeros(T, d, device="cuda", dtype=torch.bfloat16)
    ws = torch.zeros(T, 3 * d, device="cuda")
    abi.check(lib.deer_mpt_attn_small(abi.ptr(slab), s_in, mpad * 3 * d, d, H, abi.ptr(qw) if qk_ln else None, abi.ptr(kw) if qk_ln else None,
                                      1e-5, abi.ptr(mask), 8.0, abi.ptr(ws), abi.ptr(out), 0, d, T, 1, None, st()), "mpt attn")
    outf = torch.zeros(T, d, device="cuda")
    abi.check(lib.deer_mpt_attn_small(abi.ptr(slab), s_in, mpad * 3 * d, d, H, abi.ptr(qw) if qk_ln else None, abi.ptr(kw) if qk_ln else None,
                                      1e-5, abi.ptr(mask), 8.0, abi.ptr(ws), abi.ptr(outf), 1, d, T, 1, None, st()), "mpt attn")
    torch.cuda.synchronize()
    qkv = slab.sum(0)[:T]
    q, k, v = qkv.chunk(3, -1)
    if qk_ln:
        q = torch.nn.functional.layer_norm(q, (d,), qw)
        k = torch.nn.functional.layer_norm(k, (d,), kw)
    hd = d // H
    q, k, v = (t.view(T, H, hd).transpose(0, 1) for t in (q, k, v))
    slopes = 2.0 ** (-8.0 * torch.arange(1, H + 1, device="cuda") / H)
    bias = -(T - 1 - torch.arange(T, device="cuda")).float().view(1, 1, T) * slopes.view(H, 1, 1)
    w = q @ k.transpose(-1, -2) * hd ** -0.5 + bias
    w = w.masked_fill(~mask.bool().view(1, 1, T), float("-inf"))
    w = w.masked_fill(torch.ones(T, T, device="cuda").triu(1).bool(), float("-inf"))
    ref = (torch.softmax(w, -1) @ v).transpose(0, 1).reshape(T, d)
    assert rel_err(out.float(), ref) < 4e-3                  # output rounded to bf16
    assert rel_err(outf, ref) < 2e-5


# ---------------------------------------------------------------------------------------------- row ops
def test_layernorm_rows_and_strides(lib):
    B, R, C = 2, 5, 1024
    x = dev(rnd(B, R + 1, C, seed=19))
    g, b = dev(1 + 0.1 * rnd(C, seed=20)), dev(0.1 * rnd(C, seed=21))
    out = torch.zeros(B, R + 3, C, device="cuda", dtype=torch.bfloat16)
    outf = torch.zeros(B, R + 3, C, device="cuda")
    # read rows 1..R of each batch, write them at rows 2..R+1
    abi.check(lib.deer_layernorm_rows(abi.ptr(x, C * 4), C, (R + 1) * C, R, B, abi.ptr(g), abi.ptr(b), abi.ptr(out, 2 * C * 2),
                                      abi.ptr(outf, 2 * C * 4), C, (R + 3) * C, C, 1e-5, st()), "ln")
    torch.cuda.synchronize()
    ref = torch.nn.functional.layer_norm(x[:, 1:], (C,), g, b)
    assert float((outf[:, 2:R + 2] - ref).abs().max()) < 2e-5
    assert rel_err(out[:, 2:R + 2].float(), ref) < 4e-3
    assert float(out[:, :2].abs().max()) == 0 and float(out[:, R + 2:].abs().max()) == 0


def test_layernorm_rows_multi(lib):
    """One statistics pass, L affine outputs (Perceiver norm_media of every layer); skips the class token row."""
    N, P, C, L = 2, 16, 1024, 6
    x = dev(rnd(N, P + 1, C, seed=81))
    g, b = dev(1 + 0.1 * rnd(L, C, seed=82)), dev(0.1 * rnd(L, C, seed=83))
    out = torch.zeros(L, N * P, C, device="cuda", dtype=torch.bfloat16)
    abi.check(lib.deer_layernorm_rows_multi(abi.ptr(x, C * 4), C, (P + 1) * C, P, N, abi.ptr(g), abi.ptr(b), L, C, abi.ptr(out),
                                            N * P * C, C, P * C, C, 1e-5, st()), "ln multi")
    torch.cuda.synchronize()
    for l in range(L):
        ref = torch.nn.functional.layer_norm(x[:, 1:], (C,), g[l], b[l]).reshape(N * P, C)
        assert rel_err(out[l].float(), ref) < 4e-3


def test_gemm_weight_batched(lib):
    """C[z] = A[z] W[z]^T with one weight per batch entry (Perceiver to_kv of all layers)."""
    L, M, N, K = 6, 96, 128, 256
    A = dev(rnd(L, M, K, seed=84)).bfloat16()
    W = dev(rnd(L, N, K, seed=85, scale=K ** -0.5)).bfloat16()
    C = torch.zeros(L, M, N, device="cuda", dtype=torch.bfloat16)
    abi.check(lib.deer_gemm_bf16_nt_wbatch(abi.ptr(A), K, M * K, abi.ptr(W), K, N * K, None, abi.ptr(C), N, M * N, M, N, K, L,
                                           abi.EPI_BF16, 0, None, st()), "wbatch")
    torch.cuda.synchronize()
    ref = torch.einsum("lmk,lnk->lmn", A.float(), W.float())
    assert rel_err(C.float(), ref) < 6e-3


def test_resadd_ln(lib):
    T, d, s_in = 14, 2048, 5
    x0 = dev(rnd(T, d, seed=22))
    slab = dev(rnd(s_in, 16, d, seed=23))
    gate = torch.tensor([-0.3], device="cuda")
    g = dev(1 + 0.1 * rnd(d, seed=24))
    x = x0.clone()
    out = torch.zeros(T, d, device="cuda", dtype=torch.bfloat16)
    outf = torch.zeros(T, d, device="cuda")
    cp = torch.zeros(T, d, device="cuda")
    abi.check(lib.deer_resadd_ln(abi.ptr(x), abi.ptr(slab), s_in, 16 * d, abi.ptr(gate), None, abi.ptr(g), None, abi.ptr(out), abi.ptr(outf), abi.ptr(cp), T, d, 1e-5,
                                 None, st()), "resadd")
    torch.cuda.synchronize()
    xr = x0 + math.tanh(-0.3) * slab.sum(0)[:T]
    assert float((x - xr).abs().max()) < 1e-5 and torch.equal(x, cp)
    assert rel_err(out.float(), torch.nn.functional.layer_norm(xr, (d,), g)) < 4e-3
    assert rel_err(outf, torch.nn.functional.layer_norm(xr, (d,), g)) < 1e-5
    # no slab, no LN: pure copy
    x2 = x0.clone()
    abi.check(lib.deer_resadd_ln(abi.ptr(x2), None, 0, 0, None, None, None, None, None, None, abi.ptr(cp), T, d, 1e-5, None, st()), "resadd")
    torch.cuda.synchronize()
    assert torch.equal(cp, x0)


@pytest.mark.parametrize("T,s_in,R", [(4112, 2, 2), (4112, 1, 4), (2057, 2, 2), (2059, 1, 4), (5, 2, 4), (4112, 0, 2)])
def test_resadd_ln_multirow_is_bit_identical_to_one_row_per_workgroup(lib, T, s_in, R):
    """Env-batch vision rows (d = 1024): R rows per workgroup, every load of every row in flight before the first use, one pair of block
    reductions for the R rows - x and LN(x) must equal the one-row kernel BIT for bit (ragged last workgroup, with / without slabs, gate)."""
    d = 1024
    x0 = dev(rnd(T, d, seed=31))
    slab = dev(rnd(max(s_in, 1), T, d, seed=32))
    bias, g, be = dev(rnd(d, seed=33)), dev(1 + 0.1 * rnd(d, seed=34)), dev(0.1 * rnd(d, seed=35))
    gate = torch.tensor([0.7], device="cuda")
    sp = abi.ptr(slab) if s_in else None
    for gp in (None, abi.ptr(gate)):
        xa, xb = x0.clone(), x0.clone()
        oa, ob = (torch.zeros(T, d, device="cuda", dtype=torch.bfloat16) for _ in range(2))
        abi.check(lib.deer_resadd_ln_multirow(abi.ptr(xb), sp, s_in, T * d, gp, abi.ptr(bias), abi.ptr(g), abi.ptr(be), abi.ptr(ob), T, d, 1e-5, R, st()), "multirow")
        if T < 2048:                                  # below the auto threshold deer_resadd_ln IS the one-row kernel
            abi.check(lib.deer_resadd_ln(abi.ptr(xa), sp, s_in, T * d, gp, abi.ptr(bias), abi.ptr(g), abi.ptr(be), abi.ptr(oa), None, None, T, d, 1e-5, None, st()), "one row")
        else:                                         # above it: the one-row kernel through its f32-output form (never routed to the multi-row kernel)
            of = torch.zeros(T, d, device="cuda")
            abi.check(lib.deer_resadd_ln(abi.ptr(xa), sp, s_in, T * d, gp, abi.ptr(bias), abi.ptr(g), abi.ptr(be), abi.ptr(oa), abi.ptr(of), None, T, d, 1e-5, None, st()),
                      "one row")
        torch.cuda.synchronize()
        assert torch.equal(xa, xb) and torch.equal(oa, ob), (T, s_in, R)
        ref = x0 + (math.tanh(0.7) if gp else 1.0) * (slab[:s_in].sum(0) + bias) if s_in else x0
        assert float((xb - ref).abs().max()) < 1e-5
        assert rel_err(ob.float(), torch.nn.functional.layer_norm(ref, (d,), g, be)) < 4e-3


@pytest.mark.parametrize("M,N,K,S", [(514, 1024, 4096, 4), (514, 1024, 1024, 2), (128, 1024, 4096, 8), (33, 256, 512, 2),
                                     (70, 128, 48, 2)])
def test_gemm_splitk_resadd_bias(lib, M, N, K, S):
    """Split-K slabs + deer_resadd_ln (bias, residual, LayerNorm) == x + A W^T + b, then LN (ViT c_proj / out_proj)."""
    A = dev(rnd(M, K, seed=61)).bfloat16()
    W = dev(rnd(N, K, seed=62, scale=K ** -0.5)).bfloat16()
    b = dev(rnd(N, seed=63))
    g, be = dev(1 + 0.1 * rnd(N, seed=64)), dev(0.1 * rnd(N, seed=65))
    x0 = dev(rnd(M, N, seed=66))
    slab = torch.full((S, M, N), float("nan"), device="cuda")
    abi.check(lib.deer_gemm_bf16_nt_splitk(abi.ptr(A), K, abi.ptr(W), K, abi.ptr(slab), M, N, K, S, 0, None, st()), "splitk")
    x = x0.clone()
    out = torch.zeros(M, N, device="cuda", dtype=torch.bfloat16)
    abi.check(lib.deer_resadd_ln(abi.ptr(x), abi.ptr(slab), S, M * N, None, abi.ptr(b), abi.ptr(g), abi.ptr(be), abi.ptr(out), None, None,
                                 M, N, 1e-5, None, st()), "resadd")
    torch.cuda.synchronize()
    ref = x0 + A.float() @ W.float().t() + b
    assert rel_err(x, ref) < 2e-5
    assert rel_err(out.float(), torch.nn.functional.layer_norm(ref, (N,), g, be)) < 4e-3
    assert lib.deer_gemm_bf16_nt_splitk(abi.ptr(A), K, abi.ptr(W), K, abi.ptr(slab), M, N, K, 3 if K % 3 else 7, 0, None, st()) == 1


def test_vit_patch_embed(lib):
    N, S, p, W = 2, 56, 14, 128
    gw, P, kk, Kpad = 4, 16, 588, 640
    img = dev(rnd(N, 3, S, S, seed=25))
    conv = dev(rnd(W, 3, p, p, seed=26, scale=kk ** -0.5))
    col = torch.full((N * P, Kpad), 9.0, device="cuda", dtype=torch.bfloat16)
    abi.check(lib.deer_vit_im2col(abi.ptr(img), 0, N, S, p, abi.ptr(col), Kpad, st()), "im2col")
    torch.cuda.synchronize()
    ref_col = torch.nn.functional.unfold(img, p, stride=p).transpose(1, 2).reshape(N * P, kk)
    assert torch.equal(col[:, :kk].float(), ref_col.to(torch.bfloat16).float())
    assert float(col[:, kk:].abs().max()) == 0.0
    imgb = img.to(torch.bfloat16)
    abi.check(lib.deer_vit_im2col(abi.ptr(imgb), 1, N, S, p, abi.ptr(col), Kpad, st()), "im2col")
    torch.cuda.synchronize()
    assert torch.equal(col[:, :kk].float(), ref_col.to(torch.bfloat16).float())
    patch = dev(rnd(N * P, W, seed=27))
    cls, pos = dev(rnd(W, seed=28)), dev(rnd(P + 1, W, seed=29))
    g, b = dev(1 + 0.1 * rnd(W, seed=30)), dev(0.1 * rnd(W, seed=31))
    x = torch.zeros(N, P + 1, W, device="cuda")
    abi.check(lib.deer_vit_embed_lnpre(abi.ptr(patch), abi.ptr(cls), abi.ptr(pos), abi.ptr(g), abi.ptr(b), abi.ptr(x), N, P, W, 1e-5, st()), "embed")
    torch.cuda.synchronize()
    ref = torch.cat([cls.view(1, 1, W).expand(N, 1, W), patch.view(N, P, W)], 1) + pos
    ref = torch.nn.functional.layer_norm(ref, (W,), g, b)
    assert float((x - ref).abs().max()) < 2e-5


def test_fp16_family_of_the_vision_tower(lib):
    """Round 6: the fp16 twins of the tower's remaining entry points (include/deer_hip.h: "the vision tower's fp16 arithmetic") against fp32
    torch math on fp16-exact inputs - two-segment attention, LayerNorm rows (one and several affine sets), the slab-reducing residual +
    LayerNorm row op (one row per workgroup and the multi-row form at >= 2048 rows), im2col from f32 / bf16 / fp16 frames, the
    weight-batched and the split-K GEMM.  One 16-bit rounding of a result is 2^-12 relative (bf16: 2^-9)."""
    h = torch.float16
    # two-segment attention (Perceiver)
    B, H, q_len, kv1, kv2, hd = 2, 8, 64, 256, 64, 64
    inner = H * hd
    qkv = dev(rnd(B, q_len, 3 * inner, seed=71), h)
    mkv = dev(rnd(B, kv1, 2 * inner, seed=73), h)
    o = torch.zeros(B, q_len, inner, device="cuda", dtype=h)
    abi.check(lib.deer_attn_f16_hd64_2seg(abi.ptr(qkv), abi.ptr(mkv), abi.ptr(mkv, inner * 2), abi.ptr(qkv, inner * 2), abi.ptr(qkv, 2 * inner * 2),
                                          abi.ptr(o), B, H, q_len, kv1, kv2, 3 * inner, 2 * inner, 3 * inner, inner, q_len * 3 * inner,
                                          kv1 * 2 * inner, kv2 * 3 * inner, q_len * inner, hd ** -0.5, st()), "attn 2seg f16")
    torch.cuda.synchronize()
    qf = qkv[..., :inner].float().view(B, q_len, H, hd).transpose(1, 2)
    kf = torch.cat([mkv[..., :inner], qkv[..., inner:2 * inner]], 1).float().view(B, kv1 + kv2, H, hd).transpose(1, 2)
    vf = torch.cat([mkv[..., inner:], qkv[..., 2 * inner:]], 1).float().view(B, kv1 + kv2, H, hd).transpose(1, 2)
    ref = (torch.softmax(qf @ kf.transpose(-1, -2) * hd ** -0.5, -1) @ vf).transpose(1, 2).reshape(B, q_len, inner)
    assert rel_err(o.float(), ref) < 1e-3
    # LayerNorm rows, one and several affine sets
    N, P, C, L = 2, 16, 1024, 6
    x = dev(rnd(N, P + 1, C, seed=81))
    g, b = dev(1 + 0.1 * rnd(L, C, seed=82)), dev(0.1 * rnd(L, C, seed=83))
    out = torch.zeros(L, N * P, C, device="cuda", dtype=h)
    abi.check(lib.deer_layernorm_rows_multi_f16(abi.ptr(x, C * 4), C, (P + 1) * C, P, N, abi.ptr(g), abi.ptr(b), L, C, abi.ptr(out),
                                                N * P * C, C, P * C, C, 1e-5, st()), "ln multi f16")
    out1 = torch.zeros(N * (P + 1), C, device="cuda", dtype=h)
    abi.check(lib.deer_layernorm_rows_f16(abi.ptr(x), C, 0, N * (P + 1), 1, abi.ptr(g), abi.ptr(b), abi.ptr(out1), None, C, 0, C, 1e-5, st()), "ln f16")
    torch.cuda.synchronize()
    for l in range(L):
        assert rel_err(out[l].float(), torch.nn.functional.layer_norm(x[:, 1:], (C,), g[l], b[l]).reshape(N * P, C)) < 5e-4
    assert rel_err(out1.float(), torch.nn.functional.layer_norm(x, (C,), g[0], b[0]).reshape(-1, C)) < 5e-4
    # residual + LayerNorm row op: 14 rows (one per workgroup) and 2057 rows (multi-row form), bit-identical x in both 16-bit formats
    for T in (14, 2057):
        d, s_in = 1024, 2
        x0 = dev(rnd(T, d, seed=22))
        slab = dev(rnd(s_in, T, d, seed=23))
        bias = dev(0.1 * rnd(d, seed=25))
        gm = dev(1 + 0.1 * rnd(d, seed=24))
        xa, xb = x0.clone(), x0.clone()
        oh = torch.zeros(T, d, device="cuda", dtype=h)
        ob = torch.zeros(T, d, device="cuda", dtype=torch.bfloat16)
        abi.check(lib.deer_resadd_ln_f16(abi.ptr(xa), abi.ptr(slab), s_in, T * d, None, abi.ptr(bias), abi.ptr(gm), None, abi.ptr(oh), None, None, T, d, 1e-5, None, st()), "resadd f16")
        abi.check(lib.deer_resadd_ln(abi.ptr(xb), abi.ptr(slab), s_in, T * d, None, abi.ptr(bias), abi.ptr(gm), None, abi.ptr(ob), None, None, T, d, 1e-5, None, st()), "resadd")
        torch.cuda.synchronize()
        xr = x0 + slab.sum(0) + bias
        assert torch.equal(xa, xb) and float((xa - xr).abs().max()) < 1e-5
        ln = torch.nn.functional.layer_norm(xr, (d,), gm)
        assert rel_err(oh.float(), ln) < 5e-4 and rel_err(ob.float(), ln) < 4e-3
    # im2col: f32 / bf16 / fp16 frames -> fp16 patches
    Ni, S, p = 2, 56, 14
    kk, Kpad = 588, 640
    img = dev(rnd(Ni, 3, S, S, seed=25))
    ref_col = torch.nn.functional.unfold(img, p, stride=p).transpose(1, 2).reshape(Ni * 16, kk)
    col = torch.full((Ni * 16, Kpad), 9.0, device="cuda", dtype=h)
    for kind, src in ((0, img), (1, img.to(torch.bfloat16)), (2, img.to(h))):
        col.fill_(9.0)
        abi.check(lib.deer_vit_im2col_f16(abi.ptr(src), kind, Ni, S, p, abi.ptr(col), Kpad, st()), "im2col f16")
        torch.cuda.synchronize()
        want = ref_col.to(h) if kind != 1 else ref_col.to(torch.bfloat16).to(h)
        assert torch.equal(col[:, :kk], want) and float(col[:, kk:].abs().max()) == 0.0
    # weight-batched and split-K GEMM
    Lb, M, Nn, K = 6, 96, 128, 256
    A = dev(rnd(Lb, M, K, seed=84), h)
    W = dev(rnd(Lb, Nn, K, seed=85, scale=K ** -0.5), h)
    Cc = torch.zeros(Lb, M, Nn, device="cuda", dtype=h)
    abi.check(lib.deer_gemm_f16_nt_wbatch(abi.ptr(A), K, M * K, abi.ptr(W), K, Nn * K, None, abi.ptr(Cc), Nn, M * Nn, M, Nn, K, Lb, abi.EPI_BF16, 0, None, st()), "wbatch f16")
    M2, N2, K2, S2 = 514, 1024, 4096, 2
    A2 = dev(rnd(M2, K2, seed=64), h)
    W2 = dev(rnd(N2, K2, seed=65, scale=K2 ** -0.5), h)
    slab2 = torch.zeros(S2, M2, N2, device="cuda")
    abi.check(lib.deer_gemm_f16_nt_splitk(abi.ptr(A2), K2, abi.ptr(W2), K2, abi.ptr(slab2), M2, N2, K2, S2, 0, None, st()), "splitk f16")
    torch.cuda.synchronize()
    assert rel_err(Cc.float(), torch.einsum("lmk,lnk->lmn", A.float(), W.float())) < 5e-4
    assert rel_err(slab2.sum(0), A2.float() @ W2.float().t()) < 2e-5


def test_embed_tokens_and_text_time(lib):
    T, d, V = 9, 256, 515
    wte = dev(rnd(V, d, seed=32), torch.bfloat16)
    ids = torch.tensor([513, 4, 77, 513, 5, 6, 512, 0, 514], device="cuda")
    x = torch.zeros(T, d, device="cuda")
    tt = torch.zeros(T, dtype=torch.int32, device="cuda")
    abi.check(lib.deer_embed_tokens(abi.ptr(ids), abi.ptr(wte), abi.ptr(x), abi.ptr(tt), T, 1, d, V, 513, st()), "embed")
    torch.cuda.synchronize()
    assert torch.equal(x, wte[ids].float())
    assert tt.tolist() == [1, 1, 1, 2, 2, 2, 2, 2, 2]
    lat = dev(rnd(64, d, seed=33))
    dst = torch.zeros(3, 64, d, device="cuda")
    abi.check(lib.deer_broadcast_rows(abi.ptr(lat), abi.ptr(dst), 64 * d, 3, st()), "bcast")
    torch.cuda.synchronize()
    assert torch.equal(dst, lat.expand(3, 64, d))


def test_batched_small_attention_and_embedding(lib):
    """batch = several environments per launch: rows [env][T]; each env has its own media rows / key mask / media count."""
    B, T, d, H = 3, 9, 256, 2
    s_in, mpad = 2, 32
    slab = dev(rnd(s_in, mpad, 3 * d, seed=51))
    qw, kw = dev(1 + 0.1 * rnd(d, seed=52)), dev(1 + 0.1 * rnd(d, seed=53))
    mask = torch.ones(B, T, dtype=torch.uint8, device="cuda")
    mask[1, T - 3:] = 0
    ws = torch.zeros(B * T, 3 * d, device="cuda")
    out = torch.zeros(B * T, d, device="cuda")
    abi.check(lib.deer_mpt_attn_small(abi.ptr(slab), s_in, mpad * 3 * d, d, H, abi.ptr(qw), abi.ptr(kw), 1e-5, abi.ptr(mask), 8.0, abi.ptr(ws),
                                      abi.ptr(out), 1, d, T, B, None, st()), "mpt attn")
    torch.cuda.synchronize()
    qkv = slab.sum(0)[:B * T].view(B, T, 3 * d)
    hd = d // H
    slopes = 2.0 ** (-8.0 * torch.arange(1, H + 1, device="cuda") / H)
    bias = -(T - 1 - torch.arange(T, device="cuda")).float().view(1, 1, T) * slopes.view(H, 1, 1)
    for b in range(B):
        q, k, v = qkv[b].chunk(3, -1)
        q = torch.nn.functional.layer_norm(q, (d,), qw)
        k = torch.nn.functional.layer_norm(k, (d,), kw)
        q, k, v = (t.view(T, H, hd).transpose(0, 1) for t in (q, k, v))
        w = q @ k.transpose(-1, -2) * hd ** -0.5 + bias
        w = w.masked_fill(~mask[b].bool().view(1, 1, T), float("-inf"))
        w = w.masked_fill(torch.ones(T, T, device="cuda").triu(1).bool(), float("-inf"))
        ref = (torch.softmax(w, -1) @ v).transpose(0, 1).reshape(T, d)
        assert rel_err(out[b * T:(b + 1) * T], ref) < 2e-5, b
    # x-attn: env b attends only to ITS media rows
    n_kv, heads, inner = 128, 8, 512
    qs = dev(rnd(1, 32, inner, seed=54))
    kv = dev(rnd(B * n_kv, 2 * inner, seed=55), torch.bfloat16)
    tt = torch.ones(B * T, dtype=torch.int32, device="cuda")
    xo = torch.zeros(B * T, inner, device="cuda")
    abi.check(lib.deer_xattn_small(abi.ptr(qs), 1, 32 * inner, inner, abi.ptr(kv), 2 * inner, inner, abi.ptr(tt), 128, abi.ptr(xo), 1, inner, T,
                                   n_kv, heads, B, 64 ** -0.5, None, st()), "xattn")
    torch.cuda.synchronize()
    for b in range(B):
        q = qs[0, b * T:(b + 1) * T].view(T, heads, 64).transpose(0, 1) * 64 ** -0.5
        kb = kv[b * n_kv:(b + 1) * n_kv]
        k = kb[:, :inner].float().view(n_kv, heads, 64).transpose(0, 1)
        v = kb[:, inner:].float().view(n_kv, heads, 64).transpose(0, 1)
        ref = (torch.softmax(q @ k.transpose(-1, -2), -1) @ v).transpose(0, 1).reshape(T, inner)
        assert rel_err(xo[b * T:(b + 1) * T], ref) < 2e-5, b
    # embedding: media count restarts per environment
    V = 515
    wte = dev(rnd(V, d, seed=56), torch.bfloat16)
    ids = torch.tensor([[513, 4, 5, 513, 6, 7, 8, 9, 0], [1, 2, 513, 3, 4, 5, 6, 7, 8], [513, 1, 1, 1, 1, 1, 1, 1, 1]], device="cuda")
    x = torch.zeros(B * T, d, device="cuda")
    t2 = torch.zeros(B * T, dtype=torch.int32, device="cuda")
    abi.check(lib.deer_embed_tokens(abi.ptr(ids), abi.ptr(wte), abi.ptr(x), abi.ptr(t2), T, B, d, V, 513, st()), "embed")
    torch.cuda.synchronize()
    assert torch.equal(x, wte[ids.reshape(-1)].float())
    assert t2.view(B, T).tolist() == [[1, 1, 1, 2, 2, 2, 2, 2, 2], [0, 0, 1, 1, 1, 1, 1, 1, 1], [1] * 9]


def test_xattn_mfma_matches_fp32_implementation(lib):
    """The MFMA x-attn path (q reduced from f32 slabs -> bf16, P in bf16) against the fp32 VALU implementation and torch."""
    B, T, n_kv, heads, inner = 3, 14, 128, 8, 512
    ldkv = 2 * inner * 2                                     # two layers' K/V side by side; use the second
    s_in = 4
    qs = dev(rnd(s_in, 64, inner, seed=61))
    kv = dev(rnd(B * n_kv, ldkv, seed=62), torch.bfloat16)
    tt = torch.ones(B * T, dtype=torch.int32, device="cuda")
    tt[T + 2] = 0                                            # env 1, token 2: no preceding media -> zero row
    off = 2 * inner
    o_ref = torch.zeros(B * T, inner, device="cuda")
    o_mf = torch.zeros(B * T, inner, device="cuda")
    args = (abi.ptr(qs), s_in, 64 * inner, inner, abi.ptr(kv, off * 2), ldkv, inner, abi.ptr(tt), 128)
    abi.check(lib.deer_xattn_small(*args, abi.ptr(o_ref), 1, inner, T, n_kv, heads, B, 64 ** -0.5, None, st()), "small")
    abi.check(lib.deer_xattn_mfma(*args, abi.ptr(o_mf), 1, inner, T, n_kv, heads, B, 64 ** -0.5, None, st()), "mfma")
    torch.cuda.synchronize()
    assert rel_err(o_mf, o_ref) < 6e-3
    assert float(o_mf[T + 2].abs().max()) == 0.0
    for b in range(B):
        q = qs.sum(0)[b * T:(b + 1) * T].view(T, heads, 64).transpose(0, 1) * 64 ** -0.5
        kb = kv[b * n_kv:(b + 1) * n_kv, off:off + 2 * inner]
        k = kb[:, :inner].float().view(n_kv, heads, 64).transpose(0, 1)
        v = kb[:, inner:].float().view(n_kv, heads, 64).transpose(0, 1)
        ref = (torch.softmax(q @ k.transpose(-1, -2), -1) @ v).transpose(0, 1).reshape(T, inner)
        if b == 1:
            ref[2] = 0
        assert rel_err(o_mf[b * T:(b + 1) * T], ref) < 6e-3, b
    # bf16 output + exit flag
    ob = torch.full((B * T, inner), 3.0, device="cuda", dtype=torch.bfloat16)
    ctl = torch.zeros(abi.CTL_WORDS, dtype=torch.int32, device="cuda")
    ctl[abi.CTL_ALL_EXITED] = 1
    abi.check(lib.deer_xattn_mfma(*args, abi.ptr(ob), 0, inner, T, n_kv, heads, B, 64 ** -0.5, abi.ptr(ctl), st()), "mfma")
    torch.cuda.synchronize()
    assert float(ob.float().min()) == 3.0


@pytest.mark.parametrize("B,T,d", [(1, 14, 2048), (3, 11, 256), (2, 20, 2048), (8, 14, 2048), (1, 14, 4096)])
@pytest.mark.parametrize("dt", ["bf16", "f16"])
def test_xattn_fused_matches_torch_and_the_three_kernel_path(lib, dt, B, T, d):
    """to_q -> masked cross-attention -> to_out (helpers.py:184-233) as ONE launch, one f32 slab per head: against fp64 torch
    math on the same bf16 weights, and against the unfused path (skinny to_q, deer_xattn_mfma, skinny to_out)."""
    heads, inner, n_kv = 8, 512, 128
    tdt, _ = _fmt(dt)
    sfx = "_f16" if dt == "f16" else ""
    ldkv = 2 * inner * 2
    off = 2 * inner
    R = B * T
    xn = dev(rnd(R, d, seed=71))
    Wq = dev(rnd(inner, d, seed=72, scale=d ** -0.5), tdt)
    Wo = dev(rnd(d, inner, seed=73, scale=inner ** -0.5), tdt)
    kv = dev(rnd(B * n_kv, ldkv, seed=74), tdt)
    tt = torch.ones(R, dtype=torch.int32, device="cuda")
    tt[R - 3] = 0                                            # a token without preceding media -> zero attention row
    Wq_p, Wo_p = torch.empty_like(Wq), torch.empty_like(Wo)
    abi.check(lib.deer_pack_weight_mfma16(abi.ptr(Wq), abi.ptr(Wq_p), inner, d, st()), "pack")
    abi.check(lib.deer_pack_weight_mfma16(abi.ptr(Wo), abi.ptr(Wo_p), d, inner, st()), "pack")
    mpad = abi.skinny_mpad(R)
    out = torch.full((heads, mpad, d), float("nan"), device="cuda")
    abi.check(getattr(lib, "deer_xattn_fused" + sfx)(abi.ptr(xn), d, abi.ptr(Wq_p), abi.ptr(kv, off * 2), ldkv, inner, abi.ptr(tt), 128, n_kv, abi.ptr(Wo_p),
                                   abi.ptr(out), mpad * d, T, heads, B, 64 ** -0.5, None, st()), "fused")
    torch.cuda.synchronize()
    y = out[:, :R].sum(0)
    assert torch.isfinite(y).all()
    # fp64 reference
    ref = torch.zeros(R, d, dtype=torch.float64, device="cuda")
    for b in range(B):
        q = (xn[b * T:(b + 1) * T].double() @ Wq.double().t()).view(T, heads, 64).transpose(0, 1) * 64 ** -0.5
        kb = kv[b * n_kv:(b + 1) * n_kv, off:off + 2 * inner].double()
        k = kb[:, :inner].view(n_kv, heads, 64).transpose(0, 1)
        v = kb[:, inner:].view(n_kv, heads, 64).transpose(0, 1)
        a = torch.softmax(q @ k.transpose(-1, -2), -1)
        o = (a @ v).transpose(0, 1).reshape(T, inner)
        for t in range(T):
            if int(tt[b * T + t]) == 0:
                o[t] = 0
        ref[b * T:(b + 1) * T] = o @ Wo.double().t()
    assert rel_err(y, ref.float()) < (8e-3 if dt == "bf16" else 1.2e-3)   # q and P enter the MFMAs in the 16-bit format (like the unfused kernels)
    assert float(y[R - 3].abs().max()) == 0.0
    # unfused path on the same inputs
    S = lib.deer_skinny_splitk(R, inner, d)
    qslab = torch.zeros(S, mpad, inner, device="cuda")
    abi.check(getattr(lib, "deer_gemm_skinny" + sfx)(abi.ptr(xn), d, None, 0, 0, abi.A_F32, abi.ptr(Wq_p), abi.ptr(qslab), R, inner, d, S, None, st()), "q")
    ao = torch.zeros(R, inner, device="cuda")
    abi.check(getattr(lib, "deer_xattn_mfma" + sfx)(abi.ptr(qslab), S, mpad * inner, inner, abi.ptr(kv, off * 2), ldkv, inner, abi.ptr(tt), 128, abi.ptr(ao), 1, inner,
                                  T, n_kv, heads, B, 64 ** -0.5, None, st()), "xattn")
    S2 = lib.deer_skinny_splitk(R, d, inner)
    yslab = torch.zeros(S2, mpad, d, device="cuda")
    abi.check(getattr(lib, "deer_gemm_skinny" + sfx)(abi.ptr(ao), inner, None, 0, 0, abi.A_F32, abi.ptr(Wo_p), abi.ptr(yslab), R, d, inner, S2, None, st()), "o")
    torch.cuda.synchronize()
    assert rel_err(y, yslab.sum(0)[:R]) < (2e-3 if dt == "bf16" else 4e-4)   # same arithmetic, different summation order of the q projection
    # exit flag: nothing is written
    ctl = torch.zeros(abi.CTL_WORDS, dtype=torch.int32, device="cuda")
    ctl[abi.CTL_ALL_EXITED] = 1
    out2 = torch.full((heads, mpad, d), 7.0, device="cuda")
    abi.check(getattr(lib, "deer_xattn_fused" + sfx)(abi.ptr(xn), d, abi.ptr(Wq_p), abi.ptr(kv, off * 2), ldkv, inner, abi.ptr(tt), 128, n_kv, abi.ptr(Wo_p),
                                   abi.ptr(out2), mpad * d, T, heads, B, 64 ** -0.5, abi.ptr(ctl), st()), "fused")
    torch.cuda.synchronize()
    assert float(out2.min()) == 7.0


# ------------------------------------------------------------------------------------------ action head, LSTM layer
@pytest.mark.parametrize("B,in_dim,ln", [(1, 2048, False), (3, 1024, True), (8, 2048, False), (8, 1024, True), (8, 4096, False), (5, 4096, False)])
def test_head_lstm_layer_env_batch(lib, B, in_dim, ln):
    """One LSTM cell step for B environments sharing the weight stream (action_head.py:548-558; torch.nn.LSTM gate order i,f,g,o),
    optional LayerNorm of the input (LayerNormLSTM).  8 x 4096 inputs do not fit the LDS: the launcher splits the batch."""
    H = 1024
    x = dev(rnd(B, in_dim, seed=1))
    h0, c0 = dev(rnd(B, H, seed=2, scale=0.5)), dev(rnd(B, H, seed=3, scale=0.5))
    wih = dev(rnd(4 * H, in_dim, seed=4, scale=in_dim ** -0.5), torch.bfloat16)
    whh = dev(rnd(4 * H, H, seed=5, scale=H ** -0.5), torch.bfloat16)
    bih, bhh = dev(rnd(4 * H, seed=6, scale=0.1)), dev(rnd(4 * H, seed=7, scale=0.1))
    lw, lb = dev(1.0 + rnd(in_dim, seed=8, scale=0.1)), dev(rnd(in_dim, seed=9, scale=0.1))
    h1 = torch.full((B, H), float("nan"), device="cuda")
    c1 = torch.full((B, H), float("nan"), device="cuda")
    mode = 3 if ln else 0                                  # X_LN / X_RAW
    abi.check(lib.deer_head_lstm_layer(abi.ptr(x), in_dim, mode, 1, in_dim, abi.ptr(lw) if ln else None, abi.ptr(lb) if ln else None,
                                       abi.ptr(wih), abi.ptr(whh), abi.ptr(bih), abi.ptr(bhh), abi.ptr(h0), abi.ptr(c0), abi.ptr(h1),
                                       abi.ptr(c1), H, B, 1e-5, None, 2, 0, 0, st()), "lstm")
    torch.cuda.synchronize()
    xin = torch.nn.functional.layer_norm(x, (in_dim,), lw, lb, 1e-5) if ln else x
    gates = xin @ wih.float().t() + bih + h0 @ whh.float().t() + bhh
    i, f, g, o = gates.chunk(4, dim=1)
    c_ref = torch.sigmoid(f) * c0 + torch.sigmoid(i) * torch.tanh(g)
    h_ref = torch.sigmoid(o) * torch.tanh(c_ref)
    assert rel_err(c1, c_ref) < 1e-5 and rel_err(h1, h_ref) < 1e-5          # fp32 arithmetic on both sides: summation order only


@pytest.mark.parametrize("B,L", [(1, 4), (8, 4), (3, 2)])
def test_lstm_recurrent_half_once_per_step_equals_the_fused_layer(lib, B, L):
    """deer_head_lstm_hh (W_hh h_prev + b_hh of every layer in one launch) + deer_head_lstm_layer_pre (streams W_ih only) against the fused
    deer_head_lstm_layer and the torch arithmetic: every head evaluation of a control step starts from the state the previous step committed."""
    H, in_dim = 1024, 2048
    hs, cs = dev(rnd(L, B, H, seed=2, scale=0.5)), dev(rnd(L, B, H, seed=3, scale=0.5))
    whh = [dev(rnd(4 * H, H, seed=50 + l, scale=H ** -0.5), torch.bfloat16) for l in range(L)]
    bhh = [dev(rnd(4 * H, seed=60 + l, scale=0.1)) for l in range(L)]
    ghh = torch.full((L, B, 4 * H), float("nan"), device="cuda")
    wp = (ctypes.c_void_p * L)(*[w.data_ptr() for w in whh])
    bp = (ctypes.c_void_p * L)(*[b.data_ptr() for b in bhh])
    abi.check(lib.deer_head_lstm_hh(wp, bp, L, abi.ptr(hs), abi.ptr(ghh), H, B, 0, st()), "lstm_hh")
    torch.cuda.synchronize()
    for l in range(L):
        assert rel_err(ghh[l], hs[l] @ whh[l].float().t() + bhh[l]) < 1e-5
    x = dev(rnd(B, in_dim, seed=1))
    wih = dev(rnd(4 * H, in_dim, seed=4, scale=in_dim ** -0.5), torch.bfloat16)
    bih = dev(rnd(4 * H, seed=6, scale=0.1))
    out = []
    for pre in (False, True):
        h1, c1 = (torch.full((B, H), float("nan"), device="cuda") for _ in range(2))
        if pre:
            abi.check(lib.deer_head_lstm_layer_pre(abi.ptr(x), in_dim, 0, 1, in_dim, None, None, abi.ptr(wih), abi.ptr(bih), abi.ptr(ghh[0]), abi.ptr(cs[0]),
                                                   abi.ptr(h1), abi.ptr(c1), H, B, 1e-5, None, 2, 0, 0, st()), "lstm pre")
        else:
            abi.check(lib.deer_head_lstm_layer(abi.ptr(x), in_dim, 0, 1, in_dim, None, None, abi.ptr(wih), abi.ptr(whh[0]), abi.ptr(bih), abi.ptr(bhh[0]),
                                               abi.ptr(hs[0]), abi.ptr(cs[0]), abi.ptr(h1), abi.ptr(c1), H, B, 1e-5, None, 2, 0, 0, st()), "lstm")
        torch.cuda.synchronize()
        out.append((h1, c1))
    assert rel_err(out[1][0], out[0][0]) < 1e-6 and rel_err(out[1][1], out[0][1]) < 1e-6      # summation grouping only
    gates = x @ wih.float().t() + bih + hs[0] @ whh[0].float().t() + bhh[0]
    i, f, g, o = gates.chunk(4, dim=1)
    c_ref = torch.sigmoid(f) * cs[0] + torch.sigmoid(i) * torch.tanh(g)
    assert rel_err(out[1][1], c_ref) < 1e-5 and rel_err(out[1][0], torch.sigmoid(o) * torch.tanh(c_ref)) < 1e-5


# ------------------------------------------------------------------------------------------ fp32-activation arithmetic (precise.hip)
@pytest.mark.parametrize("M,N,K,epi", [(257, 1024, 1024, 0), (514, 4096, 1024, 1), (128, 1024, 4096, 2), (70, 132, 72, 0), (514, 1024, 1024, 3)])
def test_gemm_f32_exact_products(lib, M, N, K, epi):
    """f32 x f32 with the exact-f32 MFMA: differs from torch fp64 math by fp32 rounding of the accumulation only (1e-6)."""
    A = dev(rnd(M, K, seed=1))
    W = dev(rnd(N, K, seed=2, scale=K ** -0.5))
    bias = dev(rnd(N, seed=3, scale=0.1))
    C0 = dev(rnd(M, N, seed=4))
    C = C0.clone()
    abi.check(lib.deer_gemm_f32_nt(abi.ptr(A), K, abi.ptr(W), K, abi.ptr(bias), abi.ptr(C), N, M, N, K, epi, st()), "gemm_f32")
    torch.cuda.synchronize()
    y = A.double() @ W.double().t() + bias.double()
    if epi == 1:
        y = y * torch.sigmoid(1.702 * y)
    elif epi == 2:
        y = torch.nn.functional.gelu(y)
    elif epi == 3:
        y = C0.double() + y
    assert rel_err(C, y) < 2e-6


@pytest.mark.parametrize("q_len,kv1,kv2,heads,batch", [(257, 257, 0, 16, 2), (64, 256, 64, 8, 2), (5, 7, 0, 2, 1),
                                                       (64, 512, 64, 8, 2), (33, 353, 0, 2, 1), (40, 300, 271, 1, 2)])   # > 352 keys: the chunked kernel
def test_attn_f32_one_and_two_segments(lib, q_len, kv1, kv2, heads, batch):
    C = heads * 64
    q = dev(rnd(batch, q_len, C, seed=1))
    k1, v1 = dev(rnd(batch, kv1, C, seed=2)), dev(rnd(batch, kv1, C, seed=3))
    k2 = dev(rnd(batch, max(kv2, 1), C, seed=4))
    v2 = dev(rnd(batch, max(kv2, 1), C, seed=5))
    out = torch.full((batch, q_len, C), float("nan"), device="cuda")
    abi.check(lib.deer_attn_f32(abi.ptr(q), abi.ptr(k1), abi.ptr(v1), abi.ptr(k2) if kv2 else None, abi.ptr(v2) if kv2 else None, abi.ptr(out), batch,
                                heads, q_len, kv1, kv2, C, C, C, C, q_len * C, kv1 * C, max(kv2, 1) * C, q_len * C, 0.125, st()), "attn_f32")
    torch.cuda.synchronize()
    kk = torch.cat([k1, k2], 1) if kv2 else k1
    vv = torch.cat([v1, v2], 1) if kv2 else v1
    sp = lambda t: t.view(batch, -1, heads, 64).transpose(1, 2).double()
    ref = (torch.softmax(sp(q) @ sp(kk).transpose(-1, -2) * 0.125, -1) @ sp(vv)).transpose(1, 2).reshape(batch, q_len, C)
    assert rel_err(out, ref) < 2e-6


def test_xattn_f32_matches_the_masked_cross_attention(lib):
    T, n_kv, heads, batch, npm = 14, 128, 8, 2, 64
    inner = heads * 64
    S = 2
    qs = dev(rnd(S, batch * T, inner, seed=1))
    kv = dev(rnd(batch, n_kv, 2 * inner, seed=2))
    tt = torch.tensor([[0, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2, 2, 2], [1] * 14], dtype=torch.int32, device="cuda")
    out = torch.full((batch, T, inner), float("nan"), device="cuda")
    abi.check(lib.deer_xattn_f32(abi.ptr(qs), S, batch * T * inner, inner, abi.ptr(kv), 2 * inner, inner, abi.ptr(tt), npm, abi.ptr(out), inner, T, n_kv,
                                 heads, batch, 0.125, None, st()), "xattn_f32")
    torch.cuda.synchronize()
    q = qs.sum(0).view(batch, T, heads, 64).transpose(1, 2).double() * 0.125
    k = kv[..., :inner].reshape(batch, n_kv, heads, 64).transpose(1, 2).double()
    v = kv[..., inner:].reshape(batch, n_kv, heads, 64).transpose(1, 2).double()
    sim = q @ k.transpose(-1, -2)
    media_time = (torch.arange(n_kv, device="cuda") // npm + 1).view(1, 1, 1, n_kv)
    sim = sim.masked_fill(tt.view(batch, 1, T, 1) != media_time, -torch.finfo(torch.float32).max)
    att = torch.softmax(sim, -1).masked_fill((tt == 0).view(batch, 1, T, 1), 0.0)
    ref = (att @ v).transpose(1, 2).reshape(batch, T, inner)
    assert rel_err(out, ref) < 2e-6


# ------------------------------------------------------------------------------------------ one-environment trunk (csrc/trunk_r16.hip)
def _pack(lib, W):
    Wp = torch.empty_like(W)
    abi.check(lib.deer_pack_weight_mfma16(abi.ptr(W), abi.ptr(Wp), W.shape[0], W.shape[1], st()), "pack")
    return Wp


def _ln(x, g, b, eps=1e-5):
    return torch.nn.functional.layer_norm(x.double(), (x.shape[-1],), g.double(), None if b is None else b.double(), eps)


def _pack_planes(a, dt=torch.bfloat16):
    """f32 [T <= 16, K] -> 16-bit hi / lo planes in MFMA-fragment order [K/32][64][8], lane = 16 * (k % 32 / 8) + row (rows >= T zero)"""
    T, K = a.shape
    hi = a.to(dt)
    lo = (a - hi.float()).to(dt)
    out = []
    for p in (hi, lo):
        full = torch.zeros(16, K, dtype=dt, device=a.device)
        full[:T] = p
        out.append(full.view(16, K // 32, 4, 8).permute(1, 2, 0, 3).contiguous().view(-1))     # [kt][g][row][8]
    return out[0], out[1], hi, lo


@pytest.mark.parametrize("T", [1, 14, 16])
@pytest.mark.parametrize("K,N", [(2048, 8192), (2048, 6144), (2048, 512), (256, 768), (256, 1024), (4096, 12288), (4096, 16384), (4096, 1024)])
@pytest.mark.parametrize("epi", [0, 1, 2])
@pytest.mark.parametrize("dt", ["bf16", "f16"])
def test_trunk_wide_gemm(lib, dt, T, K, N, epi):
    """The wide Linears at <= 16 rows with the K split inside the workgroup (final results, no slabs): plain f32 / exact GELU -> bf16
    hi + lo planes / f32 + 32-column moments, against fp64 torch math on the same bf16 operands (hi + lo).  3e-5 relative (fp32
    accumulation order); GELU planes reproduce the f32 value to bf16^2; rows >= T are never written."""
    if K == 4096 and epi == 2:
        pytest.skip("K = 4096 (MPT-7B: 16 columns per workgroup) has no 32-column moments epilogue - MPT-7B has no q/k LayerNorm")
    if dt == "f16" and (T == 1 or (K, N) in ((2048, 512), (256, 768), (4096, 1024))):
        pytest.skip("fp16 twin: boundary shapes only")
    tdt, _ = _fmt(dt)
    A = dev(rnd(T, K, seed=3))
    W = dev(rnd(N, K, seed=7, scale=K ** -0.5), tdt)
    Wp = _pack(lib, W)
    ph, pl, hi, lo = _pack_planes(A, tdt)
    out = torch.full((16, N), float("nan"), device="cuda")
    oh = torch.zeros(16, N, device="cuda", dtype=tdt)
    ol = torch.zeros(16, N, device="cuda", dtype=tdt)
    stats = torch.full((N // 32, 16, 2), float("nan"), device="cuda")
    abi.check(getattr(lib, "deer_trunk_wide_gemm" + ("_f16" if dt == "f16" else ""))(abi.ptr(ph), abi.ptr(pl), abi.ptr(Wp), N, K, epi, abi.ptr(out), abi.ptr(oh), abi.ptr(ol), N, abi.ptr(stats), T, None,
                                       st()), "trunk_wide_gemm")
    torch.cuda.synchronize()
    y = (hi.double() + lo.double()) @ W.double().t()
    if epi == 1:
        ref = torch.nn.functional.gelu(y)
        assert rel_err(oh[:T].double() + ol[:T].double(), ref) < 1e-4
        assert float((oh[:T].float() - ref.float()).abs().max()) <= float(ref.abs().max()) * (2 ** -7 if dt == "bf16" else 2 ** -10)       # hi = fmt(gelu)
        assert T == 16 or (float(oh[T:].abs().max()) == 0.0 and float(ol[T:].abs().max()) == 0.0)
    else:
        assert rel_err(out[:T], y.float()) < 3e-5
        assert torch.isnan(out[T:]).all()
        if epi == 2:
            yg = out[:T].double().view(T, N // 32, 32)
            mu = yg.mean(-1)
            m2 = ((yg - mu.unsqueeze(-1)) ** 2).sum(-1)
            assert float((stats[:, :T, 0].double() - mu.t()).abs().max()) < 1e-5
            assert rel_err(stats[:, :T, 1], m2.t().float()) < 1e-5


@pytest.mark.parametrize("T", [1, 14, 16])
@pytest.mark.parametrize("d", [256, 2048, 4096])
@pytest.mark.parametrize("dt", ["bf16", "f16"])
def test_resadd_ln_packed_is_the_split_form_in_fragment_order(lib, dt, T, d):
    """deer_resadd_ln_packed = deer_resadd_ln_split with the planes permuted into MFMA-fragment order: the residual stream is
    bit-identical; the LayerNorm output (hi + lo) agrees to fp32 rounding (the packed form reduces the row statistics over 512 threads
    instead of 256: another summation order)."""
    tdt, _ = _fmt(dt)
    sfx = "_f16" if dt == "f16" else ""
    x0 = dev(rnd(T, d, seed=31))
    slab = dev(rnd(3, 16, d, seed=32, scale=0.2))
    gate = dev(torch.tensor([0.3]))
    gamma, beta = dev(1.0 + 0.1 * rnd(d, seed=33)), dev(0.1 * rnd(d, seed=34))
    outs = []
    for packed in (False, True):
        x = x0.clone()
        hi = torch.zeros(16 * d, device="cuda", dtype=tdt)
        lo = torch.zeros(16 * d, device="cuda", dtype=tdt)
        fn = getattr(lib, ("deer_resadd_ln_packed" if packed else "deer_resadd_ln_split") + sfx)
        abi.check(fn(abi.ptr(x), abi.ptr(slab), 3, 16 * d, abi.ptr(gate), None, abi.ptr(gamma), abi.ptr(beta), abi.ptr(hi), abi.ptr(lo), None, None, T, d,
                     1e-5, None, st()), "resadd")
        torch.cuda.synchronize()
        outs.append((x, hi, lo))
    assert torch.equal(outs[0][0], outs[1][0])
    unpack = lambda p: p.view(d // 32, 4, 16, 8).permute(2, 0, 1, 3).reshape(16, d)
    y_rm = outs[0][1].view(16, d).double() + outs[0][2].view(16, d).double()
    y_pk = unpack(outs[1][1]).double() + unpack(outs[1][2]).double()
    assert float((y_rm[:T] - y_pk[:T]).abs().max()) < 2e-5 * float(y_rm[:T].abs().max())
    xn = x0.double() + math.tanh(0.3) * slab[:, :T].double().sum(0)
    assert rel_err(y_pk[:T], _ln(xn, gamma, beta).float()) < 1e-5


@pytest.mark.parametrize("T", [3, 14, 16])
@pytest.mark.parametrize("d", [256, 2048])
def test_xattn_fused_packed_matches_f32_operand_form(lib, T, d):
    """deer_xattn_fused_packed (LN(x) as fragment-ordered bf16 hi / lo planes) against deer_xattn_fused on the f32 LN(x): the same
    hi + lo values enter the same MFMAs in the same order -> bit-identical head slabs."""
    heads, inner, n_kv = 8, 512, 128
    xn = dev(rnd(T, d, seed=51))
    Wq = _pack(lib, dev(rnd(inner, d, seed=52, scale=d ** -0.5), torch.bfloat16))
    Wo = _pack(lib, dev(rnd(d, inner, seed=53, scale=inner ** -0.5), torch.bfloat16))
    kv = dev(rnd(n_kv, 2 * inner, seed=54), torch.bfloat16)
    tt = dev(torch.ones(T, dtype=torch.int32))
    ph, pl, _, _ = _pack_planes(xn)
    a = torch.full((heads, 16, d), float("nan"), device="cuda")
    b = torch.full((heads, 16, d), float("nan"), device="cuda")
    abi.check(lib.deer_xattn_fused(abi.ptr(xn), d, abi.ptr(Wq), abi.ptr(kv), 2 * inner, inner, abi.ptr(tt), 64, n_kv, abi.ptr(Wo), abi.ptr(a), 16 * d, T,
                                   heads, 1, 0.125, None, st()), "xattn_fused")
    abi.check(lib.deer_xattn_fused_packed(abi.ptr(ph), abi.ptr(pl), d, abi.ptr(Wq), abi.ptr(kv), 2 * inner, inner, abi.ptr(tt), 64, n_kv, abi.ptr(Wo),
                                          abi.ptr(b), 16 * d, T, heads, 0.125, None, st()), "xattn_fused_packed")
    torch.cuda.synchronize()
    assert torch.isfinite(a[:, :T]).all() and torch.equal(a[:, :T], b[:, :T])


@pytest.mark.parametrize("T", [1, 5, 14, 16])
@pytest.mark.parametrize("d,heads", [(2048, 16), (256, 2), (4096, 32), (256, 4), (768, 32)])   # head widths 128 (unrolled form) / 64 / 24; 16 moment groups per part at 4096
@pytest.mark.parametrize("qk_ln,mask", [(True, False), (True, True), (False, False)])
@pytest.mark.parametrize("dt", ["bf16", "f16"])
def test_trunk_mpt_attn(lib, dt, T, d, heads, qk_ln, mask):
    """MPT attention on final q|k|v with the q/k LayerNorm over d_model reconstructed from 32-column moments (Chan's combination) -
    against fp64 torch math (LayerNorm over the full row, ALiBi slopes 2^(-8(h+1)/H), causal + key-padding mask)."""
    if dt == "f16" and (T in (1, 5) or d in (256, 768)):
        pytest.skip("fp16 twin: boundary shapes only")
    tdt, _ = _fmt(dt)
    qkv = dev(rnd(T, 3 * d, seed=21) * 1.5 + 0.3)
    gq, gk = dev(1.0 + 0.1 * rnd(d, seed=22)), dev(1.0 + 0.1 * rnd(d, seed=23))
    g32 = qkv.double().view(T, 3 * d // 32, 32)
    mu = g32.mean(-1)
    stats = torch.zeros(3 * d // 32, 16, 2, device="cuda")
    stats[:, :T, 0] = mu.t().float()
    stats[:, :T, 1] = ((g32 - mu.unsqueeze(-1)) ** 2).sum(-1).t().float()
    km = torch.ones(T, dtype=torch.uint8)
    if mask and T > 2:
        km[T - 2:] = 0                                           # right padding
    hi = torch.zeros(16, d, device="cuda", dtype=tdt)
    lo = torch.zeros(16, d, device="cuda", dtype=tdt)
    abi.check(getattr(lib, "deer_trunk_mpt_attn" + ("_f16" if dt == "f16" else ""))(abi.ptr(qkv), abi.ptr(stats), d, heads, abi.ptr(gq) if qk_ln else None, abi.ptr(gk) if qk_ln else None, 1e-5,
                                      abi.ptr(dev(km)) if mask else None, 8.0, abi.ptr(hi), abi.ptr(lo), d, T, None, st()), "trunk_mpt_attn")
    torch.cuda.synchronize()
    q, k, v = qkv.double()[:, :d], qkv.double()[:, d:2 * d], qkv.double()[:, 2 * d:]
    if qk_ln:
        q, k = _ln(q, gq, None), _ln(k, gk, None)
    hd = d // heads
    out = torch.zeros(T, d, dtype=torch.float64, device="cuda")
    for h in range(heads):
        s = q[:, h * hd:(h + 1) * hd] @ k[:, h * hd:(h + 1) * hd].t() / math.sqrt(hd)
        slope = 2.0 ** (-8.0 * (h + 1) / heads)
        s = s - (T - 1 - torch.arange(T, device="cuda", dtype=torch.float64)).unsqueeze(0) * slope
        bad = torch.triu(torch.ones(T, T, dtype=torch.bool, device="cuda"), 1)
        if mask:
            bad = bad | (dev(km) == 0).unsqueeze(0)
        s = s.masked_fill(bad, float("-inf"))
        out[:, h * hd:(h + 1) * hd] = torch.softmax(s, -1) @ v[:, h * hd:(h + 1) * hd]
    rows = [t for t in range(T) if not (mask and T > 2 and False)]
    got = hi[:T].double() + lo[:T].double()
    ok = torch.isfinite(out).all(dim=1)                          # a row whose every visible key is padding is NaN in both arms
    assert rel_err(got[ok], out[ok].float()) < 2e-5
