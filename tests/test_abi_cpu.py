"""CPU-side checks of the C-ABI boundary: libdeer_hip.so builds for gfx950, loads without a GPU, exports every
symbol include/deer_hip.h declares, and the ctypes signatures in deer_vla_amd/_abi.py match the header."""
import os
import re

import pytest

from deer_vla_amd import _abi as abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_decls():
    """every function declared in include/*.h (the kernel-level ABI deer_hip.h and the native spine deer_model.h)"""
    src = "".join(open(os.path.join(ROOT, "include", h)).read() for h in ("deer_hip.h", "deer_model.h"))
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    decls = {}
    for m in re.finditer(r"(?:int|long|void|const char\*)\s+(deer_\w+)\s*\(([^;{}]*?)\)\s*;", src, flags=re.S):
        args = m.group(2).strip()
        n = 0 if args in ("", "void") else len([a for a in args.split(",") if a.strip()])
        decls[m.group(1)] = (n, args)
    return decls


@pytest.fixture(scope="module")
def built():
    return abi.build()


def test_library_builds_and_exports_every_declared_symbol(built):
    lib = abi.lib()
    decls = header_decls()
    assert len(decls) >= 19
    for name in decls:
        assert hasattr(lib, name), name
    assert set(decls) == set(abi.SIGNATURES), set(decls) ^ set(abi.SIGNATURES)
    assert lib.deer_hip_arch() == b"gfx950"
    assert lib.deer_hip_abi_version() == 1


def test_ctypes_signatures_match_header(built):
    import ctypes
    decls = header_decls()
    for name, (n, args) in decls.items():
        sig = abi.SIGNATURES[name]
        assert len(sig) == n, (name, len(sig), n)
        for a, t in zip([x.strip() for x in args.split(",")] if n else [], sig):
            if a.startswith("const char*"):
                assert t is ctypes.c_char_p, (name, a)
            elif "*" in a:
                assert t is ctypes.c_void_p, (name, a)
            elif a.startswith("long "):
                assert t is ctypes.c_long, (name, a)
            elif a.startswith("float "):
                assert t is ctypes.c_float, (name, a)
            else:
                assert a.startswith("int "), (name, a)
                assert t is ctypes.c_int, (name, a)


def test_host_helpers_without_gpu(built):
    lib = abi.lib()
    # split-K heuristic is a pure host function: K-slice fits LDS, divides K, deterministic
    for (M, N, K) in [(14, 6144, 2048), (14, 2048, 8192), (14, 512, 2048), (14, 2048, 512), (32, 8192, 2048), (1, 64, 32)]:
        s = lib.deer_skinny_splitk(M, N, K)
        assert s >= 1 and K % (s * 32) == 0 and (K // s >= 128 or s == 1)
        assert s == lib.deer_skinny_splitk(M, N, K)


def test_engine_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from deer_vla_amd.config import deer_tiny
    from deer_vla_amd.engine import DeerEngine
    with pytest.raises(abi.DeerHipError):
        DeerEngine(deer_tiny(), {})


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "deer_vla_amd")
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                s = open(os.path.join(dp, f)).read()
                assert "import oracle" not in s and "from oracle" not in s, os.path.join(dp, f)


def test_spine_parameter_inventory_matches_the_reference_state_dict_names(built):
    """The native model object (csrc/model.hip) ingests tensors by the REFERENCE's state-dict names: its inventory must be
    exactly deer_vla_amd.synthetic.param_shapes (which tests/test_oracle_golden.py pins against the reference's own
    ``state_dict()`` through strict loads).  Host-only calls: no GPU needed."""
    import ctypes
    from deer_vla_amd.config import deer_3b, deer_9b, deer_tiny
    from deer_vla_amd.engine import config_to_c
    from deer_vla_amd.synthetic import param_shapes
    lib = abi.lib()
    for cfg in (deer_tiny(), deer_3b(), deer_3b(max_layer=4), deer_9b(), deer_tiny(lstm_layernorm=False, mlp_layernorm=False)):
        h = ctypes.c_void_p()
        cc = config_to_c(cfg, 1, 32)
        assert lib.deer_model_create(ctypes.byref(cc), ctypes.byref(h)) == 0
        ps = param_shapes(cfg)
        assert all(lib.deer_model_knows_tensor(h, k.encode()) for k in ps)
        buf = ctypes.create_string_buffer(1 << 20)
        n = lib.deer_model_missing_tensors(h, buf, len(buf))
        assert n == len(ps) and set(buf.value.decode().split()) == set(ps)
        assert lib.deer_model_arena_bytes(h) > 0 and lib.deer_model_workspace_bytes(h) > 0
        assert not lib.deer_model_knows_tensor(h, b"lang_encoder.transformer.blocks.99.decoder_layer.attn.Wqkv.weight")
        # exit configuration is validated like the reference's controller would fail (KeyError on thresholds[i])
        ids = (ctypes.c_int * 4)(1, 3, 5, 7)
        if cfg.n_layers >= 8:
            assert lib.deer_model_configure_exit(h, ids, 4, 7, 0, 1) != 0      # deepest reachable layer 6 is not an exit
            assert lib.deer_model_configure_exit(h, ids, 4, 8, 0, 1) == 0
        lib.deer_model_destroy(h)
