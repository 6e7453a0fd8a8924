"""CPU-side checks of the C-ABI boundary: libdeer_hip.so builds for gfx950, loads without a GPU, exports every
symbol include/deer_hip.h declares, and the ctypes signatures in deer_vla_amd/_abi.py match the header."""
import os
import re

import pytest

from deer_vla_amd import _abi as abi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_decls():
    src = open(os.path.join(ROOT, "include", "deer_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    decls = {}
    for m in re.finditer(r"(?:int|const char\*)\s+(deer_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
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
            if "*" in a:
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
        assert s >= 1 and K % (s * 32) == 0
        assert (K // s) * (32 if M > 16 else 16) * 2 <= 64 * 1024
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
