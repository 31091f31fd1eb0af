"""CPU-side checks of the C-ABI shared library: it loads and exports every symbol include/r3d_hip.h declares
(no compute calls -- there is no GPU here)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from r3d_amd import build, _lib
    build.build(verbose=False)
    return _lib.load()


def test_header_symbols_are_exported(lib):
    hdr = open(os.path.join(ROOT, "include", "r3d_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(r3d_[a-z0-9_]+)\s*\(", hdr))
    declared.discard("r3d_gemm_desc")
    assert len(declared) >= 25
    from r3d_amd import _lib
    assert declared == set(_lib.EXPORTS), declared ^ set(_lib.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name


def test_abi_version_and_build_info(lib):
    from r3d_amd import _lib
    assert lib.r3d_abi_version() == _lib.ABI_VERSION
    assert _lib.build_info().startswith("gfx950;")


def test_gemm_plan_is_host_only_and_sane(lib):
    import ctypes as C
    from r3d_amd._lib import GemmDesc
    d = GemmDesc()
    d.M, d.N, d.K = 128, 128, 50176            # depth_projection forward at B=8, S=16, H=128
    assert lib.r3d_gemm_plan(C.byref(d)) == 0
    assert d.tile in (1, 2, 3, 4, 5) and d.splitk >= 8 and d.k_per_split % 16 == 0
    assert d.k_per_split * d.splitk >= 50176
    d2 = GemmDesc()
    d2.M, d2.N, d2.K = 128, 50176, 128         # depth_projection weight gradient
    assert lib.r3d_gemm_plan(C.byref(d2)) == 0 and d2.splitk == 1
    assert lib.r3d_gemm_plan(None) == -1


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from r3d_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.R3DHipError, match="no CPU/PyTorch fallback"):
        _lib.load()
