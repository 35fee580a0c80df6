"""The product library builds for gfx950, loads, and exports every symbol include/bbs_sign_amd.h
declares (no compute calls: there is no GPU in the CPU test tier)."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_library_exports_header_symbols():
    from bbs_sign_amd import _lib, build
    path = build.build(twin=False, verbose=False)
    lib = _lib.load_library(path)
    header = open(os.path.join(ROOT, "include", "bbs_sign_amd.h")).read()
    declared = set(re.findall(r"\b(bbs_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations found"
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.bbs_version()


def test_product_has_no_cpu_fallback():
    """Without a GPU a context cannot be created: the call fails loudly instead of computing on the CPU."""
    import ctypes
    from bbs_sign_amd import _lib, build
    lib = _lib.load_library(build.build(twin=False, verbose=False))
    if lib.bbs_device_count() > 0:
        return                      # running on a GPU box
    h = ctypes.c_void_p()
    assert lib.bbs_ctx_create(0, 0, ctypes.byref(h)) == -104      # BBS_E_NO_DEVICE


def test_missing_library_raises():
    import pytest
    from bbs_sign_amd import _lib
    with pytest.raises(_lib.LibraryMissing):
        _lib.load_library("/nonexistent/libbbs_sign_amd.so")


def test_only_the_c_abi_is_exported():
    """Everything the library exports is a function of the header, a kernel launch stub (which the HIP runtime needs by
    name), a HIP compilation-unit marker or a weak instantiation of a standard-library template (libstdc++ gives namespace std default visibility):
    the C++ internals (templates of runtime.hpp, op_*.hpp) stay hidden."""
    import subprocess
    from bbs_sign_amd import _lib, build
    path = build.build(twin=False, verbose=False)
    out = subprocess.run(["nm", "-D", "--defined-only", path], stdout=subprocess.PIPE, text=True, check=True).stdout
    names = [ln.split()[-1] for ln in out.splitlines() if ln.strip()]
    api = [n for n in names if n.startswith("bbs_")]
    assert set(api) == set(_lib.SIGNATURES), set(api) ^ set(_lib.SIGNATURES)
    other = [n for n in names if not n.startswith("bbs_") and "k_stage" not in n and "k_pip_window" not in n and not n.startswith(("_ZNSt", "_ZSt", "_ZNKSt", "_ZTISt", "_ZTSSt", "_ZTVSt", "_ZZNSt", "__hip_"))]
    assert not other, other[:10]
