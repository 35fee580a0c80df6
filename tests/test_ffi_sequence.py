"""The call sequence of the Rust host shim (bindings/rust/src/lib.rs) from a plain-C client: tests/cpp/ffi_sequence.c is
compiled with gcc against include/bbs_sign_amd.h and performs exactly the calls the shim makes, step for step -- the
image has no Rust toolchain, so this is how the ABI the shim binds is exercised by a non-Python, non-C++ caller.
On the CPU it links the test build of the stage code (tests/hosttwin), on the GPU the product library."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build_and_run(lib_path, exe):
    src = os.path.join(ROOT, "tests", "cpp", "ffi_sequence.c")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    libdir, libname = os.path.dirname(lib_path), os.path.basename(lib_path)
    cmd = ["gcc", "-std=c99", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), src, "-o", exe, "-L", libdir, "-l:" + libname,
           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-L", "/opt/rocm/lib"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0 and "all checks passed" in r.stdout, r.stdout[-3000:]


def test_shim_binds_only_declared_symbols_with_the_header_arity():
    """Every `fn bbs_*` in the shim's extern block is declared in the header with the same number of parameters, and
    every one of them is called by ffi_sequence.c."""
    hdr = open(os.path.join(ROOT, "include", "bbs_sign_amd.h")).read()
    rs = open(os.path.join(ROOT, "bindings", "rust", "src", "lib.rs")).read()
    csrc = open(os.path.join(ROOT, "tests", "cpp", "ffi_sequence.c")).read()
    ext = rs[rs.index('extern "C" {'):]
    ext = ext[:ext.index("\n}\n")]
    fns = re.findall(r"fn (bbs_\w+)\s*\(([^;]*?)\)\s*(?:->\s*[\w:]+)?;", ext, flags=re.S)
    assert len(fns) >= 16
    for name, params in fns:
        m = re.search(r"^(?:int|void|size_t|const char\*)\s+%s\s*\(([^;]*?)\);" % name, hdr, flags=re.S | re.M)
        assert m, "%s is not declared in include/bbs_sign_amd.h" % name
        n_rs = len([p for p in params.split(",") if p.strip()])
        n_h = len([p for p in m.group(1).split(",") if p.strip() and p.strip() != "void"])
        assert n_rs == n_h, (name, n_rs, n_h)
        assert re.search(r"\b%s\s*\(" % name, csrc), "%s is bound by the shim but not exercised by ffi_sequence.c" % name


def test_ffi_sequence_cpu_twin():
    sys.path.insert(0, ROOT)
    from bbs_sign_amd import build as b
    twin = b.build(twin=True, verbose=False)
    _build_and_run(twin, os.path.join(ROOT, "bbs_sign_amd", "build", "ffi_sequence_twin"))


@pytest.mark.gpu
def test_ffi_sequence_gpu():
    sys.path.insert(0, ROOT)
    from bbs_sign_amd import build as b
    lib = b.build(twin=False, verbose=False)
    _build_and_run(lib, os.path.join(ROOT, "bbs_sign_amd", "build", "ffi_sequence"))
