"""The C++ host mirror of the reference's public interface (include/bbs_sign_amd.hpp) running the reference's own
public-interface tests (tests/cpp/public_interface.cpp): on the CPU against the test build of the stage code, on
the GPU against the product library."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build_and_run(lib_path, exe):
    src = os.path.join(ROOT, "tests", "cpp", "public_interface.cpp")
    os.makedirs(os.path.dirname(exe), exist_ok=True)
    libdir, libname = os.path.dirname(lib_path), os.path.basename(lib_path)
    cmd = ["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), src, "-o", exe, "-L", libdir, "-l:" + libname,
           "-Wl,-rpath," + libdir, "-Wl,-rpath,/opt/rocm/lib", "-L", "/opt/rocm/lib"]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0 and "all checks passed" in r.stdout, r.stdout[-3000:]


def test_cpp_public_interface_cpu_twin():
    sys.path.insert(0, ROOT)
    from bbs_sign_amd import build as b
    twin = b.build(twin=True, verbose=False)
    _build_and_run(twin, os.path.join(ROOT, "bbs_sign_amd", "build", "cpp_public_interface_twin"))


@pytest.mark.gpu
def test_cpp_public_interface_gpu():
    sys.path.insert(0, ROOT)
    from bbs_sign_amd import build as b
    lib = b.build(twin=False, verbose=False)
    _build_and_run(lib, os.path.join(ROOT, "bbs_sign_amd", "build", "cpp_public_interface"))
