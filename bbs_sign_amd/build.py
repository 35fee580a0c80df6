"""Builds the C-ABI library from bbs_sign_amd/csrc with hipcc (parallel over translation units).

    python -m bbs_sign_amd.build            # product: gfx950 -> bbs_sign_amd/libbbs_sign_amd.so
    python -m bbs_sign_amd.build --twin     # TEST-ONLY host twin -> tests/hosttwin/libbbs_hosttwin_TESTONLY.so
"""
import concurrent.futures as cf
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
PRODUCT = os.path.join(HERE, "libbbs_sign_amd.so")
TWIN = os.path.join(ROOT, "tests", "hosttwin", "libbbs_hosttwin_TESTONLY.so")


def _newest_src():
    files = glob.glob(os.path.join(CSRC, "*")) + [os.path.join(ROOT, "include", "bbs_sign_amd.h")]
    return max(os.path.getmtime(f) for f in files)


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode:
        raise RuntimeError("command failed: %s\n%s" % (" ".join(cmd), r.stdout[-4000:]))
    return r.stdout


def build(twin=False, force=False, jobs=None, verbose=True):
    out = TWIN if twin else PRODUCT
    if not force and os.path.exists(out) and os.path.getmtime(out) >= _newest_src():
        return out
    objdir = os.path.join(HERE, "build", "twin" if twin else "gfx950")
    os.makedirs(objdir, exist_ok=True)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    tus = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    if twin:
        flags = ["-O2", "--offload-host-only", "-x", "hip", "-DBBS_HOST_TWIN", "-DBBS_CHECK_BOUNDS", "-fPIC"]
    else:
        flags = ["-O3", "--offload-arch=gfx950", "-fPIC"]
    # only the C ABI of include/bbs_sign_amd.h is exported (capi.hip raises the visibility of its extern "C" blocks)
    flags += ["-fvisibility=hidden", "-fvisibility-inlines-hidden"]
    jobs = jobs or min(8, os.cpu_count() or 1)

    def one(tu):
        obj = os.path.join(objdir, os.path.basename(tu)[:-4] + ".o")
        if not force and os.path.exists(obj) and os.path.getmtime(obj) >= _newest_src():
            return obj
        _run(["hipcc"] + flags + ["-c", tu, "-o", obj])
        return obj

    with cf.ThreadPoolExecutor(max_workers=jobs) as ex:
        objs = list(ex.map(one, tus))
    link = ["hipcc", "-shared", "-fPIC"] + (["--offload-host-only"] if twin else ["--offload-arch=gfx950"]) + objs + ["-o", out]
    _run(link)
    if verbose:
        print("built", out)
    return out


if __name__ == "__main__":
    build(twin="--twin" in sys.argv, force="--force" in sys.argv)
