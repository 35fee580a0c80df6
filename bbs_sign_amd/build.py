"""Builds the C-ABI library from bbs_sign_amd/csrc with hipcc (parallel over translation units).

    python -m bbs_sign_amd.build            # product: gfx950 -> bbs_sign_amd/libbbs_sign_amd.so
    python -m bbs_sign_amd.build --twin     # TEST-ONLY host twin -> tests/hosttwin/libbbs_hosttwin_TESTONLY.so
"""
import concurrent.futures as cf
import glob
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
PRODUCT = os.path.join(HERE, "libbbs_sign_amd.so")
TWIN = os.path.join(ROOT, "tests", "hosttwin", "libbbs_hosttwin_TESTONLY.so")


def _sources():
    return sorted(glob.glob(os.path.join(CSRC, "*"))) + [os.path.join(ROOT, "include", "bbs_sign_amd.h")]


def source_hash():
    """SHA-256 over the names and contents of every file the library is built from (csrc/* and the C header).  It is
    compiled into the library (bbs_source_hash()) so that a prebuilt .so that travelled with the tree -- the GPU box
    receives the built library, and file times do not survive the copy reliably -- is recognised as current or stale
    by content, not by modification time."""
    h = hashlib.sha256()
    for f in _sources():
        h.update(os.path.basename(f).encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:32]


def built_hash(path):
    """The source hash embedded in an existing library (marker string in its read-only data), or None.  Read from the
    file, not through dlopen: a library that is about to be rebuilt must not be mapped into this process."""
    if not os.path.exists(path):
        return None
    with open(path, "rb") as f:
        data = f.read()
    i = data.find(b"BBS_SRC_HASH=")
    if i < 0:
        return None
    return data[i + 13:i + 45].decode("ascii", "replace")


def _run(cmd):
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode:
        raise RuntimeError("command failed: %s\n%s" % (" ".join(cmd), r.stdout[-4000:]))
    return r.stdout


def build(twin=False, force=False, jobs=None, verbose=True):
    out = TWIN if twin else PRODUCT
    want = source_hash()
    if not force and built_hash(out) == want:
        if verbose:
            print("up to date (source hash %s): %s" % (want, out))
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
    flags += ["-fvisibility=hidden", "-fvisibility-inlines-hidden", '-DBBS_SRC_HASH="%s"' % want]
    # an object is reused only if ITS OWN stamp (written after it compiled) names this source hash and these flags: a build
    # that failed half way leaves no object that claims a hash it was not built from
    key = hashlib.sha256((want + "\0" + "\0".join(flags)).encode()).hexdigest()
    jobs = jobs or min(8, os.cpu_count() or 1)

    def one(tu):
        obj = os.path.join(objdir, os.path.basename(tu)[:-4] + ".o")
        stamp = obj + ".stamp"
        if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == key:
            return obj
        if os.path.exists(stamp):
            os.remove(stamp)
        _run(["hipcc"] + flags + ["-c", tu, "-o", obj])
        with open(stamp, "w") as f:
            f.write(key)
        return obj

    with cf.ThreadPoolExecutor(max_workers=jobs) as ex:
        objs = list(ex.map(one, tus))
    link = ["hipcc", "-shared", "-fPIC"] + (["--offload-host-only"] if twin else ["--offload-arch=gfx950"]) + objs + ["-o", out]
    _run(link)
    assert built_hash(out) == want, "the library just built does not report the source hash it was built from"
    if verbose:
        print("built (source hash %s): %s" % (want, out))
    return out


if __name__ == "__main__":
    build(twin="--twin" in sys.argv, force="--force" in sys.argv)
