"""Ad-hoc (GPU box): sign from host buffers, record form vs octet form, alternating (order effects)."""
import os, sys, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import parity_cases as pc
from bbs_sign_amd import _lib, Job
from bbs_sign_amd.engine import _ragged_bytes

n, L = 4096, 32
suite, eng, gens, sk = pc.bench_engine("bls12_381", L, None, 16)
msgs, _, _ = pc.bench_items(suite, eng, n, L, 8, 0)
ms, mo = eng._scalars(msgs)
hb, ho = _ragged_bytes([b""] * n)
u8 = lambda x: x.ctypes.data_as(_lib.c_u8p)
u64 = lambda x: x.ctypes.data_as(_lib.c_u64p)
outs = [np.ones(n * (2 * eng.fpb + 32), dtype=np.uint8) for _ in range(9)]
turn = [0]


def submit(fn):
    st = np.full(n, -128, dtype=np.int8)
    jh = ctypes.c_void_p()
    o = outs[turn[0] % 9]; turn[0] += 1
    eng._chk(fn(eng.h, n, u8(ms), u64(mo), u8(hb), u64(ho), u8(o), st.ctypes.data_as(_lib.c_i8p), ctypes.byref(jh)), "submit")
    j = Job(eng, jh, n); j.result = st
    return j


def loop(fn, steps=64, depth=8):
    pend = []
    def retire():
        j = pend.pop(0); j.wait(); assert (j.result == 1).all(); j.free()
    t = time.perf_counter()
    for _ in range(steps):
        if len(pend) >= depth:
            retire()
        pend.append(submit(fn))
    while pend:
        retire()
    return steps * n / (time.perf_counter() - t)


for rep in range(3):
    for name, fn in (("records", eng.lib.bbs_core_sign_submit), ("octets", eng.lib.bbs_sign_octets_submit)):
        print("rep %d %-8s %.2f M sign/s" % (rep, name, loop(fn) / 1e6), flush=True)
