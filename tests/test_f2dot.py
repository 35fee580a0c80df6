"""Lazily reduced Fp2 dot products (tower.hpp F2Acc: the arithmetic core of the lane-sliced pairing kernel) on the
host, against Python big integers: random and extreme operands, every weight pattern up to the design limit of 6,
with the bound assertions of the host twin switched on (BBS_CHECK_BOUNDS)."""
import ctypes
import os
import random
import sys

import numpy as np
import pytest

from oracle.curves import BLS12_381, BN254

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="session")
def libs():
    sys.path.insert(0, ROOT)
    from bbs_sign_amd import _lib, build as b
    return [_lib.load_library(b.build(twin=True, verbose=False)), _lib.load_library(b.build(twin=False, verbose=False))]


def _u8(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_uint8))


@pytest.mark.parametrize("cid,curve", [(0, BLS12_381), (1, BN254)])
def test_f2dot(libs, cid, curve):
    p = curve.p
    fpb = curve.fp_bytes
    rng = random.Random(77 + cid)
    edge = [0, 1, 2, p - 1, p - 2, (p - 1) // 2, (1 << (8 * fpb - 1)) % p]
    patterns = [[1], [2], [0], [1] * 6, [2, 2, 2], [2, 2, 1, 1], [1, 1, 0], [2, 1, 1, 0, 1], [0] * 6, [2, 2, 0, 0]]
    for lib in libs:
        for pat in patterns:
            for trial in range(6):
                pick = (lambda: rng.choice(edge)) if trial < 2 else (lambda: rng.randrange(p))
                a = [(pick(), pick()) for _ in pat]
                b = [(pick(), pick()) for _ in pat]
                if trial == 1:
                    a = [(p - 1, p - 1)] * len(pat)
                    b = [(p - 1, p - 1)] * len(pat)
                re = im = 0
                for (a0, a1), (b0, b1), w in zip(a, b, pat):
                    if w == 0:
                        re += a0 * b0; im += a1 * b0
                    else:
                        re += w * (a0 * b0 - a1 * b1); im += w * (a0 * b1 + a1 * b0)
                ab = np.frombuffer(b"".join(x.to_bytes(fpb, "little") + y.to_bytes(fpb, "little") for x, y in a), dtype=np.uint8).copy()
                bb = np.frombuffer(b"".join(x.to_bytes(fpb, "little") + y.to_bytes(fpb, "little") for x, y in b), dtype=np.uint8).copy()
                wb = np.array(pat, dtype=np.uint8)
                out = np.zeros(2 * fpb, dtype=np.uint8)
                assert lib.bbs_selftest_f2dot(cid, len(pat), _u8(ab), _u8(bb), _u8(wb), _u8(out)) == 0
                o = out.tobytes()
                assert (int.from_bytes(o[:fpb], "little"), int.from_bytes(o[fpb:], "little")) == (re % p, im % p), (cid, pat, trial)


@pytest.mark.parametrize("cid,curve", [(0, BLS12_381), (1, BN254)])
def test_safegcd_inversion(libs, cid, curve):
    """fe_inv (Bernstein-Yang divsteps) against Python's modular inverse and the library's own Fermat power, base
    and scalar field, edge values (0 -> 0, 1, p-1, powers of two, all-ones patterns) and random values; the host
    twin also asserts that g reached 0 within the fixed number of batches."""
    rng = random.Random(5 + cid)
    for sf, mod, nb in ((0, curve.p, curve.fp_bytes), (1, curve.r, 32)):
        vals = [0, 1, 2, 3, mod - 1, mod - 2, (mod + 1) // 2, (mod - 1) // 2, (1 << (mod.bit_length() - 1)),
                (1 << (mod.bit_length() - 1)) - 1, 0x5555555555555555555555555555555555555555555555555555555555555555 % mod,
                (1 << 30) - 1, 1 << 30, (1 << 60) + 1]
        vals += [1 << k for k in range(1, mod.bit_length() - 1, 37)]
        vals += [rng.randrange(mod) for _ in range(300)]
        for lib in libs:
            for x in vals:
                xb = np.frombuffer(x.to_bytes(nb, "little"), dtype=np.uint8).copy()
                o1, o2 = np.zeros(nb, dtype=np.uint8), np.zeros(nb, dtype=np.uint8)
                assert lib.bbs_selftest_inv(cid, sf, _u8(xb), _u8(o1), _u8(o2)) == 0
                want = pow(x, -1, mod) if x else 0
                assert int.from_bytes(o1.tobytes(), "little") == want, (cid, sf, hex(x))
                assert int.from_bytes(o2.tobytes(), "little") == want


@pytest.mark.parametrize("cid,curve,xi0,nl", [(0, BLS12_381, 1, 14), (1, BN254, 9, 10)])
def test_fp4_half_square(libs, cid, curve, xi0, nl):
    """The four-column Fp4 half-square of the cyclotomic squaring (low half a^2 + xi b^2, high half 2 a b) against
    big integers; operands chosen through their INTERNAL representation (x R^-1 is passed so that the Montgomery
    form is x) to put all-ones limb patterns and the largest admissible values into the column sums."""
    p = curve.p
    fpb = curve.fp_bytes
    R = 1 << (28 * nl)
    Rinv = pow(R, -1, p)
    rng = random.Random(31 + cid)
    top = p.bit_length()
    internal = [p - 1, (1 << (top - 1)) - 1, (1 << (top - 2)) - 1, 0, 1, int("0fffffff" * nl, 16) % p, (1 << (top - 1)) + 12345]
    def f2mul(x, y):
        return ((x[0] * y[0] - x[1] * y[1]) % p, (x[0] * y[1] + x[1] * y[0]) % p)
    for lib in libs:
        for trial in range(40):
            if trial < 12:
                vals = [rng.choice(internal) * Rinv % p for _ in range(4)]
            else:
                vals = [rng.randrange(p) for _ in range(4)]
            a, b = (vals[0], vals[1]), (vals[2], vals[3])
            for hi in (0, 1):
                if hi:
                    t = f2mul(a, b)
                    want = (2 * t[0] % p, 2 * t[1] % p)
                else:
                    bb = f2mul(b, b)
                    xb = ((xi0 * bb[0] - bb[1]) % p, (bb[0] + xi0 * bb[1]) % p)
                    aa = f2mul(a, a)
                    want = ((aa[0] + xb[0]) % p, (aa[1] + xb[1]) % p)
                ab = np.frombuffer(a[0].to_bytes(fpb, "little") + a[1].to_bytes(fpb, "little"), dtype=np.uint8).copy()
                bbuf = np.frombuffer(b[0].to_bytes(fpb, "little") + b[1].to_bytes(fpb, "little"), dtype=np.uint8).copy()
                out = np.zeros(2 * fpb, dtype=np.uint8)
                assert lib.bbs_selftest_fp4sqr(cid, hi, _u8(ab), _u8(bbuf), _u8(out)) == 0
                o = out.tobytes()
                assert (int.from_bytes(o[:fpb], "little"), int.from_bytes(o[fpb:], "little")) == want, (cid, hi, trial)
