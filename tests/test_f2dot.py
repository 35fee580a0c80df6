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
                # the two-pass accumulators (low columns, quotients, high columns) give the same limbs
                out2 = np.zeros(2 * fpb, dtype=np.uint8)
                assert lib.bbs_selftest_f2dot2(cid, len(pat), _u8(ab), _u8(bb), _u8(wb), _u8(out2)) == 0
                assert out2.tobytes() == o, (cid, pat, trial, "two-pass")


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
                if cid == 0:                     # BLS12-381: the two-pass form (hi | 2) gives the same limbs
                    out2 = np.zeros(2 * fpb, dtype=np.uint8)
                    assert lib.bbs_selftest_fp4sqr(cid, hi | 2, _u8(ab), _u8(bbuf), _u8(out2)) == 0
                    assert out2.tobytes() == o, (cid, hi, trial, "two-pass")


def _glv_edge_scalars(r, lam, rng, count):
    edge = [0, 1, 2, lam - 1, lam, lam + 1, 2 * lam, 2 * lam + 1, lam * lam % r, r - 1, r - 2, (1 << 128) - 1, 1 << 128,
            (lam - 1) + lam * (lam - 1), 3 * lam - 1, r - lam, r - lam - 1]
    return [k % r for k in edge] + [rng.randrange(r) for _ in range(count)]


def _bn_lambda():
    """The cube root of unity mod r the BN254 split is built on (tools/gen_params.py picks it the same way)."""
    r = BN254.r
    g = 2
    while pow(g, (r - 1) // 3, r) == 1:
        g += 1
    return pow(g, (r - 1) // 3, r)


@pytest.mark.parametrize("cid,curve", [(0, BLS12_381), (1, BN254)])
def test_glv_split(libs, cid, curve):
    """k = (+-k1) + (+-k2) lambda mod r with both halves below 2^128 (g1.hpp glv_split).  BLS12-381: Barrett quotient
    by lambda = x^2 - 1 and two corrections, exact floor / remainder; BN254: rounding against a short lattice basis."""
    c = curve
    lam = (c.x_param * c.x_param - 1) if cid == 0 else _bn_lambda()
    assert (lam * lam + lam + 1) % c.r == 0
    rng = random.Random(4242 + cid)
    for lib in libs:
        for k in _glv_edge_scalars(c.r, lam, rng, 400):
            kb = np.frombuffer(k.to_bytes(32, "little"), dtype=np.uint8).copy()
            k1 = np.zeros(16, dtype=np.uint8)
            k2 = np.zeros(16, dtype=np.uint8)
            n1, n2 = ctypes.c_int(0), ctypes.c_int(0)
            assert lib.bbs_selftest_glv_split(cid, _u8(kb), _u8(k1), _u8(k2), ctypes.byref(n1), ctypes.byref(n2)) == 0
            a, b = int.from_bytes(k1.tobytes(), "little"), int.from_bytes(k2.tobytes(), "little")
            if cid == 0:
                assert (a, b, n1.value, n2.value) == (k % lam, k // lam, 0, 0), hex(k)
            else:
                sa, sb = (-a if n1.value else a), (-b if n2.value else b)
                assert (sa + sb * lam - k) % c.r == 0 and a < (1 << 127) and b < (1 << 127), hex(k)
        bad = np.frombuffer(c.r.to_bytes(32, "little"), dtype=np.uint8).copy()
        assert lib.bbs_selftest_glv_split(cid, _u8(bad), _u8(k1), _u8(k2), ctypes.byref(n1), ctypes.byref(n2)) != 0
        assert lib.bbs_selftest_glv_split(2, _u8(kb), _u8(k1), _u8(k2), ctypes.byref(n1), ctypes.byref(n2)) != 0


@pytest.mark.parametrize("cid,curve", [(0, BLS12_381), (1, BN254)])
def test_joint_multiplication(libs, cid, curve):
    """k0 P0 + k1 P1 + k2 P2 on the joint chain of proof_verify's T1, plain and with the GLV split, against the
    oracle's double-and-add: random and edge scalars (zero, even, below lambda, r - 1), the identity and repeated
    points among the inputs (those take the per-point fallback)."""
    c = curve
    fpb = c.fp_bytes
    rng = random.Random(99 + cid)
    lam = (BLS12_381.x_param * BLS12_381.x_param - 1) if cid == 0 else _bn_lambda()
    sc = _glv_edge_scalars(c.r, lam, rng, 8)

    def rec(P):
        return bytes(2 * fpb) if P is None else P[0].to_bytes(fpb, "little") + P[1].to_bytes(fpb, "little")

    pts = [c.g1_mul(c.g1, rng.randrange(1, c.r)) for _ in range(4)]
    cases = [([pts[0], pts[1], pts[2]], [rng.choice(sc) for _ in range(3)]) for _ in range(5)]
    cases += [([pts[0], pts[1], pts[2]], [sc[i], sc[(i + 5) % len(sc)], sc[(i + 11) % len(sc)]]) for i in range(0, len(sc), 4)]
    cases.append(([pts[0], pts[0], c.g1_neg(pts[0])], [5, 7, 12]))             # sums to the identity
    cases.append(([pts[3], None, pts[1]], [rng.randrange(c.r), 3, rng.randrange(c.r)]))
    for lib in libs:
        for glv in (0, 1):
            for P, k in cases:
                pb = np.frombuffer(b"".join(rec(q) for q in P), dtype=np.uint8).copy()
                kb = np.frombuffer(b"".join(x.to_bytes(32, "little") for x in k), dtype=np.uint8).copy()
                out = np.zeros(2 * fpb, dtype=np.uint8)
                assert lib.bbs_selftest_mul3(cid, glv, _u8(pb), _u8(kb), _u8(out)) == 0
                want = None
                for q, x in zip(P, k):
                    want = c.g1_add(want, c.g1_mul(q, x))
                assert out.tobytes() == rec(want), (glv, k)


@pytest.mark.parametrize("cid,curve", [(0, BLS12_381), (1, BN254)])
def test_lin_pm(libs, cid, curve):
    """3 x0 +- 2 x1 with the sign chosen at run time (field.hpp lin_pm: the last step of the cyclotomic square)."""
    p = curve.p
    fpb = curve.fp_bytes
    rng = random.Random(5 + cid)
    edge = [0, 1, 2, p - 1, p - 2, (p - 1) // 2, (p + 1) // 2, (1 << (8 * fpb - 1)) % p]
    vals = [(a, b) for a in edge for b in edge] + [(rng.randrange(p), rng.randrange(p)) for _ in range(300)]
    for lib in libs:
        for a, b in vals:
            for plus in (0, 1):
                xa = np.frombuffer(a.to_bytes(fpb, "little"), dtype=np.uint8).copy()
                xb = np.frombuffer(b.to_bytes(fpb, "little"), dtype=np.uint8).copy()
                out = np.zeros(fpb, dtype=np.uint8)
                assert lib.bbs_selftest_lin_pm(cid, plus, _u8(xa), _u8(xb), _u8(out)) == 0
                want = (3 * a + 2 * b) % p if plus else (3 * a - 2 * b) % p
                assert int.from_bytes(out.tobytes(), "little") == want, (hex(a), hex(b), plus)
