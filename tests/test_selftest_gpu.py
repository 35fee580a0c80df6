"""GPU self-test: the six-lane (wavefront-cooperative) Fp12 arithmetic on the GPU against the one-lane code (run on the host),
operation by operation, and the one-lane multiplication against the oracle's Fp12."""
import ctypes
import random

import numpy as np
import pytest

from bbs_sign_amd import Engine, _lib
from oracle.curves import CURVES

pytestmark = pytest.mark.gpu

OPS = {0: "mul", 1: "frob1", 2: "frob2", 3: "frob3", 4: "inv", 5: "conj", 6: "line", 7: "final_exp", 8: "sqr",
       10: "cyclo_sqr", 11: "pow_x"}


def _run(eng, op, a, b):
    fpb = eng.fpb
    ab = np.frombuffer(b"".join(int(x).to_bytes(fpb, "little") for x in a), dtype=np.uint8).copy()
    bb = np.frombuffer(b"".join(int(x).to_bytes(fpb, "little") for x in b), dtype=np.uint8).copy()
    o1 = np.zeros(12 * fpb, dtype=np.uint8)
    o2 = np.zeros(12 * fpb, dtype=np.uint8)
    rc = eng.lib.bbs_selftest_f12(eng.h, op, ab.ctypes.data_as(_lib.c_u8p), bb.ctypes.data_as(_lib.c_u8p),
                                  o1.ctypes.data_as(_lib.c_u8p), o2.ctypes.data_as(_lib.c_u8p))
    assert rc == 0
    dec = lambda o: [int.from_bytes(o.tobytes()[k * fpb:(k + 1) * fpb], "little") for k in range(12)]
    return dec(o1), dec(o2)


def _tower_to_w(c, t):
    # tower order: c0.c0, c0.c1, c0.c2, c1.c0, c1.c1, c1.c2 (each Fp2 = 2 Fp) ; w-basis g0..g5
    f2 = [(t[2 * k], t[2 * k + 1]) for k in range(6)]
    return [f2[0], f2[3], f2[1], f2[4], f2[2], f2[5]]


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_six_lane_fp12_matches_one_lane(curve):
    c = CURVES[curve]
    rng = random.Random(5)
    eng = Engine(curve)
    for op, name in OPS.items():
        for rep in range(2):
            a = [rng.randrange(c.p) for _ in range(12)]
            b = [rng.randrange(c.p) for _ in range(12)]
            if op == 6:
                P = c.g1_mul(c.g1, rng.randrange(1, c.r))
                b[0], b[1] = P
            s, d = _run(eng, op, a, b)
            assert s == d, (curve, name, rep)
            # ... and against the ORACLE's Fp12 (oracle/curves.py, w-basis), operation by operation: the one-lane and
            # the six-lane code share tower.hpp, so their agreement alone would not localise a break in it
            x = _tower_to_w(c, a)
            if op >= 10:                                  # the host makes the input cyclotomic: x^((p^6-1)(p^2+1))
                x = c.f12_mul(c.f12_conj(x), c.f12_inv(x))
                x = c.f12_mul(c.f12_frob(c.f12_frob(x)), x)
            frobk = lambda v, k: v if k == 0 else frobk(c.f12_frob(v), k - 1)
            want = None
            if op == 0:
                want = c.f12_mul(x, _tower_to_w(c, b))
            elif op in (1, 2, 3):
                want = frobk(x, op)
            elif op == 4:
                want = c.f12_inv(x)
            elif op == 5:
                want = c.f12_conj(x)
            elif op == 7:                                 # BLS12-381 raises to 3 (p^12-1)/r, BN254 to (p^12-1)/r
                want = c.final_exp(x)
                if curve == "bls12_381":
                    want = c.f12_pow(want, 3)
            elif op in (8, 10):
                want = c.f12_sqr(x)
            elif op == 11:                                # x^(curve parameter), the sign by conjugation (x unitary)
                want = c.f12_pow(x, abs(c.x_param))
                if c.x_param < 0:
                    want = c.f12_conj(want)
            if want is not None:
                assert _tower_to_w(c, d) == [tuple(v) for v in want], (curve, name, rep, "six-lane vs oracle")
    eng.close()
