"""expand_message_xmd(SHA-256), hash_to_scalar, hash-to-G1 (BLS12-381 SSWU).

ORACLE (test infrastructure, see oracle/__init__.py).

Follows:
* src/utils/utilities_helper.rs:42-97  expand_message  (RFC 9380 5.3.1, SHA-256)
* src/utils/utilities_helper.rs:15-40  FromOkm         (48 bytes BE mod r)
* src/utils/core_utilities.rs:11-21    hash_to_scalar
* src/utils/interface_utilities.rs:30-44 HashToG1Bls12381 -> zkcrypto bls12_381
  (rev 9ea427c) ``hash_to_curve`` with ExpandMsgXmd<Sha256>: RFC 9380 suite
  BLS12381G1_XMD:SHA-256_SSWU_RO_.  The crate is not vendored; this restates
  the RFC algorithm.  The 11-isogeny E' -> E is *derived* here with Velu's
  formulas from the kernel of order 11 instead of transcribing the RFC's
  coefficient tables; the one remaining freedom (an automorphism of E, j = 0)
  is fixed by the constant ISO_SCALE below and pinned by the reference's
  generator vectors (src/tests/test_vector.rs:66-68,123-136).
"""

from __future__ import annotations

import hashlib

from .curves import BLS12_381, BN254


def i2osp(v: int, n: int) -> bytes:
    return int(v).to_bytes(n, "big")


def expand_message(msg: bytes, dst: bytes, len_in_bytes: int) -> bytes:
    """utilities_helper.rs:42-97 (panics there become ValueError here)."""
    b_in_bytes = 32
    ell = (len_in_bytes + b_in_bytes - 1) // b_in_bytes
    if ell > 255:
        raise ValueError("ell was too big in expand_message_xmd")
    if len(dst) > 255:
        raise ValueError("dst size is invalid")
    dst_prime = dst + bytes([len(dst)])
    b0 = hashlib.sha256(
        bytes(64) + msg + bytes([(len_in_bytes >> 8) & 0xFF, len_in_bytes & 0xFF, 0]) + dst_prime
    ).digest()
    bi = hashlib.sha256(b0 + b"\x01" + dst_prime).digest()
    out = bi
    for i in range(2, ell + 1):
        bi = hashlib.sha256(bytes(x ^ y for x, y in zip(b0, bi)) + bytes([i]) + dst_prime).digest()
        out += bi
    return out[:len_in_bytes]


def from_okm(curve, data: bytes) -> int:
    """utilities_helper.rs:15-40: big-endian integer mod r."""
    return int.from_bytes(data, "big") % curve.r


def hash_to_scalar(curve, msg: bytes, dst: bytes) -> int:
    """core_utilities.rs:11-21 with L = 48."""
    return from_okm(curve, expand_message(msg, dst, 48))


# --------------------------------------------------------------------------
# BLS12-381 G1 hash_to_curve (RFC 9380 8.8.1)
# --------------------------------------------------------------------------
_P = BLS12_381.p
ISO_A = 0x144698A3B8E9433D693A02C96D4982B0EA985383EE66A8D8E8981AEFD881AC98936F8DA0E0F97F5CF428082D584C1D
ISO_B = 0x12E2908D11688030018B12E8753EEE3B2016C1F0F24F4070A0B9C14FCEF35EF55A23215A316CEAA5D1CC48E98E172BE0
SSWU_Z = 11
H_EFF = 0xD201000000010001
_COFACTOR = 0x396C8C005555E1568C00AAAB0000AAAB
# which of the six automorphism-twisted normalisations of the Velu isogeny is the
# RFC's iso_map: index into the sorted list of sixth roots (fixed by the KATs).
ISO_SCALE_INDEX = 0  # smallest sixth root; pinned by test_vector.rs:123-136


def _ep_add(P, Q):
    """Affine addition on E' : y^2 = x^3 + ISO_A x + ISO_B."""
    p = _P
    if P is None:
        return Q
    if Q is None:
        return P
    x1, y1 = P
    x2, y2 = Q
    if x1 == x2:
        if (y1 + y2) % p == 0:
            return None
        lam = (3 * x1 * x1 + ISO_A) * pow(2 * y1, -1, p) % p
    else:
        lam = (y2 - y1) * pow(x2 - x1, -1, p) % p
    x3 = (lam * lam - x1 - x2) % p
    return (x3, (lam * (x1 - x3) - y1) % p)


def _ep_mul(P, k):
    R = None
    for bit in bin(k)[2:]:
        R = _ep_add(R, R)
        if bit == "1":
            R = _ep_add(R, P)
    return R


def _fp_sqrt(a):
    s = pow(a, (_P + 1) // 4, _P)
    return s if s * s % _P == a % _P else None


_ISO = None


def _derive_isogeny():
    """Velu's formulas for the 11-isogeny with kernel the rational subgroup of order 11
    of E'.  Returns (kernel data, s2inv, s3inv): the isogeny is
        X = x + sum_Q [ vQ/(x-xQ) + uQ/(x-xQ)^2 ],  Y = y * dX/dx
    followed by the isomorphism (X, Y) -> (X*s2inv, Y*s3inv) onto y^2 = x^3 + 4."""
    global _ISO
    if _ISO is not None:
        return _ISO
    p = _P
    n = _COFACTOR * BLS12_381.r
    assert n % 11 == 0
    # deterministic point of order 11
    x = 0
    K = None
    while K is None:
        x += 1
        y = _fp_sqrt((x * x * x + ISO_A * x + ISO_B) % p)
        if y is None:
            continue
        Pt = (x, y)
        assert _ep_mul(Pt, n) is None, "E' does not have the order of E: wrong ISO_A/ISO_B"
        K = _ep_mul(Pt, n // 11)
    assert _ep_mul(K, 11) is None
    ker = []
    Q = K
    v = w = 0
    for _ in range(5):                       # K, 2K, .., 5K : representatives mod +-
        xQ, yQ = Q
        gx = (3 * xQ * xQ + ISO_A) % p
        gy = (-2 * yQ) % p
        vQ = 2 * gx % p
        uQ = gy * gy % p
        ker.append((xQ, vQ, uQ))
        v = (v + vQ) % p
        w = (w + uQ + xQ * vQ) % p
        Q = _ep_add(Q, K)
    A2 = (ISO_A - 5 * v) % p
    B2 = (ISO_B - 7 * w) % p
    assert A2 == 0, "codomain is not j = 0"
    # B2 = 4 * s^6 ; all six sixth roots s
    t = B2 * pow(4, -1, p) % p
    # p = 1 mod 3 ; find one sixth root by trying sqrt then cube root via exponent search
    roots = []
    # brute: s^6 = t.  Use a generator-free approach: factor x^6 - t via sqrt + cube roots.
    sq = _fp_sqrt(t)
    assert sq is not None
    for sgn in (sq, (-sq) % p):
        # cube roots of sgn: p = 1 mod 3.  (p-1)/3 exponent test, then Adleman-Manders-Miller
        # shortcut: since p = 10 mod 27?  fall back to a simple search using a primitive cube
        # root of unity and exponent inverse when 3 || (p-1) fails -> general AMM below.
        for c in _cube_roots(sgn):
            roots.append(c)
    roots = sorted(set(roots))
    assert len(roots) == 6 and all(pow(s, 6, p) == t for s in roots)
    _ISO = (ker, roots)
    return _ISO


def _cube_roots(a):
    """All cube roots of a in Fp (p = 1 mod 3), possibly none."""
    p = _P
    if pow(a, (p - 1) // 3, p) != 1:
        return []
    # write p - 1 = 3^e * m
    m = p - 1
    e = 0
    while m % 3 == 0:
        m //= 3
        e += 1
    # non-residue g
    g = 2
    while pow(g, (p - 1) // 3, p) == 1:
        g += 1
    gm = pow(g, m, p)                        # generator of the 3-Sylow subgroup (order 3^e)
    # Tonelli-Shanks style for cube roots
    # find k with a^m = gm^(k), k multiple of 3 ; then root = a^((m*? +1)/3) ...
    # Simple approach: solve discrete log in the 3-group (order 3^e, e small).
    am = pow(a, m, p)
    order = 3 ** e
    # brute-force dlog, e is tiny for this prime
    k = None
    acc = 1
    for i in range(order):
        if acc == am:
            k = i
            break
        acc = acc * gm % p
    assert k is not None and k % 3 == 0
    # a = a^(m*inv) ... choose t with 3*t = 1 mod m
    tinv = pow(3, -1, m)
    # a^(3 tinv) = a^(1 + j m) = a * am^j  with j = (3*tinv - 1)/m
    j = (3 * tinv - 1) // m
    # root0^3 = a * am^j  => correct by gm^(-k j / 3)
    root0 = pow(a, tinv, p)
    corr = pow(gm, (order - (k * j // 3) % order) % order, p)
    root = root0 * corr % p
    assert pow(root, 3, p) == a % p
    w3 = pow(g, (p - 1) // 3, p)
    return [root, root * w3 % p, root * w3 * w3 % p]


def iso_map(Pt, scale_index=None):
    """E' -> E (y^2 = x^3 + 4)."""
    if Pt is None:
        return None
    p = _P
    ker, roots = _derive_isogeny()
    idx = ISO_SCALE_INDEX if scale_index is None else scale_index
    s = roots[idx]
    x, y = Pt
    X = x
    dX = 1
    for xQ, vQ, uQ in ker:
        d = (x - xQ) % p
        if d == 0:
            return None                      # kernel point -> identity
        di = pow(d, -1, p)
        di2 = di * di % p
        X = (X + vQ * di + uQ * di2) % p
        dX = (dX - vQ * di2 - 2 * uQ * di2 * di) % p
    Y = y * dX % p
    s2i = pow(s * s, -1, p)
    s3i = pow(s * s * s, -1, p)
    return (X * s2i % p, Y * s3i % p)


def _sgn0(x):
    return x & 1


def map_to_curve_sswu(u):
    """Simplified SWU onto E' (RFC 9380 6.6.2)."""
    p = _P
    A, B, Z = ISO_A, ISO_B, SSWU_Z
    u2 = u * u % p
    tv1 = (Z * Z * u2 * u2 + Z * u2) % p
    if tv1 == 0:
        x1 = B * pow(Z * A, -1, p) % p
    else:
        x1 = (-B) * pow(A, -1, p) % p * (1 + pow(tv1, -1, p)) % p
    gx1 = (x1 * x1 * x1 + A * x1 + B) % p
    y1 = _fp_sqrt(gx1)
    if y1 is not None:
        x, y = x1, y1
    else:
        x2 = Z * u2 % p * x1 % p
        gx2 = (x2 * x2 * x2 + A * x2 + B) % p
        x, y = x2, _fp_sqrt(gx2)
        assert y is not None
    if _sgn0(u) != _sgn0(y):
        y = (-y) % p
    return (x, y)


def hash_to_g1_bls(msg: bytes, dst: bytes, scale_index=None):
    """interface_utilities.rs:30-44 (zkcrypto hash_to_curve, RO variant)."""
    p = _P
    uniform = expand_message(msg, dst, 128)
    u0 = int.from_bytes(uniform[:64], "big") % p
    u1 = int.from_bytes(uniform[64:], "big") % p
    Q0 = iso_map(map_to_curve_sswu(u0), scale_index)
    Q1 = iso_map(map_to_curve_sswu(u1), scale_index)
    R = BLS12_381.g1_add(Q0, Q1)
    return BLS12_381.g1_mul(R, H_EFF)


# ---------------------------------------------------------------------------------------------
# BN254 hash-to-G1: interface_utilities.rs:24-28 calls crate bn254_hash2curve 0.1.2 (not vendored).  Restated as
# RFC 9380 hash_to_curve with expand_message_xmd(SHA-256), L = 48, the Shallue-van de Woestijne map (6.6.1,
# straight-line version) with Z = 1 on y^2 = x^3 + 3 and cofactor 1.  PINNED by the reference's one BN254
# known answer: P1 of constants.rs:39-51 is, per test_vector.rs:21-25, the first generator under the seed
# "...BP_MESSAGE_GENERATOR_SEED" -- reproduced exactly (tests/test_oracle_kat.py).  That vector does not exercise
# the sign of the constant c3 (both candidate x are never squares at once in it); the RFC's sgn0(c3) = 0 is used,
# which gives c3 = 8815841940592487685674414971303048083897117035520822607866, the constant of gnark-crypto's BN254
# hash-to-curve (c2 and c4 coincide with it as well).
from .curves import BN254  # noqa: E402

_BN_P = BN254.p
SVDW_Z = 1


def _bn_sqrt(a):
    a %= _BN_P
    r = pow(a, (_BN_P + 1) // 4, _BN_P)          # p = 3 mod 4
    return r if r * r % _BN_P == a else None


def _bn_g(x):
    return (x * x * x + 3) % _BN_P


_gz = _bn_g(SVDW_Z)
SVDW_C1 = _gz
SVDW_C2 = (-SVDW_Z * pow(2, _BN_P - 2, _BN_P)) % _BN_P
SVDW_C3 = _bn_sqrt(-_gz * 3 * SVDW_Z * SVDW_Z)
if SVDW_C3 % 2:
    SVDW_C3 = _BN_P - SVDW_C3
SVDW_C4 = (-4 * _gz * pow(3 * SVDW_Z * SVDW_Z, _BN_P - 2, _BN_P)) % _BN_P


def map_to_curve_svdw(u):
    p = _BN_P
    tv1 = u * u % p * SVDW_C1 % p
    tv2 = (1 + tv1) % p
    tv1 = (1 - tv1) % p
    tv3 = pow(tv1 * tv2 % p, p - 2, p)            # inv0
    tv4 = u * tv1 % p * tv3 % p * SVDW_C3 % p
    x1 = (SVDW_C2 - tv4) % p
    e1 = _bn_sqrt(_bn_g(x1)) is not None
    x2 = (SVDW_C2 + tv4) % p
    e2 = (_bn_sqrt(_bn_g(x2)) is not None) and not e1
    x3 = tv2 * tv2 % p * tv3 % p
    x3 = (x3 * x3 % p * SVDW_C4 + SVDW_Z) % p
    x = x1 if e1 else (x2 if e2 else x3)
    y = _bn_sqrt(_bn_g(x))
    assert y is not None
    if (u % 2) != (y % 2):
        y = p - y
    return (x, y)


def hash_to_g1_bn(msg: bytes, dst: bytes):
    uniform = expand_message(msg, dst, 96)
    u0 = int.from_bytes(uniform[:48], "big") % _BN_P
    u1 = int.from_bytes(uniform[48:], "big") % _BN_P
    return BN254.g1_add(map_to_curve_svdw(u0), map_to_curve_svdw(u1))     # cofactor 1
