"""Field towers, G1/G2 groups and optimal-ate pairings for BLS12-381 and BN254.

ORACLE (test infrastructure, see oracle/__init__.py).  Plain Python integers,
affine coordinates, textbook formulas -- written to be obviously correct, not
fast.  It restates the published algorithms of the third-party crates the
reference delegates to (they are not vendored under /root/reference):

* ark-ff / ark-ec 0.4.2 + ark-bls12-381 0.4.0 + ark-bn254 0.4.0:
  Fp, Fp2 = Fp[u]/(u^2+1), Fp12 = Fp2[w]/(w^6 - xi), short-Weierstrass groups,
  ``E::pairing`` (optimal ate + final exponentiation).  Reference call sites:
  src/verify.rs:88-92, src/proof_verify.rs:112-115 (pairings),
  src/sign.rs:120-130, src/proof_gen.rs:249-263, src/proof_verify.rs:163-182
  (G1 scalar multiplications), src/key_gen.rs:83-89 (G2 scalar multiplication).
* Only the *boolean* ``pairing product == ONE`` ever leaves the reference, so
  any fixed non-degenerate power of the reduced pairing is equivalent; we use
  the plain exponent (p^12-1)/r.

Fp12 is represented as 6 Fp2 coefficients over the basis 1, w, .., w^5 with
w^6 = xi.  A point is ``None`` (identity) or an (x, y) tuple.
"""

from __future__ import annotations


class Curve:
    """Parameters + arithmetic for one pairing-friendly curve."""

    def __init__(self, name, p, r, b, xi, twist, g1, g2, x_param, fp_bytes):
        self.name = name
        self.p = p
        self.r = r
        self.b = b                  # E : y^2 = x^3 + b over Fp
        self.xi = xi                # Fp2 non-residue, w^6 = xi
        self.twist = twist          # "M" (b' = b*xi) or "D" (b' = b/xi)
        self.g1 = g1
        self.g2 = g2
        self.x_param = x_param      # curve parameter (signed)
        self.fp_bytes = fp_bytes
        if twist == "M":
            self.b2 = self.f2_mul((b, 0), xi)
        else:
            self.b2 = self.f2_mul((b, 0), self.f2_inv(xi))

    # ------------------------------------------------------------------ Fp2
    def f2_add(self, a, b):
        p = self.p
        return ((a[0] + b[0]) % p, (a[1] + b[1]) % p)

    def f2_sub(self, a, b):
        p = self.p
        return ((a[0] - b[0]) % p, (a[1] - b[1]) % p)

    def f2_neg(self, a):
        p = self.p
        return ((-a[0]) % p, (-a[1]) % p)

    def f2_mul(self, a, b):
        p = self.p
        return ((a[0] * b[0] - a[1] * b[1]) % p, (a[0] * b[1] + a[1] * b[0]) % p)

    def f2_sqr(self, a):
        return self.f2_mul(a, a)

    def f2_muls(self, a, s):
        p = self.p
        return ((a[0] * s) % p, (a[1] * s) % p)

    def f2_inv(self, a):
        p = self.p
        n = pow(a[0] * a[0] + a[1] * a[1], -1, p)
        return ((a[0] * n) % p, (-a[1] * n) % p)

    def f2_conj(self, a):
        return (a[0], (-a[1]) % self.p)

    def f2_pow(self, a, e):
        res = (1, 0)
        base = a
        while e:
            if e & 1:
                res = self.f2_mul(res, base)
            base = self.f2_sqr(base)
            e >>= 1
        return res

    def f2_sqrt(self, a):
        """Some square root of a in Fp2, or None (u^2 = -1, p = 3 mod 4)."""
        p = self.p
        if a == (0, 0):
            return (0, 0)
        # complex method: a = a0 + a1 u ; norm n = a0^2 + a1^2
        n = (a[0] * a[0] + a[1] * a[1]) % p
        s = pow(n, (p + 1) // 4, p)
        if s * s % p != n:
            return None
        inv2 = pow(2, -1, p)
        for sg in (s, (-s) % p):
            t = (a[0] + sg) * inv2 % p
            x0 = pow(t, (p + 1) // 4, p)
            if x0 * x0 % p != t:
                continue
            if x0 == 0:
                continue
            x1 = a[1] * pow(2 * x0, -1, p) % p
            cand = (x0, x1)
            if self.f2_sqr(cand) == (a[0] % p, a[1] % p):
                return cand
        # a1 == 0 and a0 a non-residue: root is purely imaginary
        if a[1] % p == 0:
            t = (-a[0]) % p
            x1 = pow(t, (p + 1) // 4, p)
            if x1 * x1 % p == t:
                return (0, x1)
        return None

    # ----------------------------------------------------------------- Fp12
    def f12_one(self):
        return [(1, 0)] + [(0, 0)] * 5

    def f12_mul(self, a, b):
        z = (0, 0)
        t = [z] * 11
        for i in range(6):
            ai = a[i]
            if ai == z:
                continue
            for j in range(6):
                bj = b[j]
                if bj == z:
                    continue
                t[i + j] = self.f2_add(t[i + j], self.f2_mul(ai, bj))
        out = []
        for k in range(6):
            if k + 6 < 11:
                out.append(self.f2_add(t[k], self.f2_mul(t[k + 6], self.xi)))
            else:
                out.append(t[k])
        return out

    def f12_sqr(self, a):
        return self.f12_mul(a, a)

    def f12_pow(self, a, e):
        res = self.f12_one()
        for bit in bin(e)[2:]:
            res = self.f12_sqr(res)
            if bit == "1":
                res = self.f12_mul(res, a)
        return res

    def f12_conj(self, a):
        """a^(p^6): w -> -w."""
        return [a[i] if i % 2 == 0 else self.f2_neg(a[i]) for i in range(6)]

    def f12_frob(self, a):
        """a^p.  (c w^i)^p = conj(c) * xi^(i (p-1)/6) * w^i."""
        if not hasattr(self, "_frob"):
            self._frob = [self.f2_pow(self.xi, i * (self.p - 1) // 6) for i in range(6)]
        return [self.f2_mul(self.f2_conj(a[i]), self._frob[i]) for i in range(6)]

    def f12_inv(self, a):
        """Inverse via the norm down to Fp6-free route: a^-1 = conj-products / norm.

        Uses a^(p^6) trick: a * a^(p^6) lies in Fp6 = even powers of w; then
        invert that cubic extension element over Fp2 by the adjugate formula.
        """
        ac = self.f12_conj(a)
        n = self.f12_mul(a, ac)          # only even coefficients non-zero
        c0, c1, c2 = n[0], n[2], n[4]    # element of Fp2[v]/(v^3 - xi), v = w^2
        m, s, xi = self.f2_mul, self.f2_sub, self.xi
        t0 = s(self.f2_sqr(c0), m(xi, m(c1, c2)))
        t1 = s(m(xi, self.f2_sqr(c2)), m(c0, c1))
        t2 = s(self.f2_sqr(c1), m(c0, c2))
        d = self.f2_add(m(c0, t0), m(xi, self.f2_add(m(c2, t1), m(c1, t2))))
        di = self.f2_inv(d)
        ninv = [m(t0, di), (0, 0), m(t1, di), (0, 0), m(t2, di), (0, 0)]
        return self.f12_mul(ac, ninv)

    # ------------------------------------------------------------- G1 (Fp)
    def g1_is_on_curve(self, P):
        if P is None:
            return True
        x, y = P
        return (y * y - x * x * x - self.b) % self.p == 0

    def g1_neg(self, P):
        if P is None:
            return None
        return (P[0], (-P[1]) % self.p)

    def g1_add(self, P, Q):
        p = self.p
        if P is None:
            return Q
        if Q is None:
            return P
        x1, y1 = P
        x2, y2 = Q
        if x1 == x2:
            if (y1 + y2) % p == 0:
                return None
            lam = 3 * x1 * x1 * pow(2 * y1, -1, p) % p
        else:
            lam = (y2 - y1) * pow(x2 - x1, -1, p) % p
        x3 = (lam * lam - x1 - x2) % p
        return (x3, (lam * (x1 - x3) - y1) % p)

    def g1_mul(self, P, k):
        """k*P for an integer k >= 0 (not reduced mod r: works for any curve point)."""
        R = None
        for bit in bin(k)[2:] if k else "":
            R = self.g1_add(R, R)
            if bit == "1":
                R = self.g1_add(R, P)
        return R

    # ------------------------------------------------------------ G2 (Fp2)
    def g2_is_on_curve(self, Q):
        if Q is None:
            return True
        x, y = Q
        lhs = self.f2_sqr(y)
        rhs = self.f2_add(self.f2_mul(self.f2_sqr(x), x), self.b2)
        return lhs == rhs

    def g2_neg(self, Q):
        if Q is None:
            return None
        return (Q[0], self.f2_neg(Q[1]))

    def g2_add(self, P, Q):
        if P is None:
            return Q
        if Q is None:
            return P
        x1, y1 = P
        x2, y2 = Q
        if x1 == x2:
            if self.f2_add(y1, y2) == (0, 0):
                return None
            lam = self.f2_mul(self.f2_muls(self.f2_sqr(x1), 3), self.f2_inv(self.f2_muls(y1, 2)))
        else:
            lam = self.f2_mul(self.f2_sub(y2, y1), self.f2_inv(self.f2_sub(x2, x1)))
        x3 = self.f2_sub(self.f2_sub(self.f2_sqr(lam), x1), x2)
        y3 = self.f2_sub(self.f2_mul(lam, self.f2_sub(x1, x3)), y1)
        return (x3, y3)

    def g2_mul(self, Q, k):
        R = None
        for bit in bin(k)[2:] if k else "":
            R = self.g2_add(R, R)
            if bit == "1":
                R = self.g2_add(R, Q)
        return R

    def g2_frob(self, Q):
        """The p-power Frobenius of the untwisted point, mapped back to the twist."""
        if Q is None:
            return None
        x, y = Q
        e = (self.p - 1)
        if self.twist == "D":
            gx = self.f2_pow(self.xi, e // 3)
            gy = self.f2_pow(self.xi, e // 2)
        else:
            gx = self.f2_inv(self.f2_pow(self.xi, e // 3))
            gy = self.f2_inv(self.f2_pow(self.xi, e // 2))
        return (self.f2_mul(self.f2_conj(x), gx), self.f2_mul(self.f2_conj(y), gy))

    # -------------------------------------------------------------- pairing
    def _line(self, T, Q2, P):
        """Line through twist points T and Q2 (tangent if equal), evaluated at the G1
        point P, scaled by a factor lying in a proper subfield (killed by the final
        exponentiation).  Returns (value as sparse Fp12, T+Q2)."""
        z = (0, 0)
        xP, yP = P
        xT, yT = T
        if T == Q2:
            lam = self.f2_mul(self.f2_muls(self.f2_sqr(xT), 3), self.f2_inv(self.f2_muls(yT, 2)))
        else:
            if xT == Q2[0]:
                # vertical line: T + Q2 = O ; value x_P - x_T (in a proper subfield after
                # untwisting scale) -> contributes 1 after final exponentiation.
                return self.f12_one(), None
            lam = self.f2_mul(self.f2_sub(Q2[1], yT), self.f2_inv(self.f2_sub(Q2[0], xT)))
        x3 = self.f2_sub(self.f2_sub(self.f2_sqr(lam), xT), Q2[0])
        y3 = self.f2_sub(self.f2_mul(lam, self.f2_sub(xT, x3)), yT)
        c = self.f2_sub(self.f2_mul(lam, xT), yT)          # lam*xT - yT
        lx = self.f2_muls(self.f2_neg(lam), xP)            # -lam*xP
        if self.twist == "M":
            # l * w^3 = (lam xT - yT) + (-lam xP) w^2 + yP w^3
            val = [c, z, lx, (yP % self.p, 0), z, z]
        else:
            # l = yP + (-lam xP) w + (lam xT - yT) w^3
            val = [(yP % self.p, 0), lx, z, c, z, z]
        return val, (x3, y3)

    def miller_loop(self, P, Q):
        """Optimal-ate Miller function f(P, Q), P in G1, Q on the twist.  Identity in
        either slot gives 1 (ark-ec skips such pairs)."""
        if P is None or Q is None:
            return self.f12_one()
        f = self.f12_one()
        T = Q
        if self.name == "bls12_381":
            n = abs(self.x_param)
            for bit in bin(n)[3:]:
                f = self.f12_sqr(f)
                l, T = self._line(T, T, P)
                f = self.f12_mul(f, l)
                if bit == "1":
                    l, T = self._line(T, Q, P)
                    f = self.f12_mul(f, l)
            if self.x_param < 0:
                f = self.f12_conj(f)
            return f
        # BN254: loop over 6x+2, then two Frobenius line steps
        n = 6 * self.x_param + 2
        for bit in bin(n)[3:]:
            f = self.f12_sqr(f)
            l, T = self._line(T, T, P)
            f = self.f12_mul(f, l)
            if bit == "1":
                l, T = self._line(T, Q, P)
                f = self.f12_mul(f, l)
        Q1 = self.g2_frob(Q)
        Q2 = self.g2_neg(self.g2_frob(Q1))
        l, T = self._line(T, Q1, P)
        f = self.f12_mul(f, l)
        l, T = self._line(T, Q2, P)
        f = self.f12_mul(f, l)
        return f

    def final_exp(self, f):
        # easy part with Frobenius maps, hard part by plain exponentiation
        p, r = self.p, self.r
        t = self.f12_mul(self.f12_conj(f), self.f12_inv(f))          # f^(p^6-1)
        t = self.f12_mul(self.f12_frob(self.f12_frob(t)), t)         # ^(p^2+1)
        return self.f12_pow(t, (p ** 4 - p ** 2 + 1) // r)

    def pairing(self, P, Q):
        return self.final_exp(self.miller_loop(P, Q))

    def pairing_product_is_one(self, pairs):
        """prod e(P_i, Q_i) == 1 ?  (reference: verify.rs:88-92, proof_verify.rs:112-115
        multiply two full pairings in GT and compare with ONE)."""
        f = self.f12_one()
        for P, Q in pairs:
            f = self.f12_mul(f, self.miller_loop(P, Q))
        return self.final_exp(f) == self.f12_one()


def _bls12_381():
    p = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
    r = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
    g1 = (
        0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB,
        0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1,
    )
    g2 = (
        (
            0x024AA2B2F08F0A91260805272DC51051C6E47AD4FA403B02B4510B647AE3D1770BAC0326A805BBEFD48056C8C121BDB8,
            0x13E02B6052719F607DACD3A088274F65596BD0D09920B61AB5DA61BBDC7F5049334CF11213945D57E5AC7D055D042B7E,
        ),
        (
            0x0CE5D527727D6E118CC9CDC6DA2E351AADFD9BAA8CBDD3A76D429A695160D12C923AC9CC3BACA289E193548608B82801,
            0x0606C4A02EA734CC32ACD2B02BC28B99CB3E287E85A763AF267492AB572E99AB3F370D275CEC1DA1AAA9075FF05F79BE,
        ),
    )
    return Curve("bls12_381", p, r, 4, (1, 1), "M", g1, g2, -0xD201000000010000, 48)


def _bn254():
    p = 21888242871839275222246405745257275088696311157297823662689037894645226208583
    r = 21888242871839275222246405745257275088548364400416034343698204186575808495617
    g1 = (1, 2)
    g2 = (
        (
            10857046999023057135944570762232829481370756359578518086990519993285655852781,
            11559732032986387107991004021392285783925812861821192530917403151452391805634,
        ),
        (
            8495653923123431417604973247489272438418190587263600148770280649306958101930,
            4082367875863433681332203403145435568316851327593401208105741076214120093531,
        ),
    )
    return Curve("bn254", p, r, 3, (9, 1), "D", g1, g2, 4965661367192848881, 32)


BLS12_381 = _bls12_381()
BN254 = _bn254()
CURVES = {"bls12_381": BLS12_381, "bn254": BN254}
