#!/usr/bin/env python3
"""Generates bbs_sign_amd/csrc/params_gen.hpp: Montgomery constants, curve constants and
Frobenius coefficients for BLS12-381 and BN254 as 32-bit little-endian limb arrays.

Self-contained (plain Python integers; does not import the oracle).  Re-run after editing:
    python tools/gen_params.py
"""
import os

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                   "bbs_sign_amd", "csrc", "params_gen.hpp")


def limbs(v, n):
    assert 0 <= v < (1 << (32 * n))
    return [(v >> (32 * i)) & 0xFFFFFFFF for i in range(n)]


def arr(name, v, n):
    body = ", ".join("0x%08xu" % w for w in limbs(v, n))
    return "    static constexpr uint32_t %s[%d] = {%s};\n" % (name, n, body)


def limbs28(v, n):
    assert 0 <= v < (1 << (28 * n))
    return [(v >> (28 * i)) & 0xFFFFFFF for i in range(n)]


def arr28(name, v, n):
    body = ", ".join("0x%07xu" % w for w in limbs28(v, n))
    return "    static constexpr uint32_t %s[%d] = {%s};\n" % (name, n, body)


def raw_arr(name, ws):
    body = ", ".join("0x%08xu" % w for w in ws)
    return "    static constexpr uint32_t %s[%d] = {%s};\n" % (name, len(ws), body)


def field28_struct(name, mod, n, nc, bound_mult):
    """Base field in radix 2^28 (carry-free multiply-accumulate columns, see field.hpp)."""
    R = 1 << (28 * n)
    inv = (-pow(mod, -1, 1 << 28)) % (1 << 28)
    E = mod.bit_length()
    qshift = E - 28 * (n - 1)
    assert qshift > 0
    # closure of the lazy add/sub: results stay < bound_mult * p (see field.hpp)
    alpha = (1 << E) / mod
    qmax = int(2 * bound_mult / alpha)
    assert alpha + qmax * (alpha - 1) < bound_mult, (name, alpha, qmax)
    # subtraction constant: SUBM = bound_mult * p with limbs adjusted so that every lower limb has
    # 2^28 borrowed from the limb above (limbs >= 2^28 - 1 >= any normalised limb of b)
    m = limbs28(bound_mult * mod, n)
    adj = [m[0] + (1 << 28)] + [m[i] + (1 << 28) - 1 for i in range(1, n - 1)] + [m[n - 1] - 1]
    assert sum(a << (28 * i) for i, a in enumerate(adj)) == bound_mult * mod
    assert all(0 <= a < (1 << 30) for a in adj)
    s = "struct %s {\n" % name
    s += "    static constexpr int W = 28;            // limb width (bits)\n"
    s += "    static constexpr int N = %d;            // internal limbs\n" % n
    s += "    static constexpr int NC = %d;           // canonical 32-bit words\n" % nc
    s += "    static constexpr int BITS = %d;\n" % E
    s += "    static constexpr int BOUND = %d;        // values are kept < BOUND * p\n" % bound_mult
    s += "    static constexpr int QSHIFT = %d;       // top limb >> QSHIFT estimates floor(v / 2^BITS)\n" % qshift
    s += "    static constexpr uint32_t INV = 0x%07xu;   // -mod^-1 mod 2^28\n" % inv
    s += arr28("MOD", mod, n)
    s += arr28("MOD2", 2 * mod, n)
    s += arr28("MODB", bound_mult * mod, n)   # BOUND * p, added before a subtraction
    # 16 p^2 in 2N-1 product columns.  Column k is lifted by L_k (a power of two >= twice the largest
    # possible column sum of a product of two normal elements), borrowed from column k+1: added to
    # t0 - t1 in the fused Fp2 multiplication so that every column stays non-negative.
    ncol = 2 * n - 1
    amax = [(1 << 28) - 1] * (n - 1) + [((bound_mult * mod) >> (28 * (n - 1))) + 1]
    maxcol = [sum(amax[i] * amax[k - i] for i in range(n) if 0 <= k - i < n) for k in range(ncol)]
    lift = []
    for k in range(ncol - 1):
        L = 1 << 28
        while L < 2 * maxcol[k]:
            L <<= 1
        lift.append(L)
    adj = None
    for K in (16, 32, 64, 128):
        w16 = K * mod * mod
        cols = [(w16 >> (28 * k)) & 0xFFFFFFF for k in range(ncol - 1)] + [w16 >> (28 * (ncol - 1))]
        cand = [cols[k] + (lift[k] if k < ncol - 1 else 0) - ((lift[k - 1] >> 28) if k > 0 else 0) for k in range(ncol)]
        assert sum(v << (28 * k) for k, v in enumerate(cand)) == w16
        ok = all(cand[k] >= maxcol[k] and cand[k] + maxcol[k] + n * (1 << 56) + (1 << 40) < (1 << 63) for k in range(ncol))
        # value bound of the fused product: (t0 - t1 + K p^2) / R + p must stay below 2p
        ok = ok and (4 * bound_mult * bound_mult + K) * mod * mod < (mod << (28 * n))
        if ok:
            adj = cand
            break
    assert adj is not None, name
    s += "    // K p^2 (K = %d) in lifted product columns, see r28::f2mul\n" % K
    s += "    static constexpr uint64_t WP2[%d] = {%s};\n" % (2 * n - 1, ", ".join("0x%016xull" % v for v in adj))
    # the same for a SUM of products of total weight <= 6 accumulated column-wise BEFORE any reduction (r28::cols_*,
    # tower.hpp F2Acc): column k lifted by >= 6 * maxcol[k] (a multiple of 2^28, borrowed from column k+1)
    WSUM = 6
    lift6 = [-(-(WSUM * maxcol[k] + (1 << 40)) // (1 << 28)) * (1 << 28) for k in range(ncol - 1)]     # margin covers the borrow
    adj6 = None
    for K6 in (32, 64, 128, 256, 512, 1024, 2048):
        w = K6 * mod * mod
        cols = [(w >> (28 * k)) & 0xFFFFFFF for k in range(ncol - 1)] + [w >> (28 * (ncol - 1))]
        cand = [cols[k] + (lift6[k] if k < ncol - 1 else 0) - ((lift6[k - 1] >> 28) if k > 0 else 0) for k in range(ncol)]
        assert sum(v << (28 * k) for k, v in enumerate(cand)) == w
        # every column: lift covers the subtracted sum; t0 + lift + the reduction's additions stay below 2^64
        ok = all(cand[k] >= WSUM * maxcol[k] and cand[k] + WSUM * maxcol[k] + n * (1 << 56) + (1 << 40) < (1 << 64) for k in range(ncol))
        # the imaginary columns sum (a0 b1 + a1 b0) of total weight 6 fit 64 bits (the Karatsuba sum column may wrap
        # modulo 2^64; the subtraction is done modulo 2^64 too, see tower.hpp f2acc_mac)
        ok = ok and 2 * WSUM * n * ((1 << 28) - 1) ** 2 < (1 << 64)
        # value bound: (6 (BOUND p)^2 + K p^2) / R + p < 2p
        ok = ok and (WSUM * bound_mult * bound_mult + K6) * mod * mod < (mod << (28 * n))
        if ok:
            adj6 = cand
            break
    assert adj6 is not None, name
    s += "    // K p^2 (K = %d) lifted for column sums of total weight <= 6, see r28::cols_* / F2Acc\n" % K6
    s += "    static constexpr uint64_t WP2X[%d] = {%s};\n" % (2 * n - 1, ", ".join("0x%016xull" % v for v in adj6))
    # BOUND * p with 2^28 borrowed into every lower limb: a - b + SUBM is non-negative limb by limb
    mb = limbs28(bound_mult * mod, n)
    sub = [mb[0] + (1 << 28)] + [mb[i] + (1 << 28) - 1 for i in range(1, n - 1)] + [mb[n - 1] - 1]
    assert sum(a << (28 * i) for i, a in enumerate(sub)) == bound_mult * mod
    s += raw_arr("SUBM", sub)
    # quotient estimate for fe_lin: q = (T * RECIP) >> RSHIFT with T = the top QK limbs of the value
    qk = 1 if (mod >> (28 * (n - 1))) >= (1 << 12) else 2
    unit_bits = 28 * (n - qk)
    pt = mod / float(1 << unit_bits)
    rshift = 40 if qk == 1 else 58
    recip = (1 << (rshift + unit_bits)) // mod
    assert recip < (1 << 32) and pt > 1000, (name, recip, pt)
    s += "    static constexpr int QK = %d;               // limbs used for the quotient estimate of fe_lin\n" % qk
    s += "    static constexpr int RSHIFT = %d;\n" % rshift
    s += "    static constexpr uint32_t RECIP = 0x%08xu;  // floor(2^RSHIFT * 2^(28 (N - QK)) / p)\n" % recip
    s += arr28("ONE", R % mod, n)
    s += arr28("R2", R * R % mod, n)
    s += arr("MODC", mod, nc)                # canonical 32-bit limbs
    s += arr("HALF", (mod - 1) // 2, nc)
    s += arr("MOD_M2", mod - 2, nc)
    s += modinv30_consts(mod)
    s += "};\n\n"
    return s


def f2_mul(a, b, p):
    return ((a[0] * b[0] - a[1] * b[1]) % p, (a[0] * b[1] + a[1] * b[0]) % p)


def f2_pow(a, e, p):
    r = (1, 0)
    while e:
        if e & 1:
            r = f2_mul(r, a, p)
        a = f2_mul(a, a, p)
        e >>= 1
    return r



def modinv30_consts(mod):
    """Constants of the safegcd inversion (field.hpp modinv30): modulus in 30-bit limbs, its inverse mod 2^30, and
    the number of 30-step batches >= the Bernstein-Yang bound floor((49 d + 80) / 17) for d-bit inputs."""
    bits = mod.bit_length()
    nl = (bits + 1 + 29) // 30
    limbs30 = [(mod >> (30 * i)) & 0x3FFFFFFF for i in range(nl)]
    minv = pow(mod, -1, 1 << 30)
    batches = -(-((49 * bits + 80) // 17) // 30)
    s = "    static constexpr int NL30 = %d;            // 30-bit limbs of the safegcd inversion\n" % nl
    s += "    static constexpr int INV_BATCHES = %d;     // batches of 30 divsteps\n" % batches
    s += "    static constexpr uint32_t MINV30 = 0x%08xu; // mod^-1 mod 2^30\n" % minv
    s += "    static constexpr uint32_t MOD30[%d] = {%s};\n" % (nl, ", ".join("0x%08xu" % v for v in limbs30))
    return s


def field_struct(name, mod, n):
    R = 1 << (32 * n)
    inv = (-pow(mod, -1, 1 << 32)) % (1 << 32)
    s = "struct %s {\n" % name
    s += "    static constexpr int W = 32;\n"
    s += "    static constexpr int N = %d;\n" % n
    s += "    static constexpr int NC = %d;\n" % n
    s += "    static constexpr int BITS = %d;\n" % mod.bit_length()
    s += "    static constexpr uint32_t INV = 0x%08xu;   // -mod^-1 mod 2^32\n" % inv
    s += arr("MOD", mod, n)
    s += arr("ONE", R % mod, n)            # R mod p  (Montgomery 1)
    s += arr("R2", R * R % mod, n)         # R^2 mod p
    s += arr("R3", R * R * R % mod, n)     # R^3 mod p
    s += arr("MODC", mod, n)
    s += arr("HALF", (mod - 1) // 2, n)    # (p-1)/2 : y > HALF <=> lexicographically largest
    s += arr("MOD_M2", mod - 2, n)         # exponent for Fermat inversion
    s += modinv30_consts(mod)
    s += "};\n\n"
    return s


def curve_struct(tag, p, r, n, b, xi, twist, g1, g2, p1, xabs, xneg, loop_bits_desc):
    R = 1 << (28 * n)
    m = lambda v: v * R % p
    arr = arr28
    limbs = limbs28
    s = "struct %sConsts {\n" % tag
    s += "    static constexpr int N = %d;\n" % n
    s += "    static constexpr bool TWIST_M = %s;\n" % ("true" if twist == "M" else "false")
    s += "    static constexpr bool X_NEG = %s;\n" % ("true" if xneg else "false")
    s += "    static constexpr uint64_t X_ABS = 0x%016xull;\n" % xabs
    s += "    static constexpr uint32_t XI_C0 = %d;   // xi = XI_C0 + u\n" % xi[0]
    s += arr("B_M", m(b), n)
    # cube root of unity beta with (beta x, y) = [-x^2] (x, y) on the prime-order subgroup (BLS12: the
    # endomorphism subgroup test phi(P) == -[x^2] P; validated numerically against [r] P below); unused for cofactor 1
    g_ = 2
    while pow(g_, (p - 1) // 3, p) == 1:
        g_ += 1
    beta = pow(g_, (p - 1) // 3, p)
    s += arr("BETA_M", m(beta), n)
    # GLV split (g1.hpp g1_mul_aff_glv; BLS12 only): lambda = x^2 - 1 is a root of X^2 + X + 1 mod r of half the
    # length of r, so k = k1 + k2 lambda with k2 = floor(k / lambda), k1 = k mod lambda, both < 2^128.
    # BETA_L: the cube root of unity with (BETA_L x, y) = [lambda] (x, y) on G1 (checked on the generator below).
    def aadd(P, Q):
        if P is None:
            return Q
        if Q is None:
            return P
        (x1, y1), (x2, y2) = P, Q
        if x1 == x2:
            if (y1 + y2) % p == 0:
                return None
            l_ = 3 * x1 * x1 * pow(2 * y1, -1, p) % p
        else:
            l_ = (y2 - y1) * pow(x2 - x1, -1, p) % p
        x3 = (l_ * l_ - x1 - x2) % p
        return (x3, (l_ * (x1 - x3) - y1) % p)

    def amul(P, k):
        R_ = None
        while k:
            if k & 1:
                R_ = aadd(R_, P)
            P = aadd(P, P)
            k >>= 1
        return R_

    def beta_for(lam_):
        lg = amul(g1, lam_)
        cands = [c for c in (beta, beta * beta % p) if lg == (c * g1[0] % p, g1[1])]
        assert len(cands) == 1
        return cands[0]

    lam = xabs * xabs - 1
    simple = (lam * lam + lam + 1) % r == 0 and lam.bit_length() <= 128
    s += "    static constexpr bool HAS_GLV = true;\n"
    import math as _m
    cof1 = r > p + 1 - 2 * _m.isqrt(p) - 2          # r within the Hasse interval: the group order is r itself (cofactor 1)
    assert amul(g1, r) is None
    s += "    static constexpr bool GLV_ALWAYS = %s;    // cofactor 1: every on-curve point is in the subgroup, no vouching needed\n" % ("true" if cof1 else "false")
    s += "    static constexpr bool GLV_LATTICE = %s;   // false: k2 = floor(k / lambda); true: rounding against a short basis\n" % ("false" if simple else "true")
    if simple:
        mu = (1 << 256) // lam
        assert mu.bit_length() <= 160
        s += raw_arr("GLV_LAMBDA", [(lam >> (32 * i)) & 0xFFFFFFFF for i in range(4)])
        s += "    // GLV_MU = floor(2^256 / lambda)\n"
        s += raw_arr("GLV_MU", [(mu >> (32 * i)) & 0xFFFFFFFF for i in range(5)])
        s += arr("BETA_L_M", m(beta_for(lam)), n)
    else:
        # BN: lambda is a full-length root of X^2 + X + 1 mod r; short basis (a1, b1), (a2, b2) of the lattice
        # {(x, y): x + y lambda = 0 mod r} from the extended Euclidean algorithm on (r, lambda) (Gallant-Lambert-
        # Vanstone); k = k1 + k2 lambda with (k1, k2) = (k, 0) - c1 (a1, b1) - c2 (a2, b2), c1 = round(b2 k / r),
        # c2 = round(-b1 k / r), computed as (k G + 2^319) >> 320 with G = round(2^320 |b| / r): the identity holds for
        # ANY integers c1, c2, the rounding only decides how short (k1, k2) are -- asserted below over edge and random k.
        gq = 2
        while pow(gq, (r - 1) // 3, r) == 1:
            gq += 1
        lam = pow(gq, (r - 1) // 3, r)
        assert (lam * lam + lam + 1) % r == 0
        rows = [(r, 1, 0), (lam, 0, 1)]                    # (remainder, s, t): remainder = s r + t lambda
        while rows[-1][0] != 0:
            q_ = rows[-2][0] // rows[-1][0]
            rows.append((rows[-2][0] - q_ * rows[-1][0], rows[-2][1] - q_ * rows[-1][1], rows[-2][2] - q_ * rows[-1][2]))
        import math
        sq = math.isqrt(r)
        l_ = max(i for i, row in enumerate(rows) if row[0] >= sq)
        v1 = (rows[l_ + 1][0], -rows[l_ + 1][2])
        c_a, c_b = (rows[l_][0], -rows[l_][2]), (rows[l_ + 2][0], -rows[l_ + 2][2])
        v2 = c_a if c_a[0] ** 2 + c_a[1] ** 2 <= c_b[0] ** 2 + c_b[1] ** 2 else c_b
        (a1, b1), (a2, b2) = v1, v2
        assert (a1 + b1 * lam) % r == 0 and (a2 + b2 * lam) % r == 0 and a1 * b2 - a2 * b1 in (r, -r)
        if a1 * b2 - a2 * b1 == -r:
            (a1, b1), (a2, b2) = (a2, b2), (a1, b1)
        SH = 320
        G1c, G2c = (abs(b2) << SH) // r, (abs(b1) << SH) // r
        G1c += 1 if ((abs(b2) << SH) % r) * 2 >= r else 0
        G2c += 1 if ((abs(b1) << SH) % r) * 2 >= r else 0
        c1neg, c2neg = b2 < 0, -b1 < 0

        def split(k):
            c1 = (k * G1c + (1 << (SH - 1))) >> SH
            c2 = (k * G2c + (1 << (SH - 1))) >> SH
            c1 = -c1 if c1neg else c1
            c2 = -c2 if c2neg else c2
            return k - c1 * a1 - c2 * a2, -c1 * b1 - c2 * b2
        import random
        rng = random.Random(1)
        worst = 0
        for k in [0, 1, 2, r - 1, r - 2, lam, lam + 1, r - lam, (r - 1) // 2, 1 << 128, (1 << 253) + 12345] + [rng.randrange(r) for _ in range(20000)]:
            k1, k2 = split(k)
            assert (k1 + k2 * lam - k) % r == 0
            worst = max(worst, abs(k1), abs(k2))
        assert worst.bit_length() <= 127, worst.bit_length()        # one bit of margin below the 128-bit recoding
        for v in (a1, b1, a2, b2):
            assert abs(v).bit_length() <= 128
        assert G1c.bit_length() <= 224 and G2c.bit_length() <= 224
        w32 = lambda v, cnt: [(abs(v) >> (32 * i)) & 0xFFFFFFFF for i in range(cnt)]
        s += raw_arr("GLV_LAMBDA", w32(lam, 8))
        s += raw_arr("GLV_A1", w32(a1, 4)) + raw_arr("GLV_B1", w32(b1, 4)) + raw_arr("GLV_A2", w32(a2, 4)) + raw_arr("GLV_B2", w32(b2, 4))
        s += "    static constexpr bool GLV_A1_NEG = %s, GLV_B1_NEG = %s, GLV_A2_NEG = %s, GLV_B2_NEG = %s;\n" % tuple(
            "true" if v < 0 else "false" for v in (a1, b1, a2, b2))
        s += "    // c_i = +-((k GLV_Gi + 2^319) >> 320)\n"
        s += raw_arr("GLV_G1", w32(G1c, 7)) + raw_arr("GLV_G2", w32(G2c, 7))
        s += "    static constexpr bool GLV_C1_NEG = %s, GLV_C2_NEG = %s;\n" % ("true" if c1neg else "false", "true" if c2neg else "false")
        s += arr("BETA_L_M", m(beta_for(lam)), n)
    s += arr("B3_M", m(3 * b), n)
    s += arr("G1X_M", m(g1[0]), n) + arr("G1Y_M", m(g1[1]), n)
    s += arr("P1X_M", m(p1[0]), n) + arr("P1Y_M", m(p1[1]), n)
    s += arr("G2X0_M", m(g2[0][0]), n) + arr("G2X1_M", m(g2[0][1]), n)
    s += arr("G2Y0_M", m(g2[1][0]), n) + arr("G2Y1_M", m(g2[1][1]), n)
    # twist curve coefficient b' (Fp2)
    if twist == "M":
        b2 = f2_mul((b, 0), xi, p)
    else:
        d = pow(xi[0] * xi[0] + xi[1] * xi[1], -1, p)
        b2 = f2_mul((b, 0), (xi[0] * d % p, -xi[1] * d % p), p)
    s += arr("B2_C0_M", m(b2[0]), n) + arr("B2_C1_M", m(b2[1]), n)
    # Frobenius coefficients gamma[k][i] = xi^(i (p^k - 1)/6), k = 1..3, i = 0..5
    s += "    // FROB[k-1][i][c] : xi^(i*(p^k-1)/6), Montgomery form, c = 0 real / 1 imaginary\n"
    s += "    static constexpr uint32_t FROB[3][6][2][%d] = {\n" % n
    for k in (1, 2, 3):
        s += "      {\n"
        for i in range(6):
            g = f2_pow(xi, i * (p ** k - 1) // 6, p)
            s += "        {{%s}, {%s}},\n" % (
                ", ".join("0x%07xu" % w for w in limbs(m(g[0]), n)),
                ", ".join("0x%07xu" % w for w in limbs(m(g[1]), n)))
        s += "      },\n"
    s += "    };\n"
    s += "};\n\n"
    return s


def sswu_struct(p, r, n):
    """BLS12-381 G1 simplified-SWU constants (RFC 9380 8.8.1).  The 11-isogeny E' -> E is derived with
    Velu's formulas from the rational subgroup of order 11 of E' (kernel x-coordinates xQ with the
    classical vQ, uQ), followed by the isomorphism (X, Y) -> (X s^-2, Y s^-3) onto y^2 = x^3 + 4 with
    s the smallest sixth root of B''/4 -- the choice the reference's generator vectors pin
    (src/tests/test_vector.rs:123-136)."""
    A = 0x144698A3B8E9433D693A02C96D4982B0EA985383EE66A8D8E8981AEFD881AC98936F8DA0E0F97F5CF428082D584C1D
    B = 0x12E2908D11688030018B12E8753EEE3B2016C1F0F24F4070A0B9C14FCEF35EF55A23215A316CEAA5D1CC48E98E172BE0
    cof = 0x396C8C005555E1568C00AAAB0000AAAB

    def add(P, Q):
        if P is None:
            return Q
        if Q is None:
            return P
        x1, y1 = P
        x2, y2 = Q
        if x1 == x2:
            if (y1 + y2) % p == 0:
                return None
            lam = (3 * x1 * x1 + A) * pow(2 * y1, -1, p) % p
        else:
            lam = (y2 - y1) * pow(x2 - x1, -1, p) % p
        x3 = (lam * lam - x1 - x2) % p
        return (x3, (lam * (x1 - x3) - y1) % p)

    def mul(P, k):
        R = None
        for bit in bin(k)[2:]:
            R = add(R, R)
            if bit == "1":
                R = add(R, P)
        return R

    n_pts = cof * r
    x = 0
    K = None
    while K is None:
        x += 1
        y2 = (x * x * x + A * x + B) % p
        y = pow(y2, (p + 1) // 4, p)
        if y * y % p != y2:
            continue
        assert mul((x, y), n_pts) is None
        K = mul((x, y), n_pts // 11)
    ker = []
    Q = K
    v = w = 0
    for _ in range(5):
        xQ, yQ = Q
        gx = (3 * xQ * xQ + A) % p
        vQ = 2 * gx % p
        uQ = 4 * yQ * yQ % p
        ker.append((xQ, vQ, uQ))
        v = (v + vQ) % p
        w = (w + uQ + xQ * vQ) % p
        Q = add(Q, K)
    assert (A - 5 * v) % p == 0
    t = (B - 7 * w) * pow(4, -1, p) % p
    # all sixth roots of t: brute force over the 3-Sylow/2-Sylow structure via exponent search
    roots = []
    g = 2
    while pow(g, (p - 1) // 2, p) == 1 or pow(g, (p - 1) // 3, p) == 1:
        g += 1
    z6 = pow(g, (p - 1) // 6, p)          # primitive sixth root of unity
    # one sixth root: t^(e) with 6 e = 1 mod (p-1)/gcd..; p-1 = 2 * 3^k * m -> use Tonelli-like search
    # simple approach: s = t^(inv6 mod m') corrected by a power of z; search small corrections
    m = p - 1
    e3 = 0
    while m % 3 == 0:
        m //= 3
        e3 += 1
    e2 = 0
    while m % 2 == 0:
        m //= 2
        e2 += 1
    inv6 = pow(6, -1, m)
    s0 = pow(t, inv6, p)                   # s0^6 = t * (element of the 2,3-Sylow part)
    gen = pow(g, m, p)                     # generates the subgroup of order 2^e2 * 3^e3
    order = (2 ** e2) * (3 ** e3)
    found = None
    acc = 1
    for k in range(order):
        if pow(s0 * acc % p, 6, p) == t:
            found = s0 * acc % p
            break
        acc = acc * gen % p
    assert found is not None
    roots = sorted({found * pow(z6, k, p) % p for k in range(6)})
    assert len(roots) == 6 and all(pow(sx, 6, p) == t for sx in roots)
    sroot = roots[0]
    R = 1 << (28 * n)
    mm = lambda val: val * R % p
    s = "// BLS12-381 G1 hash-to-curve constants (simplified SWU on E', 11-isogeny by Velu, see tools/gen_params.py)\n"
    s += "struct BlsSswu {\n"
    s += arr28("A_M", mm(A), n) + arr28("B_M", mm(B), n) + arr28("Z_M", mm(11), n)
    s += arr28("S2INV_M", mm(pow(sroot * sroot, -1, p)), n) + arr28("S3INV_M", mm(pow(sroot ** 3, -1, p)), n)
    for name, idx in (("KX", 0), ("KV", 1), ("KU", 2)):
        s += "    static constexpr uint32_t %s_M[5][%d] = {\n" % (name, n)
        for k in ker:
            s += "        {%s},\n" % ", ".join("0x%07xu" % wv for wv in limbs28(mm(k[idx]), n))
        s += "    };\n"
    s += arr("SQRT_EXP", (p + 1) // 4, 12)
    s += "    static constexpr uint64_t H_EFF = 0xd201000000010001ull;\n"
    s += "};\n\n"
    return s


def svdw_struct(p, n):
    """BN254 hash-to-curve constants: Shallue-van de Woestijne map (RFC 9380 6.6.1) with Z = 1 on y^2 = x^3 + 3."""
    Z, A, B = 1, 0, 3
    g = lambda x: (x * x * x + A * x + B) % p
    gz = g(Z)
    h = (3 * Z * Z + 4 * A) % p
    assert gz != 0 and h != 0
    t = (-h * pow(4 * gz, -1, p)) % p
    assert t != 0 and pow(t, (p - 1) // 2, p) == 1                    # -(3Z^2+4A)/(4g(Z)) is a non-zero square
    assert pow(gz, (p - 1) // 2, p) == 1 or pow(g((-Z * pow(2, -1, p)) % p), (p - 1) // 2, p) == 1
    c1 = gz
    c2 = (-Z * pow(2, -1, p)) % p
    c3 = pow((-gz * h) % p, (p + 1) // 4, p)
    assert c3 * c3 % p == (-gz * h) % p
    if c3 & 1:
        c3 = p - c3                                                    # sgn0(c3) = 0
    c4 = (-4 * gz * pow(h, -1, p)) % p
    R = 1 << (28 * n)
    mm = lambda val: val * R % p
    s = "// BN254 G1 hash-to-curve constants (Shallue-van de Woestijne, Z = 1; see tools/gen_params.py)\n"
    s += "struct BnSvdw {\n"
    s += arr28("Z_M", mm(Z), n) + arr28("C1_M", mm(c1), n) + arr28("C2_M", mm(c2), n) + arr28("C3_M", mm(c3), n) + arr28("C4_M", mm(c4), n)
    s += arr("SQRT_EXP", (p + 1) // 4, 8)
    s += "};\n\n"
    return s


def main():
    bls_p = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
    bls_r = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
    bn_p = 21888242871839275222246405745257275088696311157297823662689037894645226208583
    bn_r = 21888242871839275222246405745257275088548364400416034343698204186575808495617
    bls_g1 = (0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB,
              0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1)
    bls_g2 = ((0x024AA2B2F08F0A91260805272DC51051C6E47AD4FA403B02B4510B647AE3D1770BAC0326A805BBEFD48056C8C121BDB8,
               0x13E02B6052719F607DACD3A088274F65596BD0D09920B61AB5DA61BBDC7F5049334CF11213945D57E5AC7D055D042B7E),
              (0x0CE5D527727D6E118CC9CDC6DA2E351AADFD9BAA8CBDD3A76D429A695160D12C923AC9CC3BACA289E193548608B82801,
               0x0606C4A02EA734CC32ACD2B02BC28B99CB3E287E85A763AF267492AB572E99AB3F370D275CEC1DA1AAA9075FF05F79BE))
    bls_p1 = (1355253221325668152696183518801331769866100080859571110928822005264442742039790254588065001486134245057142899747017,
              2563071790429735027383427649950865259619709115697058137448106859255609577834149037543606665262210555960464099235249)
    bn_g2 = ((10857046999023057135944570762232829481370756359578518086990519993285655852781,
              11559732032986387107991004021392285783925812861821192530917403151452391805634),
             (8495653923123431417604973247489272438418190587263600148770280649306958101930,
              4082367875863433681332203403145435568316851327593401208105741076214120093531))
    bn_p1 = (7738860219269362160002109478394842060990190871738832255540382874922375322334,
             8255268479661695615178834896135584953541182794935974658059743263102507888551)

    s = "// GENERATED by tools/gen_params.py -- do not edit.\n#pragma once\n#include <cstdint>\n\nnamespace bbs {\n\n"
    s += field28_struct("BlsFpParams", bls_p, 14, 12, 2)
    s += field_struct("BlsFrParams", bls_r, 8)
    s += field28_struct("BnFpParams", bn_p, 10, 8, 3)
    s += field_struct("BnFrParams", bn_r, 8)
    s += curve_struct("Bls", bls_p, bls_r, 14, 4, (1, 1), "M", bls_g1, bls_g2, bls_p1,
                      0xD201000000010000, True, "")
    s += curve_struct("Bn", bn_p, bn_r, 10, 3, (9, 1), "D", (1, 2), bn_g2, bn_p1,
                      4965661367192848881, False, "")
    s += sswu_struct(bls_p, bls_r, 14)
    s += svdw_struct(bn_p, 10)
    s += "}  // namespace bbs\n"
    with open(OUT, "w") as f:
        f.write(s)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
