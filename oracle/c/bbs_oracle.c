/*
 * ORACLE (test infrastructure, see oracle/__init__.py) -- plain-C restatement of the reference's
 * path, 64-bit limbs, reference operation order.  One source, two builds: BLS12-381 (default) and BN254 (-DORC_BN254:
 * 4-limb Fp, xi = 9 + u, D-type twist, Miller loop over 6x + 2 plus the two Frobenius line steps, Devegili-Scott-Dahab
 * hard part, ark-serialize's little-endian compressed encodings -- the BN254 byte formats rest on crate knowledge, see
 * the reference holds no BN254 byte vector).  Used (a) as the CPU baseline of
 * bench.py (`cpu_baseline.kind = "port"`), (b) to check FULL 4096-item GPU batches item by item.
 * It is pinned against the pure-Python oracle (itself pinned by the reference's known-answer
 * vectors) in tests/test_oracle_c.py.  Never linked into the product.
 *
 * Follows /root/reference:
 *   src/utils/utilities_helper.rs:15-97   expand_message, FromOkm
 *   src/utils/core_utilities.rs:11-63     hash_to_scalar, calculate_domain
 *   src/sign.rs:63-133  src/verify.rs:53-93  src/proof_gen.rs:116-365  src/proof_verify.rs:64-188
 * Arithmetic restated from the published algorithms of ark-ff / ark-ec 0.4.2 / ark-bls12-381 0.4.0
 * (not vendored): Montgomery Fp/Fr, Fp2/Fp6/Fp12 tower, Jacobian G1 with plain MSB-first
 * double-and-add (`Projective::mul_bigint`), optimal-ate Miller loop (projective line steps) and a
 * final exponentiation by 3 (p^12-1)/r (easy part + the BLS12 x-chain for the hard part).
 * Each `E::pairing` is a full pairing (own final exponentiation), exactly as the reference calls it.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef uint64_t u64;

/* ------------------------------------------------------------------ Fp (6 limbs) / Fr (4 limbs) */
#ifdef ORC_BN254
#define NP 4
#define NR 4
static const u64 P[NP] = {0x3c208c16d87cfd47ULL, 0x97816a916871ca8dULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
static const u64 RM[NR] = {0x43e1f593f0000001ULL, 0x2833e84879b97091ULL, 0xb85045b68181585dULL, 0x30644e72e131a029ULL};
#else
#define NP 6
#define NR 4
static const u64 P[NP] = {0xb9feffffffffaaabULL, 0x1eabfffeb153ffffULL, 0x6730d2a0f6b0f624ULL,
                          0x64774b84f38512bfULL, 0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL};
static const u64 RM[NR] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL};
#endif
#define FPB (8 * NP)                 /* bytes of a canonical Fp element */
static u64 P_INV, R_INV;            /* -m^-1 mod 2^64 */
static u64 P_R2[NP], P_ONE[NP], R_R2[NR], R_ONE[NR], R_R3[NR];

typedef struct { u64 v[NP]; } fp;
typedef struct { u64 v[NR]; } fr;

static u64 neg_inv64(u64 m) { u64 x = 1; for (int i = 0; i < 6; i++) x *= 2 - m * x; return (u64)0 - x; }

static int geq(const u64* a, const u64* m, int n) {
    for (int i = n - 1; i >= 0; i--) { if (a[i] > m[i]) return 1; if (a[i] < m[i]) return 0; }
    return 1;
}
static void sub_n(u64* r, const u64* a, const u64* b, int n) {
    u64 bw = 0;
    for (int i = 0; i < n; i++) { u128 t = (u128)a[i] - b[i] - bw; r[i] = (u64)t; bw = (u64)(t >> 64) & 1; }
}
static u64 add_n(u64* r, const u64* a, const u64* b, int n) {
    u64 c = 0;
    for (int i = 0; i < n; i++) { u128 t = (u128)a[i] + b[i] + c; r[i] = (u64)t; c = (u64)(t >> 64); }
    return c;
}
static void mont_mul(u64* r, const u64* a, const u64* b, const u64* m, u64 inv, int n) {
    u64 t[8] = {0};
    for (int i = 0; i < n; i++) {
        u128 c = 0;
        for (int j = 0; j < n; j++) { c += (u128)a[j] * b[i] + t[j]; t[j] = (u64)c; c >>= 64; }
        c += t[n]; t[n] = (u64)c; t[n + 1] = (u64)(c >> 64);
        u64 q = t[0] * inv;
        c = (u128)q * m[0] + t[0]; c >>= 64;
        for (int j = 1; j < n; j++) { c += (u128)q * m[j] + t[j]; t[j - 1] = (u64)c; c >>= 64; }
        c += t[n]; t[n - 1] = (u64)c; t[n] = t[n + 1] + (u64)(c >> 64);
    }
    if (t[n] || geq(t, m, n)) sub_n(t, t, m, n);
    memcpy(r, t, n * 8);
}
static void mod_add(u64* r, const u64* a, const u64* b, const u64* m, int n) {
    u64 t[6]; u64 c = add_n(t, a, b, n);
    if (c || geq(t, m, n)) sub_n(t, t, m, n);
    memcpy(r, t, n * 8);
}
static void mod_sub(u64* r, const u64* a, const u64* b, const u64* m, int n) {
    u64 t[6]; u64 bw = 0;
    for (int i = 0; i < n; i++) { u128 x = (u128)a[i] - b[i] - bw; t[i] = (u64)x; bw = (u64)(x >> 64) & 1; }
    if (bw) add_n(t, t, m, n);
    memcpy(r, t, n * 8);
}

/* fully unrolled 6-limb Montgomery product (the generic mont_mul above is kept for Fr) */
static inline fp fp_mul(fp a, fp b) {
    u64 t[NP + 2] = {0};
#pragma GCC unroll 6
    for (int i = 0; i < NP; i++) {
        u128 c = 0;
#pragma GCC unroll 6
        for (int j = 0; j < NP; j++) { c += (u128)a.v[j] * b.v[i] + t[j]; t[j] = (u64)c; c >>= 64; }
        c += t[NP]; t[NP] = (u64)c; t[NP + 1] = (u64)(c >> 64);
        u64 q = t[0] * P_INV;
        c = (u128)q * P[0] + t[0]; c >>= 64;
#pragma GCC unroll 5
        for (int j = 1; j < NP; j++) { c += (u128)q * P[j] + t[j]; t[j - 1] = (u64)c; c >>= 64; }
        c += t[NP]; t[NP - 1] = (u64)c; t[NP] = t[NP + 1] + (u64)(c >> 64);
    }
    fp r;
    if (t[NP] || geq(t, P, NP)) sub_n(t, t, P, NP);
    memcpy(r.v, t, sizeof r.v);
    return r;
}
static fp fp_sqr(fp a) { return fp_mul(a, a); }
static fp fp_add(fp a, fp b) { fp r; mod_add(r.v, a.v, b.v, P, NP); return r; }
static fp fp_sub(fp a, fp b) { fp r; mod_sub(r.v, a.v, b.v, P, NP); return r; }
static fp fp_zero(void) { fp r; memset(&r, 0, sizeof r); return r; }
static fp fp_one(void) { fp r; memcpy(r.v, P_ONE, sizeof r.v); return r; }
static fp fp_neg(fp a) { return fp_sub(fp_zero(), a); }
static fp fp_dbl(fp a) { return fp_add(a, a); }
static int fp_is_zero(fp a) { u64 x = 0; for (int i = 0; i < NP; i++) x |= a.v[i]; return x == 0; }
static int fp_eq(fp a, fp b) { return memcmp(a.v, b.v, sizeof a.v) == 0; }
static fp fp_pow(fp a, const u64* e, int n) {
    fp r = fp_one();
    for (int i = n - 1; i >= 0; i--) for (int b = 63; b >= 0; b--) { r = fp_sqr(r); if ((e[i] >> b) & 1) r = fp_mul(r, a); }
    return r;
}
static fp fp_inv(fp a) { u64 e[NP]; memcpy(e, P, sizeof e); e[0] -= 2; return fp_pow(a, e, NP); }
static fp fp_from_raw(const u64* w) { fp a, r2; memcpy(a.v, w, sizeof a.v); memcpy(r2.v, P_R2, sizeof r2.v); return fp_mul(a, r2); }
static void fp_to_raw(fp a, u64* w) { fp one; memset(&one, 0, sizeof one); one.v[0] = 1; fp r = fp_mul(a, one); memcpy(w, r.v, sizeof r.v); }
static fp fp_from_le(const uint8_t* b) { u64 w[NP]; memcpy(w, b, FPB); return fp_from_raw(w); }
static void fp_to_le(fp a, uint8_t* b) { u64 w[NP]; fp_to_raw(a, w); memcpy(b, w, FPB); }

static fr fr_mul(fr a, fr b) { fr r; mont_mul(r.v, a.v, b.v, RM, R_INV, NR); return r; }
static fr fr_add(fr a, fr b) { fr r; mod_add(r.v, a.v, b.v, RM, NR); return r; }
static fr fr_sub(fr a, fr b) { fr r; mod_sub(r.v, a.v, b.v, RM, NR); return r; }
static int fr_is_zero(fr a) { return (a.v[0] | a.v[1] | a.v[2] | a.v[3]) == 0; }
static fr fr_from_raw(const u64* w) { fr a, r2; memcpy(a.v, w, 32); memcpy(r2.v, R_R2, 32); return fr_mul(a, r2); }
static void fr_to_raw(fr a, u64* w) { fr one; memset(&one, 0, sizeof one); one.v[0] = 1; fr r = fr_mul(a, one); memcpy(w, r.v, 32); }
static fr fr_from_le(const uint8_t* b) { u64 w[NR]; memcpy(w, b, 32); return fr_from_raw(w); }
static void fr_to_le(fr a, uint8_t* b) { u64 w[NR]; fr_to_raw(a, w); memcpy(b, w, 32); }
static void fr_to_be(fr a, uint8_t* b) { uint8_t le[32]; fr_to_le(a, le); for (int i = 0; i < 32; i++) b[i] = le[31 - i]; }
static fr fr_inv(fr a) {
    u64 e[NR]; memcpy(e, RM, sizeof e); e[0] -= 2;
    fr r; memcpy(r.v, R_ONE, 32);
    for (int i = NR - 1; i >= 0; i--) for (int b = 63; b >= 0; b--) { r = fr_mul(r, r); if ((e[i] >> b) & 1) r = fr_mul(r, a); }
    return r;
}

static void init_consts(void) {
    static int done = 0;
    if (done) return;
    P_INV = neg_inv64(P[0]); R_INV = neg_inv64(RM[0]);
    /* R mod m and R^2 mod m by repeated doubling */
    u64 t[NP] = {1};
    for (int i = 0; i < 2 * 64 * NP; i++) { mod_add(t, t, t, P, NP); if (i == 64 * NP - 1) memcpy(P_ONE, t, sizeof t); }
    memcpy(P_R2, t, sizeof t);
    u64 s[NR] = {1, 0, 0, 0};
    for (int i = 0; i < 3 * 64 * NR; i++) {
        mod_add(s, s, s, RM, NR);
        if (i == 64 * NR - 1) memcpy(R_ONE, s, sizeof s);
        if (i == 2 * 64 * NR - 1) memcpy(R_R2, s, sizeof s);
    }
    memcpy(R_R3, s, sizeof s);
    done = 1;
}

/* ------------------------------------------------------------------------------- Fp2/6/12 */
typedef struct { fp c0, c1; } fp2;
typedef struct { fp2 c0, c1, c2; } fp6;
typedef struct { fp6 c0, c1; } fp12;

static fp2 f2_add(fp2 a, fp2 b) { fp2 r = {fp_add(a.c0, b.c0), fp_add(a.c1, b.c1)}; return r; }
static fp2 f2_sub(fp2 a, fp2 b) { fp2 r = {fp_sub(a.c0, b.c0), fp_sub(a.c1, b.c1)}; return r; }
static fp2 f2_neg(fp2 a) { fp2 r = {fp_neg(a.c0), fp_neg(a.c1)}; return r; }
static fp2 f2_conj(fp2 a) { fp2 r = {a.c0, fp_neg(a.c1)}; return r; }
static fp2 f2_zero(void) { fp2 r = {fp_zero(), fp_zero()}; return r; }
static fp2 f2_one(void) { fp2 r = {fp_one(), fp_zero()}; return r; }
static int f2_is_zero(fp2 a) { return fp_is_zero(a.c0) && fp_is_zero(a.c1); }
static int f2_eq(fp2 a, fp2 b) { return fp_eq(a.c0, b.c0) && fp_eq(a.c1, b.c1); }
static fp2 f2_mul(fp2 a, fp2 b) {
    fp t0 = fp_mul(a.c0, b.c0), t1 = fp_mul(a.c1, b.c1);
    fp s = fp_mul(fp_add(a.c0, a.c1), fp_add(b.c0, b.c1));
    fp2 r = {fp_sub(t0, t1), fp_sub(fp_sub(s, t0), t1)};
    return r;
}
static fp2 f2_sqr(fp2 a) { return f2_mul(a, a); }
static fp2 f2_mul_fp(fp2 a, fp s) { fp2 r = {fp_mul(a.c0, s), fp_mul(a.c1, s)}; return r; }
#ifdef ORC_BN254
static fp fp_mul9(fp a) { fp t = fp_dbl(fp_dbl(fp_dbl(a))); return fp_add(t, a); }
static fp2 f2_mul_xi(fp2 a) { fp2 r = {fp_sub(fp_mul9(a.c0), a.c1), fp_add(fp_mul9(a.c1), a.c0)}; return r; }   /* xi = 9 + u */
#else
static fp2 f2_mul_xi(fp2 a) { fp2 r = {fp_sub(a.c0, a.c1), fp_add(a.c0, a.c1)}; return r; }   /* xi = 1 + u */
#endif
static fp2 f2_inv(fp2 a) {
    fp n = fp_inv(fp_add(fp_sqr(a.c0), fp_sqr(a.c1)));
    fp2 r = {fp_mul(a.c0, n), fp_neg(fp_mul(a.c1, n))};
    return r;
}
static fp6 f6_add(fp6 a, fp6 b) { fp6 r = {f2_add(a.c0, b.c0), f2_add(a.c1, b.c1), f2_add(a.c2, b.c2)}; return r; }
static fp6 f6_sub(fp6 a, fp6 b) { fp6 r = {f2_sub(a.c0, b.c0), f2_sub(a.c1, b.c1), f2_sub(a.c2, b.c2)}; return r; }
static fp6 f6_neg(fp6 a) { fp6 r = {f2_neg(a.c0), f2_neg(a.c1), f2_neg(a.c2)}; return r; }
static fp6 f6_mul_v(fp6 a) { fp6 r = {f2_mul_xi(a.c2), a.c0, a.c1}; return r; }
static fp6 f6_mul(fp6 a, fp6 b) {
    fp2 v0 = f2_mul(a.c0, b.c0), v1 = f2_mul(a.c1, b.c1), v2 = f2_mul(a.c2, b.c2);
    fp2 t0 = f2_sub(f2_sub(f2_mul(f2_add(a.c1, a.c2), f2_add(b.c1, b.c2)), v1), v2);
    fp2 t1 = f2_sub(f2_sub(f2_mul(f2_add(a.c0, a.c1), f2_add(b.c0, b.c1)), v0), v1);
    fp2 t2 = f2_sub(f2_sub(f2_mul(f2_add(a.c0, a.c2), f2_add(b.c0, b.c2)), v0), v2);
    fp6 r = {f2_add(v0, f2_mul_xi(t0)), f2_add(t1, f2_mul_xi(v2)), f2_add(t2, v1)};
    return r;
}
static fp6 f6_inv(fp6 a) {
    fp2 t0 = f2_sub(f2_sqr(a.c0), f2_mul_xi(f2_mul(a.c1, a.c2)));
    fp2 t1 = f2_sub(f2_mul_xi(f2_sqr(a.c2)), f2_mul(a.c0, a.c1));
    fp2 t2 = f2_sub(f2_sqr(a.c1), f2_mul(a.c0, a.c2));
    fp2 d = f2_add(f2_mul(a.c0, t0), f2_mul_xi(f2_add(f2_mul(a.c2, t1), f2_mul(a.c1, t2))));
    fp2 di = f2_inv(d);
    fp6 r = {f2_mul(t0, di), f2_mul(t1, di), f2_mul(t2, di)};
    return r;
}
static fp12 f12_one(void) { fp12 r; memset(&r, 0, sizeof r); r.c0.c0 = f2_one(); return r; }
static fp12 f12_mul(fp12 a, fp12 b) {
    fp6 v0 = f6_mul(a.c0, b.c0), v1 = f6_mul(a.c1, b.c1);
    fp6 s = f6_mul(f6_add(a.c0, a.c1), f6_add(b.c0, b.c1));
    fp12 r = {f6_add(v0, f6_mul_v(v1)), f6_sub(f6_sub(s, v0), v1)};
    return r;
}
static fp12 f12_sqr(fp12 a) { return f12_mul(a, a); }
static fp12 f12_conj(fp12 a) { fp12 r = {a.c0, f6_neg(a.c1)}; return r; }
static fp12 f12_inv(fp12 a) {
    fp6 d = f6_sub(f6_mul(a.c0, a.c0), f6_mul_v(f6_mul(a.c1, a.c1)));
    fp6 di = f6_inv(d);
    fp12 r = {f6_mul(a.c0, di), f6_neg(f6_mul(a.c1, di))};
    return r;
}
static int f12_is_one(fp12 a) {
    fp12 o = f12_one();
    return memcmp(&a, &o, sizeof a) == 0;
}
/* Frobenius: coefficient of w^i (i = 0..5) is conj(g_i) * xi^(i (p-1)/6); tower index c0=(g0,g2,g4), c1=(g1,g3,g5) */
static fp2 FROB1[6];
static fp2 f2_pow_u64s(fp2 a, const u64* e, int n) {
    fp2 r = f2_one();
    for (int i = n - 1; i >= 0; i--) for (int b = 63; b >= 0; b--) { r = f2_sqr(r); if ((e[i] >> b) & 1) r = f2_mul(r, a); }
    return r;
}
static void init_frob(void) {
    static int done = 0;
    if (done) return;
    /* e = (p - 1) / 6 */
    u64 e[NP]; memcpy(e, P, sizeof e); e[0] -= 1;
    u128 rem = 0;
    for (int i = NP - 1; i >= 0; i--) { u128 cur = (rem << 64) | e[i]; e[i] = (u64)(cur / 6); rem = cur % 6; }
    fp2 xi = f2_mul_xi(f2_one());
    fp2 g = f2_pow_u64s(xi, e, NP);
    FROB1[0] = f2_one();
    for (int i = 1; i < 6; i++) FROB1[i] = f2_mul(FROB1[i - 1], g);
    done = 1;
}
static fp12 f12_frob(fp12 a) {
    fp12 r;
    r.c0.c0 = f2_conj(a.c0.c0);
    r.c1.c0 = f2_mul(f2_conj(a.c1.c0), FROB1[1]);
    r.c0.c1 = f2_mul(f2_conj(a.c0.c1), FROB1[2]);
    r.c1.c1 = f2_mul(f2_conj(a.c1.c1), FROB1[3]);
    r.c0.c2 = f2_mul(f2_conj(a.c0.c2), FROB1[4]);
    r.c1.c2 = f2_mul(f2_conj(a.c1.c2), FROB1[5]);
    return r;
}

/* ----------------------------------------------------------------------------------- G1 */
typedef struct { fp x, y; int inf; } g1a;
typedef struct { fp x, y, z; } g1j;       /* z == 0: identity */
static fp FP_B;                            /* 4 */
static g1j g1j_inf(void) { g1j r = {fp_one(), fp_one(), fp_zero()}; return r; }
static int g1j_is_inf(g1j p) { return fp_is_zero(p.z); }
static g1j g1j_dbl(g1j p) {
    if (g1j_is_inf(p)) return p;
    fp A = fp_sqr(p.x), B = fp_sqr(p.y), C = fp_sqr(B);
    fp t = fp_add(p.x, B);
    fp D = fp_dbl(fp_sub(fp_sub(fp_sqr(t), A), C));
    fp E = fp_add(fp_dbl(A), A), F = fp_sqr(E);
    g1j r;
    r.x = fp_sub(F, fp_dbl(D));
    r.y = fp_sub(fp_mul(E, fp_sub(D, r.x)), fp_dbl(fp_dbl(fp_dbl(C))));
    r.z = fp_dbl(fp_mul(p.y, p.z));
    return r;
}
static g1j g1j_add(g1j p, g1j q) {
    if (g1j_is_inf(q)) return p;
    if (g1j_is_inf(p)) return q;
    fp z1z1 = fp_sqr(p.z), z2z2 = fp_sqr(q.z);
    fp u1 = fp_mul(p.x, z2z2), u2 = fp_mul(q.x, z1z1);
    fp s1 = fp_mul(fp_mul(p.y, q.z), z2z2), s2 = fp_mul(fp_mul(q.y, p.z), z1z1);
    fp h = fp_sub(u2, u1), rr = fp_sub(s2, s1);
    if (fp_is_zero(h)) { if (fp_is_zero(rr)) return g1j_dbl(p); return g1j_inf(); }
    rr = fp_dbl(rr);
    fp i = fp_sqr(fp_dbl(h)), j = fp_mul(h, i), v = fp_mul(u1, i);
    g1j r;
    r.x = fp_sub(fp_sub(fp_sqr(rr), j), fp_dbl(v));
    r.y = fp_sub(fp_mul(rr, fp_sub(v, r.x)), fp_dbl(fp_mul(s1, j)));
    r.z = fp_mul(fp_sub(fp_sub(fp_sqr(fp_add(p.z, q.z)), z1z1), z2z2), h);
    return r;
}
static g1j g1j_from_aff(g1a p) { if (p.inf) return g1j_inf(); g1j r = {p.x, p.y, fp_one()}; return r; }
static g1a g1j_to_aff(g1j p) {
    g1a r; memset(&r, 0, sizeof r);
    if (g1j_is_inf(p)) { r.inf = 1; return r; }
    fp zi = fp_inv(p.z), zi2 = fp_sqr(zi);
    r.x = fp_mul(p.x, zi2); r.y = fp_mul(fp_mul(p.y, zi2), zi); r.inf = 0;
    return r;
}
static g1j g1j_neg(g1j p) { p.y = fp_neg(p.y); return p; }
/* ark-ec `Projective::mul_bigint`: MSB-first double-and-add over the canonical scalar */
static g1j g1_mul(g1j p, fr k) {
    u64 w[NR]; fr_to_raw(k, w);
    g1j r = g1j_inf();
    int started = 0;
    for (int i = 255; i >= 0; i--) {
        if (started) r = g1j_dbl(r);
        if ((w[i >> 6] >> (i & 63)) & 1) { r = g1j_add(r, p); started = 1; }
    }
    return r;
}
static g1a g1a_from_le(const uint8_t* b) {
    g1a r; int z = 1;
    for (int i = 0; i < 2 * FPB; i++) if (b[i]) z = 0;
    r.inf = z;
    r.x = fp_from_le(b); r.y = fp_from_le(b + FPB);
    return r;
}
static void g1a_to_le(g1a p, uint8_t* b) {
    if (p.inf) { memset(b, 0, 2 * FPB); return; }
    fp_to_le(p.x, b); fp_to_le(p.y, b + FPB);
}
static int fp_gt_half(fp a) {             /* canonical a > (p-1)/2 */
    u64 w[NP], h[NP]; fp_to_raw(a, w);
    u64 c = 0;                             /* h = (p-1)/2 */
    memcpy(h, P, sizeof h); h[0] -= 1;
    for (int i = NP - 1; i >= 0; i--) { u64 n = h[i] & 1; h[i] = (h[i] >> 1) | (c << 63); c = n; }
    for (int i = NP - 1; i >= 0; i--) { if (w[i] > h[i]) return 1; if (w[i] < h[i]) return 0; }
    return 0;
}
#ifdef ORC_BN254
static void g1_compress(g1a p, uint8_t* out) {      /* ark-serialize: 32 B little-endian x, flags in the LAST byte */
    if (p.inf) { memset(out, 0, FPB); out[FPB - 1] = 0x40; return; }
    fp_to_le(p.x, out);
    if (fp_gt_half(p.y)) out[FPB - 1] |= 0x80;
}
#else
static void g1_compress(g1a p, uint8_t* out) {      /* 48 B big-endian, flags in the first byte */
    if (p.inf) { memset(out, 0, 48); out[0] = 0xC0; return; }
    uint8_t le[48]; fp_to_le(p.x, le);
    for (int i = 0; i < 48; i++) out[i] = le[47 - i];
    out[0] |= 0x80;
    if (fp_gt_half(p.y)) out[0] |= 0x20;
}
#endif

/* ----------------------------------------------------------------------------------- G2 */
typedef struct { fp2 x, y; int inf; } g2a;
static g2a G2_GEN;
static g2a g2_add(g2a t, g2a q) {
    if (t.inf) return q;
    if (q.inf) return t;
    fp2 lam;
    if (f2_eq(t.x, q.x)) {
        if (!f2_eq(t.y, q.y) || f2_is_zero(t.y)) { g2a r; memset(&r, 0, sizeof r); r.inf = 1; return r; }
        fp2 x2 = f2_sqr(t.x);
        lam = f2_mul(f2_add(f2_add(x2, x2), x2), f2_inv(f2_add(t.y, t.y)));
    } else {
        lam = f2_mul(f2_sub(q.y, t.y), f2_inv(f2_sub(q.x, t.x)));
    }
    g2a r; r.inf = 0;
    r.x = f2_sub(f2_sub(f2_sqr(lam), t.x), q.x);
    r.y = f2_sub(f2_mul(lam, f2_sub(t.x, r.x)), t.y);
    return r;
}
static g2a g2_neg(g2a q) { q.y = f2_neg(q.y); return q; }
static g2a g2_mul(g2a q, fr k) {
    u64 w[NR]; fr_to_raw(k, w);
    g2a r; memset(&r, 0, sizeof r); r.inf = 1;
    for (int i = 255; i >= 0; i--) { r = g2_add(r, r); if ((w[i >> 6] >> (i & 63)) & 1) r = g2_add(r, q); }
    return r;
}
static g2a g2a_from_le(const uint8_t* b, int inf) {
    g2a r; r.inf = inf;
    r.x.c0 = fp_from_le(b); r.x.c1 = fp_from_le(b + FPB); r.y.c0 = fp_from_le(b + 2 * FPB); r.y.c1 = fp_from_le(b + 3 * FPB);
    return r;
}
#ifdef ORC_BN254
static void g2_compress(g2a q, uint8_t* out) {      /* x.c0 || x.c1 little-endian, flags in the last byte */
    if (q.inf) { memset(out, 0, 2 * FPB); out[2 * FPB - 1] = 0x40; return; }
    fp_to_le(q.x.c0, out); fp_to_le(q.x.c1, out + FPB);
    int big = fp_is_zero(q.y.c1) ? fp_gt_half(q.y.c0) : fp_gt_half(q.y.c1);
    if (big) out[2 * FPB - 1] |= 0x80;
}
#else
static void g2_compress(g2a q, uint8_t* out) {      /* x.c1 || x.c0 big-endian, flags first byte */
    if (q.inf) { memset(out, 0, 96); out[0] = 0xC0; return; }
    uint8_t le[48];
    fp_to_le(q.x.c1, le); for (int i = 0; i < 48; i++) out[i] = le[47 - i];
    fp_to_le(q.x.c0, le); for (int i = 0; i < 48; i++) out[48 + i] = le[47 - i];
    out[0] |= 0x80;
    int big = fp_is_zero(q.y.c1) ? fp_gt_half(q.y.c0) : fp_gt_half(q.y.c1);
    if (big) out[0] |= 0x20;
}
#endif

/* ------------------------------------------------------------------------------- pairing */
#ifdef ORC_BN254
#define X_ABS 0x44e992b44a6909f1ULL            /* x = 4965661367192848881 > 0 */
#else
#define X_ABS 0xd201000000010000ULL
#endif
/* Miller loop with homogeneous projective G2 steps (no inversions).  Lines are scaled by factors in
 * Fp2 (killed by the final exponentiation); M-type twist, line * w^3 = l0 + l1 v + l4 v w:
 *   tangent at T = (X, Y, Z):  l0 = 3 X^3 - 2 Y^2 Z,  l1 = -3 X^2 Z xP,  l4 = 2 Y Z^2 yP
 *   chord through T and affine Q = (x2, y2) with u = y2 Z - Y, v = x2 Z - X:
 *                              l0 = u x2 - v y2,       l1 = -u xP,        l4 = v yP          */
typedef struct { fp2 x, y, z; } g2p;
static fp12 line_val(fp2 l0, fp2 l1, fp2 l4) {
    fp12 l; memset(&l, 0, sizeof l);
#ifdef ORC_BN254
    /* D-type twist: the same three values sit at  yP-term w^0,  xP-term w^1,  constant term w^3
     * (l = yP + (-lambda xP) w + (lambda xT - yT) w^3, scaled by the same Fp2 factor as below) */
    l.c0.c0 = l4; l.c1.c0 = l1; l.c1.c1 = l0;
#else
    l.c0.c0 = l0; l.c0.c1 = l1; l.c1.c1 = l4;
#endif
    return l;
}
static void dbl_step(fp12* f, g2p* T, g1a Pt) {
    fp2 X = T->x, Y = T->y, Z = T->z;
    fp2 XX = f2_sqr(X), YY = f2_sqr(Y), YZ = f2_mul(Y, Z);
    fp2 W = f2_add(f2_add(XX, XX), XX);                       /* 3 X^2 */
    fp2 l0 = f2_sub(f2_mul(W, X), f2_add(f2_mul(YY, Z), f2_mul(YY, Z)));
    fp2 l1 = f2_neg(f2_mul_fp(f2_mul(W, Z), Pt.x));
    fp2 l4 = f2_mul_fp(f2_add(f2_mul(YZ, Z), f2_mul(YZ, Z)), Pt.y);
    *f = f12_mul(*f, line_val(l0, l1, l4));
    /* dbl-2007-bl (a = 0), homogeneous projective */
    fp2 S = YZ, B = f2_mul(f2_mul(X, Y), S);
    fp2 B4 = f2_add(f2_add(B, B), f2_add(B, B)), B8 = f2_add(B4, B4);
    fp2 Hh = f2_sub(f2_sqr(W), B8);
    fp2 SS = f2_sqr(S);
    fp2 YYSS8 = f2_mul(YY, SS);
    YYSS8 = f2_add(YYSS8, YYSS8); YYSS8 = f2_add(YYSS8, YYSS8); YYSS8 = f2_add(YYSS8, YYSS8);
    g2p r;
    r.x = f2_mul(f2_add(Hh, Hh), S);
    r.y = f2_sub(f2_mul(W, f2_sub(B4, Hh)), YYSS8);
    fp2 S3 = f2_mul(SS, S);
    S3 = f2_add(S3, S3); S3 = f2_add(S3, S3); S3 = f2_add(S3, S3);
    r.z = S3;
    *T = r;
}
static void add_step(fp12* f, g2p* T, g2a Q, g1a Pt) {
    fp2 u = f2_sub(f2_mul(Q.y, T->z), T->y), v = f2_sub(f2_mul(Q.x, T->z), T->x);
    fp2 l0 = f2_sub(f2_mul(u, Q.x), f2_mul(v, Q.y));
    fp2 l1 = f2_neg(f2_mul_fp(u, Pt.x));
    fp2 l4 = f2_mul_fp(v, Pt.y);
    *f = f12_mul(*f, line_val(l0, l1, l4));
    /* madd-1998-cmo */
    fp2 uu = f2_sqr(u), vv = f2_sqr(v), vvv = f2_mul(v, vv), Rr = f2_mul(vv, T->x);
    fp2 A = f2_sub(f2_sub(f2_mul(uu, T->z), vvv), f2_add(Rr, Rr));
    g2p r;
    r.x = f2_mul(v, A);
    r.y = f2_sub(f2_mul(u, f2_sub(Rr, A)), f2_mul(vvv, T->y));
    r.z = f2_mul(vvv, T->z);
    *T = r;
}
#ifdef ORC_BN254
static fp2 G2FROB_X, G2FROB_Y;                 /* xi^((p-1)/3), xi^((p-1)/2): p-power Frobenius on the D-twist */
static g2a g2_frob(g2a q) { g2a r = q; r.x = f2_mul(f2_conj(q.x), G2FROB_X); r.y = f2_mul(f2_conj(q.y), G2FROB_Y); return r; }
static fp12 miller_loop(g1a Pt, g2a Q) {
    fp12 f = f12_one();
    if (Pt.inf || Q.inf) return f;            /* ark-ec skips identity pairs */
    g2p T = {Q.x, Q.y, f2_one()};
    /* optimal ate: loop over 6x + 2 = 0x1 9d797039be763ba8 (65 bits), then the lines through pi(Q) and -pi^2(Q) */
    const u64 lo = 0x9d797039be763ba8ULL;
    for (int i = 63; i >= 0; i--) {
        f = f12_sqr(f);
        dbl_step(&f, &T, Pt);
        if ((lo >> i) & 1) add_step(&f, &T, Q, Pt);
    }
    g2a Q1 = g2_frob(Q), Q2 = g2_neg(g2_frob(Q1));
    add_step(&f, &T, Q1, Pt);
    add_step(&f, &T, Q2, Pt);
    return f;
}
#else
static fp12 miller_loop(g1a Pt, g2a Q) {
    fp12 f = f12_one();
    if (Pt.inf || Q.inf) return f;            /* ark-ec skips identity pairs */
    g2p T = {Q.x, Q.y, f2_one()};
    for (int i = 62; i >= 0; i--) {
        f = f12_sqr(f);
        dbl_step(&f, &T, Pt);
        if ((X_ABS >> i) & 1) add_step(&f, &T, Q, Pt);
    }
    return f12_conj(f);                        /* x < 0 */
}
#endif
static fp12 f12_pow_big(fp12 a, const u64* e, int n) {
    fp12 r = f12_one();
    int started = 0;
    for (int i = n - 1; i >= 0; i--) for (int b = 63; b >= 0; b--) {
        if (started) r = f12_sqr(r);
        if ((e[i] >> b) & 1) { r = started ? f12_mul(r, a) : a; started = 1; }
    }
    return r;
}
static void init_hard_exp(void) {}
#ifdef ORC_BN254
static fp12 f12_pow_x(fp12 a) {                /* a^x, x > 0 */
    fp12 r = a;
    for (int i = 61; i >= 0; i--) { r = f12_sqr(r); if ((X_ABS >> i) & 1) r = f12_mul(r, a); }
    return r;
}
/* f^((p^12-1)/r): easy part, then the hard part (p^4-p^2+1)/r by the Devegili-Scott-Dahab chain
 * y0 y1^2 y2^6 y3^12 y4^18 y5^30 y6^36 (f unitary after the easy part: inverse = conjugate) */
static fp12 final_exp(fp12 f) {
    fp12 t = f12_mul(f12_conj(f), f12_inv(f));                    /* ^(p^6 - 1) */
    t = f12_mul(f12_frob(f12_frob(t)), t);                        /* ^(p^2 + 1) */
    fp12 fu = f12_pow_x(t), fu2 = f12_pow_x(fu), fu3 = f12_pow_x(fu2);
    fp12 tp = f12_frob(t), tp2 = f12_frob(tp), tp3 = f12_frob(tp2);
    fp12 y0 = f12_mul(f12_mul(tp, tp2), tp3);
    fp12 y1 = f12_conj(t);
    fp12 y2 = f12_frob(f12_frob(fu2));
    fp12 y3 = f12_conj(f12_frob(fu));
    fp12 y4 = f12_conj(f12_mul(fu, f12_frob(fu2)));
    fp12 y5 = f12_conj(fu2);
    fp12 y6 = f12_conj(f12_mul(fu3, f12_frob(fu3)));
    fp12 t0 = f12_sqr(y6);
    t0 = f12_mul(t0, y4); t0 = f12_mul(t0, y5);
    fp12 t1 = f12_mul(y3, y5);
    t1 = f12_mul(t1, t0); t0 = f12_mul(t0, y2);
    t1 = f12_sqr(t1); t1 = f12_mul(t1, t0); t1 = f12_sqr(t1);
    t0 = f12_mul(t1, y1); t1 = f12_mul(t1, y0);
    t0 = f12_sqr(t0);
    return f12_mul(t0, t1);
}
#else
static fp12 f12_pow_x(fp12 a) {                /* a^x, x = -X_ABS, a unitary */
    fp12 r = a;
    for (int i = 62; i >= 0; i--) { r = f12_sqr(r); if ((X_ABS >> i) & 1) r = f12_mul(r, a); }
    return f12_conj(r);
}
/* f^(3 (p^12-1)/r) with 3 (p^4-p^2+1)/r = (x-1)^2 (x+p) (x^2+p^2-1) + 3  (same boolean as (p^12-1)/r) */
static fp12 final_exp(fp12 f) {
    fp12 t = f12_mul(f12_conj(f), f12_inv(f));                    /* ^(p^6 - 1) */
    t = f12_mul(f12_frob(f12_frob(t)), t);                        /* ^(p^2 + 1) */
    fp12 a = f12_mul(f12_pow_x(t), f12_conj(t));
    a = f12_mul(f12_pow_x(a), f12_conj(a));
    fp12 b = f12_mul(f12_pow_x(a), f12_frob(a));
    fp12 c = f12_pow_x(f12_pow_x(b));
    c = f12_mul(c, f12_frob(f12_frob(b)));
    c = f12_mul(c, f12_conj(b));
    return f12_mul(c, f12_mul(f12_sqr(t), t));
}
#endif
static fp12 pairing(g1a Pt, g2a Q) { return final_exp(miller_loop(Pt, Q)); }

/* ------------------------------------------------------------------------------- SHA-256 */
typedef struct { uint32_t h[8]; uint8_t buf[64]; uint64_t len; } sha;
static const uint32_t K256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
    0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
    0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
    0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
    0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
    0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
    0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
#define ROR(x, n) (((x) >> (n)) | ((x) << (32 - (n))))
static void sha_block(sha* s, const uint8_t* b) {
    uint32_t w[64];
    for (int i = 0; i < 16; i++) w[i] = ((uint32_t)b[4 * i] << 24) | ((uint32_t)b[4 * i + 1] << 16) | ((uint32_t)b[4 * i + 2] << 8) | b[4 * i + 3];
    for (int i = 16; i < 64; i++) {
        uint32_t s0 = ROR(w[i - 15], 7) ^ ROR(w[i - 15], 18) ^ (w[i - 15] >> 3), s1 = ROR(w[i - 2], 17) ^ ROR(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = s->h[0], bb = s->h[1], c = s->h[2], d = s->h[3], e = s->h[4], f = s->h[5], g = s->h[6], h = s->h[7];
    for (int i = 0; i < 64; i++) {
        uint32_t t1 = h + (ROR(e, 6) ^ ROR(e, 11) ^ ROR(e, 25)) + ((e & f) ^ (~e & g)) + K256[i] + w[i];
        uint32_t t2 = (ROR(a, 2) ^ ROR(a, 13) ^ ROR(a, 22)) + ((a & bb) ^ (a & c) ^ (bb & c));
        h = g; g = f; f = e; e = d + t1; d = c; c = bb; bb = a; a = t1 + t2;
    }
    s->h[0] += a; s->h[1] += bb; s->h[2] += c; s->h[3] += d; s->h[4] += e; s->h[5] += f; s->h[6] += g; s->h[7] += h;
}
static void sha_init(sha* s) {
    static const uint32_t iv[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    memcpy(s->h, iv, sizeof iv); s->len = 0;
}
static void sha_upd(sha* s, const uint8_t* p, size_t n) {
    for (size_t i = 0; i < n; i++) { s->buf[s->len & 63] = p[i]; s->len++; if ((s->len & 63) == 0) sha_block(s, s->buf); }
}
static void sha_fin(sha* s, uint8_t* out) {
    uint64_t bits = s->len * 8; uint8_t b = 0x80;
    sha_upd(s, &b, 1); b = 0;
    while ((s->len & 63) != 56) sha_upd(s, &b, 1);
    uint8_t l[8]; for (int i = 0; i < 8; i++) l[i] = (uint8_t)(bits >> (56 - 8 * i));
    sha_upd(s, l, 8);
    for (int i = 0; i < 8; i++) { out[4 * i] = (uint8_t)(s->h[i] >> 24); out[4 * i + 1] = (uint8_t)(s->h[i] >> 16); out[4 * i + 2] = (uint8_t)(s->h[i] >> 8); out[4 * i + 3] = (uint8_t)s->h[i]; }
}
/* utilities_helper.rs:42-97 with len_in_bytes = 48 */
static int expand48(const uint8_t* msg, size_t mlen, const uint8_t* dst, size_t dlen, uint8_t* out48) {
    if (dlen > 255) return -1;
    uint8_t z[64] = {0}, lib[3] = {0, 48, 0}, dl = (uint8_t)dlen, b0[32], b1[32], b2[32], t[32], c;
    sha s; sha_init(&s);
    sha_upd(&s, z, 64); sha_upd(&s, msg, mlen); sha_upd(&s, lib, 3); sha_upd(&s, dst, dlen); sha_upd(&s, &dl, 1); sha_fin(&s, b0);
    sha_init(&s); c = 1; sha_upd(&s, b0, 32); sha_upd(&s, &c, 1); sha_upd(&s, dst, dlen); sha_upd(&s, &dl, 1); sha_fin(&s, b1);
    for (int i = 0; i < 32; i++) t[i] = b0[i] ^ b1[i];
    sha_init(&s); c = 2; sha_upd(&s, t, 32); sha_upd(&s, &c, 1); sha_upd(&s, dst, dlen); sha_upd(&s, &dl, 1); sha_fin(&s, b2);
    memcpy(out48, b1, 32); memcpy(out48 + 32, b2, 16);
    return 0;
}
/* core_utilities.rs:11-21 + FromOkm: 48 bytes big-endian mod r */
static int hash_to_scalar(const uint8_t* msg, size_t mlen, const uint8_t* dst, size_t dlen, fr* out) {
    uint8_t okm[48];
    if (expand48(msg, mlen, dst, dlen, okm)) return -1;
    u64 lo[4], hi[4] = {0};
    for (int i = 0; i < 4; i++) { u64 w = 0; for (int b = 0; b < 8; b++) w = (w << 8) | okm[16 + 8 * (3 - i) + b]; lo[i] = w; }
    for (int i = 0; i < 2; i++) { u64 w = 0; for (int b = 0; b < 8; b++) w = (w << 8) | okm[8 * (1 - i) + b]; hi[i] = w; }
    fr l, h, r2, r3; memcpy(l.v, lo, 32); memcpy(h.v, hi, 32); memcpy(r2.v, R_R2, 32); memcpy(r3.v, R_R3, 32);
    *out = fr_add(fr_mul(l, r2), fr_mul(h, r3));          /* lo*R + hi*2^256*R */
    return 0;
}

/* --------------------------------------------------------------------------------- BBS */
typedef struct { uint8_t* p; size_t n, cap; } buf;
static void bput(buf* b, const void* s, size_t n) {
    if (b->n + n > b->cap) { b->cap = (b->n + n) * 2 + 64; b->p = (uint8_t*)realloc(b->p, b->cap); }
    memcpy(b->p + b->n, s, n); b->n += n;
}
static void bput_u64be(buf* b, u64 v) { uint8_t t[8]; for (int i = 0; i < 8; i++) t[i] = (uint8_t)(v >> (56 - 8 * i)); bput(b, t, 8); }
static void bput_fr(buf* b, fr x) { uint8_t t[32]; fr_to_be(x, t); bput(b, t, 32); }
static void bput_g1(buf* b, g1j p) { uint8_t t[FPB]; g1_compress(g1j_to_aff(p), t); bput(b, t, FPB); }

static g1j P1_PT;
#ifdef ORC_BN254
static void div_small(u64* e, u64 d) { u128 rem = 0; for (int i = NP - 1; i >= 0; i--) { u128 cur = (rem << 64) | e[i]; e[i] = (u64)(cur / d); rem = cur % d; } }
static void init_all(void) {
    static int done = 0;
    if (done) return;
    init_consts(); init_frob(); init_hard_exp();
    u64 three[NP] = {3}; FP_B = fp_from_raw(three);
    static const u64 g2x0[NP] = {0x46debd5cd992f6edULL, 0x674322d4f75edaddULL, 0x426a00665e5c4479ULL, 0x1800deef121f1e76ULL};
    static const u64 g2x1[NP] = {0x97e485b7aef312c2ULL, 0xf1aa493335a9e712ULL, 0x7260bfb731fb5d25ULL, 0x198e9393920d483aULL};
    static const u64 g2y0[NP] = {0x4ce6cc0166fa7daaULL, 0xe3d1e7690c43d37bULL, 0x4aab71808dcb408fULL, 0x12c85ea5db8c6debULL};
    static const u64 g2y1[NP] = {0x55acdadcd122975bULL, 0xbc4b313370b38ef3ULL, 0xec9e99ad690c3395ULL, 0x090689d0585ff075ULL};
    G2_GEN.inf = 0; G2_GEN.x.c0 = fp_from_raw(g2x0); G2_GEN.x.c1 = fp_from_raw(g2x1); G2_GEN.y.c0 = fp_from_raw(g2y0); G2_GEN.y.c1 = fp_from_raw(g2y1);
    /* P1 (src/constants.rs:39-51), the affine coordinates the reference hard-codes */
    static const u64 p1x[NP] = {0x098204f045e61adeULL, 0x0db2146c9bde3376ULL, 0x7ce8f44e877c5f6cULL, 0x111c0a273f09aa94ULL};
    static const u64 p1y[NP] = {0xb420b35cc0ef2fa7ULL, 0x6fb06e2ad7890eaaULL, 0xdec23067d05d165dULL, 0x124050fe34102928ULL};
    g1a p1; p1.inf = 0; p1.x = fp_from_raw(p1x); p1.y = fp_from_raw(p1y);
    P1_PT = g1j_from_aff(p1);
    fp2 xi = f2_mul_xi(f2_one());
    u64 e[NP]; memcpy(e, P, sizeof e); e[0] -= 1; div_small(e, 3);
    G2FROB_X = f2_pow_u64s(xi, e, NP);
    memcpy(e, P, sizeof e); e[0] -= 1; div_small(e, 2);
    G2FROB_Y = f2_pow_u64s(xi, e, NP);
    done = 1;
}
#else
static void init_all(void) {
    static int done = 0;
    if (done) return;
    init_consts(); init_frob(); init_hard_exp();
    u64 four[NP] = {4, 0, 0, 0, 0, 0}; FP_B = fp_from_raw(four);
    static const u64 g2x0[NP] = {0xd48056c8c121bdb8ULL, 0x0bac0326a805bbefULL, 0xb4510b647ae3d177ULL, 0xc6e47ad4fa403b02ULL, 0x260805272dc51051ULL, 0x024aa2b2f08f0a91ULL};
    static const u64 g2x1[NP] = {0xe5ac7d055d042b7eULL, 0x334cf11213945d57ULL, 0xb5da61bbdc7f5049ULL, 0x596bd0d09920b61aULL, 0x7dacd3a088274f65ULL, 0x13e02b6052719f60ULL};
    static const u64 g2y0[NP] = {0xe193548608b82801ULL, 0x923ac9cc3baca289ULL, 0x6d429a695160d12cULL, 0xadfd9baa8cbdd3a7ULL, 0x8cc9cdc6da2e351aULL, 0x0ce5d527727d6e11ULL};
    static const u64 g2y1[NP] = {0xaaa9075ff05f79beULL, 0x3f370d275cec1da1ULL, 0x267492ab572e99abULL, 0xcb3e287e85a763afULL, 0x32acd2b02bc28b99ULL, 0x0606c4a02ea734ccULL};
    G2_GEN.inf = 0; G2_GEN.x.c0 = fp_from_raw(g2x0); G2_GEN.x.c1 = fp_from_raw(g2x1); G2_GEN.y.c0 = fp_from_raw(g2y0); G2_GEN.y.c1 = fp_from_raw(g2y1);
    /* P1 (src/constants.rs:73-79), decimal constants as limbs */
    static const u64 p1x[NP] = {0x11406d161b4e28c9ULL, 0x5e7c59698588e70dULL, 0x66c872b948f1fd22ULL, 0xb205762f9776b3a7ULL, 0xa3e94ea9025e4662ULL, 0x08ce256102840821ULL};
    g1a p1; p1.inf = 0; p1.x = fp_from_raw(p1x);
    /* y from the curve equation with the sign bit of the compressed vector a8ce.. (0x20 set: largest) */
    fp y2 = fp_add(fp_mul(fp_sqr(p1.x), p1.x), FP_B);
    u64 e[NP]; memcpy(e, P, sizeof e); e[0] += 1;            /* (p+1)/4 */
    u64 c = 0; for (int i = NP - 1; i >= 0; i--) { u64 n = e[i] & 3; e[i] = (e[i] >> 2) | (c << 62); c = n; }
    fp y = fp_pow(y2, e, NP);
    if (!fp_gt_half(y)) y = fp_neg(y);
    p1.y = y;
    P1_PT = g1j_from_aff(p1);
    done = 1;
}
#endif

typedef struct { int L; g1j* gens; const uint8_t* api_id; size_t alen; } gen_ctx;

/* calculate_domain, core_utilities.rs:24-63 */
static int calc_domain(g2a pk, const gen_ctx* g, const uint8_t* hdr, size_t hlen, fr* out) {
    buf b = {0};
    uint8_t t[2 * FPB];
    g2_compress(pk, t); bput(&b, t, 2 * FPB);
    bput_u64be(&b, (u64)g->L);
    for (int i = 0; i <= g->L; i++) bput_g1(&b, g->gens[i]);
    bput(&b, g->api_id, g->alen);
    bput_u64be(&b, (u64)hlen); bput(&b, hdr, hlen);
    uint8_t dst[300]; memcpy(dst, g->api_id, g->alen); memcpy(dst + g->alen, "H2S_", 4);
    int rc = hash_to_scalar(b.p, b.n, dst, g->alen + 4, out);
    free(b.p);
    return rc;
}
/* proof_challenge_calculate, proof_gen.rs:272-328 */
static int calc_challenge(const g1j pts[5], fr domain, const fr* dmsgs, const uint64_t* didx, size_t R, const uint8_t* ph, size_t plen,
                          const gen_ctx* g, fr* out) {
    buf b = {0};
    bput_u64be(&b, (u64)R);
    for (size_t k = 0; k < R; k++) { bput_u64be(&b, didx[k]); bput_fr(&b, dmsgs[k]); }
    for (int k = 0; k < 5; k++) bput_g1(&b, pts[k]);
    bput_fr(&b, domain);
    bput_u64be(&b, (u64)plen); bput(&b, ph, plen);
    uint8_t dst[300]; memcpy(dst, g->api_id, g->alen); memcpy(dst + g->alen, "H2S_", 4);
    int rc = hash_to_scalar(b.p, b.n, dst, g->alen + 4, out);
    free(b.p);
    return rc;
}
static g1j compute_b(const gen_ctx* g, fr domain, const fr* msgs) {   /* sign.rs:120-126 */
    g1j b = g1j_add(P1_PT, g1_mul(g->gens[0], domain));
    for (int i = 1; i <= g->L; i++) b = g1j_add(b, g1_mul(g->gens[i], msgs[i - 1]));
    return b;
}
static gen_ctx make_ctx(int L, const uint8_t* gens_le, const uint8_t* api_id, size_t alen) {
    gen_ctx g; g.L = L; g.api_id = api_id; g.alen = alen;
    g.gens = (g1j*)malloc(sizeof(g1j) * (size_t)(L + 1));
    for (int i = 0; i <= L; i++) g.gens[i] = g1j_from_aff(g1a_from_le(gens_le + 2 * FPB * (size_t)i));
    return g;
}

/* ---- exported entry points (ctypes) ---------------------------------------------------------- */
int orc_fp_bytes(void) { return FPB; }
void orc_sk_to_pk(const uint8_t* sk32, uint8_t* pk192) {
    init_all();
    g2a pk = g2_mul(G2_GEN, fr_from_le(sk32));
    fp_to_le(pk.x.c0, pk192); fp_to_le(pk.x.c1, pk192 + FPB); fp_to_le(pk.y.c0, pk192 + 2 * FPB); fp_to_le(pk.y.c1, pk192 + 3 * FPB);
}

/* core_sign, sign.rs:63-133.  out: A (96 LE) || e (32 LE).  returns 1, or -20 for the unwrap panic */
int orc_core_sign(const uint8_t* sk32, int L, const uint8_t* gens_le, const uint8_t* api_id, size_t alen,
                  const uint8_t* hdr, size_t hlen, const uint8_t* msgs_le, uint8_t* out128) {
    init_all();
    gen_ctx g = make_ctx(L, gens_le, api_id, alen);
    fr sk = fr_from_le(sk32);
    g2a pk = g2_mul(G2_GEN, sk);                                     /* sign.rs:81 */
    fr* m = (fr*)malloc(sizeof(fr) * (size_t)(L + 1));
    for (int i = 0; i < L; i++) m[i] = fr_from_le(msgs_le + 32 * (size_t)i);
    fr domain; calc_domain(pk, &g, hdr, hlen, &domain);
    buf b = {0};
    bput_fr(&b, sk); for (int i = 0; i < L; i++) bput_fr(&b, m[i]); bput_fr(&b, domain);
    uint8_t dst[300]; memcpy(dst, api_id, alen); memcpy(dst + alen, "H2S_", 4);
    fr e; hash_to_scalar(b.p, b.n, dst, alen + 4, &e);
    free(b.p);
    g1j B = compute_b(&g, domain, m);
    fr spe = fr_add(sk, e);
    int rc = 1;
    if (fr_is_zero(spe)) rc = -20;
    else { g1a_to_le(g1j_to_aff(g1_mul(B, fr_inv(spe))), out128); fr_to_le(e, out128 + 2 * FPB); }
    free(m); free(g.gens);
    return rc;
}

/* core_verify, verify.rs:53-93 */
int orc_core_verify(const uint8_t* pk192, int pk_inf, int L, const uint8_t* gens_le, const uint8_t* api_id, size_t alen,
                    const uint8_t* hdr, size_t hlen, const uint8_t* msgs_le, const uint8_t* sig128) {
    init_all();
    gen_ctx g = make_ctx(L, gens_le, api_id, alen);
    g2a pk = g2a_from_le(pk192, pk_inf);
    fr* m = (fr*)malloc(sizeof(fr) * (size_t)(L + 1));
    for (int i = 0; i < L; i++) m[i] = fr_from_le(msgs_le + 32 * (size_t)i);
    fr domain; calc_domain(pk, &g, hdr, hlen, &domain);
    g1j B = compute_b(&g, domain, m);
    g1a A = g1a_from_le(sig128);
    fr e = fr_from_le(sig128 + 2 * FPB);
    g2a q = g2_add(pk, g2_mul(G2_GEN, e));
    fp12 gt = f12_mul(pairing(A, q), pairing(g1j_to_aff(B), g2_neg(G2_GEN)));
    free(m); free(g.gens);
    return f12_is_one(gt);
}

/* core_proof_gen, proof_gen.rs:116-365.  disclosed must already be validated/sorted/deduped by the
 * caller (the Python oracle covers the error paths); rnd = 5 + U scalars.  proof_fixed: 3 points + 4 scalars */
int orc_core_proof_gen(const uint8_t* pk192, int pk_inf, int L, const uint8_t* gens_le, const uint8_t* api_id, size_t alen,
                       const uint8_t* hdr, size_t hlen, const uint8_t* ph, size_t plen, const uint8_t* msgs_le,
                       const uint8_t* sig128, const uint64_t* disclosed, size_t R, const uint8_t* rnd_le,
                       uint8_t* proof_fixed, uint8_t* commitments_le) {
    init_all();
    gen_ctx g = make_ctx(L, gens_le, api_id, alen);
    g2a pk = g2a_from_le(pk192, pk_inf);
    size_t U = (size_t)L - R;
    fr* m = (fr*)malloc(sizeof(fr) * (size_t)(L + 1));
    fr* rs = (fr*)malloc(sizeof(fr) * (5 + U));
    for (int i = 0; i < L; i++) m[i] = fr_from_le(msgs_le + 32 * (size_t)i);
    for (size_t i = 0; i < 5 + U; i++) rs[i] = fr_from_le(rnd_le + 32 * i);
    uint8_t* isd = (uint8_t*)calloc((size_t)L + 1, 1);
    for (size_t k = 0; k < R; k++) isd[disclosed[k]] = 1;
    size_t* und = (size_t*)malloc(sizeof(size_t) * (U + 1)); size_t nu = 0;
    for (int j = 0; j < L; j++) if (!isd[j]) und[nu++] = (size_t)j;
    fr domain; calc_domain(pk, &g, hdr, hlen, &domain);
    g1j A = g1j_from_aff(g1a_from_le(sig128));
    fr e = fr_from_le(sig128 + 2 * FPB);
    g1j B = compute_b(&g, domain, m);                                  /* proof_init :249-263 */
    g1j D = g1_mul(B, rs[1]);
    g1j Abar = g1_mul(A, fr_mul(rs[0], rs[1]));
    g1j Bbar = g1j_add(g1_mul(D, rs[0]), g1j_neg(g1_mul(Abar, e)));
    g1j T1 = g1j_add(g1_mul(Abar, rs[2]), g1_mul(D, rs[3]));
    g1j T2 = g1_mul(D, rs[4]);
    for (size_t i = 0; i < U; i++) T2 = g1j_add(T2, g1_mul(g.gens[1 + und[i]], rs[5 + i]));
    g1j pts[5] = {Abar, Bbar, D, T1, T2};
    fr* dm = (fr*)malloc(sizeof(fr) * (R + 1));
    for (size_t k = 0; k < R; k++) dm[k] = m[disclosed[k]];
    fr c; calc_challenge(pts, domain, dm, disclosed, R, ph, plen, &g, &c);
    int rc = 1;
    if (fr_is_zero(rs[1])) rc = -21;
    else {
        fr r3 = fr_inv(rs[1]);                                         /* proof_finalize :346-353 */
        g1a_to_le(g1j_to_aff(Abar), proof_fixed); g1a_to_le(g1j_to_aff(Bbar), proof_fixed + 2 * FPB); g1a_to_le(g1j_to_aff(D), proof_fixed + 4 * FPB);
        fr_to_le(fr_add(rs[2], fr_mul(e, c)), proof_fixed + 6 * FPB);
        fr_to_le(fr_sub(rs[3], fr_mul(rs[0], c)), proof_fixed + 6 * FPB + 32);
        fr_to_le(fr_sub(rs[4], fr_mul(r3, c)), proof_fixed + 6 * FPB + 64);
        fr_to_le(c, proof_fixed + 6 * FPB + 96);
        for (size_t i = 0; i < U; i++) fr_to_le(fr_add(rs[5 + i], fr_mul(m[und[i]], c)), commitments_le + 32 * i);
    }
    free(m); free(rs); free(isd); free(und); free(dm); free(g.gens);
    return rc;
}

/* core_proof_verify, proof_verify.rs:64-188 (inputs already validated: distinct in-range indexes) */
int orc_core_proof_verify(const uint8_t* pk192, int pk_inf, int L, const uint8_t* gens_le, const uint8_t* api_id, size_t alen,
                          const uint8_t* hdr, size_t hlen, const uint8_t* ph, size_t plen, const uint8_t* proof_fixed,
                          const uint8_t* commitments_le, const uint8_t* dmsgs_le, const uint64_t* didx, size_t R) {
    init_all();
    gen_ctx g = make_ctx(L, gens_le, api_id, alen);
    g2a pk = g2a_from_le(pk192, pk_inf);
    size_t U = (size_t)L - R;
    g1j Abar = g1j_from_aff(g1a_from_le(proof_fixed)), Bbar = g1j_from_aff(g1a_from_le(proof_fixed + 2 * FPB)), D = g1j_from_aff(g1a_from_le(proof_fixed + 4 * FPB));
    fr e_cap = fr_from_le(proof_fixed + 6 * FPB), r1_cap = fr_from_le(proof_fixed + 6 * FPB + 32), r3_cap = fr_from_le(proof_fixed + 6 * FPB + 64), c = fr_from_le(proof_fixed + 6 * FPB + 96);
    fr* dm = (fr*)malloc(sizeof(fr) * (R + 1));
    for (size_t k = 0; k < R; k++) dm[k] = fr_from_le(dmsgs_le + 32 * k);
    uint8_t* isd = (uint8_t*)calloc((size_t)L + 1, 1);
    for (size_t k = 0; k < R; k++) isd[didx[k]] = 1;
    fr domain; calc_domain(pk, &g, hdr, hlen, &domain);
    g1j T1 = g1j_add(g1j_add(g1_mul(Bbar, c), g1_mul(Abar, e_cap)), g1_mul(D, r1_cap));      /* :163-164 */
    g1j Bv = g1j_add(P1_PT, g1_mul(g.gens[0], domain));
    for (size_t k = 0; k < R; k++) Bv = g1j_add(Bv, g1_mul(g.gens[1 + didx[k]], dm[k]));
    g1j T2 = g1j_add(g1_mul(Bv, c), g1_mul(D, r3_cap));
    size_t ci = 0;
    for (int j = 0; j < L; j++) if (!isd[j]) { T2 = g1j_add(T2, g1_mul(g.gens[1 + j], fr_from_le(commitments_le + 32 * ci))); ci++; }
    (void)U;
    g1j pts[5] = {Abar, Bbar, D, T1, T2};
    fr ch; calc_challenge(pts, domain, dm, didx, R, ph, plen, &g, &ch);
    int res;
    uint8_t a[32], b[32]; fr_to_le(ch, a); fr_to_le(c, b);
    if (memcmp(a, b, 32) != 0) res = 0;                                 /* :108-110 */
    else res = f12_is_one(f12_mul(pairing(g1j_to_aff(Abar), pk), pairing(g1j_to_aff(Bbar), g2_neg(G2_GEN))));
    free(dm); free(isd); free(g.gens);
    return res;
}

/* Plain sum  S = sum_i k_i * P_i  as the reference writes every such sum (sign.rs:120-126, proof_verify.rs:165-182):
   independent MSB-first double-and-add multiplications, added one by one.  The checker of the device's bucket-method
   (Pippenger) MSM at sizes the Python oracle is too slow for.  pts_le: n affine points (x || y little-endian, all-zero =
   identity), scal_le: n canonical 32-byte little-endian scalars, out: the affine sum (all-zero = identity). */
void orc_g1_msm_plain(size_t n, const uint8_t* pts_le, const uint8_t* scal_le, uint8_t* out) {
    init_all();
    g1j acc = g1j_inf();
    for (size_t i = 0; i < n; i++)
        acc = g1j_add(acc, g1_mul(g1j_from_aff(g1a_from_le(pts_le + 2 * FPB * i)), fr_from_le(scal_le + 32 * i)));
    g1a_to_le(g1j_to_aff(acc), out);
}

#ifdef ORC_BN254
/* ---- BN254 hash_to_curve (interface_utilities.rs:24-28 calls crate bn254_hash2curve 0.1.2, not vendored): RFC 9380
   hash_to_curve, expand_message_xmd(SHA-256), L = 48, Shallue-van de Woestijne map (section 6.6.1, straight-line form),
   Z = 1 on y^2 = x^3 + 3, cofactor 1.  A THIRD statement of the map (after oracle/hashing.py and the product's host_h2c.hpp),
   with its constants derived here from Z: c1 = g(Z), c2 = -Z / 2, c3 = sqrt(-g(Z) (3 Z^2)) with sgn0(c3) = 0,
   c4 = -4 g(Z) / (3 Z^2).  The reference's one BN254 known answer (P1, constants.rs:39-51) never has both candidates x1, x2
   on the curve at once, so it does not see the sign of c3; orc_bn_svdw_map reports which candidates were squares so that
   tests can pick inputs that do. */
static void xmd(const uint8_t* msg, size_t mlen, const uint8_t* dst, size_t dlen, uint8_t* out, size_t len) {
    uint8_t z[64] = {0}, lib[3], dl = (uint8_t)dlen, b0[32], bi[32], t[32], c;
    lib[0] = (uint8_t)(len >> 8); lib[1] = (uint8_t)len; lib[2] = 0;
    sha s; sha_init(&s);
    sha_upd(&s, z, 64); sha_upd(&s, msg, mlen); sha_upd(&s, lib, 3); sha_upd(&s, dst, dlen); sha_upd(&s, &dl, 1); sha_fin(&s, b0);
    memset(bi, 0, 32);
    size_t at = 0;
    for (c = 1; at < len; c++) {
        for (int i = 0; i < 32; i++) t[i] = b0[i] ^ bi[i];
        sha_init(&s); sha_upd(&s, t, 32); sha_upd(&s, &c, 1); sha_upd(&s, dst, dlen); sha_upd(&s, &dl, 1); sha_fin(&s, bi);
        size_t k = len - at < 32 ? len - at : 32;
        memcpy(out + at, bi, k); at += k;
    }
}
static fp fp_small(u64 v) { u64 w[NP] = {0}; w[0] = v; return fp_from_raw(w); }
static fp fp_sqrt_exp(fp a) {                       /* a^((p+1)/4): p = 3 mod 4 */
    u64 e[NP]; memcpy(e, P, sizeof e); e[0] += 1;
    u64 c = 0; for (int i = NP - 1; i >= 0; i--) { u64 n = e[i] & 3; e[i] = (e[i] >> 2) | (c << 62); c = n; }
    return fp_pow(a, e, NP);
}
static int fp_is_square(fp a, fp* root) { fp r = fp_sqrt_exp(a); if (root) *root = r; return fp_eq(fp_sqr(r), a); }
static int fp_sgn0(fp a) { u64 w[NP]; fp_to_raw(a, w); return (int)(w[0] & 1); }
static fp bn_g(fp x) { return fp_add(fp_mul(fp_sqr(x), x), fp_small(3)); }
/* 48 big-endian bytes mod p */
static fp fp_from_be48(const uint8_t* b) {
    fp acc = fp_zero(), k256 = fp_small(256);
    for (int i = 0; i < 48; i++) acc = fp_add(fp_mul(acc, k256), fp_small(b[i]));
    return acc;
}
static g1a bn_svdw(fp u, int* squares) {
    const fp Z = fp_one(), gz = bn_g(Z), one = fp_one();
    const fp z2x3 = fp_mul(fp_small(3), fp_sqr(Z));
    const fp c1 = gz, c2 = fp_neg(fp_mul(Z, fp_inv(fp_small(2))));
    fp c3; int ok = fp_is_square(fp_neg(fp_mul(gz, z2x3)), &c3); (void)ok;
    if (fp_sgn0(c3)) c3 = fp_neg(c3);
    const fp c4 = fp_neg(fp_mul(fp_mul(fp_small(4), gz), fp_inv(z2x3)));
    fp tv1 = fp_mul(fp_sqr(u), c1), tv2 = fp_add(one, tv1);
    tv1 = fp_sub(one, tv1);
    fp tv3 = fp_mul(tv1, tv2);
    tv3 = fp_is_zero(tv3) ? tv3 : fp_inv(tv3);                         /* inv0 */
    fp tv4 = fp_mul(fp_mul(fp_mul(u, tv1), tv3), c3);
    fp x1 = fp_sub(c2, tv4), x2 = fp_add(c2, tv4);
    fp x3 = fp_mul(fp_sqr(tv2), tv3);
    x3 = fp_add(fp_mul(fp_sqr(x3), c4), Z);
    const int e1 = fp_is_square(bn_g(x1), NULL), e2 = fp_is_square(bn_g(x2), NULL), e3 = fp_is_square(bn_g(x3), NULL);
    if (squares) *squares = e1 | (e2 << 1) | (e3 << 2);
    fp x = e1 ? x1 : (e2 ? x2 : x3), y;
    fp_is_square(bn_g(x), &y);
    if (fp_sgn0(u) != fp_sgn0(y)) y = fp_neg(y);
    g1a r; r.inf = 0; r.x = x; r.y = y;
    return r;
}
/* u: 48 big-endian bytes (reduced mod p); out: x || y little-endian; returns the squares mask (bit 0: g(x1), 1: g(x2), 2: g(x3)) */
int orc_bn_svdw_map(const uint8_t* u_be48, uint8_t* out) {
    init_all();
    int sq = 0;
    g1a_to_le(bn_svdw(fp_from_be48(u_be48), &sq), out);
    return sq;
}
/* hash_to_curve; squares_out[2] (may be NULL): the masks of the two maps */
void orc_bn_hash_to_g1(const uint8_t* msg, size_t mlen, const uint8_t* dst, size_t dlen, uint8_t* out, int* squares_out) {
    init_all();
    uint8_t uni[96];
    xmd(msg, mlen, dst, dlen, uni, 96);
    int s0 = 0, s1 = 0;
    g1a q0 = bn_svdw(fp_from_be48(uni), &s0), q1 = bn_svdw(fp_from_be48(uni + 48), &s1);
    if (squares_out) { squares_out[0] = s0; squares_out[1] = s1; }
    g1a_to_le(g1j_to_aff(g1j_add(g1j_from_aff(q0), g1j_from_aff(q1))), out);
}
#endif
