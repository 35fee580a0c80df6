// G1 group law (y^2 = x^3 + b, a = 0) in Jacobian coordinates, complete (every exceptional case
// is handled so results are the exact group elements the reference's ark-ec `Projective` produces,
// for any on-curve inputs, identity included).  Z == 0 encodes the identity.
//
// Reference call sites: every `generators[i] * scalar`, `b + ..`, `d * r - a_bar * e` in
// /root/reference/src/sign.rs:120-130, verify.rs:81-86, proof_gen.rs:249-263,
// proof_verify.rs:163-182.
#pragma once
#include "tower.hpp"

namespace bbs {

#define FP typename C::FpP
// point doubling / mixed addition with the multipliers INLINED: operands stay in VGPRs; measured
// on MI355X the MSM stage drops 7.4 -> 5.8 ms (the non-inlined form, -DBBS_G1_CALL_MUL, pays for
// argument traffic through scratch on every call)
#ifndef BBS_G1_CALL_MUL
#define G1MUL fe_mul_i
#define G1SQR fe_sqr_i
#else
#define G1MUL fe_mul
#define G1SQR fe_sqr
#endif

template <class C>
struct G1Aff {        // (0,0) encodes the identity (not on either curve: b != 0)
    Fp<C> x, y;
};

template <class C>
struct G1Jac {
    Fp<C> x, y, z;
};

template <class C> BBS_HD bool g1a_is_inf(const G1Aff<C>& p) { return fe_is_zero<FP>(p.x) & fe_is_zero<FP>(p.y); }
template <class C> BBS_HD bool g1j_is_inf(const G1Jac<C>& p) { return fe_is_zero<FP>(p.z); }
template <class C> BBS_HD G1Jac<C> g1j_inf() { return {fe_one<FP>(), fe_one<FP>(), fe_zero<FP>()}; }
template <class C> BBS_HD G1Aff<C> g1a_inf() { return {fe_zero<FP>(), fe_zero<FP>()}; }
template <class C> BBS_HD G1Aff<C> g1a_neg(const G1Aff<C>& p) { return {p.x, fe_neg<FP>(p.y)}; }
template <class C> BBS_HD G1Jac<C> g1j_neg(const G1Jac<C>& p) { return {p.x, fe_neg<FP>(p.y), p.z}; }

template <class C>
BBS_HD G1Jac<C> g1j_from_aff(const G1Aff<C>& p) {
    if (g1a_is_inf<C>(p)) return g1j_inf<C>();
    return {p.x, p.y, fe_one<FP>()};
}

template <class C>
BBS_HD Fp<C> curve_b() {
    Fp<C> b;
#pragma unroll
    for (int i = 0; i < C::FpP::N; i++) b.v[i] = C::K::B_M[i];
    return b;
}

template <class C>
BBS_HD bool g1a_on_curve(const G1Aff<C>& p) {
    if (g1a_is_inf<C>(p)) return true;
    Fp<C> lhs = fe_sqr<FP>(p.y);
    Fp<C> rhs = fe_add<FP>(fe_mul<FP>(fe_sqr<FP>(p.x), p.x), curve_b<C>());
    return fe_eq<FP>(lhs, rhs);
}

// dbl-2009-l (a = 0): 2M + 5S; the linear steps are single reduction chains (fe_lin)
template <class C>
BBS_HD G1Jac<C> g1j_dbl(const G1Jac<C>& p) {
    // identity (Z=0) maps to Z3 = 2*Y*0 = 0 : stays the identity.  Y = 0 cannot happen on
    // these curves (no point of order 2: x^3 = -b has no root in Fp for b = 4 / b = 3).
    Fp<C> A = G1SQR<FP>(p.x);
    Fp<C> B = G1SQR<FP>(p.y);
    Fp<C> Cc = G1SQR<FP>(B);
    Fp<C> t2 = G1SQR<FP>(fe_add_nr<FP>(p.x, B));              // (X + B)^2, lazy sum feeds the squarer
    Fp<C> D = fe_lin<FP, 2, -2, -2>(t2, A, Cc);               // 2((X+B)^2 - A - C)
    Fp<C> E = fe_scale<FP, 3>(A);
    Fp<C> F = G1SQR<FP>(E);
    G1Jac<C> r;
    r.x = fe_lin<FP, 1, -2>(F, D);
    Fp<C> c4 = fe_scale<FP, 4>(Cc);
    Fp<C> m = G1MUL<FP>(E, fe_sub<FP>(D, r.x));
    r.y = fe_lin<FP, 1, -2>(m, c4);                            // E (D - X3) - 8 C
    r.z = fe_scale<FP, 2>(G1MUL<FP>(p.y, p.z));
    return r;
}

// madd-2007-bl: Jacobian + affine, 7M + 4S, with the exceptional cases resolved
template <class C>
BBS_HD G1Jac<C> g1j_add_aff(const G1Jac<C>& p, const G1Aff<C>& q) {
    if (g1a_is_inf<C>(q)) return p;
    if (g1j_is_inf<C>(p)) return {q.x, q.y, fe_one<FP>()};
    Fp<C> Z1Z1 = G1SQR<FP>(p.z);
    Fp<C> U2 = G1MUL<FP>(q.x, Z1Z1);
    Fp<C> S2 = G1MUL<FP>(G1MUL<FP>(q.y, p.z), Z1Z1);
    Fp<C> H = fe_sub<FP>(U2, p.x);
    Fp<C> rr = fe_lin<FP, 2, -2>(S2, p.y);                     // 2 (S2 - Y1); zero iff S2 == Y1 (p odd)
    if (fe_is_zero<FP>(H)) {
        if (fe_is_zero<FP>(rr)) return g1j_dbl<C>(p);
        return g1j_inf<C>();
    }
    Fp<C> HH = G1SQR<FP>(H);
    Fp<C> I = fe_scale<FP, 4>(HH);
    Fp<C> J = G1MUL<FP>(H, I);
    Fp<C> V = G1MUL<FP>(p.x, I);
    G1Jac<C> r;
    r.x = fe_lin<FP, 1, -1, -2>(G1SQR<FP>(rr), J, V);
    Fp<C> m = G1MUL<FP>(rr, fe_sub<FP>(V, r.x));
    r.y = fe_lin<FP, 1, -2>(m, G1MUL<FP>(p.y, J));
    r.z = fe_lin<FP, 1, -1, -1>(G1SQR<FP>(fe_add_nr<FP>(p.z, H)), Z1Z1, HH);
    return r;
}

// add-2007-bl: Jacobian + Jacobian, 11M + 5S, with the exceptional cases resolved
template <class C>
BBS_HD_NOINLINE G1Jac<C> g1j_add(const G1Jac<C>& p, const G1Jac<C>& q) {
    if (g1j_is_inf<C>(q)) return p;
    if (g1j_is_inf<C>(p)) return q;
    Fp<C> Z1Z1 = fe_sqr<FP>(p.z);
    Fp<C> Z2Z2 = fe_sqr<FP>(q.z);
    Fp<C> U1 = fe_mul<FP>(p.x, Z2Z2);
    Fp<C> U2 = fe_mul<FP>(q.x, Z1Z1);
    Fp<C> S1 = fe_mul<FP>(fe_mul<FP>(p.y, q.z), Z2Z2);
    Fp<C> S2 = fe_mul<FP>(fe_mul<FP>(q.y, p.z), Z1Z1);
    Fp<C> H = fe_sub<FP>(U2, U1);
    Fp<C> rr = fe_sub<FP>(S2, S1);
    if (fe_is_zero<FP>(H)) {
        if (fe_is_zero<FP>(rr)) return g1j_dbl<C>(p);
        return g1j_inf<C>();
    }
    rr = fe_dbl<FP>(rr);
    Fp<C> I = fe_sqr<FP>(fe_dbl<FP>(H));
    Fp<C> J = fe_mul<FP>(H, I);
    Fp<C> V = fe_mul<FP>(U1, I);
    G1Jac<C> r;
    r.x = fe_sub<FP>(fe_sub<FP>(fe_sqr<FP>(rr), J), fe_dbl<FP>(V));
    r.y = fe_sub<FP>(fe_mul<FP>(rr, fe_sub<FP>(V, r.x)), fe_dbl<FP>(fe_mul<FP>(S1, J)));
    r.z = fe_mul<FP>(fe_sub<FP>(fe_sub<FP>(fe_sqr<FP>(fe_add<FP>(p.z, q.z)), Z1Z1), Z2Z2), H);
    return r;
}

template <class C>
BBS_HD_NOINLINE G1Aff<C> g1j_to_aff(const G1Jac<C>& p) {
    if (g1j_is_inf<C>(p)) return g1a_inf<C>();
    Fp<C> zi = fe_inv<FP>(p.z);
    Fp<C> zi2 = fe_sqr<FP>(zi);
    return {fe_mul<FP>(p.x, zi2), fe_mul<FP>(fe_mul<FP>(p.y, zi2), zi)};
}

// bit i of a canonical (non-Montgomery) scalar held as N 32-bit limbs
template <int N>
BBS_HD uint32_t limb_bit(const uint32_t* s, int i) {
    return (s[i >> 5] >> (i & 31)) & 1u;
}

// k * P for an affine P and a canonical 256-bit scalar: MSB-first double-and-add over the
// non-adjacent form of k (digits in {-1,0,1}: digit_{i-1} = bit_i(3k) - bit_i(k)), mixed additions
// of +-P.  ~256 doublings + ~85 additions; the group element equals ark-ec's
// `Projective::mul_bigint` result (plain double-and-add in the reference).
template <class C>
BBS_HD_NOINLINE G1Jac<C> g1_mul_aff(const G1Aff<C>& p, const uint32_t* k) {
    uint32_t h[9];
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {           // h = 3k = k + 2k
        const uint32_t two = (k[i] << 1) | (i ? (k[i - 1] >> 31) : 0u);
        c += (uint64_t)k[i] + two;
        h[i] = (uint32_t)c;
        c >>= 32;
    }
    h[8] = (uint32_t)c + (k[7] >> 31);
    const G1Aff<C> np = g1a_neg<C>(p);
    G1Jac<C> r = g1j_inf<C>();
    bool started = false;
    for (int i = 257; i >= 1; i--) {
        if (started) r = g1j_dbl<C>(r);
        const uint32_t hb = (h[i >> 5] >> (i & 31)) & 1u;
        const uint32_t kb = (i < 256) ? ((k[i >> 5] >> (i & 31)) & 1u) : 0u;
        if (hb != kb) {
            r = g1j_add_aff<C>(r, hb ? p : np);
            started = true;
        }
    }
    return r;
}

#undef FP
#undef G1MUL
#undef G1SQR
}  // namespace bbs
