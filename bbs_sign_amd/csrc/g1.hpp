// G1 group law (y^2 = x^3 + b, a = 0) in Jacobian coordinates, complete (every exceptional case
// is handled so results are the exact group elements the reference's ark-ec `Projective` produces,
// for any on-curve inputs, identity included).  Z == 0 encodes the identity.
//
// Reference call sites: every `generators[i] * scalar`, `b + ..`, `d * r - a_bar * e` in
// /root/reference/src/sign.rs:120-130, verify.rs:81-86, proof_gen.rs:249-263,
// proof_verify.rs:163-182.
#pragma once
#include <type_traits>
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
// TAG: a separate instance for callers that live in register-capped kernels (the register budget of a device function is
// the loosest one among the kernels that reach it, and the kernel is charged the maximum over everything it can reach)
// (_i: the body, inlined -- for the lane-per-item stages that sum a handful of partial sums: a call hands both operands and
// the result through the caller's frame, 624 bytes of the callee's frame on top; the multipliers inside stay calls)
template <class C>
BBS_HD G1Jac<C> g1j_add_i(const G1Jac<C>& p, const G1Jac<C>& q) {
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
template <class C, int TAG = 0>
BBS_HD_NOINLINE G1Jac<C> g1j_add(const G1Jac<C>& p, const G1Jac<C>& q) { return g1j_add_i<C>(p, q); }

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

// Generic path (any on-curve P, small-order points included):
// k * P for an affine P and a canonical 256-bit scalar: MSB-first double-and-add over the
// non-adjacent form of k (digits in {-1,0,1}: digit_{i-1} = bit_i(3k) - bit_i(k)), mixed additions
// of +-P.  ~256 doublings + ~85 additions; the group element equals ark-ec's
// `Projective::mul_bigint` result (plain double-and-add in the reference).
// (INLINED, round 5.  As a called function it was reached from two places of one kernel -- stage PvChains: the fall-back of
// T1's joint chain and the fall-back of the single multiplications, both inlined into the kernel, both in divergent code -- and
// that kernel then computed wrong sums for EVERY lane of a wavefront in which one lane took the call (MI355X, ROCm 7.2: all
// items of tests/parity_cases.py::check_batch_verification's identity-point batch read "challenge mismatch"; the kernels with
// one such call site each, and the form that reaches it through g1_mul_aff_tab_to, were right).  No undefined behaviour was
// found in the source, the ISA at the call sites looks orderly; inlining removes the call and the difference
// (profiles/r05_b_pvchains_identity_point.log).  The body is one doubling and one mixed addition in a rolled loop.)
template <class C>
BBS_HD G1Jac<C> g1_mul_aff_naf(const G1Aff<C>& p, const uint32_t* k) {
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

// ---- regular fixed-window scalar multiplication ------------------------------------------------
// On a GPU the 64 lanes of a wavefront run one instruction stream: with a sparse recoding (NAF) some lane needs
// the addition at almost every bit, so the wavefront pays ~256 doublings + ~256 additions per multiplication.
// Here every lane adds at the same 64 steps: k (made odd) is written with 64 ODD digits d_i in {+-1, .., +-15}
// (u = (k >> 1) | 2^255, d_i = 2 * nibble_i(u) - 15), the table {1,3,..,15} P is built with 1 doubling + 7 mixed
// additions and brought to ONE common Z, i.e. to affine points of the isomorphic curve y^2 = x^3 + b Z^6 on
// which the whole loop then runs with mixed additions (the a = 0 formulas do not involve b); the result's Z is
// multiplied by the common Z at the end.  ~252 doublings + 64 mixed additions + ~125 multiplications of set-up.
// An even k is handled as (k + 1) P - P.  The group element is the one ark-ec's double-and-add produces.
// Inputs for which the table would hit an exceptional case (identity, points of order < 16: on-curve points
// outside the prime-order subgroup) take the generic path.

// mixed addition that also returns Z3 / Z1 (= 2 H); false in an exceptional case
template <class C>
BBS_HD bool g1j_add_aff_zr(const G1Jac<C>& p, const G1Aff<C>& q, G1Jac<C>& r, Fp<C>& zr) {
    Fp<C> Z1Z1 = G1SQR<FP>(p.z);
    Fp<C> U2 = G1MUL<FP>(q.x, Z1Z1);
    Fp<C> S2 = G1MUL<FP>(G1MUL<FP>(q.y, p.z), Z1Z1);
    Fp<C> H = fe_sub<FP>(U2, p.x);
    if (fe_is_zero<FP>(H)) return false;
    Fp<C> rr = fe_lin<FP, 2, -2>(S2, p.y);
    Fp<C> HH = G1SQR<FP>(H);
    Fp<C> I = fe_scale<FP, 4>(HH);
    Fp<C> J = G1MUL<FP>(H, I);
    Fp<C> V = G1MUL<FP>(p.x, I);
    r.x = fe_lin<FP, 1, -1, -2>(G1SQR<FP>(rr), J, V);
    Fp<C> m = G1MUL<FP>(rr, fe_sub<FP>(V, r.x));
    r.y = fe_lin<FP, 1, -2>(m, G1MUL<FP>(p.y, J));
    r.z = fe_lin<FP, 1, -1, -1>(G1SQR<FP>(fe_add_nr<FP>(p.z, H)), Z1Z1, HH);
    zr = fe_dbl<FP>(H);
    return true;
}

constexpr int G1_TAB = 8;          // odd multiples 1, 3, .., 15

// where a table lives: in the lane's private memory (one multiplication: 896 B of scratch for BLS12-381) or in
// a caller-provided HBM buffer laid out [entry][word][item] (the three tables of a joint multiplication would be
// 2.7 KB of scratch per lane; scratch is reserved per hardware queue for every wave slot of the chip, and 16
// queues x 5.4 KB x 64 lanes x 8192 slots did not fit -- HSA_STATUS_ERROR_OUT_OF_RESOURCES)
template <class C>
struct TabPriv {
    G1Aff<C> t[G1_TAB];
    BBS_HD G1Aff<C> ld(uint32_t e) const { return t[e]; }
    BBS_HD void st(uint32_t e, const G1Aff<C>& p) { t[e] = p; }
};
template <class C>
struct TabHbm {
    uint32_t* base;          // word w of entry e at base[(e * 2N + w) * stride]
    size_t stride;
    BBS_HD G1Aff<C> ld(uint32_t e) const {
        constexpr int N = C::FpP::N;
        G1Aff<C> p;
        const uint32_t* b = base + (size_t)e * 2 * N * stride;
#pragma unroll
        for (int w = 0; w < N; w++) { p.x.v[w] = b[(size_t)w * stride]; p.y.v[w] = b[(size_t)(N + w) * stride]; }
        return p;
    }
    BBS_HD void st(uint32_t e, const G1Aff<C>& p) {
        constexpr int N = C::FpP::N;
        uint32_t* b = base + (size_t)e * 2 * N * stride;
#pragma unroll
        for (int w = 0; w < N; w++) { b[(size_t)w * stride] = p.x.v[w]; b[(size_t)(N + w) * stride] = p.y.v[w]; }
    }
};

// "where the table of a single multiplication lives", passed to the (non-inlined) multiplication routines
template <class C>
struct AtPriv {
    using Tab = TabPriv<C>;
    BBS_HD void init(Tab&) const {}
};
template <class C>
struct AtHbm {
    using Tab = TabHbm<C>;
    uint32_t* base;
    size_t stride;
    BBS_HD void init(Tab& t) const { t.base = base; t.stride = stride; }
};

// table of the odd multiples of P as affine points of the curve isomorphic by `zc` (point (x', y') there is the
// Jacobian point (x', y', zc) here); false if P is the identity or an exceptional case occurred.
// (Inlined into its caller: as a separate function writing the caller's private table through pointers it
// faulted on MI355X with a memory aperture violation -- tools/repro_mul.py -- while the inlined form is clean.)
template <class C, class T>
BBS_HD bool g1_odd_table(const G1Aff<C>& p, T& tab, Fp<C>& zc) {
    if (g1a_is_inf<C>(p)) return false;
    const G1Jac<C> d = g1j_dbl<C>(G1Jac<C>{p.x, p.y, fe_one<FP>()});          // Z = 2 y != 0 (no 2-torsion)
    const Fp<C> zd2 = fe_sqr<FP>(d.z), zd3 = fe_mul<FP>(zd2, d.z);
    const G1Aff<C> dd = {d.x, d.y};                                            // 2P, affine on the curve scaled by d.z
    G1Jac<C> cur = {fe_mul<FP>(p.x, zd2), fe_mul<FP>(p.y, zd3), fe_one<FP>()}; // P on that curve
    Fp<C> zr[G1_TAB];
    tab.st(0, G1Aff<C>{cur.x, cur.y});
#pragma unroll 1
    for (int j = 1; j < G1_TAB; j++) {
        G1Jac<C> nxt;
        if (!g1j_add_aff_zr<C>(cur, dd, nxt, zr[j])) return false;
        cur = nxt;
        tab.st(j, G1Aff<C>{cur.x, cur.y});                                     // (2j+1) P with Z_j; rescaled below
    }
    Fp<C> sc = zr[G1_TAB - 1];                                                 // Z_7 / Z_j
#pragma unroll 1
    for (int j = G1_TAB - 2; j >= 0; j--) {
        const Fp<C> s2 = fe_sqr<FP>(sc);
        const G1Aff<C> e = tab.ld(j);
        tab.st(j, G1Aff<C>{fe_mul<FP>(e.x, s2), fe_mul<FP>(fe_mul<FP>(e.y, s2), sc)});
        if (j) sc = fe_mul<FP>(sc, zr[j]);
    }
    zc = fe_mul<FP>(cur.z, d.z);
    return true;
}

// u = ((k | 1) >> 1) | 2^255 : nibble i of u gives the odd digit d_i = 2 U - 15
BBS_HD void g1_recode(const uint32_t* k, uint32_t* u) {
#pragma unroll
    for (int i = 0; i < 7; i++) u[i] = (k[i] >> 1) | (k[i + 1] << 31);
    u[7] = (k[7] >> 1) | 0x80000000u;
}
template <class C, class T>
BBS_HD G1Aff<C> g1_tab_digit(const T& tab, uint32_t U, bool flip = false) {      // flip: the digit of a negative half
    const bool low = U < 8, neg = low != flip;
    G1Aff<C> q = tab.ld(low ? 7u - U : U - 8u);
    q.y = fe_select<FP>(neg, fe_neg<FP>(q.y), q.y);
    return q;
}

// (the table lives where the caller says: AtPriv = the lane's private memory, AtHbm = a caller-provided buffer)
// (result through `out`, the running point a plain local: as a named return value it is the caller's memory and every
// doubling of the loop then starts with a scratch round trip -- DESIGN.md 5 rule 7b)
// (_inl: the body, for a kernel that is nothing but this multiplication -- stage PvVarMul -- where a call would only add the
// callee's frame and its saved registers to the kernel's scratch; everyone else calls the non-inlined wrapper below)
template <class C, class W>
BBS_HD void g1_mul_aff_tab_inl(const G1Aff<C>& p, const uint32_t* k, const W where, G1Jac<C>& out) {
#ifdef BBS_G1_MUL_NAF
    out = g1_mul_aff_naf<C>(p, k);
    return;
#endif
    typename W::Tab tab;
    where.init(tab);
    Fp<C> zc;
    if (!g1_odd_table<C>(p, tab, zc)) { out = g1_mul_aff_naf<C>(p, k); return; }
    uint32_t u[8];
    g1_recode(k, u);
    const bool even = (k[0] & 1u) == 0;
    G1Jac<C> r = g1j_from_aff<C>(g1_tab_digit<C>(tab, u[7] >> 28));
    // entry of step i (i = -1: the correction (k + 1) P - P of an even k); requested one step ahead of its addition
    auto fetch = [&](int i) -> G1Aff<C> {
        if (i >= 0) return g1_tab_digit<C>(tab, (u[i >> 3] >> (4 * (i & 7))) & 15u);
        return even ? g1a_neg<C>(tab.ld(0)) : g1a_inf<C>();
    };
    G1Aff<C> qn = fetch(62);
#pragma unroll 1
    for (int i = 62; i >= -1; i--) {
        const G1Aff<C> q = qn;
        if (i > -1) qn = fetch(i - 1);
        if (i >= 0) {
#pragma unroll 1
            for (int t = 0; t < 4; t++) r = g1j_dbl<C>(r);
        }
        r = g1j_add_aff<C>(r, q);
    }
    r.z = fe_mul<FP>(r.z, zc);
    out = r;
}
template <class C, class W>
BBS_HD_NOINLINE void g1_mul_aff_tab_to(const G1Aff<C>& p, const uint32_t* k, const W where, G1Jac<C>& out) { g1_mul_aff_tab_inl<C, W>(p, k, where, out); }
template <class C, class W>
BBS_HD G1Jac<C> g1_mul_aff_tab(const G1Aff<C>& p, const uint32_t* k, const W where) {
    G1Jac<C> r;
    g1_mul_aff_tab_to<C, W>(p, k, where, r);
    return r;
}

template <class C>
BBS_HD G1Jac<C> g1_mul_aff(const G1Aff<C>& p, const uint32_t* k) { return g1_mul_aff_tab<C, AtPriv<C>>(p, k, AtPriv<C>{}); }

// ---- GLV split (BLS12-381; opt-in: bbs_ctx_set_points_in_subgroup) ----------------------------
// On the prime-order subgroup (beta x, y) = [lambda] (x, y) with lambda = x^2 - 1 ~ 2^127.4 a root of X^2 + X + 1
// mod r, so k P = k1 P + k2 phi(P) with k2 = floor(k / lambda), k1 = k mod lambda, both below 2^128: half the
// doublings (124 instead of 252) for the same number of table additions, the phi-table being the same table with
// x multiplied by beta.  The identity phi(P) = lambda P holds ONLY on the subgroup: for an on-curve point outside it
// the result is not k P, which is why the default path does not use it (the reference's types guarantee membership,
// this ABI takes raw coordinates).  Same group element as the plain chain for every P in G1.
template <class C>
BBS_HD void glv_split_simple(const uint32_t* k, uint32_t* k1, uint32_t* k2) {
    using K = typename C::K;
    uint32_t mu[5], lam[4];
#pragma unroll
    for (int j = 0; j < 5; j++) mu[j] = K::GLV_MU[j];
#pragma unroll
    for (int j = 0; j < 4; j++) lam[j] = K::GLV_LAMBDA[j];
    // q = floor(k mu / 2^256) with mu = floor(2^256 / lambda):  floor(k / lambda) - 2 < q <= floor(k / lambda)
    uint32_t pr[13];
#pragma unroll
    for (int j = 0; j < 13; j++) pr[j] = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < 5; j++) {
            c += (uint64_t)pr[i + j] + (uint64_t)k[i] * mu[j];
            pr[i + j] = (uint32_t)c;
            c >>= 32;
        }
        pr[i + 5] = (uint32_t)c;
    }
    uint32_t q[4] = {pr[8], pr[9], pr[10], pr[11]};
    // rem = k - q lambda  (< 3 lambda < 2^130: five words)
    uint32_t ql[8];
#pragma unroll
    for (int j = 0; j < 8; j++) ql[j] = 0;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        uint64_t c = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            c += (uint64_t)ql[i + j] + (uint64_t)q[i] * lam[j];
            ql[i + j] = (uint32_t)c;
            c >>= 32;
        }
        ql[i + 4] = (uint32_t)c;
    }
    uint32_t rem[5];
    {
        int64_t b = 0;
#pragma unroll
        for (int j = 0; j < 5; j++) {
            b += (int64_t)k[j] - (int64_t)ql[j];
            rem[j] = (uint32_t)b;
            b >>= 32;
        }
    }
#pragma unroll
    for (int rep = 0; rep < 2; rep++) {
        uint32_t d[5];
        int64_t b = 0;
#pragma unroll
        for (int j = 0; j < 5; j++) {
            b += (int64_t)rem[j] - (int64_t)(j < 4 ? lam[j] : 0u);
            d[j] = (uint32_t)b;
            b >>= 32;
        }
        const bool ge = b == 0;                 // no borrow: rem >= lambda
        uint64_t c = ge ? 1u : 0u;
#pragma unroll
        for (int j = 0; j < 5; j++) rem[j] = ge ? d[j] : rem[j];
#pragma unroll
        for (int j = 0; j < 4; j++) {
            c += q[j];
            q[j] = (uint32_t)c;
            c >>= 32;
        }
    }
#pragma unroll
    for (int j = 0; j < 4; j++) { k1[j] = rem[j]; k2[j] = q[j]; }
}

// BN254: lambda is a full-length root of X^2 + X + 1 mod r; (k1, k2) = (k, 0) - c1 (a1, b1) - c2 (a2, b2) with a short
// basis of {(x, y): x + y lambda = 0 mod r} and c_i = round(k b / r) taken as (k G_i + 2^319) >> 320 (params_gen.hpp,
// derived and bounded in tools/gen_params.py: |k1|, |k2| < 2^127).  k = k1 + k2 lambda mod r holds for any integers
// c1, c2, so the rounding only affects the lengths.  Halves come out as magnitude + sign.  Every on-curve point of a
// cofactor-1 curve is in the subgroup, so this split needs no vouching (K::GLV_ALWAYS).
template <class C>
BBS_HD void glv_split_lattice(const uint32_t* k, uint32_t* k1, uint32_t* k2, bool& neg1, bool& neg2) {
    using K = typename C::K;
    // c = (k G + 2^319) >> 320 : words 10..13 of the 15-word product
    auto quot = [&](const uint32_t* G, uint32_t* c) {
        uint32_t pr[15];
#pragma unroll
        for (int j = 0; j < 15; j++) pr[j] = 0;
#pragma unroll
        for (int i = 0; i < 8; i++) {
            uint64_t cy = 0;
#pragma unroll
            for (int j = 0; j < 7; j++) {
                cy += (uint64_t)pr[i + j] + (uint64_t)k[i] * G[j];
                pr[i + j] = (uint32_t)cy;
                cy >>= 32;
            }
            pr[i + 7] = (uint32_t)cy;
        }
        uint64_t cy = (uint64_t)pr[9] + 0x80000000u;            // + 2^319
        cy >>= 32;
#pragma unroll
        for (int j = 0; j < 4; j++) { cy += pr[10 + j]; c[j] = (uint32_t)cy; cy >>= 32; }
    };
    // acc (192-bit two's complement) -= / += c * a  (low six words; the final values are below 2^128 in magnitude)
    auto mulacc = [&](uint32_t* acc, const uint32_t* c, const uint32_t* a, bool subtract) {
        uint32_t t[8];
#pragma unroll
        for (int j = 0; j < 8; j++) t[j] = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            uint64_t cy = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                cy += (uint64_t)t[i + j] + (uint64_t)c[i] * a[j];
                t[i + j] = (uint32_t)cy;
                cy >>= 32;
            }
            t[i + 4] = (uint32_t)cy;
        }
        int64_t b = 0;
#pragma unroll
        for (int j = 0; j < 6; j++) {
            b += (int64_t)acc[j] + (subtract ? -(int64_t)t[j] : (int64_t)t[j]);
            acc[j] = (uint32_t)b;
            b >>= 32;
        }
    };
    auto finish = [&](uint32_t* acc, uint32_t* out, bool& neg) {
        neg = (acc[5] >> 31) != 0;
        uint64_t cy = neg ? 1u : 0u;
#pragma unroll
        for (int j = 0; j < 4; j++) {
            cy += neg ? (uint32_t)~acc[j] : acc[j];
            out[j] = (uint32_t)cy;
            cy >>= 32;
        }
    };
    uint32_t G1[7], G2[7], A1[4], B1[4], A2[4], B2[4];
#pragma unroll
    for (int j = 0; j < 7; j++) { G1[j] = K::GLV_G1[j]; G2[j] = K::GLV_G2[j]; }
#pragma unroll
    for (int j = 0; j < 4; j++) { A1[j] = K::GLV_A1[j]; B1[j] = K::GLV_B1[j]; A2[j] = K::GLV_A2[j]; B2[j] = K::GLV_B2[j]; }
    uint32_t c1[4], c2[4];
    quot(G1, c1);
    quot(G2, c2);
    uint32_t x[6], y[6];
#pragma unroll
    for (int j = 0; j < 6; j++) { x[j] = k[j]; y[j] = 0; }
    // k1 = k - c1 a1 - c2 a2 ; k2 = -c1 b1 - c2 b2   (c_i = +-|c_i|, a, b = +-|.|: a product is subtracted when its signs agree)
    mulacc(x, c1, A1, K::GLV_C1_NEG == K::GLV_A1_NEG);
    mulacc(x, c2, A2, K::GLV_C2_NEG == K::GLV_A2_NEG);
    mulacc(y, c1, B1, K::GLV_C1_NEG == K::GLV_B1_NEG);
    mulacc(y, c2, B2, K::GLV_C2_NEG == K::GLV_B2_NEG);
    finish(x, k1, neg1);
    finish(y, k2, neg2);
}

// k = (+-k1) + (+-k2) lambda mod r, k1, k2 < 2^128
template <class C>
BBS_HD void glv_split(const uint32_t* k, uint32_t* k1, uint32_t* k2, bool& neg1, bool& neg2) {
    if constexpr (C::K::GLV_LATTICE) {
        glv_split_lattice<C>(k, k1, k2, neg1, neg2);
    } else {
        glv_split_simple<C>(k, k1, k2);
        neg1 = false;
        neg2 = false;
    }
}

// 32 odd digits of a 128-bit h (made odd): u = ((h | 1) >> 1) | 2^127
BBS_HD void g1_recode128(const uint32_t* h, uint32_t* u) {
#pragma unroll
    for (int i = 0; i < 3; i++) u[i] = (h[i] >> 1) | (h[i + 1] << 31);
    u[3] = (h[3] >> 1) | 0x80000000u;
}

template <class C>
BBS_HD Fp<C> glv_beta() {
    Fp<C> b;
#pragma unroll
    for (int i = 0; i < C::FpP::N; i++) b.v[i] = C::K::BETA_L_M[i];
    return b;
}

// k * P for P in the prime-order subgroup: two 128-bit halves on one doubling chain (see above)
template <class C, class W>
BBS_HD void g1_mul_aff_glv_tab_inl(const G1Aff<C>& p, const uint32_t* k, const W where, G1Jac<C>& out) {
    typename W::Tab tab;
    where.init(tab);
    Fp<C> zc;
    if (!g1_odd_table<C>(p, tab, zc)) { out = g1_mul_aff_naf<C>(p, k); return; }
    uint32_t h[2][4], u[2][4];
    bool neg[2];
    glv_split<C>(k, h[0], h[1], neg[0], neg[1]);
    g1_recode128(h[0], u[0]);
    g1_recode128(h[1], u[1]);
    const bool even[2] = {(h[0][0] & 1u) == 0, (h[1][0] & 1u) == 0};
    const Fp<C> beta = glv_beta<C>();
    // 33 rounds (digits 31 .. 0, then the even-half corrections) x (P, phi P): step s = round * 2 + half
    constexpr int STEPS = 33 * 2;
    auto fetch = [&](int s) -> G1Aff<C> {
        const int rd = s >> 1, j = s & 1, i = 31 - rd;
        G1Aff<C> q;
        if (i >= 0) q = g1_tab_digit<C>(tab, (u[j][i >> 3] >> (4 * (i & 7))) & 15u, neg[j]);
        else q = even[j] ? (neg[j] ? tab.ld(0) : g1a_neg<C>(tab.ld(0))) : g1a_inf<C>();
        return q;
    };
    G1Jac<C> r = g1j_inf<C>();
    G1Aff<C> qn = fetch(0);
#pragma unroll 1
    for (int s = 0; s < STEPS; s++) {
        G1Aff<C> q = qn;
        if (s + 1 < STEPS) qn = fetch(s + 1);
        if (s & 1) q.x = G1MUL<FP>(q.x, beta);          // phi, applied after the next entry has been requested
        const int rd = s >> 1;
        if ((s & 1) == 0 && rd >= 1 && rd <= 31) {
#pragma unroll 1
            for (int t = 0; t < 4; t++) r = g1j_dbl<C>(r);
        }
        r = g1j_add_aff<C>(r, q);
    }
    r.z = fe_mul<FP>(r.z, zc);
    out = r;
}
template <class C, class W>
BBS_HD_NOINLINE void g1_mul_aff_glv_tab_to(const G1Aff<C>& p, const uint32_t* k, const W where, G1Jac<C>& out) { g1_mul_aff_glv_tab_inl<C, W>(p, k, where, out); }
template <class C, class W>
BBS_HD G1Jac<C> g1_mul_aff_glv_tab(const G1Aff<C>& p, const uint32_t* k, const W where) {
    G1Jac<C> r;
    g1_mul_aff_glv_tab_to<C, W>(p, k, where, r);
    return r;
}

// k0 P0 + k1 P1 + k2 P2 on ONE shared doubling chain (Straus): the three tables are brought to one common curve
// (scale of table j times the other two tables' scales), ~252 doublings + 3 * 64 mixed additions.
// The tables live in the caller's HBM buffer `tabs` (3 * G1_TAB * 2N words, stride apart); ENTRY 0 OF TABLE j HOLDS P_j ON
// ENTRY (stored there by the caller: the points are then no memory objects of this function).
// Returns false (out untouched) when a table hit an exceptional case (identity, point of small order): the caller then
// takes three separate multiplications.
// GLV = true (points in the prime-order subgroup, BLS12-381): six 128-bit halves, ~124 doublings + 6 * 33 additions.
// Round 5: INLINED into its one device caller (stage PvT1Chain, a kernel of its own), loops over the tables rolled and free
// of indexed locals.  As one non-inlined function it held the points, the scalars, the tables' scales and its callee-saved
// registers in a 1264-byte frame below a 1456-byte kernel frame: 2.7 KB of scratch per lane, and scratch x hardware queues
// is a budget (DESIGN.md 5 rule 6).
template <class C, bool GLV = false>
BBS_HD bool g1_mul3_tabs_fast(const uint32_t* k0, const uint32_t* k1, const uint32_t* k2, uint32_t* tabs, size_t stride, G1Jac<C>& out) {
#ifdef BBS_G1_MUL_NAF
    return false;
#endif
    constexpr int N = C::FpP::N;
    constexpr size_t TW = (size_t)G1_TAB * 2 * N;
    Fp<C> zc0 = fe_one<FP>(), zc1 = zc0, zc2 = zc0;
    bool ok = true;
#pragma unroll 1
    for (int j = 0; j < 3; j++) {
        TabHbm<C> tab{tabs + (size_t)j * TW * stride, stride};
        Fp<C> z = fe_one<FP>();
        ok = g1_odd_table<C>(tab.ld(0), tab, z) && ok;
        zc0 = fe_select<FP>(j == 0, z, zc0);
        zc1 = fe_select<FP>(j == 1, z, zc1);
        zc2 = fe_select<FP>(j == 2, z, zc2);
    }
    if (!ok) return false;
    // point (x, y) of table j is Jacobian (x, y, zc_j) = (x t^2, y t^3, zc_0 zc_1 zc_2) with t = product of the other two
    const Fp<C> z01 = fe_mul<FP>(zc0, zc1), z12 = fe_mul<FP>(zc1, zc2), z02 = fe_mul<FP>(zc0, zc2);
    const Fp<C> zall = fe_mul<FP>(z01, zc2);
#pragma unroll 1
    for (int j = 0; j < 3; j++) {
        TabHbm<C> tab{tabs + (size_t)j * TW * stride, stride};
        const Fp<C> tj = fe_select<FP>(j == 0, z12, fe_select<FP>(j == 1, z02, z01));
        const Fp<C> t2 = fe_sqr<FP>(tj), t3 = fe_mul<FP>(t2, tj);
#pragma unroll 1
        for (int e = 0; e < G1_TAB; e++) {
            const G1Aff<C> q = tab.ld(e);
            tab.st(e, G1Aff<C>{fe_mul<FP>(q.x, t2), fe_mul<FP>(q.y, t3)});
        }
    }
    G1Jac<C> r = g1j_inf<C>();
    if constexpr (GLV) {
        // terms t = 2 * table + half: (P_j, k_j mod lambda), (phi P_j, floor(k_j / lambda))
        uint32_t u[6][4];
        bool even[6], neg[6];
        const uint32_t* ks[3] = {k0, k1, k2};
#pragma unroll
        for (int j = 0; j < 3; j++) {
            uint32_t h0[4], h1[4];
            glv_split<C>(ks[j], h0, h1, neg[2 * j], neg[2 * j + 1]);
            g1_recode128(h0, u[2 * j]);
            g1_recode128(h1, u[2 * j + 1]);
            even[2 * j] = (h0[0] & 1u) == 0;
            even[2 * j + 1] = (h1[0] & 1u) == 0;
        }
        const Fp<C> beta = glv_beta<C>();
        constexpr int STEPS = 33 * 6;
        auto fetch = [&](int s) -> G1Aff<C> {
            const int rd = s / 6, t = s - 6 * rd, i = 31 - rd;
            const TabHbm<C> tab{tabs + (size_t)(t >> 1) * TW * stride, stride};
            G1Aff<C> q;
            if (i >= 0) q = g1_tab_digit<C>(tab, (u[t][i >> 3] >> (4 * (i & 7))) & 15u, neg[t]);
            else q = even[t] ? (neg[t] ? tab.ld(0) : g1a_neg<C>(tab.ld(0))) : g1a_inf<C>();
            return q;
        };
        G1Aff<C> qn = fetch(0);
#pragma unroll 1
        for (int s = 0; s < STEPS; s++) {
            G1Aff<C> q = qn;
            if (s + 1 < STEPS) qn = fetch(s + 1);
            if (s & 1) q.x = G1MUL<FP>(q.x, beta);      // phi (odd terms), applied after the next entry has been requested
            const int rd = s / 6;
            if (s == 6 * rd && rd >= 1 && rd <= 31) {
#pragma unroll 1
                for (int t = 0; t < 4; t++) r = g1j_dbl<C>(r);
            }
            r = g1j_add_aff<C>(r, q);
        }
    } else {
        uint32_t u[3][8];
        g1_recode(k0, u[0]); g1_recode(k1, u[1]); g1_recode(k2, u[2]);
        const bool even[3] = {(k0[0] & 1u) == 0, (k1[0] & 1u) == 0, (k2[0] & 1u) == 0};
        // 65 rounds (digits 63 .. 0, then the even-scalar corrections) x 3 tables, flattened: step s = round * 3 + table;
        // the entry of step s + 1 is requested before the addition of step s (HBM table reads hidden behind it)
        constexpr int STEPS = 65 * 3;
        auto fetch = [&](int s) -> G1Aff<C> {
            const int rd = s / 3, j = s - 3 * rd, i = 63 - rd;
            const TabHbm<C> tab{tabs + (size_t)j * TW * stride, stride};
            if (i >= 0) return g1_tab_digit<C>(tab, (u[j][i >> 3] >> (4 * (i & 7))) & 15u);
            return even[j] ? g1a_neg<C>(tab.ld(0)) : g1a_inf<C>();
        };
        G1Aff<C> qn = fetch(0);
#pragma unroll 1
        for (int s = 0; s < STEPS; s++) {
            const G1Aff<C> q = qn;
            if (s + 1 < STEPS) qn = fetch(s + 1);
            const int rd = s / 3;
            if (s == 3 * rd && rd >= 1 && rd <= 63) {
#pragma unroll 1
                for (int t = 0; t < 4; t++) r = g1j_dbl<C>(r);
            }
            r = g1j_add_aff<C>(r, q);
        }
    }
    r.z = fe_mul<FP>(r.z, zall);
    out = r;
    return true;
}
// host self-test form (bbs_selftest_mul3): the joint chain, or -- a table hit an exceptional case -- the sum of three
// separate multiplications, each of which falls back to the generic chain on its own (tables in the caller's buffer).
// The device does the same split across stages: PvT1Chain stores the three products, PvChallenge sums them.
template <class C>
BBS_HD G1Jac<C> g1_mul3_aff(const G1Aff<C>& p0, const uint32_t* k0, const G1Aff<C>& p1, const uint32_t* k1,
                             const G1Aff<C>& p2, const uint32_t* k2, uint32_t* tabs, size_t stride, bool glv = false) {
    constexpr size_t TW = (size_t)G1_TAB * 2 * C::FpP::N;
    { TabHbm<C> t0{tabs, stride}, t1{tabs + TW * stride, stride}, t2{tabs + 2 * TW * stride, stride};
      t0.st(0, p0); t1.st(0, p1); t2.st(0, p2); }
    G1Jac<C> r;
    if constexpr (C::K::HAS_GLV) {
        if (glv && g1_mul3_tabs_fast<C, true>(k0, k1, k2, tabs, stride, r)) return r;
    }
    if (!glv && g1_mul3_tabs_fast<C>(k0, k1, k2, tabs, stride, r)) return r;
    const AtHbm<C> w0{tabs, stride}, w1{tabs + TW * stride, stride}, w2{tabs + 2 * TW * stride, stride};
    return g1j_add<C, 1>(g1j_add<C, 1>(g1_mul_aff_tab<C, AtHbm<C>>(p0, k0, w0), g1_mul_aff_tab<C, AtHbm<C>>(p1, k1, w1)),
                      g1_mul_aff_tab<C, AtHbm<C>>(p2, k2, w2));
}

// k0 P0 + k1 P1 on ONE shared doubling chain: the two-term form of g1_mul3_tabs_fast.  Used by proof_gen's throughput
// form: Bbar = (r1 r2) B - (e r1 r2) A and T1 = (r1~ r2) B + (e~ r1 r2) A (src/proof_gen.rs:254-258 restructured over A and
// B) are two chains of ~252 doublings instead of four.  Tables in the caller's HBM buffer (2 * G1_TAB * 2N words, stride apart);
// ENTRY 0 OF TABLE j HOLDS P_j ON ENTRY.  Inlined into its one device caller (stage PgVarPart), loops over the tables rolled
// and free of indexed locals, as g1_mul3_tabs_fast (round 5: its frame was 1216 bytes below the kernel's).
template <class C, bool GLV = false>
BBS_HD bool g1_mul2_tabs_fast(const uint32_t* k0, const uint32_t* k1, uint32_t* tabs, size_t stride, G1Jac<C>& out) {
#ifdef BBS_G1_MUL_NAF
    return false;
#endif
    constexpr int N = C::FpP::N;
    constexpr size_t TW = (size_t)G1_TAB * 2 * N;
    Fp<C> zc0 = fe_one<FP>(), zc1 = zc0;
    bool ok = true;
#pragma unroll 1
    for (int j = 0; j < 2; j++) {
        TabHbm<C> tab{tabs + (size_t)j * TW * stride, stride};
        Fp<C> z = fe_one<FP>();
        ok = g1_odd_table<C>(tab.ld(0), tab, z) && ok;
        zc0 = fe_select<FP>(j == 0, z, zc0);
        zc1 = fe_select<FP>(j == 1, z, zc1);
    }
    if (!ok) return false;
    // point (x, y) of table j is Jacobian (x, y, zc_j) = (x t^2, y t^3, zc_0 zc_1) with t = the other table's scale
    const Fp<C> zall = fe_mul<FP>(zc0, zc1);
#pragma unroll 1
    for (int j = 0; j < 2; j++) {
        TabHbm<C> tab{tabs + (size_t)j * TW * stride, stride};
        const Fp<C> tj = fe_select<FP>(j == 0, zc1, zc0);
        const Fp<C> t2 = fe_sqr<FP>(tj), t3 = fe_mul<FP>(t2, tj);
#pragma unroll 1
        for (int e = 0; e < G1_TAB; e++) {
            const G1Aff<C> q = tab.ld(e);
            tab.st(e, G1Aff<C>{fe_mul<FP>(q.x, t2), fe_mul<FP>(q.y, t3)});
        }
    }
    G1Jac<C> r = g1j_inf<C>();
    if constexpr (GLV) {
        // terms t = 2 * table + half: (P_j, k_j mod lambda), (phi P_j, floor(k_j / lambda))
        uint32_t u[4][4];
        bool even[4], neg[4];
        const uint32_t* ks[2] = {k0, k1};
#pragma unroll 1
        for (int j = 0; j < 2; j++) {
            uint32_t h0[4], h1[4];
            glv_split<C>(ks[j], h0, h1, neg[2 * j], neg[2 * j + 1]);
            g1_recode128(h0, u[2 * j]);
            g1_recode128(h1, u[2 * j + 1]);
            even[2 * j] = (h0[0] & 1u) == 0;
            even[2 * j + 1] = (h1[0] & 1u) == 0;
        }
        const Fp<C> beta = glv_beta<C>();
        constexpr int STEPS = 33 * 4;
        auto fetch = [&](int s) -> G1Aff<C> {
            const int rd = s / 4, t = s - 4 * rd, i = 31 - rd;
            const TabHbm<C> tab{tabs + (size_t)(t >> 1) * G1_TAB * 2 * N * stride, stride};
            G1Aff<C> q;
            if (i >= 0) q = g1_tab_digit<C>(tab, (u[t][i >> 3] >> (4 * (i & 7))) & 15u, neg[t]);
            else q = even[t] ? (neg[t] ? tab.ld(0) : g1a_neg<C>(tab.ld(0))) : g1a_inf<C>();
            return q;
        };
        G1Aff<C> qn = fetch(0);
#pragma unroll 1
        for (int s = 0; s < STEPS; s++) {
            G1Aff<C> q = qn;
            if (s + 1 < STEPS) qn = fetch(s + 1);
            if (s & 1) q.x = G1MUL<FP>(q.x, beta);      // phi (odd terms), applied after the next entry has been requested
            const int rd = s / 4;
            if (s == 4 * rd && rd >= 1 && rd <= 31) {
#pragma unroll 1
                for (int t = 0; t < 4; t++) r = g1j_dbl<C>(r);
            }
            r = g1j_add_aff<C>(r, q);
        }
    } else {
        uint32_t u[2][8];
        g1_recode(k0, u[0]); g1_recode(k1, u[1]);
        const bool even[2] = {(k0[0] & 1u) == 0, (k1[0] & 1u) == 0};
        // 65 rounds (digits 63 .. 0, then the even-scalar corrections) x 2 tables, flattened: step s = round * 2 + table
        constexpr int STEPS = 65 * 2;
        auto fetch = [&](int s) -> G1Aff<C> {
            const int rd = s >> 1, j = s & 1, i = 63 - rd;
            const TabHbm<C> tab{tabs + (size_t)j * G1_TAB * 2 * N * stride, stride};
            if (i >= 0) return g1_tab_digit<C>(tab, (u[j][i >> 3] >> (4 * (i & 7))) & 15u);
            return even[j] ? g1a_neg<C>(tab.ld(0)) : g1a_inf<C>();
        };
        G1Aff<C> qn = fetch(0);
#pragma unroll 1
        for (int s = 0; s < STEPS; s++) {
            const G1Aff<C> q = qn;
            if (s + 1 < STEPS) qn = fetch(s + 1);
            const int rd = s >> 1;
            if (s == 2 * rd && rd >= 1 && rd <= 63) {
#pragma unroll 1
                for (int t = 0; t < 4; t++) r = g1j_dbl<C>(r);
            }
            r = g1j_add_aff<C>(r, q);
        }
    }
    r.z = fe_mul<FP>(r.z, zall);
    out = r;
    return true;
}
// host self-test / reference form: the joint chain, or two separate multiplications when a table hit an exceptional case
template <class C>
BBS_HD G1Jac<C> g1_mul2_aff(const G1Aff<C>& p0, const uint32_t* k0, const G1Aff<C>& p1, const uint32_t* k1,
                             uint32_t* tabs, size_t stride, bool glv = false) {
    constexpr size_t TW = (size_t)G1_TAB * 2 * C::FpP::N;
    { TabHbm<C> t0{tabs, stride}, t1{tabs + TW * stride, stride}; t0.st(0, p0); t1.st(0, p1); }
    G1Jac<C> r;
    if constexpr (C::K::HAS_GLV) {
        if (glv && g1_mul2_tabs_fast<C, true>(k0, k1, tabs, stride, r)) return r;
    }
    if (!glv && g1_mul2_tabs_fast<C>(k0, k1, tabs, stride, r)) return r;
    const AtHbm<C> w0{tabs, stride}, w1{tabs + TW * stride, stride};
    return g1j_add<C, 1>(g1_mul_aff_tab<C, AtHbm<C>>(p0, k0, w0), g1_mul_aff_tab<C, AtHbm<C>>(p1, k1, w1));
}

// ---- comb over 64-bit pieces (round 4; proof_gen) ------------------------------------------------------------------
// A point that is multiplied by SEVERAL scalars of the same item (proof_gen: B by four, A by three; src/proof_gen.rs:254-258
// restructured over A and B) gets its doublings done ONCE: the sub-bases Q_j = 2^(64 j) P, j = 0 .. 3 (192 doublings), each with
// its table of odd multiples 1, 3, .., 15 in TRUE affine form, live in HBM ([piece][entry][2N words], `stride` apart; built by
// stage PgTables).  A scalar k = sum_j k_j 2^(64 j) then costs 60 doublings and 4 x 17 mixed additions instead of 252 and 65:
//   k P = sum_j k_j Q_j,  k_j = sum_{i<16} d_(j,i) 16^i with odd digits d = 2 U - 15 read from u_j = (k_j >> 1) | 2^63 (the
//   regular recoding of g1_recode, per piece; an even piece is corrected by one more addition of -Q_j)
// and any number of (point, scalar) terms share the one chain of 60 doublings.  The additions are the complete mixed addition
// (g1j_add_aff), so the chain is right for any table contents that are the multiples they claim to be.
constexpr int COMB_PIECES = 4;
constexpr size_t comb_table_words(int n_fp_limbs) { return (size_t)COMB_PIECES * G1_TAB * 2 * (size_t)n_fp_limbs; }
struct CombTerm {
    const uint32_t* tab;      // this point's [COMB_PIECES][G1_TAB][2N] table (word w of entry e of piece j at tab[((j * G1_TAB + e) * 2N + w) * stride])
    uint32_t u[COMB_PIECES][2];
    bool even[COMB_PIECES];
    bool neg[COMB_PIECES];    // piece j enters with a minus sign
};
BBS_HD void comb_recode_piece(uint32_t lo, uint32_t hi, bool neg, int j, CombTerm& t) {
    t.u[j][0] = (lo >> 1) | (hi << 31);
    t.u[j][1] = (hi >> 1) | 0x80000000u;
    t.even[j] = (lo & 1u) == 0;
    t.neg[j] = neg;
}
BBS_HD void comb_recode(const uint32_t* k, bool neg, const uint32_t* tab, CombTerm& t) {
    t.tab = tab;
#pragma unroll
    for (int j = 0; j < COMB_PIECES; j++) comb_recode_piece(k[2 * j], k[2 * j + 1], neg, j, t);
}
// The same comb where the curve has the GLV endomorphism and the point is known to be in the prime-order subgroup:
// k P = k1 P + k2 phi(P) with |k1|, |k2| < 2^128 (glv_split), so the four pieces are the 64-bit halves of k1 and of k2 and the
// four sub-bases P, 2^64 P, phi(P), phi(2^64 P) = 2^64 phi(P): 64 doublings of preparation per point instead of 192 (the tables
// of the phi images are the first two with x multiplied by beta).
template <class C>
BBS_HD void comb_recode_glv(const uint32_t* k, bool neg, const uint32_t* tab, CombTerm& t) {
    uint32_t h0[4], h1[4];
    bool n0 = false, n1 = false;
    glv_split<C>(k, h0, h1, n0, n1);
    t.tab = tab;
    comb_recode_piece(h0[0], h0[1], neg != n0, 0, t);
    comb_recode_piece(h0[2], h0[3], neg != n0, 1, t);
    comb_recode_piece(h1[0], h1[1], neg != n1, 2, t);
    comb_recode_piece(h1[2], h1[3], neg != n1, 3, t);
}
template <class C, int NT>
BBS_HD_NOINLINE void g1_comb_sum_to(const CombTerm* terms, size_t stride, G1Jac<C>& out) {
    constexpr int N = C::FpP::N;
    constexpr int PER_ROUND = NT * COMB_PIECES;
    constexpr int STEPS = 17 * PER_ROUND;            // rounds 0 .. 15: digits 15 .. 0; round 16: the even-piece corrections
    auto fetch = [&](int s) -> G1Aff<C> {
        const int rd = s / PER_ROUND, w = s - rd * PER_ROUND, t = w / COMB_PIECES, j = w - t * COMB_PIECES, i = 15 - rd;
        const CombTerm& T = terms[t];
        const TabHbm<C> tab{const_cast<uint32_t*>(T.tab) + (size_t)j * G1_TAB * 2 * N * stride, stride};
        if (i >= 0) return g1_tab_digit<C>(tab, (T.u[j][i >> 3] >> (4 * (i & 7))) & 15u, T.neg[j]);
        if (!T.even[j]) return g1a_inf<C>();
        const G1Aff<C> q = tab.ld(0);
        return T.neg[j] ? q : g1a_neg<C>(q);            // k_j even: (k_j + 1) Q_j was summed, take Q_j off again (with the term's sign)
    };
    G1Jac<C> r = g1j_inf<C>();
    G1Aff<C> qn = fetch(0);
#pragma unroll 1
    for (int s = 0; s < STEPS; s++) {
        const G1Aff<C> q = qn;
        if (s + 1 < STEPS) qn = fetch(s + 1);        // the next entry is requested before this addition (HBM latency hidden)
        const int rd = s / PER_ROUND;
        if (s == rd * PER_ROUND && rd >= 1 && rd <= 15) {
#pragma unroll 1
            for (int t = 0; t < 4; t++) r = g1j_dbl<C>(r);
        }
        r = g1j_add_aff<C>(r, q);
    }
    out = r;
}

template <class C>
BBS_HD G1Jac<C> g1_mul_aff_glv(const G1Aff<C>& p, const uint32_t* k) { return g1_mul_aff_glv_tab<C, AtPriv<C>>(p, k, AtPriv<C>{}); }

// k * P with the GLV split where the curve has it and the caller vouches for subgroup membership
template <class C>
BBS_HD G1Jac<C> g1_mul_aff_sel(const G1Aff<C>& p, const uint32_t* k, bool glv) {
    if constexpr (C::K::HAS_GLV) {
        if (glv) return g1_mul_aff_glv<C>(p, k);
    }
    return g1_mul_aff<C>(p, k);
}
// inlined form of g1_mul_aff_sel_hbm (result through `out`)
template <class C>
BBS_HD void g1_mul_aff_sel_hbm_inl(const G1Aff<C>& p, const uint32_t* k, bool glv, uint32_t* tab, size_t stride, G1Jac<C>& out) {
    if constexpr (C::K::HAS_GLV) {
        if (glv) { g1_mul_aff_glv_tab_inl<C, AtHbm<C>>(p, k, AtHbm<C>{tab, stride}, out); return; }
    }
    g1_mul_aff_tab_inl<C, AtHbm<C>>(p, k, AtHbm<C>{tab, stride}, out);
}
// the same with the window table in a caller-provided HBM buffer (G1_TAB * 2N words, stride apart)
template <class C>
BBS_HD G1Jac<C> g1_mul_aff_sel_hbm(const G1Aff<C>& p, const uint32_t* k, bool glv, uint32_t* tab, size_t stride) {
    if constexpr (C::K::HAS_GLV) {
        if (glv) return g1_mul_aff_glv_tab<C, AtHbm<C>>(p, k, AtHbm<C>{tab, stride});
    }
    return g1_mul_aff_tab<C, AtHbm<C>>(p, k, AtHbm<C>{tab, stride});
}

#undef FP
#undef G1MUL
#undef G1SQR
}  // namespace bbs
