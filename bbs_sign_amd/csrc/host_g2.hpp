// Host-side (once per issuer key, never per item) G2 arithmetic on the twist: public-key
// derivation pk = sk * BP2 (/root/reference/src/key_gen.rs:83-90), G2 compression for the domain
// hash (src/utils/core_utilities.rs:51-52) and precomputation of the Miller-loop line tables that
// the device pairing kernel consumes (pairing.hpp).  Uses the same field templates as the device.
#pragma once
#include <cstring>
#include <vector>

#include "pairing.hpp"

namespace bbs {

template <class C>
struct G2Aff {
    Fp2<C> x, y;
    bool inf;
};

template <class C>
inline Fp2<C> f2_from_consts(const uint32_t* c0, const uint32_t* c1) {
    Fp2<C> r;
    for (int i = 0; i < C::FpP::N; i++) { r.c0.v[i] = c0[i]; r.c1.v[i] = c1[i]; }
    return r;
}

template <class C>
inline G2Aff<C> g2_generator() {
    return {f2_from_consts<C>(C::K::G2X0_M, C::K::G2X1_M), f2_from_consts<C>(C::K::G2Y0_M, C::K::G2Y1_M), false};
}

template <class C>
inline Fp2<C> g2_b() { return f2_from_consts<C>(C::K::B2_C0_M, C::K::B2_C1_M); }

template <class C>
inline bool g2_on_curve(const G2Aff<C>& q) {
    if (q.inf) return true;
    Fp2<C> lhs = f2_sqr<C>(q.y);
    Fp2<C> rhs = f2_add<C>(f2_mul<C>(f2_sqr<C>(q.x), q.x), g2_b<C>());
    return f2_eq<C>(lhs, rhs);
}

template <class C>
inline G2Aff<C> g2_neg(const G2Aff<C>& q) { return {q.x, f2_neg<C>(q.y), q.inf}; }

// slope of the line through t and q (tangent when equal); false when the line is vertical
template <class C>
inline bool g2_slope(const G2Aff<C>& t, const G2Aff<C>& q, Fp2<C>& lam) {
    if (f2_eq<C>(t.x, q.x)) {
        if (!f2_eq<C>(t.y, q.y) || f2_is_zero<C>(t.y)) return false;
        Fp2<C> x2 = f2_sqr<C>(t.x);
        Fp2<C> num = f2_add<C>(f2_dbl<C>(x2), x2);
        lam = f2_mul<C>(num, f2_inv<C>(f2_dbl<C>(t.y)));
        return true;
    }
    lam = f2_mul<C>(f2_sub<C>(q.y, t.y), f2_inv<C>(f2_sub<C>(q.x, t.x)));
    return true;
}

template <class C>
inline G2Aff<C> g2_add(const G2Aff<C>& t, const G2Aff<C>& q) {
    if (t.inf) return q;
    if (q.inf) return t;
    Fp2<C> lam;
    if (!g2_slope<C>(t, q, lam)) return {f2_zero<C>(), f2_zero<C>(), true};
    Fp2<C> x3 = f2_sub<C>(f2_sub<C>(f2_sqr<C>(lam), t.x), q.x);
    Fp2<C> y3 = f2_sub<C>(f2_mul<C>(lam, f2_sub<C>(t.x, x3)), t.y);
    return {x3, y3, false};
}

// k * q, canonical 256-bit scalar limbs
template <class C>
inline G2Aff<C> g2_mul(const G2Aff<C>& q, const uint32_t* k) {
    G2Aff<C> r = {f2_zero<C>(), f2_zero<C>(), true};
    for (int i = 255; i >= 0; i--) {
        r = g2_add<C>(r, r);
        if ((k[i >> 5] >> (i & 31)) & 1) r = g2_add<C>(r, q);
    }
    return r;
}

template <class C>
inline bool g2_in_subgroup(const G2Aff<C>& q) {
    if (q.inf) return true;
    uint32_t rl[8];
    for (int i = 0; i < 8; i++) rl[i] = C::FrP::MOD[i];
    return g2_mul<C>(q, rl).inf;
}

// p-power Frobenius of the untwisted point, mapped back to the (D-type) twist
template <class C>
inline G2Aff<C> g2_frob_D(const G2Aff<C>& q) {
    Fp2<C> gx = f2_from_consts<C>(C::K::FROB[0][2][0], C::K::FROB[0][2][1]);
    Fp2<C> gy = f2_from_consts<C>(C::K::FROB[0][3][0], C::K::FROB[0][3][1]);
    return {f2_mul<C>(f2_conj<C>(q.x), gx), f2_mul<C>(f2_conj<C>(q.y), gy), q.inf};
}

// bits (MSB first, leading 1 dropped) of the Miller loop count: |x| for BLS12, 6x+2 for BN
template <class C>
inline std::vector<int> miller_bits() {
    std::vector<int> bits;
    unsigned __int128 n = C::K::X_ABS;
    if (C::ID == 1) n = 6 * n + 2;
    int top = 127;
    while (!((n >> top) & 1)) top--;
    for (int i = top - 1; i >= 0; i--) bits.push_back((int)((n >> i) & 1));
    return bits;
}

template <class C>
inline void build_schedule(MillerSchedule& s) {
    int k = 0;
    for (int b : miller_bits<C>()) {
        s.op[k++] = 0;
        s.op[k++] = 1;
        if (b) s.op[k++] = 1;
    }
    if (C::ID == 1) { s.op[k++] = 1; s.op[k++] = 1; }
    s.n_ops = k;
}

// returns false if a degenerate (vertical) line is met: Q is not a point of order r
template <class C>
inline bool build_line_table(const G2Aff<C>& Q, LineTable<C>& tab) {
    tab.n_lines = 0;
    tab.q_is_identity = Q.inf ? 1 : 0;
    if (Q.inf) return true;
    G2Aff<C> T = Q;
    auto step = [&](const G2Aff<C>& other) -> bool {
        Fp2<C> lam;
        if (T.inf || !g2_slope<C>(T, other, lam)) return false;
        LineEntry<C>& e = tab.e[tab.n_lines++];
        e.c = f2_sub<C>(f2_mul<C>(lam, T.x), T.y);
        e.nl = f2_neg<C>(lam);
        Fp2<C> x3 = f2_sub<C>(f2_sub<C>(f2_sqr<C>(lam), T.x), other.x);
        Fp2<C> y3 = f2_sub<C>(f2_mul<C>(lam, f2_sub<C>(T.x, x3)), T.y);
        T = {x3, y3, false};
        return true;
    };
    for (int b : miller_bits<C>()) {
        G2Aff<C> Tc = T;
        if (!step(Tc)) return false;
        if (b && !step(Q)) return false;
    }
    if (C::ID == 1) {
        G2Aff<C> Q1 = g2_frob_D<C>(Q);
        G2Aff<C> Q2 = g2_neg<C>(g2_frob_D<C>(Q1));
        if (!step(Q1)) return false;
        if (!step(Q2)) return false;
    }
    return true;
}

// ---- canonical byte I/O (host) -----------------------------------------------------------
// ABI field elements are little-endian canonical bytes (NB = 4*N).
template <class P>
inline bool fe_from_le_bytes(const uint8_t* b, Fe<P>& out) {
    uint32_t l[P::NC];
    for (int i = 0; i < P::NC; i++)
        l[i] = (uint32_t)b[4 * i] | ((uint32_t)b[4 * i + 1] << 8) | ((uint32_t)b[4 * i + 2] << 16) | ((uint32_t)b[4 * i + 3] << 24);
    if (!limbs_lt_mod<P>(l)) return false;
    out = fe_from_words<P>(l);
    return true;
}

template <class P>
inline void fe_to_le_bytes(const Fe<P>& a, uint8_t* b) {
    uint32_t c[P::NC];
    fe_to_words<P>(a, c);
    for (int i = 0; i < P::NC; i++) {
        b[4 * i] = (uint8_t)c[i]; b[4 * i + 1] = (uint8_t)(c[i] >> 8);
        b[4 * i + 2] = (uint8_t)(c[i] >> 16); b[4 * i + 3] = (uint8_t)(c[i] >> 24);
    }
}

template <class P>
inline void fe_to_be_bytes(const Fe<P>& a, uint8_t* b) {
    uint8_t le[4 * P::NC];
    fe_to_le_bytes<P>(a, le);
    for (int i = 0; i < 4 * P::NC; i++) b[i] = le[4 * P::NC - 1 - i];
}

template <class P>
inline bool fe_gt_half(const Fe<P>& a) {      // canonical value > (p-1)/2
    uint32_t c[P::NC];
    fe_to_words<P>(a, c);
    return words_gt_half<P>(c);
}

// ark-serialize compressed G2 (see oracle/bbs.py g2_compress for the format notes)
template <class C>
inline void g2_compress(const G2Aff<C>& q, uint8_t* out) {
    constexpr int NB = 4 * C::FpP::NC;
    using P = typename C::FpP;
    if (C::ID == 0) {
        if (q.inf) { std::memset(out, 0, 2 * NB); out[0] = 0xC0; return; }
        fe_to_be_bytes<P>(q.x.c1, out);
        fe_to_be_bytes<P>(q.x.c0, out + NB);
        out[0] |= 0x80;
        bool largest = fe_is_zero<P>(q.y.c1) ? fe_gt_half<P>(q.y.c0)
                                              : fe_gt_half<P>(q.y.c1);
        if (largest) out[0] |= 0x20;
    } else {
        if (q.inf) { std::memset(out, 0, 2 * NB); out[2 * NB - 1] = 0x40; return; }
        fe_to_le_bytes<P>(q.x.c0, out);
        fe_to_le_bytes<P>(q.x.c1, out + NB);
        bool largest = fe_is_zero<P>(q.y.c1) ? fe_gt_half<P>(q.y.c0)
                                              : fe_gt_half<P>(q.y.c1);
        if (largest) out[2 * NB - 1] |= 0x80;
    }
}

// ark-serialize compressed G1 from an affine point (host twin of the device compressor)
template <class C>
inline void g1_compress_host(const G1Aff<C>& p, uint8_t* out) {
    constexpr int NB = 4 * C::FpP::NC;
    using P = typename C::FpP;
    const bool inf = g1a_is_inf<C>(p);
    if (C::ID == 0) {
        if (inf) { std::memset(out, 0, NB); out[0] = 0xC0; return; }
        fe_to_be_bytes<P>(p.x, out);
        out[0] |= 0x80;
        if (fe_gt_half<P>(p.y)) out[0] |= 0x20;
    } else {
        if (inf) { std::memset(out, 0, NB); out[NB - 1] = 0x40; return; }
        fe_to_le_bytes<P>(p.x, out);
        if (fe_gt_half<P>(p.y)) out[NB - 1] |= 0x80;
    }
}

}  // namespace bbs
