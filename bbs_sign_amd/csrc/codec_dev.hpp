// Device side of the wire codec: decompression of G1 points (square root), on-curve and prime-order-subgroup
// checks for a whole batch of octet strings.  The host codec (host_codec.hpp) does this per item on one core
// (~0.5 ms per point: a 381-bit exponentiation for the root and a scalar multiplication for the subgroup), which
// caps an ingest path at a few hundred proofs per second per core next to an engine verifying a million.
//
//   BLS12-381: 48 bytes big-endian, flags 0x80 compressed / 0x40 infinity / 0x20 "y is the larger root" in byte 0
//              (the octet form of the reference's vectors, src/tests/test_vector.rs:56-68, 163-260).
//   BN254    : ark-serialize's compressed form, 32 bytes little-endian, flags 0x80 "y negative" / 0x40 infinity in
//              the last byte (crate knowledge, unpinned).
// Subgroup membership on BLS12-381 by the endomorphism test  phi(P) == -[x^2] P,  phi(x, y) = (beta x, y)  (126
// doublings + 10 additions instead of a 255-bit multiplication; checked against [r] P on points of every kind in
// tests/test_codec.py through both code paths); BN254 has cofactor 1.
// Per point code: 0 ok, 1 ok and the identity, -40 malformed / non-canonical, -41 not on the curve / not in the subgroup.
#pragma once
#include "pippenger.hpp"

namespace bbs {

template <class C>
struct G1DecodeArgs {
    size_t n_points;
    const uint8_t* in;        // [n_points][fp_bytes] compressed octets
    uint32_t* out;            // [2NC][n_points] canonical affine words (identity = zeros)
    int8_t* code;             // [n_points]
};

// |x| * P for the (sparse, 64-bit) curve parameter, P given in Jacobian form
template <class C>
BBS_HD_NOINLINE G1Jac<C> g1j_mul_xabs(const G1Jac<C>& p) {
    const uint64_t x = C::K::X_ABS;
    int top = 63;
    while (!((x >> top) & 1)) top--;
    G1Jac<C> r = p;
    for (int i = top - 1; i >= 0; i--) {
        r = g1j_dbl<C>(r);
        if ((x >> i) & 1) r = g1j_add<C>(r, p);
    }
    return r;
}

template <class C>
BBS_HD bool g1_in_subgroup_endo(const G1Aff<C>& p) {
    using P = typename C::FpP;
    if constexpr (C::ID != 0) return true;                              // BN254: cofactor 1
    const G1Jac<C> q = g1j_mul_xabs<C>(g1j_mul_xabs<C>(G1Jac<C>{p.x, p.y, fe_one<P>()}));   // [x^2] P
    if (g1j_is_inf<C>(q)) return false;
    // (beta x_P, y_P) == -(X / Z^2, Y / Z^3)  <=>  beta x_P Z^2 == X  and  y_P Z^3 + Y == 0
    Fp<C> beta;
#pragma unroll
    for (int i = 0; i < P::N; i++) beta.v[i] = C::K::BETA_M[i];
    const Fp<C> z2 = fe_sqr<P>(q.z), z3 = fe_mul<P>(z2, q.z);
    const bool ex = fe_eq<P>(fe_mul<P>(fe_mul<P>(beta, p.x), z2), q.x);
    const bool ey = fe_is_zero<P>(fe_add<P>(fe_mul<P>(p.y, z3), q.y));
    return ex && ey;
}

// a^((p+1)/4): the square root when a is a square (p = 3 mod 4 on both curves)
template <class P>
BBS_HD_NOINLINE Fe<P> fe_sqrt_candidate(const Fe<P>& a) {
    uint32_t e[P::NC];
    uint64_t c = 1;
#pragma unroll
    for (int i = 0; i < P::NC; i++) { c += P::MODC[i]; e[i] = (uint32_t)c; c >>= 32; }
#pragma unroll
    for (int i = 0; i < P::NC; i++) e[i] = (e[i] >> 2) | ((i + 1 < P::NC ? e[i + 1] : (uint32_t)c) << 30);
    Fe<P> r = fe_one<P>();
    bool started = false;
    for (int i = P::NC - 1; i >= 0; i--)
        for (int b = 31; b >= 0; b--) {
            if (started) r = fe_sqr<P>(r);
            if ((e[i] >> b) & 1) { r = started ? fe_mul<P>(r, a) : a; started = true; }
        }
    return r;
}

template <class C>
struct G1Decode {
    static __host__ __device__ void run(const G1DecodeArgs<C>& a, size_t t) {
        using P = typename C::FpP;
        constexpr int NC = P::NC, NB = 4 * NC;
        const uint8_t* in = a.in + t * NB;
        uint32_t w[NC], zero[2 * NC];
#pragma unroll
        for (int k = 0; k < 2 * NC; k++) zero[k] = 0;
        bool inf, ybig;
        uint32_t rest = 0;                                              // OR of the value bytes (for the identity encoding)
        if constexpr (C::ID == 0) {
            const uint32_t b0 = in[0];
            if (!(b0 & 0x80u)) { a.code[t] = -40; soa_st<2 * NC>(a.out, a.n_points, t, zero); return; }
            inf = (b0 & 0x40u) != 0; ybig = (b0 & 0x20u) != 0;
#pragma unroll
            for (int k = 0; k < NC; k++) {                              // big-endian bytes -> little-endian words
                uint32_t v = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    uint32_t byte = in[NB - 1 - (4 * k + j)];
                    if (4 * k + j == NB - 1) byte &= 0x1Fu;
                    v |= byte << (8 * j);
                }
                w[k] = v; rest |= v;
            }
        } else {
            const uint32_t bl = in[NB - 1];
            inf = (bl & 0x40u) != 0; ybig = (bl & 0x80u) != 0;
#pragma unroll
            for (int k = 0; k < NC; k++) {
                uint32_t v = 0;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    uint32_t byte = in[4 * k + j];
                    if (4 * k + j == NB - 1) byte &= 0x3Fu;
                    v |= byte << (8 * j);
                }
                w[k] = v; rest |= v;
            }
        }
        if (inf) {
            a.code[t] = (rest == 0 && !ybig) ? 1 : -40;
            soa_st<2 * NC>(a.out, a.n_points, t, zero);
            return;
        }
        if (!limbs_lt_mod<P>(w)) { a.code[t] = -40; soa_st<2 * NC>(a.out, a.n_points, t, zero); return; }
        const Fe<P> x = fe_from_words<P>(w);
        const Fe<P> rhs = fe_add<P>(fe_mul<P>(fe_sqr<P>(x), x), curve_b<C>());
        Fe<P> y = fe_sqrt_candidate<P>(rhs);
        if (!fe_eq<P>(fe_sqr<P>(y), rhs)) { a.code[t] = -41; soa_st<2 * NC>(a.out, a.n_points, t, zero); return; }
        uint32_t yw[NC];
        fe_to_words<P>(y, yw);
        if (words_gt_half<P>(yw) != ybig) { y = fe_neg<P>(y); fe_to_words<P>(y, yw); }
        const G1Aff<C> pt = {x, y};
        if (!g1_in_subgroup_endo<C>(pt)) { a.code[t] = -41; soa_st<2 * NC>(a.out, a.n_points, t, zero); return; }
        uint32_t o[2 * NC];
#pragma unroll
        for (int k = 0; k < NC; k++) { o[k] = w[k]; o[NC + k] = yw[k]; }
        soa_st<2 * NC>(a.out, a.n_points, t, o);
        a.code[t] = 0;
    }
};

}  // namespace bbs
