// Host-side, once per (ciphersuite, L): create_generators and hash-to-G1 for BLS12-381.
//
// Follows /root/reference/src/utils/interface_utilities.rs:47-73 (create_generators: chained
// expand_message + hash_to_curve) and :30-44 (HashToG1Bls12381 = zkcrypto bls12_381 @9ea427c
// `hash_to_curve` with ExpandMsgXmd<Sha256>, i.e. RFC 9380 BLS12381G1_XMD:SHA-256_SSWU_RO_).  The
// crate is not vendored; this is the RFC algorithm with the 11-isogeny evaluated by Velu's formula
// (constants: tools/gen_params.py), pinned by the reference's generator vectors
// (src/tests/test_vector.rs:66-68,123-136) in tests/test_public_api.py.
// BN254 (interface_utilities.rs:24-28, crate bn254_hash2curve 0.1.2, not vendored): RFC 9380 hash_to_curve with
// the Shallue-van de Woestijne map, Z = 1, L = 48, cofactor 1 -- pinned by the reference's one BN254 known answer,
// P1 of src/constants.rs:39-51 (tests/test_oracle_kat.py, tests/test_public_api.py).
// Per-call in the reference (sign.rs:49, verify.rs:35, proof_gen.rs:98, proof_verify.rs:40-43);
// here the result is computed once and lives in a context.
#pragma once
#include <cstring>
#include <vector>

#include "host_g2.hpp"
#include "sha256.hpp"

namespace bbs {

// expand_message_xmd(SHA-256), any output length (utilities_helper.rs:42-97); false = the reference panics
inline bool expand_message_host(const uint8_t* msg, size_t msg_len, const uint8_t* dst, size_t dst_len,
                                size_t len_in_bytes, std::vector<uint8_t>& out) {
    const size_t ell = (len_in_bytes + 31) / 32;
    if (ell > 255 || dst_len > 255) return false;
    Sha256 s;
    xmd48_begin(s);                                   // Z_pad
    for (size_t i = 0; i < msg_len; i++) sha256_byte(s, msg[i]);
    sha256_byte(s, (uint32_t)(len_in_bytes >> 8) & 0xff);
    sha256_byte(s, (uint32_t)len_in_bytes & 0xff);
    sha256_byte(s, 0);
    xmd_dst_prime(s, dst, (uint32_t)dst_len);
    uint32_t b0[8], bi[8];
    sha256_final(s, b0);
    out.clear();
    for (size_t i = 1; i <= ell; i++) {
        Sha256 t;
        sha256_init(t);
        for (int k = 0; k < 8; k++) sha256_word(t, i == 1 ? b0[k] : (b0[k] ^ bi[k]));
        sha256_byte(t, (uint32_t)i);
        xmd_dst_prime(t, dst, (uint32_t)dst_len);
        sha256_final(t, bi);
        for (int k = 0; k < 8; k++)
            for (int b = 3; b >= 0; b--) out.push_back((uint8_t)(bi[k] >> (8 * b)));
    }
    out.resize(len_in_bytes);
    return true;
}

// hash_to_scalar on the host (core_utilities.rs:11-21), canonical limbs out; false = dst too long
template <class C>
inline bool hash_to_scalar_host(const uint8_t* msg, size_t msg_len, const uint8_t* dst, size_t dst_len, uint32_t* out8) {
    if (dst_len > 255) return false;
    Sha256 s;
    xmd48_begin(s);
    for (size_t i = 0; i < msg_len; i++) sha256_byte(s, msg[i]);
    uint32_t okm[12];
    xmd48_finish(s, dst, (uint32_t)dst_len, okm);
    Fr<C> r = fe_to_canonical<typename C::FrP>(fr_from_okm<C>(okm));
    for (int i = 0; i < 8; i++) out8[i] = r.v[i];
    return true;
}

namespace h2c {

using P = BlsFpParams;
using F = Fe<P>;
using C = BlsCurve;

inline F fconst(const uint32_t* w) { F r; for (int i = 0; i < P::N; i++) r.v[i] = w[i]; return r; }

// big-endian bytes -> Fp (value reduced mod p), Horner over 16-byte chunks
inline F from_be_bytes_mod(const uint8_t* b, size_t len) {
    uint32_t w[P::NC] = {0};
    w[4] = 1;                                          // 2^128
    const F two128 = fe_from_words<P>(w);
    F acc = fe_zero<P>();
    for (size_t off = 0; off < len; off += 16) {
        const size_t n = (len - off < 16) ? len - off : 16;
        uint32_t c[P::NC] = {0};
        for (size_t k = 0; k < n; k++) {               // chunk bytes are big-endian
            const size_t bitpos = 8 * (n - 1 - k);
            c[bitpos / 32] |= (uint32_t)b[off + k] << (bitpos % 32);
        }
        F shift = two128;
        if (n < 16) { uint32_t sw[P::NC] = {0}; sw[(8 * n) / 32] = 1u << ((8 * n) % 32); shift = fe_from_words<P>(sw); }
        acc = fe_add<P>(fe_mul<P>(acc, shift), fe_from_words<P>(c));
    }
    return acc;
}

inline F pow_words(const F& a, const uint32_t* e, int nw) {
    F r = fe_one<P>();
    for (int i = nw - 1; i >= 0; i--)
        for (int b = 31; b >= 0; b--) {
            r = fe_sqr<P>(r);
            if ((e[i] >> b) & 1) r = fe_mul<P>(r, a);
        }
    return r;
}

inline bool sqrt_fp(const F& a, F& out) {              // p = 3 mod 4
    out = pow_words(a, BlsSswu::SQRT_EXP, 12);
    return fe_eq<P>(fe_sqr<P>(out), a);
}

inline uint32_t sgn0(const F& a) {
    uint32_t w[P::NC];
    fe_to_words<P>(a, w);
    return w[0] & 1u;
}

struct Pt { F x, y; bool inf; };

// simplified SWU onto E' : y^2 = x^3 + A x + B (RFC 9380 6.6.2)
inline Pt map_to_curve_sswu(const F& u) {
    const F A = fconst(BlsSswu::A_M), B = fconst(BlsSswu::B_M), Z = fconst(BlsSswu::Z_M);
    const F u2 = fe_sqr<P>(u);
    const F zu2 = fe_mul<P>(Z, u2);
    const F tv1 = fe_add<P>(fe_sqr<P>(zu2), zu2);
    F x1;
    if (fe_is_zero<P>(tv1)) {
        x1 = fe_mul<P>(B, fe_inv<P>(fe_mul<P>(Z, A)));
    } else {
        const F nba = fe_neg<P>(fe_mul<P>(B, fe_inv<P>(A)));
        x1 = fe_mul<P>(nba, fe_add<P>(fe_one<P>(), fe_inv<P>(tv1)));
    }
    auto g = [&](const F& x) { return fe_add<P>(fe_add<P>(fe_mul<P>(fe_sqr<P>(x), x), fe_mul<P>(A, x)), B); };
    F x = x1, y;
    if (!sqrt_fp(g(x1), y)) {
        x = fe_mul<P>(zu2, x1);
        const bool ok = sqrt_fp(g(x), y);
        (void)ok;
    }
    if (sgn0(u) != sgn0(y)) y = fe_neg<P>(y);
    return {x, y, false};
}

// 11-isogeny E' -> E by Velu's formula + the isomorphism onto y^2 = x^3 + 4
inline Pt iso_map(const Pt& p) {
    if (p.inf) return p;
    F X = p.x, dX = fe_one<P>();
    for (int k = 0; k < 5; k++) {
        const F d = fe_sub<P>(p.x, fconst(BlsSswu::KX_M[k]));
        if (fe_is_zero<P>(d)) return {fe_zero<P>(), fe_zero<P>(), true};      // kernel point
        const F di = fe_inv<P>(d), di2 = fe_sqr<P>(di), di3 = fe_mul<P>(di2, di);
        const F v = fconst(BlsSswu::KV_M[k]), uq = fconst(BlsSswu::KU_M[k]);
        X = fe_add<P>(X, fe_add<P>(fe_mul<P>(v, di), fe_mul<P>(uq, di2)));
        dX = fe_sub<P>(dX, fe_add<P>(fe_mul<P>(v, di2), fe_dbl<P>(fe_mul<P>(uq, di3))));
    }
    const F Y = fe_mul<P>(p.y, dX);
    return {fe_mul<P>(X, fconst(BlsSswu::S2INV_M)), fe_mul<P>(Y, fconst(BlsSswu::S3INV_M)), false};
}

inline G1Aff<C> hash_to_g1(const uint8_t* msg, size_t msg_len, const uint8_t* dst, size_t dst_len, bool& ok) {
    std::vector<uint8_t> uni;
    ok = expand_message_host(msg, msg_len, dst, dst_len, 128, uni);
    if (!ok) return g1a_inf<C>();
    const Pt q0 = iso_map(map_to_curve_sswu(from_be_bytes_mod(uni.data(), 64)));
    const Pt q1 = iso_map(map_to_curve_sswu(from_be_bytes_mod(uni.data() + 64, 64)));
    G1Jac<C> r = g1j_inf<C>();
    if (!q0.inf) r = g1j_add_aff<C>(r, G1Aff<C>{q0.x, q0.y});
    if (!q1.inf) r = g1j_add_aff<C>(r, G1Aff<C>{q1.x, q1.y});
    // clear the cofactor: h_eff = 1 - x = 0xd201000000010001
    const G1Aff<C> ra = g1j_to_aff<C>(r);
    uint32_t k[8] = {(uint32_t)BlsSswu::H_EFF, (uint32_t)(BlsSswu::H_EFF >> 32), 0, 0, 0, 0, 0, 0};
    return g1j_to_aff<C>(g1_mul_aff<C>(ra, k));
}

}  // namespace h2c

namespace h2c_bn {

using P = BnFpParams;
using F = Fe<P>;
using C = BnCurve;

inline F fconst(const uint32_t* w) { F r; for (int i = 0; i < P::N; i++) r.v[i] = w[i]; return r; }

inline F from_be_bytes_mod(const uint8_t* b, size_t len) {              // Horner over 16-byte chunks
    uint32_t w[P::NC] = {0};
    w[4] = 1;
    const F two128 = fe_from_words<P>(w);
    F acc = fe_zero<P>();
    for (size_t off = 0; off < len; off += 16) {
        uint32_t c[P::NC] = {0};
        for (size_t k = 0; k < 16; k++) {
            const size_t bitpos = 8 * (15 - k);
            c[bitpos / 32] |= (uint32_t)b[off + k] << (bitpos % 32);
        }
        acc = fe_add<P>(fe_mul<P>(acc, two128), fe_from_words<P>(c));
    }
    return acc;
}
inline F pow_words(const F& a, const uint32_t* e, int nw) {
    F r = fe_one<P>();
    for (int i = nw - 1; i >= 0; i--)
        for (int b = 31; b >= 0; b--) {
            r = fe_sqr<P>(r);
            if ((e[i] >> b) & 1) r = fe_mul<P>(r, a);
        }
    return r;
}
inline bool sqrt_fp(const F& a, F& out) {                                // p = 3 mod 4
    out = pow_words(a, BnSvdw::SQRT_EXP, 8);
    return fe_eq<P>(fe_sqr<P>(out), a);
}
inline uint32_t sgn0(const F& a) {
    uint32_t w[P::NC];
    fe_to_words<P>(a, w);
    return w[0] & 1u;
}
inline F g(const F& x) { return fe_add<P>(fe_mul<P>(fe_sqr<P>(x), x), curve_b<C>()); }

// RFC 9380 6.6.1 (straight-line Shallue-van de Woestijne)
inline G1Aff<C> map_to_curve_svdw(const F& u) {
    const F Z = fconst(BnSvdw::Z_M), c1 = fconst(BnSvdw::C1_M), c2 = fconst(BnSvdw::C2_M), c3 = fconst(BnSvdw::C3_M),
            c4 = fconst(BnSvdw::C4_M), one = fe_one<P>();
    F tv1 = fe_mul<P>(fe_sqr<P>(u), c1);
    const F tv2 = fe_add<P>(one, tv1);
    tv1 = fe_sub<P>(one, tv1);
    const F t12 = fe_mul<P>(tv1, tv2);
    const F tv3 = fe_is_zero<P>(t12) ? fe_zero<P>() : fe_inv<P>(t12);    // inv0
    const F tv4 = fe_mul<P>(fe_mul<P>(fe_mul<P>(u, tv1), tv3), c3);
    const F x1 = fe_sub<P>(c2, tv4), x2 = fe_add<P>(c2, tv4);
    F x3 = fe_mul<P>(fe_sqr<P>(tv2), tv3);
    x3 = fe_add<P>(fe_mul<P>(fe_sqr<P>(x3), c4), Z);
    F y, x = x1;
    if (!sqrt_fp(g(x1), y)) {
        x = x2;
        if (!sqrt_fp(g(x2), y)) {
            x = x3;
            const bool ok = sqrt_fp(g(x3), y);
            (void)ok;
        }
    }
    if (sgn0(u) != sgn0(y)) y = fe_neg<P>(y);
    return {x, y};
}

inline G1Aff<C> hash_to_g1(const uint8_t* msg, size_t msg_len, const uint8_t* dst, size_t dst_len, bool& ok) {
    std::vector<uint8_t> uni;
    ok = expand_message_host(msg, msg_len, dst, dst_len, 96, uni);
    if (!ok) return g1a_inf<C>();
    const G1Aff<C> q0 = map_to_curve_svdw(from_be_bytes_mod(uni.data(), 48));
    const G1Aff<C> q1 = map_to_curve_svdw(from_be_bytes_mod(uni.data() + 48, 48));
    return g1j_to_aff<C>(g1j_add_aff<C>(g1j_from_aff<C>(q0), q1));       // cofactor 1
}

}  // namespace h2c_bn

template <class C> struct H2cOf;
template <> struct H2cOf<BlsCurve> {
    static G1Aff<BlsCurve> run(const uint8_t* m, size_t ml, const uint8_t* d, size_t dl, bool& ok) { return h2c::hash_to_g1(m, ml, d, dl, ok); }
};
template <> struct H2cOf<BnCurve> {
    static G1Aff<BnCurve> run(const uint8_t* m, size_t ml, const uint8_t* d, size_t dl, bool& ok) { return h2c_bn::hash_to_g1(m, ml, d, dl, ok); }
};

// create_generators (interface_utilities.rs:47-73).  0 ok, -1 dst too long (reference panics)
template <class C>
inline int create_generators_host(size_t count, const uint8_t* api_id, size_t api_id_len, std::vector<G1Aff<C>>& out) {
    auto cat = [&](const char* suf) {
        std::vector<uint8_t> v(api_id, api_id + api_id_len);
        v.insert(v.end(), suf, suf + std::strlen(suf));
        return v;
    };
    const std::vector<uint8_t> seed_dst = cat("SIG_GENERATOR_SEED_"), gen_dst = cat("SIG_GENERATOR_DST_"),
                               gen_seed = cat("MESSAGE_GENERATOR_SEED");
    std::vector<uint8_t> v;
    if (!expand_message_host(gen_seed.data(), gen_seed.size(), seed_dst.data(), seed_dst.size(), 48, v)) return -1;
    out.clear();
    for (size_t i = 0; i < count; i++) {
        std::vector<uint8_t> m(v);
        const uint64_t idx = (uint64_t)i + 1;
        for (int b = 7; b >= 0; b--) m.push_back((uint8_t)(idx >> (8 * b)));
        if (!expand_message_host(m.data(), m.size(), seed_dst.data(), seed_dst.size(), 48, v)) return -1;
        bool ok = true;
        out.push_back(H2cOf<C>::run(v.data(), v.size(), gen_dst.data(), gen_dst.size(), ok));
        if (!ok) return -1;
    }
    return 0;
}

}  // namespace bbs
