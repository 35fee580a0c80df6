// Wire codec (host side): octet strings <-> the C ABI's canonical records, with the validation an
// ingest path needs (canonical encodings, on-curve, prime-order subgroup, non-identity where the
// draft demands it).  SURVEY.md 8(f3): the reference derives CanonicalSerialize / Deserialize for
// Signature, Proof, PublicKey (/root/reference/src/sign.rs:18, src/proof_gen.rs:29, src/key_gen.rs:12)
// but never exercises them; the byte strings it does pin are the IETF octet forms of its vectors
// (src/tests/test_vector.rs:163-260: signature = A || e, proof = Abar || Bbar || D || e^ || r1^ || r3^ ||
// m^_1.. || c, public key = compressed G2), which is what this codec reads and writes.
//   BLS12-381: 48 / 96-byte big-endian compressed points (flags 0x80 / 0x40 / 0x20 in byte 0),
//              scalars 32 bytes big-endian.
//   BN254    : ark-serialize's own compressed form (little-endian x, flags in the last byte;
//              crate knowledge, unpinned), scalars 32 bytes big-endian as the reference hashes them.
#pragma once
#include "host_h2c.hpp"

namespace bbs {
namespace codec {

template <class P>
inline Fe<P> pow_words(const Fe<P>& a, const uint32_t* e, int nw) {
    Fe<P> r = fe_one<P>();
    for (int i = nw - 1; i >= 0; i--)
        for (int b = 31; b >= 0; b--) {
            r = fe_sqr<P>(r);
            if ((e[i] >> b) & 1) r = fe_mul<P>(r, a);
        }
    return r;
}

// square root in Fp, p = 3 mod 4 (both curves): a^((p+1)/4)
template <class P>
inline bool fe_sqrt(const Fe<P>& a, Fe<P>& out) {
    uint32_t e[P::NC];
    uint64_t c = 1;                                   // e = (p + 1) / 4
    for (int i = 0; i < P::NC; i++) { c += P::MODC[i]; e[i] = (uint32_t)c; c >>= 32; }
    for (int i = 0; i < P::NC; i++) e[i] = (e[i] >> 2) | ((i + 1 < P::NC ? e[i + 1] : (uint32_t)c) << 30);
    out = pow_words<P>(a, e, P::NC);
    return fe_eq<P>(fe_sqr<P>(out), a);
}

// square root in Fp2 = Fp[u]/(u^2+1) (complex method)
template <class C>
inline bool f2_sqrt(const Fp2<C>& a, Fp2<C>& out) {
    using P = typename C::FpP;
    if (f2_is_zero<C>(a)) { out = f2_zero<C>(); return true; }
    Fe<P> n = fe_add<P>(fe_sqr<P>(a.c0), fe_sqr<P>(a.c1)), s;
    if (!fe_sqrt<P>(n, s)) return false;
    uint32_t two[P::NC] = {2};
    const Fe<P> inv2 = fe_inv<P>(fe_from_words<P>(two));
    for (int k = 0; k < 2; k++) {
        const Fe<P> sg = k ? fe_neg<P>(s) : s;
        const Fe<P> t = fe_mul<P>(fe_add<P>(a.c0, sg), inv2);
        Fe<P> x0;
        if (!fe_sqrt<P>(t, x0) || fe_is_zero<P>(x0)) continue;
        const Fe<P> x1 = fe_mul<P>(a.c1, fe_inv<P>(fe_dbl<P>(x0)));
        Fp2<C> cand = {x0, x1};
        if (f2_eq<C>(f2_sqr<C>(cand), a)) { out = cand; return true; }
    }
    if (fe_is_zero<P>(a.c1)) {                         // a0 a non-residue: purely imaginary root
        Fe<P> x1;
        if (fe_sqrt<P>(fe_neg<P>(a.c0), x1)) { out = {fe_zero<P>(), x1}; return true; }
    }
    return false;
}

template <class C>
inline bool g1_in_subgroup(const G1Aff<C>& p) {
    uint32_t r[8];
    for (int i = 0; i < 8; i++) r[i] = C::FrP::MOD[i];
    return g1j_is_inf<C>(g1_mul_aff<C>(p, r));
}

template <class P>
inline bool fe_from_be_bytes(const uint8_t* b, Fe<P>& out) {
    uint8_t le[4 * P::NC];
    for (int i = 0; i < 4 * P::NC; i++) le[i] = b[4 * P::NC - 1 - i];
    return fe_from_le_bytes<P>(le, out);
}

// ---- G1 ---------------------------------------------------------------------------------------
// returns 0 ok (point may be the identity, *is_inf set), -1 malformed / not canonical, -2 not on the
// curve, -3 not in the prime-order subgroup
template <class C>
inline int g1_decompress(const uint8_t* in, G1Aff<C>& out, bool& is_inf) {
    using P = typename C::FpP;
    constexpr int NB = 4 * P::NC;
    uint8_t buf[NB];
    std::memcpy(buf, in, NB);
    bool ybig;
    is_inf = false;
    Fe<P> x;
    if (C::ID == 0) {
        if (!(buf[0] & 0x80)) return -1;                          // compressed form only
        const bool inf = buf[0] & 0x40;
        ybig = buf[0] & 0x20;
        buf[0] &= 0x1F;
        if (inf) {
            for (int i = 0; i < NB; i++) if (buf[i]) return -1;
            if (ybig) return -1;
            out = g1a_inf<C>(); is_inf = true; return 0;
        }
        if (!fe_from_be_bytes<P>(buf, x)) return -1;
    } else {
        const bool inf = buf[NB - 1] & 0x40;
        ybig = buf[NB - 1] & 0x80;
        buf[NB - 1] &= 0x3F;
        if (inf) {
            for (int i = 0; i < NB; i++) if (buf[i]) return -1;
            if (ybig) return -1;
            out = g1a_inf<C>(); is_inf = true; return 0;
        }
        if (!fe_from_le_bytes<P>(buf, x)) return -1;
    }
    Fe<P> y;
    if (!fe_sqrt<P>(fe_add<P>(fe_mul<P>(fe_sqr<P>(x), x), curve_b<C>()), y)) return -2;
    if (fe_gt_half<P>(y) != ybig) y = fe_neg<P>(y);
    out = {x, y};
    if (!g1_in_subgroup<C>(out)) return -3;
    return 0;
}

template <class C>
inline int g2_decompress(const uint8_t* in, G2Aff<C>& out) {
    using P = typename C::FpP;
    constexpr int NB = 4 * P::NC;
    uint8_t buf[2 * NB];
    std::memcpy(buf, in, 2 * NB);
    bool ybig, inf;
    Fp2<C> x;
    if (C::ID == 0) {
        if (!(buf[0] & 0x80)) return -1;
        inf = buf[0] & 0x40; ybig = buf[0] & 0x20;
        buf[0] &= 0x1F;
        if (!inf && (!fe_from_be_bytes<P>(buf, x.c1) || !fe_from_be_bytes<P>(buf + NB, x.c0))) return -1;
    } else {
        inf = buf[2 * NB - 1] & 0x40; ybig = buf[2 * NB - 1] & 0x80;
        buf[2 * NB - 1] &= 0x3F;
        if (!inf && (!fe_from_le_bytes<P>(buf, x.c0) || !fe_from_le_bytes<P>(buf + NB, x.c1))) return -1;
    }
    if (inf) {
        for (int i = 0; i < 2 * NB; i++) if (buf[i]) return -1;
        if (ybig) return -1;
        out = {f2_zero<C>(), f2_zero<C>(), true};
        return 0;
    }
    Fp2<C> y;
    if (!f2_sqrt<C>(f2_add<C>(f2_mul<C>(f2_sqr<C>(x), x), g2_b<C>()), y)) return -2;
    const bool big = fe_is_zero<P>(y.c1) ? fe_gt_half<P>(y.c0) : fe_gt_half<P>(y.c1);
    if (big != ybig) y = f2_neg<C>(y);
    out = {x, y, false};
    if (!g2_in_subgroup<C>(out)) return -3;
    return 0;
}

// scalar: 32 bytes big-endian < r  <->  32 bytes little-endian (ABI)
template <class C>
inline bool scalar_be_to_le(const uint8_t* be, uint8_t* le) {
    uint32_t w[8];
    for (int i = 0; i < 32; i++) le[i] = be[31 - i];
    for (int i = 0; i < 8; i++) w[i] = (uint32_t)le[4 * i] | ((uint32_t)le[4 * i + 1] << 8) | ((uint32_t)le[4 * i + 2] << 16) | ((uint32_t)le[4 * i + 3] << 24);
    return limbs_lt_mod<typename C::FrP>(w);
}
inline void scalar_le_to_be(const uint8_t* le, uint8_t* be) { for (int i = 0; i < 32; i++) be[i] = le[31 - i]; }

template <class C>
inline bool g1_record_to_aff(const uint8_t* rec, G1Aff<C>& p) {       // ABI record x || y (LE), zero = identity
    using P = typename C::FpP;
    constexpr int NB = 4 * P::NC;
    return fe_from_le_bytes<P>(rec, p.x) && fe_from_le_bytes<P>(rec + NB, p.y);
}
template <class C>
inline void g1_aff_to_record(const G1Aff<C>& p, uint8_t* rec) {
    using P = typename C::FpP;
    constexpr int NB = 4 * P::NC;
    if (g1a_is_inf<C>(p)) { std::memset(rec, 0, 2 * NB); return; }
    fe_to_le_bytes<P>(p.x, rec);
    fe_to_le_bytes<P>(p.y, rec + NB);
}

}  // namespace codec
}  // namespace bbs
