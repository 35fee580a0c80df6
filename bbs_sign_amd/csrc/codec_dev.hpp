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

// one compressed point (fp_bytes octets at `in`, any alignment) -> canonical affine words o[2 NC] (zeros unless the
// code is 0).  Code: 0 ok, 1 ok and the identity, -40 malformed / non-canonical, -41 not on the curve / not in the subgroup.
template <class C>
__host__ __device__ inline int8_t g1_decode_octets(const uint8_t* in, uint32_t* o) {
    using P = typename C::FpP;
    constexpr int NC = P::NC, NB = 4 * NC;
    uint32_t w[NC];
#pragma unroll
    for (int k = 0; k < 2 * NC; k++) o[k] = 0;
    bool inf, ybig;
    uint32_t rest = 0;                                              // OR of the value bytes (for the identity encoding)
    if constexpr (C::ID == 0) {
        const uint32_t b0 = in[0];
        if (!(b0 & 0x80u)) return -40;
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
    if (inf) return (rest == 0 && !ybig) ? 1 : -40;
    if (!limbs_lt_mod<P>(w)) return -40;
    const Fe<P> x = fe_from_words<P>(w);
    const Fe<P> rhs = fe_add<P>(fe_mul<P>(fe_sqr<P>(x), x), curve_b<C>());
    Fe<P> y = fe_sqrt_candidate<P>(rhs);
    if (!fe_eq<P>(fe_sqr<P>(y), rhs)) return -41;
    uint32_t yw[NC];
    fe_to_words<P>(y, yw);
    if (words_gt_half<P>(yw) != ybig) { y = fe_neg<P>(y); fe_to_words<P>(y, yw); }
    const G1Aff<C> pt = {x, y};
    if (!g1_in_subgroup_endo<C>(pt)) return -41;
#pragma unroll
    for (int k = 0; k < NC; k++) { o[k] = w[k]; o[NC + k] = yw[k]; }
    return 0;
}

template <class C>
struct G1Decode {
    static __host__ __device__ void run(const G1DecodeArgs<C>& a, size_t t) {
        constexpr int NC = C::FpP::NC, NB = 4 * NC;
        uint32_t o[2 * NC];
        a.code[t] = g1_decode_octets<C>(a.in + t * NB, o);
        soa_st<2 * NC>(a.out, a.n_points, t, o);
    }
};

// ---- proof_verify from the wire: proof OCTET strings in, statuses out ---------------------------------------------
// The octet form of the reference's vectors (src/tests/test_vector.rs:199-260): compress(Abar) || compress(Bbar) ||
// compress(D) || e^ || r1^ || r3^ || m^_1 .. m^_U || c, scalars 32 bytes big-endian.  Two stages replace PvIngest:
//   PvOctDecode (lane per (item, point)): decompression (square root), on-curve and prime-order-subgroup checks
//   PvOctIngest (lane per item): shape of the octet string, the three point verdicts (identity rejected), scalars
//       big-endian -> words with range checks, THEN the reference's proof_verify_init checks, slots / mask / indexes.
// Verdict order = bbs_proof_from_octets followed by core_proof_verify: -42 shape, the first failing point's code, -40
// scalars, then -3 / -6 / -1 / -23 / -22.  The points are known to be in G1 afterwards, so the variable-base terms
// may use the GLV split without the caller vouching for anything.
template <class C>
struct PvOctArgs {
    size_t n;
    int L, dst_too_long;
    const uint8_t* oct;                   // proof octets, ragged
    const uint64_t *oct_off, *dm_off, *di_off, *hdr_off64, *ph_off64;   // n + 1 entries each, rebased to 0
    const uint32_t* dm;                   // disclosed messages, 8 words each (scalars, as core_proof_verify takes them)
    const uint64_t* di;
    uint32_t *pts, *sc, *slots, *dmask, *didx, *rcount, *hdr_off, *hdr_len, *ph_off, *ph_len;
    int8_t* pcode;                        // [3 n] per point: G1 decode code
    int8_t* status0;
    int msg_dst_too_long;                 // raw-message form: api_id || "MAP_MSG_TO_SCALAR_AS_HASH_" exceeds 255 bytes
};
// msg_to_scalars (interface_utilities.rs:76-88) for the raw-message form of the wire path: lane per disclosed message,
// scalar t = hash_to_scalar(message t, api_id || "MAP_MSG_TO_SCALAR_AS_HASH_"), canonical words [t][8]
struct MsgHashArgs {
    size_t nm;
    const uint64_t* off;                  // nm + 1 byte offsets, rebased to 0
    const uint8_t* bytes;
    uint8_t dst[256];
    uint32_t dst_len;
    uint32_t* out;
};
template <class C>
struct MsgHash {
    static __host__ __device__ void run(const MsgHashArgs& a, size_t t) {
        Sha256 s;
        xmd48_begin(s);
        sha256_bytes(s, a.bytes + a.off[t], (uint32_t)(a.off[t + 1] - a.off[t]));
        uint32_t okm[12];
        xmd48_finish(s, a.dst, a.dst_len, okm);
        const Fr<C> r = fe_to_canonical<typename C::FrP>(fr_from_okm<C>(okm));
#pragma unroll
        for (int k = 0; k < 8; k++) a.out[t * 8 + k] = r.v[k];
    }
};
template <class C>
BBS_HD bool pv_oct_shape(const PvOctArgs<C>& a, size_t i, size_t& u) {
    constexpr size_t NB = 4 * C::FpP::NC, FIXED = 3 * NB + 4 * 32;
    const uint64_t len = a.oct_off[i + 1] - a.oct_off[i];
    if (len < FIXED || (len - FIXED) % 32) return false;
    u = (size_t)((len - FIXED) / 32);
    return true;
}
template <class C>
struct PvOctDecode {
    static __host__ __device__ void run(const PvOctArgs<C>& a, size_t t) {
        constexpr int NC = C::FpP::NC;
        constexpr size_t NB = 4 * NC;
        const size_t i = t / 3;
        const int p = (int)(t - 3 * i);
        size_t u;
        uint32_t o[2 * NC];
        int8_t code = -42;
        if (pv_oct_shape<C>(a, i, u)) code = g1_decode_octets<C>(a.oct + a.oct_off[i] + (size_t)p * NB, o);
        else {
#pragma unroll
            for (int k = 0; k < 2 * NC; k++) o[k] = 0;
        }
        a.pcode[t] = code;
        soa_st<2 * NC>(a.pts + (size_t)p * 2 * NC * a.n, a.n, i, o);
    }
};
template <class C>
struct PvOctIngest {
    static __host__ __device__ void be32(const uint8_t* b, uint32_t* w) { be32_words(b, w); }
    static __host__ __device__ void run(const PvOctArgs<C>& a, size_t i) {
        using R = typename C::FrP;
        constexpr size_t NB = 4 * C::FpP::NC;
        const size_t n = a.n;
        a.hdr_off[i] = (uint32_t)a.hdr_off64[i];
        a.hdr_len[i] = (uint32_t)(a.hdr_off64[i + 1] - a.hdr_off64[i]);
        a.ph_off[i] = (uint32_t)a.ph_off64[i];
        a.ph_len[i] = (uint32_t)(a.ph_off64[i + 1] - a.ph_off64[i]);
        const int MW = ((a.L > 1 ? a.L : 1) + 31) / 32;
        for (int w = 0; w < MW; w++) a.dmask[(size_t)w * n + i] = 0;
        a.rcount[i] = 0;
        size_t u;
        if (!pv_oct_shape<C>(a, i, u)) { a.status0[i] = -42; return; }
        int8_t verdict = ST_PENDING;
        for (int p = 2; p >= 0; p--) {                   // the first failing point (lowest p) decides
            const int8_t c = a.pcode[3 * i + p];
            if (c == 1) verdict = -42;                    // octets_to_proof rejects identity points
            else if (c < 0) verdict = c;
        }
        const uint8_t* s = a.oct + a.oct_off[i] + 3 * NB;
        bool ok = true;
        uint32_t w[8];
        for (int k = 0; k < 3; k++) { be32(s + 32 * k, w); ok &= limbs_lt_mod<R>(w); soa_st<8>(a.sc + (size_t)k * 8 * n, n, i, w); }
        be32(s + 96 + 32 * u, w); ok &= limbs_lt_mod<R>(w); soa_st<8>(a.sc + (size_t)3 * 8 * n, n, i, w);
        for (size_t k = 0; k < u; k++) { be32(s + 96 + 32 * k, w); ok &= limbs_lt_mod<R>(w); }
        if (verdict == ST_PENDING && !ok) verdict = -40;
        if (verdict != ST_PENDING) { a.status0[i] = verdict; return; }
        // from here: proof_verify_init's checks (src/proof_verify.rs:139-150) exactly as PvIngest
        const uint64_t r = a.di_off[i + 1] - a.di_off[i], rm = a.dm_off[i + 1] - a.dm_off[i];
        // raw-message form: the reference hashes the disclosed messages before anything else (proof_verify.rs:43-47), and
        // expand_message panics on a DST longer than 255 bytes (utilities_helper.rs:46-52)
        if (a.msg_dst_too_long && rm > 0) { a.status0[i] = -23; return; }
        const uint64_t l = (uint64_t)u + r;
        const uint64_t* idx = a.di + a.di_off[i];
        bool bad = false;
        for (uint64_t k = 0; k < r; k++) bad |= idx[k] >= l;
        int8_t st = ST_PENDING;
        if (bad) st = -3;
        else if (rm != r) st = -6;
        else if (l != (uint64_t)a.L) st = -1;
        else if (a.dst_too_long) st = -23;
        else {
            uint64_t distinct = 0;
            for (uint64_t k = 0; k < r; k++) {
                const size_t j = (size_t)idx[k];
                uint32_t* wp = a.dmask + (j >> 5) * n + i;
                const uint32_t x = *wp, bit = 1u << (j & 31);
                if (!(x & bit)) { *wp = x | bit; distinct++; }
            }
            if (distinct != r) st = -22;
        }
        if (st != ST_PENDING) { a.status0[i] = st; return; }
        for (uint64_t k = 0; k < r; k++) {
            const size_t j = (size_t)idx[k];
            soa_ld<8>(a.dm + (a.dm_off[i] + k) * 8, 1, 0, w);
            ok &= limbs_lt_mod<R>(w);
            soa_st<8>(a.slots + j * 8 * n, n, i, w);
            a.didx[(size_t)k * n + i] = (uint32_t)j;
        }
        size_t cu = 0;
        for (size_t j = 0; j < (size_t)l; j++) {
            if ((a.dmask[(j >> 5) * n + i] >> (j & 31)) & 1u) continue;
            be32(s + 96 + 32 * cu, w);
            soa_st<8>(a.slots + j * 8 * n, n, i, w);
            cu++;
        }
        a.rcount[i] = (uint32_t)r;
        a.status0[i] = ok ? ST_PENDING : (int8_t)-40;
    }
};

// ---- verify from the wire: signature OCTET strings compress(A) || e (32 bytes big-endian) --------------------------
// VfOctDecode (lane per item): decompression, on-curve and subgroup checks of A into the SoA array VfIngest would have
// filled from a record; VfIngest (stages.hpp) then applies bbs_signature_from_octets' verdicts before core_verify's.
template <class C>
struct VfOctArgs {
    size_t n;
    const uint8_t* oct;                   // n strings of fp_bytes + 32 octets
    uint32_t* sig_a;                      // [2 NC][n] canonical words
    int8_t* pcode;
};
template <class C>
struct VfOctDecode {
    static __host__ __device__ void run(const VfOctArgs<C>& a, size_t i) {
        constexpr int NC = C::FpP::NC;
        constexpr size_t NB = 4 * NC;
        uint32_t o[2 * NC];
        a.pcode[i] = g1_decode_octets<C>(a.oct + i * (NB + 32), o);
        soa_st<2 * NC>(a.sig_a, a.n, i, o);
    }
};

}  // namespace bbs
