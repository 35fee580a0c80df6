// SHA-256, expand_message_xmd(SHA-256) and hash_to_scalar, streaming, one hash per lane.
//
// Follows /root/reference/src/utils/utilities_helper.rs:42-97 (expand_message),
// :15-40 (FromOkm: 48 bytes big-endian mod r) and src/utils/core_utilities.rs:11-21
// (hash_to_scalar).  The reference's sha2 0.10.6 crate is replaced by the FIPS 180-4 compression
// function below.
#pragma once
#include "tower.hpp"

namespace bbs {

struct Sha256 {
    uint32_t h[8];
    uint32_t w[16];     // current block, big-endian words
    uint32_t fill;      // bytes in the current block
    uint64_t total;     // total bytes absorbed
};

BBS_HD uint32_t rotr32(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

BBS_HD_NOINLINE void sha256_compress(uint32_t* h, const uint32_t* blk) {
    const uint32_t K[64] = {
        0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5,
        0xd807aa98, 0x12835b01, 0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174,
        0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc, 0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da,
        0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147, 0x06ca6351, 0x14292967,
        0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
        0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070,
        0x19a4c116, 0x1e376c08, 0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3,
        0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208, 0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
    uint32_t w[16];
#pragma unroll
    for (int i = 0; i < 16; i++) w[i] = blk[i];
    uint32_t a = h[0], b = h[1], c = h[2], d = h[3], e = h[4], f = h[5], g = h[6], hh = h[7];
#pragma unroll
    for (int i = 0; i < 64; i++) {
        if (i >= 16) {
            uint32_t w15 = w[(i + 1) & 15], w2 = w[(i + 14) & 15];
            uint32_t s0 = rotr32(w15, 7) ^ rotr32(w15, 18) ^ (w15 >> 3);
            uint32_t s1 = rotr32(w2, 17) ^ rotr32(w2, 19) ^ (w2 >> 10);
            w[i & 15] = w[i & 15] + s0 + w[(i + 9) & 15] + s1;
        }
        uint32_t S1 = rotr32(e, 6) ^ rotr32(e, 11) ^ rotr32(e, 25);
        uint32_t ch = (e & f) ^ (~e & g);
        uint32_t t1 = hh + S1 + ch + K[i] + w[i & 15];
        uint32_t S0 = rotr32(a, 2) ^ rotr32(a, 13) ^ rotr32(a, 22);
        uint32_t mj = (a & b) ^ (a & c) ^ (b & c);
        uint32_t t2 = S0 + mj;
        hh = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    h[0] += a; h[1] += b; h[2] += c; h[3] += d; h[4] += e; h[5] += f; h[6] += g; h[7] += hh;
}

BBS_HD void sha256_init(Sha256& s) {
    const uint32_t iv[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
#pragma unroll
    for (int i = 0; i < 8; i++) s.h[i] = iv[i];
#pragma unroll
    for (int i = 0; i < 16; i++) s.w[i] = 0;
    s.fill = 0;
    s.total = 0;
}

// resume from a midstate taken on a 64-byte boundary after `total` bytes
BBS_HD void sha256_init_mid(Sha256& s, const uint32_t* mid, uint64_t total) {
#pragma unroll
    for (int i = 0; i < 8; i++) s.h[i] = mid[i];
#pragma unroll
    for (int i = 0; i < 16; i++) s.w[i] = 0;
    s.fill = 0;
    s.total = total;
}

BBS_HD_NOINLINE void sha256_byte(Sha256& s, uint32_t b) {
    const uint32_t wi = s.fill >> 2, sh = (3 - (s.fill & 3)) * 8;
    // select-update instead of a dynamically indexed store keeps w[] in registers
#pragma unroll
    for (int i = 0; i < 16; i++) s.w[i] = (i == (int)wi) ? (s.w[i] | (b << sh)) : s.w[i];
    s.fill++;
    s.total++;
    if (s.fill == 64) {
        sha256_compress(s.h, s.w);
#pragma unroll
        for (int i = 0; i < 16; i++) s.w[i] = 0;
        s.fill = 0;
    }
}

// absorb one big-endian 32-bit word (fast path when the stream is word aligned)
BBS_HD_NOINLINE void sha256_word(Sha256& s, uint32_t wv) {
    if ((s.fill & 3) == 0) {
        const uint32_t wi = s.fill >> 2;
#pragma unroll
        for (int i = 0; i < 16; i++) s.w[i] = (i == (int)wi) ? wv : s.w[i];
        s.fill += 4;
        s.total += 4;
        if (s.fill == 64) {
            sha256_compress(s.h, s.w);
#pragma unroll
            for (int i = 0; i < 16; i++) s.w[i] = 0;
            s.fill = 0;
        }
    } else {
        sha256_byte(s, wv >> 24); sha256_byte(s, (wv >> 16) & 0xff);
        sha256_byte(s, (wv >> 8) & 0xff); sha256_byte(s, wv & 0xff);
    }
}

// big-endian word at p: one 32-bit load when p is aligned, four byte loads otherwise
BBS_HD uint32_t sha256_be32(const uint8_t* p) {
    if ((reinterpret_cast<uintptr_t>(p) & 3u) == 0) {
        const uint32_t l = *reinterpret_cast<const uint32_t*>(p);
        return (l << 24) | ((l & 0xff00u) << 8) | ((l >> 8) & 0xff00u) | (l >> 24);
    }
    return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | (uint32_t)p[3];
}

// Byte strings are absorbed a byte at a time only until the stream is word aligned and for the last n mod 4 bytes; in
// between whole 64-byte blocks go straight from memory into the compression function when the block buffer is empty, and
// whole words through sha256_word otherwise (round 3: one select chain per WORD instead of per byte -- the raw-message
// forms hash 32 messages per signature and were bound by this loop).
BBS_HD_NOINLINE void sha256_bytes(Sha256& s, const uint8_t* p, uint32_t n) {
    uint32_t i = 0;
    while (i < n && (s.fill & 3u)) sha256_byte(s, p[i++]);
    while (s.fill == 0 && n - i >= 64) {
        uint32_t w[16];
#pragma unroll
        for (int k = 0; k < 16; k++) w[k] = sha256_be32(p + i + 4 * k);
        sha256_compress(s.h, w);
        s.total += 64;
        i += 64;
    }
    while (n - i >= 4) { sha256_word(s, sha256_be32(p + i)); i += 4; }
    while (i < n) sha256_byte(s, p[i++]);
}

BBS_HD void sha256_u64be(Sha256& s, uint64_t v) {
    sha256_word(s, (uint32_t)(v >> 32));
    sha256_word(s, (uint32_t)v);
}

BBS_HD_NOINLINE void sha256_final(Sha256& s, uint32_t* out8) {
    const uint64_t bits = s.total * 8;
    sha256_byte(s, 0x80);                          // (compresses and clears the buffer by itself when this was byte 64)
    // the zero padding is already there: the buffer is cleared after every compression and bytes are OR-ed in
    if (s.fill > 56) {                             // no room for the length: one more block
        sha256_compress(s.h, s.w);
#pragma unroll
        for (int i = 0; i < 16; i++) s.w[i] = 0;
    }
    s.w[14] = (uint32_t)(bits >> 32);
    s.w[15] = (uint32_t)bits;
    sha256_compress(s.h, s.w);
#pragma unroll
    for (int i = 0; i < 8; i++) out8[i] = s.h[i];
}

// ---------------------------------------------------------------------------------------------
// expand_message_xmd with len_in_bytes = 48 (ell = 2), as hash_to_scalar uses it.
// Usage: xmd48_begin(s) ; absorb msg bytes into s ; xmd48_finish(s, dst, dst_len, out12)
// out12 = 48 uniform bytes as 12 big-endian words.
// ---------------------------------------------------------------------------------------------
BBS_HD void xmd48_begin(Sha256& s) {
    sha256_init(s);
    // Z_pad: 64 zero bytes == one all-zero block
    sha256_compress(s.h, s.w);
    s.total = 64;
}

BBS_HD_NOINLINE void xmd_dst_prime(Sha256& s, const uint8_t* dst, uint32_t dst_len) {
    sha256_bytes(s, dst, dst_len);
    sha256_byte(s, dst_len);
}

BBS_HD_NOINLINE void xmd48_finish(Sha256& s, const uint8_t* dst, uint32_t dst_len, uint32_t* out12) {
    // l_i_b_str = I2OSP(48, 2) || I2OSP(0, 1)
    sha256_byte(s, 0); sha256_byte(s, 48); sha256_byte(s, 0);
    xmd_dst_prime(s, dst, dst_len);
    uint32_t b0[8], b1[8], b2[8];
    sha256_final(s, b0);
    Sha256 t;
    sha256_init(t);
#pragma unroll
    for (int i = 0; i < 8; i++) sha256_word(t, b0[i]);
    sha256_byte(t, 1);
    xmd_dst_prime(t, dst, dst_len);
    sha256_final(t, b1);
    sha256_init(t);
#pragma unroll
    for (int i = 0; i < 8; i++) sha256_word(t, b0[i] ^ b1[i]);
    sha256_byte(t, 2);
    xmd_dst_prime(t, dst, dst_len);
    sha256_final(t, b2);
#pragma unroll
    for (int i = 0; i < 8; i++) out12[i] = b1[i];
#pragma unroll
    for (int i = 0; i < 4; i++) out12[8 + i] = b2[i];
}

// 48 big-endian bytes (12 BE words) -> scalar mod r, Montgomery form
template <class C>
BBS_HD Fr<C> fr_from_okm(const uint32_t* be12) {
    using P = typename C::FrP;
    // value = hi * 2^256 + lo ; hi = first 16 bytes, lo = last 32 bytes
    Fr<C> lo, hi, r2, r3;
#pragma unroll
    for (int i = 0; i < 8; i++) lo.v[i] = be12[11 - i];
#pragma unroll
    for (int i = 0; i < 8; i++) hi.v[i] = (i < 4) ? be12[3 - i] : 0u;
#pragma unroll
    for (int i = 0; i < 8; i++) { r2.v[i] = P::R2[i]; r3.v[i] = P::R3[i]; }
    // mont(lo, R^2) = lo*R ; mont(hi, R^3) = hi*R^2 = (hi*2^256)*R   (R = 2^256)
    return fe_add<P>(fe_mul<P>(lo, r2), fe_mul<P>(hi, r3));
}

// absorb a scalar (Montgomery form) as 32 big-endian bytes (sign.rs:92-116, proof_gen.rs:294-318)
template <class C>
BBS_HD void sha256_fr_be(Sha256& s, const Fr<C>& a) {
    Fr<C> c = fe_to_canonical<typename C::FrP>(a);
#pragma unroll
    for (int i = 7; i >= 0; i--) sha256_word(s, c.v[i]);
}

// absorb canonical limbs as 32 BE bytes
BBS_HD void sha256_limbs_be8(Sha256& s, const uint32_t* c) {
#pragma unroll
    for (int i = 7; i >= 0; i--) sha256_word(s, c[i]);
}

}  // namespace bbs
