// Prime-field arithmetic on 32-bit limbs, Montgomery form, one element per GPU lane.
//
// CDNA4 has no 64x64 multiplier: the native wide multiply is v_mad_u64_u32
// (32x32+64 -> 64).  Elements are therefore N little-endian 32-bit limbs
// (N = 12 for the BLS12-381 base field, 8 for BN254's base field and for both
// scalar fields) and the Montgomery radix is 2^(32 N).  All values handed
// between functions are fully reduced (in [0, p)).
//
// Every function is __host__ __device__: the very same code is compiled for
// gfx950 (product) and for x86 (tests/hosttwin, logic tests without a GPU).
//
// Replaces, for the hot path, what the reference gets from ark-ff 0.4.2
// (`Fp<MontBackend<..>>`): every `*`, `+`, `-`, `.inverse()` on field elements in
// /root/reference/src/{sign,verify,proof_gen,proof_verify}.rs.
#pragma once
#include <cstdint>
#include <hip/hip_runtime.h>

#define BBS_HD __host__ __device__ __forceinline__
#define BBS_HD_NOINLINE __host__ __device__ inline __attribute__((noinline))

namespace bbs {

template <class P>
struct Fe {
    static constexpr int N = P::N;
    uint32_t v[N];
};

template <class P>
BBS_HD Fe<P> fe_zero() {
    Fe<P> r;
#pragma unroll
    for (int i = 0; i < P::N; i++) r.v[i] = 0;
    return r;
}

template <class P>
BBS_HD Fe<P> fe_one() {   // Montgomery 1
    Fe<P> r;
#pragma unroll
    for (int i = 0; i < P::N; i++) r.v[i] = P::ONE[i];
    return r;
}

template <class P>
BBS_HD bool fe_is_zero(const Fe<P>& a) {
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < P::N; i++) acc |= a.v[i];
    return acc == 0;
}

template <class P>
BBS_HD bool fe_eq(const Fe<P>& a, const Fe<P>& b) {
    uint32_t acc = 0;
#pragma unroll
    for (int i = 0; i < P::N; i++) acc |= (a.v[i] ^ b.v[i]);
    return acc == 0;
}

template <class P>
BBS_HD Fe<P> fe_select(bool c, const Fe<P>& a, const Fe<P>& b) {   // c ? a : b
    Fe<P> r;
#pragma unroll
    for (int i = 0; i < P::N; i++) r.v[i] = c ? a.v[i] : b.v[i];
    return r;
}

// r = a - MOD if a >= MOD (a given with an extra carry bit), else a
template <class P>
BBS_HD void fe_cond_sub_mod(uint32_t* t, uint32_t carry) {
    uint32_t d[P::N];
    uint64_t bw = 0;
#pragma unroll
    for (int i = 0; i < P::N; i++) {
        uint64_t x = (uint64_t)t[i] - P::MOD[i] - bw;
        d[i] = (uint32_t)x;
        bw = (x >> 63) & 1;
    }
    // take the difference when there was a carry out of a, or no borrow
    bool take = (carry != 0) | (bw == 0);
#pragma unroll
    for (int i = 0; i < P::N; i++) t[i] = take ? d[i] : t[i];
}

template <class P>
BBS_HD Fe<P> fe_add(const Fe<P>& a, const Fe<P>& b) {
    Fe<P> r;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < P::N; i++) {
        c += (uint64_t)a.v[i] + b.v[i];
        r.v[i] = (uint32_t)c;
        c >>= 32;
    }
    fe_cond_sub_mod<P>(r.v, (uint32_t)c);
    return r;
}

template <class P>
BBS_HD Fe<P> fe_sub(const Fe<P>& a, const Fe<P>& b) {
    Fe<P> r;
    uint64_t bw = 0;
#pragma unroll
    for (int i = 0; i < P::N; i++) {
        uint64_t x = (uint64_t)a.v[i] - b.v[i] - bw;
        r.v[i] = (uint32_t)x;
        bw = (x >> 63) & 1;
    }
    uint32_t mask = (uint32_t)0 - (uint32_t)bw;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < P::N; i++) {
        c += (uint64_t)r.v[i] + (P::MOD[i] & mask);
        r.v[i] = (uint32_t)c;
        c >>= 32;
    }
    return r;
}

template <class P>
BBS_HD Fe<P> fe_neg(const Fe<P>& a) {
    return fe_sub<P>(fe_zero<P>(), a);
}

template <class P>
BBS_HD Fe<P> fe_dbl(const Fe<P>& a) {
    return fe_add<P>(a, a);
}

// Montgomery product a*b/R mod p (CIOS, 32-bit limbs, 64-bit accumulation = v_mad_u64_u32)
template <class P>
BBS_HD void fe_mul_raw(uint32_t* r, const uint32_t* a, const uint32_t* b) {
    constexpr int N = P::N;
    uint32_t t[N + 2];
#pragma unroll
    for (int i = 0; i < N + 2; i++) t[i] = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        uint64_t c = 0;
        const uint32_t bi = b[i];
#pragma unroll
        for (int j = 0; j < N; j++) {
            c += (uint64_t)a[j] * bi + t[j];
            t[j] = (uint32_t)c;
            c >>= 32;
        }
        c += t[N];
        t[N] = (uint32_t)c;
        t[N + 1] = (uint32_t)(c >> 32);
        const uint32_t m = t[0] * P::INV;
        c = (uint64_t)m * P::MOD[0] + t[0];
        c >>= 32;
#pragma unroll
        for (int j = 1; j < N; j++) {
            c += (uint64_t)m * P::MOD[j] + t[j];
            t[j - 1] = (uint32_t)c;
            c >>= 32;
        }
        c += t[N];
        t[N - 1] = (uint32_t)c;
        t[N] = t[N + 1] + (uint32_t)(c >> 32);
    }
    fe_cond_sub_mod<P>(t, t[N]);
#pragma unroll
    for (int i = 0; i < N; i++) r[i] = t[i];
}

// The multiplier is deliberately NOT inlined on the device: one unrolled 12-limb CIOS body is
// ~4-5 KB of ISA and the pairing kernel has thousands of call sites; keeping one copy keeps the
// kernel inside the instruction cache.
template <class P>
BBS_HD_NOINLINE Fe<P> fe_mul(const Fe<P>& a, const Fe<P>& b) {
    Fe<P> r;
    fe_mul_raw<P>(r.v, a.v, b.v);
    return r;
}

template <class P>
BBS_HD_NOINLINE Fe<P> fe_sqr(const Fe<P>& a) {
    Fe<P> r;
    fe_mul_raw<P>(r.v, a.v, a.v);
    return r;
}

template <class P>
BBS_HD Fe<P> fe_from_limbs(const uint32_t* limbs) {   // canonical limbs -> Montgomery
    Fe<P> a, r2;
#pragma unroll
    for (int i = 0; i < P::N; i++) { a.v[i] = limbs[i]; r2.v[i] = P::R2[i]; }
    return fe_mul<P>(a, r2);
}

template <class P>
BBS_HD Fe<P> fe_to_canonical(const Fe<P>& a) {        // Montgomery -> canonical limbs
    Fe<P> one = fe_zero<P>();
    one.v[0] = 1;
    return fe_mul<P>(a, one);
}

// canonical a < MOD ?
template <class P>
BBS_HD bool limbs_lt_mod(const uint32_t* a) {
    uint64_t bw = 0;
#pragma unroll
    for (int i = 0; i < P::N; i++) {
        uint64_t x = (uint64_t)a[i] - P::MOD[i] - bw;
        bw = (x >> 63) & 1;
    }
    return bw != 0;
}

// canonical value > (p-1)/2  ("lexicographically largest" / ark "negative" y)
template <class P>
BBS_HD bool canonical_gt_half(const Fe<P>& c) {
    uint64_t bw = 0;
#pragma unroll
    for (int i = 0; i < P::N; i++) {
        uint64_t x = (uint64_t)P::HALF[i] - c.v[i] - bw;
        bw = (x >> 63) & 1;
    }
    return bw != 0;   // HALF - c underflows  <=>  c > HALF
}

// a^(p-2) (Fermat inversion, square-and-multiply over the constant exponent); 0 -> 0
template <class P>
BBS_HD_NOINLINE Fe<P> fe_inv(const Fe<P>& a) {
    Fe<P> r = fe_one<P>();
    bool started = false;
#pragma unroll
    for (int i = P::N - 1; i >= 0; i--) {
        const uint32_t w = P::MOD_M2[i];
        for (int b = 31; b >= 0; b--) {
            if (started) r = fe_sqr<P>(r);
            if ((w >> b) & 1) {
                r = started ? fe_mul<P>(r, a) : a;
                started = true;
            }
        }
    }
    return r;
}

}  // namespace bbs
