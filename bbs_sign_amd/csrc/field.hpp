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

}  // namespace bbs
#include "fe_mul_asm_gen.hpp"
namespace bbs {

// 96-bit accumulator (lo:64, hi:32) += a*b.  On gfx950 this is exactly two instructions:
// v_mad_u64_u32 (32x32+64 -> 64, carry-out in VCC) and v_addc_co_u32 folding the carry into hi.
// Measured on MI355X (tools/ubench/valu_int.hip): every VALU instruction of this mix issues at the
// same ~2 ns per wave-instruction per SIMD, so the instruction COUNT is the cost -- product scanning
// with an explicit carry word (2 instr / MAC) replaces the CIOS form (4.3 instr / MAC as compiled).
BBS_HD void mac96(uint64_t& lo, uint32_t& hi, uint32_t a, uint32_t b) {
    const uint64_t p = (uint64_t)a * b;
    lo += p;
    hi += (lo < p) ? 1u : 0u;
}
// 96-bit accumulator += x (32-bit)
BBS_HD void acc96_add32(uint64_t& lo, uint32_t& hi, uint32_t x) {
    const uint64_t o = lo;
    lo += x;
    hi += (lo < o) ? 1u : 0u;
}

// Montgomery product a*b/R mod p: product scanning (column-wise, "FIPS"), 32-bit limbs.
//   columns 0..N-1 : acc += sum_{i+j=c} a_i b_j + sum_{i+j=c, i<c} m_i p_j ; m_c = acc_lo * INV ;
//                    acc += m_c p_0 (low word becomes 0) ; acc >>= 32
//   columns N..2N-1: acc += sum a_i b_j + sum m_i p_j ; r_{c-N} = acc_lo ; acc >>= 32
template <class P>
BBS_HD void fe_mul_raw(uint32_t* r, const uint32_t* a, const uint32_t* b) {
#if defined(__HIP_DEVICE_COMPILE__)
    // device: the generated asm-block form of exactly this algorithm (fe_mul_asm_gen.hpp)
    if constexpr (P::N == 12) { fe_mul_ps12<P>(r, a, b); return; }
    else if constexpr (P::N == 8) { fe_mul_ps8<P>(r, a, b); return; }
#endif
    constexpr int N = P::N;
    uint32_t m[N];
    uint32_t t[N + 1];
    uint64_t lo = 0;
    uint32_t hi = 0;
#pragma unroll
    for (int c = 0; c < N; c++) {
#pragma unroll
        for (int i = 0; i <= c; i++) mac96(lo, hi, a[i], b[c - i]);
#pragma unroll
        for (int i = 0; i < c; i++) mac96(lo, hi, m[i], P::MOD[c - i]);
        m[c] = (uint32_t)lo * P::INV;
        mac96(lo, hi, m[c], P::MOD[0]);
        lo = (lo >> 32) | ((uint64_t)hi << 32);
        hi = 0;
    }
#pragma unroll
    for (int c = N; c < 2 * N - 1; c++) {
#pragma unroll
        for (int i = c - N + 1; i < N; i++) mac96(lo, hi, a[i], b[c - i]);
#pragma unroll
        for (int i = c - N + 1; i < N; i++) mac96(lo, hi, m[i], P::MOD[c - i]);
        t[c - N] = (uint32_t)lo;
        lo = (lo >> 32) | ((uint64_t)hi << 32);
        hi = 0;
    }
    t[N - 1] = (uint32_t)lo;
    t[N] = (uint32_t)(lo >> 32);
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
