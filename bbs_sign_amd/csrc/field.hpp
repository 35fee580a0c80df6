// Prime-field arithmetic for the GPU lanes, Montgomery form, two limb layouts.
//
// Measured on MI355X (tools/ubench/valu_int.hip): v_mad_u64_u32 (32x32+64 -> 64) issues at the same
// ~2 ns per wave-instruction per SIMD as a plain add, while every carry that goes through VCC/SGPR
// costs two extra wait states (hipcc pads `v_addc` chains with `s_nop 1`).  So on CDNA4 multiplies
// are cheap and carries are expensive.  The BASE fields (Fp of BLS12-381 and BN254: >99 % of the
// work) therefore use a CARRY-FREE layout:
//
//   W = 28 : N limbs of 28 bits in 32-bit words (N = 14 for BLS12-381, 10 for BN254), radix
//            R = 2^(28 N).  A product column sums up to 28 limb products (< 2^58 each) in ONE
//            64-bit accumulator with no carry instruction at all: one v_mad_u64_u32 per
//            multiply-accumulate, then `& mask` and `>> 28` per column.  Values are kept
//            "normal": limbs < 2^28 and value < BOUND * p (BOUND = 2 / 3); the only lazy form is
//            fe_add_nr (limbs < 2^29), legal only as a direct multiplier operand.
//            add/sub are one signed limb chain with a quotient estimate from the top limb.
//   W = 32 : N = 8 limbs of 32 bits, fully reduced, plain CIOS -- the scalar fields Fr (hashing
//            glue and a few dozen products per item).
//
// Every function is __host__ __device__: the same code is compiled for gfx950 (product) and for
// x86 (tests/hosttwin).  Replaces what the reference gets from ark-ff 0.4.2 (`Fp<MontBackend>`):
// every `*`, `+`, `-`, `.inverse()` in /root/reference/src/{sign,verify,proof_gen,proof_verify}.rs.
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

constexpr uint32_t MASK28 = 0x0FFFFFFFu;

// Host-only invariant checks of the lazy-reduction bounds (tests/hosttwin builds with
// -DBBS_CHECK_BOUNDS run the whole parity suite with them; never compiled for the device).
#if defined(BBS_CHECK_BOUNDS) && !defined(__HIP_DEVICE_COMPILE__)
#include <cstdio>
#include <cstdlib>
#define BBS_BOUND_ASSERT(cond, what) do { if (!(cond)) { std::fprintf(stderr, "bound violated: %s (%s:%d)\n", what, __FILE__, __LINE__); std::abort(); } } while (0)
#else
#define BBS_BOUND_ASSERT(cond, what) do { } while (0)
#endif

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
BBS_HD Fe<P> fe_select(bool c, const Fe<P>& a, const Fe<P>& b) {   // c ? a : b
    Fe<P> r;
#pragma unroll
    for (int i = 0; i < P::N; i++) r.v[i] = c ? a.v[i] : b.v[i];
    return r;
}

// =============================================================================================
// W = 32 : fully reduced, carry chains
// =============================================================================================
namespace r32 {

template <class P>
BBS_HD void cond_sub_mod(uint32_t* t, uint32_t carry) {
    uint32_t d[P::N];
    uint64_t bw = 0;
#pragma unroll
    for (int i = 0; i < P::N; i++) {
        uint64_t x = (uint64_t)t[i] - P::MOD[i] - bw;
        d[i] = (uint32_t)x;
        bw = (x >> 63) & 1;
    }
    bool take = (carry != 0) | (bw == 0);
#pragma unroll
    for (int i = 0; i < P::N; i++) t[i] = take ? d[i] : t[i];
}

template <class P>
BBS_HD Fe<P> add(const Fe<P>& a, const Fe<P>& b) {
    Fe<P> r;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < P::N; i++) {
        c += (uint64_t)a.v[i] + b.v[i];
        r.v[i] = (uint32_t)c;
        c >>= 32;
    }
    cond_sub_mod<P>(r.v, (uint32_t)c);
    return r;
}

template <class P>
BBS_HD Fe<P> sub(const Fe<P>& a, const Fe<P>& b) {
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

// CIOS Montgomery product, result fully reduced; a < 2^(32N) arbitrary, b < p
template <class P>
BBS_HD void mul(uint32_t* r, const uint32_t* a, const uint32_t* b) {
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
    cond_sub_mod<P>(t, t[N]);
#pragma unroll
    for (int i = 0; i < N; i++) r[i] = t[i];
}

}  // namespace r32

// =============================================================================================
// W = 28 : carry-free columns
// =============================================================================================
namespace r28 {

// a * b / R mod p.  Operands: limbs < 2^29 (normal or fe_add_nr), values < 2*BOUND*p.
// Column bound: 14 * 2^58 + 14 * 2^56 + carry < 2^62.  Result: normal, value < p * (1 + 2^-5).
template <class P>
BBS_HD void mul(uint32_t* r, const uint32_t* a, const uint32_t* b) {
    constexpr int N = P::N;
    uint32_t m[N];
    uint64_t acc = 0;
    for (int i = 0; i < N; i++) BBS_BOUND_ASSERT(a[i] < (1u << 29) && b[i] < (1u << 29), "mul operand limb < 2^29");
    BBS_BOUND_ASSERT(a[N - 1] <= 2 * P::MODB[N - 1] + 2 && b[N - 1] <= 2 * P::MODB[N - 1] + 2, "mul operand value < 2*BOUND*p");
#pragma unroll
    for (int c = 0; c < N; c++) {
#pragma unroll
        for (int i = 0; i <= c; i++) acc += (uint64_t)a[i] * b[c - i];
#pragma unroll
        for (int i = 0; i < c; i++) acc += (uint64_t)m[i] * P::MOD[c - i];
        m[c] = ((uint32_t)acc * P::INV) & MASK28;
        acc += (uint64_t)m[c] * P::MOD[0];
        acc >>= 28;
    }
#pragma unroll
    for (int c = N; c < 2 * N - 1; c++) {
#pragma unroll
        for (int i = c - N + 1; i < N; i++) acc += (uint64_t)a[i] * b[c - i];
#pragma unroll
        for (int i = c - N + 1; i < N; i++) acc += (uint64_t)m[i] * P::MOD[c - i];
        r[c - N] = (uint32_t)acc & MASK28;
        acc >>= 28;
    }
    r[N - 1] = (uint32_t)acc;
    BBS_BOUND_ASSERT(acc <= P::MOD2[N - 1], "mul result < 2p");
}

// a^2 / R mod p: off-diagonal products once with a doubled operand (2 a_i < 2^30).
template <class P>
BBS_HD void sqr(uint32_t* r, const uint32_t* a) {
    constexpr int N = P::N;
    uint32_t m[N], a2[N];
    for (int i = 0; i < N; i++) BBS_BOUND_ASSERT(a[i] < (1u << 29), "sqr operand limb < 2^29");
    BBS_BOUND_ASSERT(a[N - 1] <= 2 * P::MODB[N - 1] + 2, "sqr operand value < 2*BOUND*p");
#pragma unroll
    for (int i = 0; i < N; i++) a2[i] = a[i] << 1;
    uint64_t acc = 0;
#pragma unroll
    for (int c = 0; c < 2 * N - 1; c++) {
        const int lo = (c < N) ? 0 : c - N + 1;
#pragma unroll
        for (int i = lo; 2 * i < c; i++) acc += (uint64_t)a2[i] * a[c - i];
        if ((c & 1) == 0) acc += (uint64_t)a[c >> 1] * a[c >> 1];
        if (c < N) {
#pragma unroll
            for (int i = 0; i < c; i++) acc += (uint64_t)m[i] * P::MOD[c - i];
            m[c] = ((uint32_t)acc * P::INV) & MASK28;
            acc += (uint64_t)m[c] * P::MOD[0];
        } else {
#pragma unroll
            for (int i = c - N + 1; i < N; i++) acc += (uint64_t)m[i] * P::MOD[c - i];
            r[c - N] = (uint32_t)acc & MASK28;
        }
        acc >>= 28;
    }
    r[N - 1] = (uint32_t)acc;
}

// Fused Fp2 product (r0 + r1 u) = (a0 + a1 u)(b0 + b1 u) / R, u^2 = -1, all operands normal.
// Karatsuba on the UNREDUCED column sums + only two Montgomery reductions:
//   t0 = a0 b0, t1 = a1 b1, ts = (a0+a1)(b0+b1)  column by column (3 MACs per limb pair)
//   c1 = ts - t0 - t1   is non-negative in every column (it is sum a0_i b1_j + a1_i b0_j)
//   c0 = t0 - t1 + K p^2 with K p^2 stored in lifted columns (WP2) so every column is non-negative
// 5 N^2 multiply-accumulates instead of 6 N^2, no intermediate add/sub chains.  Column bounds and
// the output bound (< 2p) are asserted numerically by tools/gen_params.py.
template <class P>
BBS_HD void f2mul(uint32_t* r0, uint32_t* r1, const uint32_t* a0, const uint32_t* a1, const uint32_t* b0, const uint32_t* b1) {
    constexpr int N = P::N;
    uint32_t sa[N], sb[N], m0[N], m1[N];
#pragma unroll
    for (int i = 0; i < N; i++) {
        BBS_BOUND_ASSERT(a0[i] <= MASK28 && a1[i] <= MASK28 && b0[i] <= MASK28 && b1[i] <= MASK28, "f2mul operands normal");
        sa[i] = a0[i] + a1[i];
        sb[i] = b0[i] + b1[i];
    }
    BBS_BOUND_ASSERT(a0[N - 1] <= P::MODB[N - 1] && a1[N - 1] <= P::MODB[N - 1] && b0[N - 1] <= P::MODB[N - 1] && b1[N - 1] <= P::MODB[N - 1], "f2mul operands < BOUND*p");
    uint64_t acc0 = 0, acc1 = 0;
#pragma unroll
    for (int c = 0; c < 2 * N - 1; c++) {
        uint64_t t0 = 0, t1 = 0, ts = 0;
        const int lo = (c < N) ? 0 : c - N + 1;
        const int hi = (c < N) ? c : N - 1;
#pragma unroll
        for (int i = lo; i <= hi; i++) {
            t0 += (uint64_t)a0[i] * b0[c - i];
            t1 += (uint64_t)a1[i] * b1[c - i];
            ts += (uint64_t)sa[i] * sb[c - i];
        }
        acc0 += t0 + P::WP2[c] - t1;
        acc1 += ts - t0 - t1;
        if (c < N) {
#pragma unroll
            for (int i = 0; i < c; i++) {
                acc0 += (uint64_t)m0[i] * P::MOD[c - i];
                acc1 += (uint64_t)m1[i] * P::MOD[c - i];
            }
            m0[c] = ((uint32_t)acc0 * P::INV) & MASK28;
            m1[c] = ((uint32_t)acc1 * P::INV) & MASK28;
            acc0 += (uint64_t)m0[c] * P::MOD[0];
            acc1 += (uint64_t)m1[c] * P::MOD[0];
        } else {
#pragma unroll
            for (int i = c - N + 1; i < N; i++) {
                acc0 += (uint64_t)m0[i] * P::MOD[c - i];
                acc1 += (uint64_t)m1[i] * P::MOD[c - i];
            }
            r0[c - N] = (uint32_t)acc0 & MASK28;
            r1[c - N] = (uint32_t)acc1 & MASK28;
        }
        acc0 >>= 28;
        acc1 >>= 28;
    }
    r0[N - 1] = (uint32_t)acc0;
    r1[N - 1] = (uint32_t)acc1;
    BBS_BOUND_ASSERT(acc0 <= P::MOD2[N - 1] && acc1 <= P::MOD2[N - 1], "f2mul result < 2p");
}

// Fused Fp2 square: r0 = (a0 + a1)(a0 - a1), r1 = 2 a0 a1  (a0 - a1 taken as a0 + (BOUND p - a1),
// limb-wise non-negative through the borrow-adjusted SUBM), two column products, two reductions.
template <class P>
BBS_HD void f2sqr(uint32_t* r0, uint32_t* r1, const uint32_t* a0, const uint32_t* a1) {
    constexpr int N = P::N;
    uint32_t sa[N], da[N], a2[N], m0[N], m1[N];
#pragma unroll
    for (int i = 0; i < N; i++) {
        BBS_BOUND_ASSERT(a0[i] <= MASK28 && a1[i] <= MASK28, "f2sqr operands normal");
        sa[i] = a0[i] + a1[i];
        da[i] = a0[i] + P::SUBM[i] - a1[i];
        a2[i] = a0[i] << 1;
    }
    BBS_BOUND_ASSERT(a0[N - 1] <= P::MODB[N - 1] && a1[N - 1] <= P::MODB[N - 1], "f2sqr operands < BOUND*p");
    uint64_t acc0 = 0, acc1 = 0;
#pragma unroll
    for (int c = 0; c < 2 * N - 1; c++) {
        const int lo = (c < N) ? 0 : c - N + 1;
        const int hi = (c < N) ? c : N - 1;
#pragma unroll
        for (int i = lo; i <= hi; i++) {
            acc0 += (uint64_t)sa[i] * da[c - i];
            acc1 += (uint64_t)a2[i] * a1[c - i];
        }
        if (c < N) {
#pragma unroll
            for (int i = 0; i < c; i++) {
                acc0 += (uint64_t)m0[i] * P::MOD[c - i];
                acc1 += (uint64_t)m1[i] * P::MOD[c - i];
            }
            m0[c] = ((uint32_t)acc0 * P::INV) & MASK28;
            m1[c] = ((uint32_t)acc1 * P::INV) & MASK28;
            acc0 += (uint64_t)m0[c] * P::MOD[0];
            acc1 += (uint64_t)m1[c] * P::MOD[0];
        } else {
#pragma unroll
            for (int i = c - N + 1; i < N; i++) {
                acc0 += (uint64_t)m0[i] * P::MOD[c - i];
                acc1 += (uint64_t)m1[i] * P::MOD[c - i];
            }
            r0[c - N] = (uint32_t)acc0 & MASK28;
            r1[c - N] = (uint32_t)acc1 & MASK28;
        }
        acc0 >>= 28;
        acc1 >>= 28;
    }
    r0[N - 1] = (uint32_t)acc0;
    r1[N - 1] = (uint32_t)acc1;
    BBS_BOUND_ASSERT(acc0 <= P::MOD2[N - 1] && acc1 <= P::MOD2[N - 1], "f2sqr result < 2p");
}

// One signed limb chain:  r = x - q*p  with limbs renormalised; x given limb-wise (may be out of
// [0, 2^28) per limb), q < 8.  The caller guarantees 0 <= x - q*p < 2^(28 N).
template <class P>
BBS_HD void chain_reduce(uint32_t* r, const int32_t* x, uint32_t q) {
    constexpr int N = P::N;
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        const int32_t t = x[i] - (int32_t)(q * P::MOD[i]) + c;
        if (i < N - 1) {
            r[i] = (uint32_t)t & MASK28;
            c = t >> 28;
        } else {
            r[i] = (uint32_t)t;
            BBS_BOUND_ASSERT(t >= 0 && (uint32_t)t <= P::MODB[N - 1], "add/sub result in [0, BOUND*p)");
        }
    }
}

// a + b for normal a, b.  s = a + b < 2*BOUND*p; q = (top limbs) >> QSHIFT <= floor(s / 2^BITS)
// so s - q p >= 0, and s - q p < 2^BITS + q (2^BITS - p) + eps < BOUND * p (checked numerically
// by tools/gen_params.py for each field).
template <class P>
BBS_HD Fe<P> add(const Fe<P>& a, const Fe<P>& b) {
    constexpr int N = P::N;
    int32_t x[N];
    for (int i = 0; i < N; i++) BBS_BOUND_ASSERT(a.v[i] <= MASK28 && b.v[i] <= MASK28, "add operands normal");
    BBS_BOUND_ASSERT(a.v[N - 1] <= P::MODB[N - 1] && b.v[N - 1] <= P::MODB[N - 1], "add operands < BOUND*p");
#pragma unroll
    for (int i = 0; i < N; i++) x[i] = (int32_t)(a.v[i] + b.v[i]);
    const uint32_t q = (uint32_t)x[N - 1] >> P::QSHIFT;
    Fe<P> r;
    chain_reduce<P>(r.v, x, q);
    return r;
}

// a - b for normal a, b:  s = a - b + BOUND*p in (0, 2*BOUND*p); the limb-wise top may over-state
// floor(s / 2^(28(N-1))) by one borrow, hence (top - 1).
template <class P>
BBS_HD Fe<P> sub(const Fe<P>& a, const Fe<P>& b) {
    constexpr int N = P::N;
    int32_t x[N];
    for (int i = 0; i < N; i++) BBS_BOUND_ASSERT(a.v[i] <= MASK28 && b.v[i] <= MASK28, "sub operands normal");
    BBS_BOUND_ASSERT(a.v[N - 1] <= P::MODB[N - 1] && b.v[N - 1] <= P::MODB[N - 1], "sub operands < BOUND*p");
#pragma unroll
    for (int i = 0; i < N; i++) x[i] = (int32_t)a.v[i] - (int32_t)b.v[i] + (int32_t)P::MODB[i];
    const int32_t top = x[N - 1] - 1;
    const uint32_t q = (top > 0 ? (uint32_t)top : 0u) >> P::QSHIFT;
    Fe<P> r;
    chain_reduce<P>(r.v, x, q);
    return r;
}

// Single-chain linear combination  C0 x0 + C1 x1 + C2 x2 + C3 x3  (compile-time integer coefficients,
// total weight sum|Ci| <= 8, operands normal): the terms are combined limb-wise, NEG * BOUND * p is
// added to make the value positive (NEG = sum of the negative coefficients' magnitudes), the quotient
// floor(v / p) is estimated from the top QK limbs with a reciprocal (never too large, at most one too
// small), and ONE signed chain subtracts q p and renormalises.  Result: normal, value < p (1 + 2^-10).
// Replaces chains of add / sub / dbl (each its own limb chain) in the point and tower formulas.
template <class P, int C0, int C1, int C2, int C3>
BBS_HD Fe<P> lin(const Fe<P>& x0, const Fe<P>& x1, const Fe<P>& x2, const Fe<P>& x3) {
    constexpr int N = P::N;
    constexpr int NEG = (C0 < 0 ? -C0 : 0) + (C1 < 0 ? -C1 : 0) + (C2 < 0 ? -C2 : 0) + (C3 < 0 ? -C3 : 0);
    constexpr int POS = (C0 > 0 ? C0 : 0) + (C1 > 0 ? C1 : 0) + (C2 > 0 ? C2 : 0) + (C3 > 0 ? C3 : 0);
    static_assert(NEG + POS <= 8, "fe_lin weight");
    int32_t x[N];
#pragma unroll
    for (int i = 0; i < N; i++) {
        BBS_BOUND_ASSERT(x0.v[i] <= MASK28 && x1.v[i] <= MASK28 && x2.v[i] <= MASK28 && x3.v[i] <= MASK28, "lin operands normal");
        x[i] = C0 * (int32_t)x0.v[i] + C1 * (int32_t)x1.v[i] + C2 * (int32_t)x2.v[i] + C3 * (int32_t)x3.v[i] + NEG * (int32_t)P::MODB[i];
    }
    BBS_BOUND_ASSERT(x0.v[N - 1] <= P::MODB[N - 1] && x1.v[N - 1] <= P::MODB[N - 1] && x2.v[N - 1] <= P::MODB[N - 1] && x3.v[N - 1] <= P::MODB[N - 1], "lin operands < BOUND*p");
    // top estimate, lowered by the largest possible borrow from the limbs below
    int64_t T = x[N - 1];
    if constexpr (P::QK == 2) T = T * (int64_t)(1 << 28) + x[N - 2];
    T -= (NEG + POS + 1);
    const uint32_t q = T > 0 ? (uint32_t)(((uint64_t)T * P::RECIP) >> P::RSHIFT) : 0u;
    Fe<P> r;
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        const int64_t tt = (int64_t)x[i] - (int64_t)q * (int64_t)P::MOD[i] + c;
        if (i < N - 1) {
            r.v[i] = (uint32_t)tt & MASK28;
            c = tt >> 28;
        } else {
            r.v[i] = (uint32_t)tt;
            BBS_BOUND_ASSERT(tt >= 0 && (uint64_t)tt <= P::MOD2[N - 1], "lin result in [0, 2p)");
        }
    }
    return r;
}

// C0 x0 + C1 x1 or C0 x0 - C1 x1, the sign chosen at run time (per lane) -- one chain instead of computing both forms
// and selecting: the lift C1 * BOUND * p is always added (harmless for the sum: the quotient estimate absorbs it).
template <class P, int C0, int C1>
BBS_HD Fe<P> lin_pm(const Fe<P>& x0, const Fe<P>& x1, bool plus) {
    constexpr int N = P::N;
    static_assert(C0 > 0 && C1 > 0 && C0 + 2 * C1 <= 8, "lin_pm weight");
    int32_t x[N];
    const int32_t m = plus ? 0 : -1;
#pragma unroll
    for (int i = 0; i < N; i++) {
        BBS_BOUND_ASSERT(x0.v[i] <= MASK28 && x1.v[i] <= MASK28, "lin_pm operands normal");
        x[i] = C0 * (int32_t)x0.v[i] + C1 * ((((int32_t)x1.v[i]) ^ m) - m) + C1 * (int32_t)P::MODB[i];
    }
    BBS_BOUND_ASSERT(x0.v[N - 1] <= P::MODB[N - 1] && x1.v[N - 1] <= P::MODB[N - 1], "lin_pm operands < BOUND*p");
    int64_t T = x[N - 1];
    if constexpr (P::QK == 2) T = T * (int64_t)(1 << 28) + x[N - 2];
    T -= (C0 + 2 * C1 + 1);
    const uint32_t q = T > 0 ? (uint32_t)(((uint64_t)T * P::RECIP) >> P::RSHIFT) : 0u;
    Fe<P> r;
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        const int64_t tt = (int64_t)x[i] - (int64_t)q * (int64_t)P::MOD[i] + c;
        if (i < N - 1) {
            r.v[i] = (uint32_t)tt & MASK28;
            c = tt >> 28;
        } else {
            r.v[i] = (uint32_t)tt;
            BBS_BOUND_ASSERT(tt >= 0 && (uint64_t)tt <= P::MOD2[N - 1], "lin_pm result in [0, 2p)");
        }
    }
    return r;
}

// Reduce a lazily accumulated value given limb-wise by `limb(i)` (signed 64-bit, |limb| < 2^40, total
// value in [0, weight * BOUND * p), `weight` = number of normal terms summed, negatives already
// compensated by multiples of BOUND*p): reciprocal quotient estimate + one signed chain.
template <class P, class F>
BBS_HD Fe<P> reduce_fn(F limb, int weight) {
    constexpr int N = P::N;
    int64_t T = limb(N - 1);
    if constexpr (P::QK == 2) T = T * (int64_t)(1 << 28) + limb(N - 2);
    T -= (weight + 1);
    const uint32_t q = T > 0 ? (uint32_t)(((uint64_t)T * P::RECIP) >> P::RSHIFT) : 0u;
    Fe<P> r;
    int64_t c = 0;
#pragma unroll
    for (int i = 0; i < N; i++) {
        const int64_t tt = limb(i) - (int64_t)q * (int64_t)P::MOD[i] + c;
        if (i < N - 1) {
            r.v[i] = (uint32_t)tt & MASK28;
            c = tt >> 28;
        } else {
            r.v[i] = (uint32_t)tt;
            BBS_BOUND_ASSERT(tt >= 0 && (uint64_t)tt <= P::MOD2[N - 1], "reduce_fn result in [0, 2p)");
        }
    }
    return r;
}

// value == 0 mod p for a normal a  (a in {0, p, .., (BOUND-1) p})
template <class P>
BBS_HD bool is_zero(const Fe<P>& a) {
    uint32_t z0 = 0, z1 = 0, z2 = 0;
#pragma unroll
    for (int i = 0; i < P::N; i++) {
        z0 |= a.v[i];
        z1 |= a.v[i] ^ P::MOD[i];
        if constexpr (P::BOUND > 2) z2 |= a.v[i] ^ P::MOD2[i];
    }
    bool z = (z0 == 0) | (z1 == 0);
    if constexpr (P::BOUND > 2) z |= (z2 == 0);
    return z;
}

// normal -> the unique representative in [0, p)
template <class P>
BBS_HD Fe<P> canon(const Fe<P>& a) {
    constexpr int N = P::N;
    Fe<P> r = a;
#pragma unroll
    for (int rep = 1; rep < P::BOUND; rep++) {
        uint32_t d[N];
        int32_t c = 0;
#pragma unroll
        for (int i = 0; i < N; i++) {
            const int32_t t = (int32_t)r.v[i] - (int32_t)P::MOD[i] + c;
            d[i] = (uint32_t)t & MASK28;
            c = t >> 28;
        }
        const bool ge = c >= 0;                 // no borrow out of the top: r >= p
#pragma unroll
        for (int i = 0; i < N; i++) r.v[i] = ge ? d[i] : r.v[i];
    }
    return r;
}

// 28-bit limbs (value < 2^(32 NC)) <-> 32-bit words
template <class P>
BBS_HD void pack32(const uint32_t* l28, uint32_t* w) {
#pragma unroll
    for (int k = 0; k < P::NC; k++) {
        const int bit = 32 * k;
        const int i = bit / 28, sh = bit % 28;
        uint64_t x = (uint64_t)l28[i] >> sh;
        if (i + 1 < P::N) x |= (uint64_t)l28[i + 1] << (28 - sh);
        if (i + 2 < P::N) x |= (uint64_t)l28[i + 2] << (56 - sh);
        w[k] = (uint32_t)x;
    }
}
template <class P>
BBS_HD void unpack32(const uint32_t* w, uint32_t* l28) {
#pragma unroll
    for (int i = 0; i < P::N; i++) {
        const int bit = 28 * i;
        const int k = bit / 32, sh = bit % 32;
        uint64_t x = 0;
        if (k < P::NC) x = (uint64_t)w[k] >> sh;
        if (k + 1 < P::NC) x |= (uint64_t)w[k + 1] << (32 - sh);
        l28[i] = (uint32_t)x & MASK28;
    }
}

// ---- column accumulators: a SUM of limb products, reduced once --------------------------------
// t[0 .. 2N-2] are 64-bit column sums (weight 2^(28 c)).  One Montgomery reduction serves a whole dot product
// instead of one per product; the caller keeps the total weight of what it accumulates <= 6 (bounds asserted
// numerically by tools/gen_params.py, constant WP2X).
template <class P>
BBS_HD void cols_zero(uint64_t* t) {
#pragma unroll
    for (int c = 0; c < 2 * P::N - 1; c++) t[c] = 0;
}
template <class P>
BBS_HD void cols_mac(uint64_t* t, const uint32_t* a, const uint32_t* b) {     // a limbs < 2^30, b limbs < 2^29
#pragma unroll
    for (int i = 0; i < P::N; i++) {
#pragma unroll
        for (int j = 0; j < P::N; j++) t[i + j] += (uint64_t)a[i] * b[j];
    }
}
// t / R mod p, normal result (< 2p); t is consumed
template <class P>
BBS_HD void cols_reduce(uint32_t* r, uint64_t* t) {
    constexpr int N = P::N;
#pragma unroll
    for (int c = 0; c < N; c++) {
        const uint32_t m = ((uint32_t)t[c] * P::INV) & MASK28;
#pragma unroll
        for (int j = 0; j < N; j++) t[c + j] += (uint64_t)m * P::MOD[j];
        t[c + 1] += t[c] >> 28;
    }
    uint64_t carry = 0;
#pragma unroll
    for (int c = N; c < 2 * N - 1; c++) {
        const uint64_t v = t[c] + carry;
        r[c - N] = (uint32_t)v & MASK28;
        carry = v >> 28;
    }
    r[N - 1] = (uint32_t)carry;
    BBS_BOUND_ASSERT(carry <= P::MOD2[N - 1], "cols_reduce result < 2p");
}


// ---- the same sum of limb products accumulated in TWO passes: columns 0 .. N-1 first, then N .. 2N-2 ---------------
// Half the accumulator registers (N or N-1 columns instead of 2N-1): what lets the lane-sliced Fp12 kernels keep two
// wavefronts per SIMD.  The multiply-accumulates are the same ones, each in the pass that owns its column; the
// Montgomery quotients m[] and the carry out of column N-1 are computed after the low pass and consumed by the high one.
template <class P>
BBS_HD void cols_lo_zero(uint64_t* t) {
#pragma unroll
    for (int c = 0; c < P::N; c++) t[c] = 0;
}
template <class P>
BBS_HD void cols_hi_zero(uint64_t* t) {
#pragma unroll
    for (int c = 0; c < P::N - 1; c++) t[c] = 0;
}
template <class P>
BBS_HD void cols_mac_lo(uint64_t* t, const uint32_t* a, const uint32_t* b) {      // columns i + j < N
#pragma unroll
    for (int i = 0; i < P::N; i++) {
#pragma unroll
        for (int j = 0; j < P::N - i; j++) t[i + j] += (uint64_t)a[i] * b[j];
    }
}
template <class P>
BBS_HD void cols_mac_hi(uint64_t* t, const uint32_t* a, const uint32_t* b) {      // columns i + j >= N, stored at i + j - N
#pragma unroll
    for (int i = 1; i < P::N; i++) {
#pragma unroll
        for (int j = P::N - i; j < P::N; j++) t[i + j - P::N] += (uint64_t)a[i] * b[j];
    }
}
// low half of cols_reduce: the N quotients and the carry into column N; t (N columns) is consumed
template <class P>
BBS_HD void cols_reduce_lo(uint64_t* t, uint32_t* m, uint64_t& carry) {
    constexpr int N = P::N;
#pragma unroll
    for (int c = 0; c < N; c++) {
        m[c] = ((uint32_t)t[c] * P::INV) & MASK28;
#pragma unroll
        for (int j = 0; j < N - c; j++) t[c + j] += (uint64_t)m[c] * P::MOD[j];
        if (c + 1 < N) t[c + 1] += t[c] >> 28;
    }
    carry = t[N - 1] >> 28;
}
// high half: the quotient products that land in columns N .. 2N-2, the carry, the result limbs; t (N-1 columns) is consumed
template <class P>
BBS_HD void cols_reduce_hi(uint32_t* r, uint64_t* t, const uint32_t* m, uint64_t carry) {
    constexpr int N = P::N;
#pragma unroll
    for (int k = 0; k < N - 1; k++) {
#pragma unroll
        for (int c = k + 1; c < N; c++) t[k] += (uint64_t)m[c] * P::MOD[N + k - c];
    }
#pragma unroll
    for (int k = 0; k < N - 1; k++) {
        const uint64_t v = t[k] + carry;
        r[k] = (uint32_t)v & MASK28;
        carry = v >> 28;
    }
    r[N - 1] = (uint32_t)carry;
    BBS_BOUND_ASSERT(carry <= P::MOD2[N - 1], "cols_reduce_hi result < 2p");
}

}  // namespace r28

// =============================================================================================
// dispatch
// =============================================================================================
template <class P>
BBS_HD bool fe_is_zero(const Fe<P>& a) {
    if constexpr (P::W == 28) return r28::is_zero<P>(a);
    else {
        uint32_t acc = 0;
#pragma unroll
        for (int i = 0; i < P::N; i++) acc |= a.v[i];
        return acc == 0;
    }
}

template <class P>
BBS_HD Fe<P> fe_add(const Fe<P>& a, const Fe<P>& b) {
    if constexpr (P::W == 28) return r28::add<P>(a, b);
    else return r32::add<P>(a, b);
}
template <class P>
BBS_HD Fe<P> fe_sub(const Fe<P>& a, const Fe<P>& b) {
    if constexpr (P::W == 28) return r28::sub<P>(a, b);
    else return r32::sub<P>(a, b);
}
template <class P> BBS_HD Fe<P> fe_neg(const Fe<P>& a) { return fe_sub<P>(fe_zero<P>(), a); }
template <class P> BBS_HD Fe<P> fe_dbl(const Fe<P>& a) { return fe_add<P>(a, a); }

// C0 x0 + C1 x1 (+ C2 x2 + C3 x3) in one reduction chain (W = 28) or by repeated add / sub (W = 32)
template <class P, int C0, int C1, int C2 = 0, int C3 = 0>
BBS_HD Fe<P> fe_lin(const Fe<P>& x0, const Fe<P>& x1, const Fe<P>& x2, const Fe<P>& x3) {
    if constexpr (P::W == 28) return r28::lin<P, C0, C1, C2, C3>(x0, x1, x2, x3);
    else {
        Fe<P> acc = fe_zero<P>();
        auto term = [&](int cf, const Fe<P>& x) {
            for (int k = 0; k < (cf < 0 ? -cf : cf); k++) acc = cf < 0 ? fe_sub<P>(acc, x) : fe_add<P>(acc, x);
        };
        term(C0, x0); term(C1, x1); term(C2, x2); term(C3, x3);
        return acc;
    }
}
// C0 x0 + C1 x1 (plus) or C0 x0 - C1 x1, chosen at run time
template <class P, int C0, int C1>
BBS_HD Fe<P> fe_lin_pm(const Fe<P>& x0, const Fe<P>& x1, bool plus) {
    if constexpr (P::W == 28) return r28::lin_pm<P, C0, C1>(x0, x1, plus);
    else return fe_select<P>(plus, fe_lin<P, C0, C1, 0, 0>(x0, x1, x1, x1), fe_lin<P, C0, -C1, 0, 0>(x0, x1, x1, x1));
}
template <class P, int C0, int C1, int C2 = 0>
BBS_HD Fe<P> fe_lin(const Fe<P>& x0, const Fe<P>& x1, const Fe<P>& x2) { return fe_lin<P, C0, C1, C2, 0>(x0, x1, x2, x2); }
template <class P, int C0, int C1>
BBS_HD Fe<P> fe_lin(const Fe<P>& x0, const Fe<P>& x1) { return fe_lin<P, C0, C1, 0, 0>(x0, x1, x1, x1); }
template <class P, int C0>
BBS_HD Fe<P> fe_scale(const Fe<P>& x0) { return fe_lin<P, C0, 0, 0, 0>(x0, x0, x0, x0); }

// lazy sum, limbs < 2^29: ONLY as an operand of fe_mul / fe_sqr (W = 28); plain add otherwise
template <class P>
BBS_HD Fe<P> fe_add_nr(const Fe<P>& a, const Fe<P>& b) {
    if constexpr (P::W == 28) {
        Fe<P> r;
#pragma unroll
        for (int i = 0; i < P::N; i++) r.v[i] = a.v[i] + b.v[i];
        return r;
    } else return r32::add<P>(a, b);
}

template <class P>
BBS_HD bool fe_eq(const Fe<P>& a, const Fe<P>& b) {
    if constexpr (P::W == 28) return r28::is_zero<P>(r28::sub<P>(a, b));
    else {
        uint32_t acc = 0;
#pragma unroll
        for (int i = 0; i < P::N; i++) acc |= (a.v[i] ^ b.v[i]);
        return acc == 0;
    }
}

// The multiplier is deliberately NOT inlined on the device: one unrolled body is 2-4 KB of ISA and
// the pairing kernel has thousands of call sites; one copy keeps the kernels in the instruction
// cache.
template <class P>
BBS_HD_NOINLINE Fe<P> fe_mul(const Fe<P> a, const Fe<P> b) {   // by value: operands travel in VGPRs, not through scratch
    Fe<P> r;
    if constexpr (P::W == 28) r28::mul<P>(r.v, a.v, b.v);
    else r32::mul<P>(r.v, a.v, b.v);
    return r;
}
// inlined forms (the caller keeps everything in VGPRs; use where the code-size cost is acceptable)
template <class P>
BBS_HD Fe<P> fe_mul_i(const Fe<P>& a, const Fe<P>& b) {
    Fe<P> r;
    if constexpr (P::W == 28) r28::mul<P>(r.v, a.v, b.v);
    else r32::mul<P>(r.v, a.v, b.v);
    return r;
}
template <class P>
BBS_HD Fe<P> fe_sqr_i(const Fe<P>& a) {
    Fe<P> r;
    if constexpr (P::W == 28) r28::sqr<P>(r.v, a.v);
    else r32::mul<P>(r.v, a.v, a.v);
    return r;
}

template <class P>
BBS_HD_NOINLINE Fe<P> fe_sqr(const Fe<P> a) {
    Fe<P> r;
    if constexpr (P::W == 28) r28::sqr<P>(r.v, a.v);
    else r32::mul<P>(r.v, a.v, a.v);
    return r;
}

// ---- canonical 32-bit words (ABI / hashing side) <-> field elements ------------------------
// words: NC little-endian 32-bit words of the canonical value in [0, p)
template <class P>
BBS_HD Fe<P> fe_from_words(const uint32_t* w) {      // canonical (or any value < 2^(32 NC)) -> Montgomery
    Fe<P> a, r2;
    if constexpr (P::W == 28) r28::unpack32<P>(w, a.v);
    else {
#pragma unroll
        for (int i = 0; i < P::N; i++) a.v[i] = w[i];
    }
#pragma unroll
    for (int i = 0; i < P::N; i++) r2.v[i] = P::R2[i];
    return fe_mul<P>(a, r2);
}
template <class P>
BBS_HD void fe_to_words(const Fe<P>& a, uint32_t* w) {   // Montgomery -> canonical words
    Fe<P> one = fe_zero<P>();
    one.v[0] = 1;
    Fe<P> c = fe_mul<P>(a, one);
    if constexpr (P::W == 28) {
        c = r28::canon<P>(c);
        r28::pack32<P>(c.v, w);
    } else {
#pragma unroll
        for (int i = 0; i < P::N; i++) w[i] = c.v[i];
    }
}

// W = 32 helpers used by the scalar-field glue: canonical limbs live in an Fe<P>
template <class P>
BBS_HD Fe<P> fe_from_limbs(const uint32_t* limbs) {
    static_assert(P::W == 32, "scalar-field helper");
    return fe_from_words<P>(limbs);
}
template <class P>
BBS_HD Fe<P> fe_to_canonical(const Fe<P>& a) {
    static_assert(P::W == 32, "scalar-field helper");
    Fe<P> r;
    fe_to_words<P>(a, r.v);
    return r;
}

// canonical words w < p ?
template <class P>
BBS_HD bool limbs_lt_mod(const uint32_t* w) {
    uint64_t bw = 0;
#pragma unroll
    for (int i = 0; i < P::NC; i++) {
        uint64_t x = (uint64_t)w[i] - P::MODC[i] - bw;
        bw = (x >> 63) & 1;
    }
    return bw != 0;
}

// canonical words > (p-1)/2  ("lexicographically largest" / ark "negative" y)
template <class P>
BBS_HD bool words_gt_half(const uint32_t* w) {
    uint64_t bw = 0;
#pragma unroll
    for (int i = 0; i < P::NC; i++) {
        uint64_t x = (uint64_t)P::HALF[i] - w[i] - bw;
        bw = (x >> 63) & 1;
    }
    return bw != 0;
}

// ---- modular inversion: Bernstein-Yang "safegcd" divsteps, 30 at a time on 30-bit signed limbs ------------------
// (the formulation of libsecp256k1's modinv32, restated).  Branch-free, so the 64 lanes of a wavefront stay in
// step; ~600 instructions per batch of 30 divsteps, 37 batches for the 381-bit field: ~20 k instructions where the
// Fermat power a^(p-2) took ~290 k (it was 13 % of the pairing kernel and half of the compression stages).
// x: canonical integer in NC 32-bit words, replaced by x^-1 mod p (0 -> 0).
template <class P>
BBS_HD void modinv30(uint32_t* x) {
    constexpr int NL = P::NL30;
    constexpr int32_t M30 = 0x3FFFFFFF;
    int32_t f[NL], g[NL], d[NL], e[NL];
#pragma unroll
    for (int i = 0; i < NL; i++) {
        const int bit = 30 * i, wi = bit >> 5, sh = bit & 31;
        uint64_t two = wi < P::NC ? x[wi] : 0u;
        if (wi + 1 < P::NC) two |= (uint64_t)x[wi + 1] << 32;
        f[i] = (int32_t)P::MOD30[i];
        g[i] = (int32_t)((two >> sh) & (uint32_t)M30);
        d[i] = 0;
        e[i] = i == 0 ? 1 : 0;
    }
    int32_t zeta = -1;                                    // -(delta + 1/2)
#pragma unroll 1
    for (int it = 0; it < P::INV_BATCHES; it++) {
        // 30 divsteps on the low limbs -> transition matrix (u v; q r), scaled by 2^30
        uint32_t u = 1, v = 0, q = 0, r = 1, ff = (uint32_t)f[0], gg = (uint32_t)g[0];
#pragma unroll
        for (int i = 0; i < 30; i++) {
            uint32_t c1 = (uint32_t)(zeta >> 31);
            const uint32_t c2 = 0u - (gg & 1u);
            const uint32_t xx = (ff ^ c1) - c1, yy = (u ^ c1) - c1, zz = (v ^ c1) - c1;
            gg += xx & c2; q += yy & c2; r += zz & c2;
            c1 &= c2;
            zeta = (int32_t)((uint32_t)zeta ^ c1) - 1;
            ff += gg & c1; u += q & c1; v += r & c1;
            gg >>= 1; u <<= 1; v <<= 1;
        }
        const int32_t tu = (int32_t)u, tv = (int32_t)v, tq = (int32_t)q, tr = (int32_t)r;
        // (d, e) <- (u d + v e, q d + r e) / 2^30 mod p
        {
            const int32_t sd = d[NL - 1] >> 31, se = e[NL - 1] >> 31;
            int32_t md = (tu & sd) + (tv & se), me = (tq & sd) + (tr & se);
            int64_t cd = (int64_t)tu * d[0] + (int64_t)tv * e[0];
            int64_t ce = (int64_t)tq * d[0] + (int64_t)tr * e[0];
            md -= (int32_t)((P::MINV30 * (uint32_t)cd + (uint32_t)md) & (uint32_t)M30);
            me -= (int32_t)((P::MINV30 * (uint32_t)ce + (uint32_t)me) & (uint32_t)M30);
            cd += (int64_t)(int32_t)P::MOD30[0] * md;
            ce += (int64_t)(int32_t)P::MOD30[0] * me;
            cd >>= 30; ce >>= 30;
#pragma unroll
            for (int i = 1; i < NL; i++) {
                cd += (int64_t)tu * d[i] + (int64_t)tv * e[i] + (int64_t)(int32_t)P::MOD30[i] * md;
                ce += (int64_t)tq * d[i] + (int64_t)tr * e[i] + (int64_t)(int32_t)P::MOD30[i] * me;
                d[i - 1] = (int32_t)cd & M30; cd >>= 30;
                e[i - 1] = (int32_t)ce & M30; ce >>= 30;
            }
            d[NL - 1] = (int32_t)cd;
            e[NL - 1] = (int32_t)ce;
        }
        // (f, g) <- (u f + v g, q f + r g) / 2^30 (exact)
        {
            int64_t cf = (int64_t)tu * f[0] + (int64_t)tv * g[0];
            int64_t cg = (int64_t)tq * f[0] + (int64_t)tr * g[0];
            cf >>= 30; cg >>= 30;
#pragma unroll
            for (int i = 1; i < NL; i++) {
                cf += (int64_t)tu * f[i] + (int64_t)tv * g[i];
                cg += (int64_t)tq * f[i] + (int64_t)tr * g[i];
                f[i - 1] = (int32_t)cf & M30; cf >>= 30;
                g[i - 1] = (int32_t)cg & M30; cg >>= 30;
            }
            f[NL - 1] = (int32_t)cf;
            g[NL - 1] = (int32_t)cg;
        }
    }
#ifdef BBS_CHECK_BOUNDS
    {
        int32_t any = 0;
        for (int i = 0; i < NL; i++) any |= g[i];
        BBS_BOUND_ASSERT(any == 0, "modinv30: g == 0 after the last batch");
    }
#endif
    // result = d * sign(f), brought to [0, p)
    {
        int32_t cond_add = d[NL - 1] >> 31;
        const int32_t cond_neg = f[NL - 1] >> 31;
#pragma unroll
        for (int i = 0; i < NL; i++) {
            d[i] += (int32_t)P::MOD30[i] & cond_add;
            d[i] = (d[i] ^ cond_neg) - cond_neg;
        }
#pragma unroll
        for (int i = 0; i < NL - 1; i++) { d[i + 1] += d[i] >> 30; d[i] &= M30; }
        cond_add = d[NL - 1] >> 31;
#pragma unroll
        for (int i = 0; i < NL; i++) d[i] += (int32_t)P::MOD30[i] & cond_add;
#pragma unroll
        for (int i = 0; i < NL - 1; i++) { d[i + 1] += d[i] >> 30; d[i] &= M30; }
    }
#pragma unroll
    for (int w = 0; w < P::NC; w++) {                     // 30-bit limbs -> 32-bit words
        const int bit = 32 * w, li = bit / 30, sh = bit - 30 * li;
        uint64_t two = (uint32_t)d[li];
        if (li + 1 < NL) two |= (uint64_t)(uint32_t)d[li + 1] << 30;
        if (li + 2 < NL) two |= (uint64_t)(uint32_t)d[li + 2] << 60;
        x[w] = (uint32_t)(two >> sh);
    }
}

// a^-1 (0 -> 0): canonical integer, safegcd, back to the internal form
template <class P>
BBS_HD_NOINLINE Fe<P> fe_inv(const Fe<P>& a) {
    uint32_t w[P::NC];
    fe_to_words<P>(a, w);
    modinv30<P>(w);
    return fe_from_words<P>(w);
}

// a^(p-2) (Fermat inversion, square-and-multiply over the constant exponent); 0 -> 0.  Kept as the independent
// reference the tests compare fe_inv with (bbs_selftest_inv).
template <class P>
BBS_HD_NOINLINE Fe<P> fe_inv_fermat(const Fe<P>& a) {
    Fe<P> r = fe_one<P>();
    bool started = false;
#pragma unroll
    for (int i = P::NC - 1; i >= 0; i--) {
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
