// Extension tower Fp2 = Fp[u]/(u^2+1), Fp6 = Fp2[v]/(v^3-xi), Fp12 = Fp6[w]/(w^2-v) for the two
// curves (xi = 1+u for BLS12-381, 9+u for BN254).  Karatsuba at every level.
//
// This is the arithmetic behind `E::pairing(..).0 * E::pairing(..).0 == E::TargetField::ONE`
// at /root/reference/src/verify.rs:88-92 and src/proof_verify.rs:112-115 (ark-ec 0.4.2 /
// ark-ff 0.4.2 Fp12 there; written from the published tower formulas here).
#pragma once
#include "field.hpp"
#include "params_gen.hpp"

namespace bbs {

struct BlsCurve {
    using FpP = BlsFpParams;
    using FrP = BlsFrParams;
    using K = BlsConsts;
    static constexpr int ID = 0;
};
struct BnCurve {
    using FpP = BnFpParams;
    using FrP = BnFrParams;
    using K = BnConsts;
    static constexpr int ID = 1;
};

template <class C> using Fp = Fe<typename C::FpP>;
template <class C> using Fr = Fe<typename C::FrP>;

#define FP typename C::FpP

// ------------------------------------------------------------------------- Fp2
template <class C>
struct Fp2 {
    Fp<C> c0, c1;
};

template <class C> BBS_HD Fp2<C> f2_zero() { return {fe_zero<FP>(), fe_zero<FP>()}; }
template <class C> BBS_HD Fp2<C> f2_one() { return {fe_one<FP>(), fe_zero<FP>()}; }
template <class C> BBS_HD bool f2_is_zero(const Fp2<C>& a) { return fe_is_zero<FP>(a.c0) & fe_is_zero<FP>(a.c1); }
template <class C> BBS_HD bool f2_eq(const Fp2<C>& a, const Fp2<C>& b) { return fe_eq<FP>(a.c0, b.c0) & fe_eq<FP>(a.c1, b.c1); }
template <class C> BBS_HD Fp2<C> f2_add(const Fp2<C>& a, const Fp2<C>& b) { return {fe_add<FP>(a.c0, b.c0), fe_add<FP>(a.c1, b.c1)}; }
template <class C> BBS_HD Fp2<C> f2_sub(const Fp2<C>& a, const Fp2<C>& b) { return {fe_sub<FP>(a.c0, b.c0), fe_sub<FP>(a.c1, b.c1)}; }
template <class C> BBS_HD Fp2<C> f2_neg(const Fp2<C>& a) { return {fe_neg<FP>(a.c0), fe_neg<FP>(a.c1)}; }
template <class C> BBS_HD Fp2<C> f2_dbl(const Fp2<C>& a) { return {fe_dbl<FP>(a.c0), fe_dbl<FP>(a.c1)}; }
template <class C> BBS_HD Fp2<C> f2_conj(const Fp2<C>& a) { return {a.c0, fe_neg<FP>(a.c1)}; }

#ifndef BBS_F2_FUSED_NOINLINE
#define BBS_F2_FUSED_ATTR BBS_HD
#else
#define BBS_F2_FUSED_ATTR BBS_HD_NOINLINE
#endif
// the fused forms: own functions (register + stack arguments) or inlined into the caller
template <class C>
BBS_F2_FUSED_ATTR Fp2<C> f2_mul_fused(const Fp2<C> a, const Fp2<C> b) {
    Fp2<C> r;
    r28::f2mul<FP>(r.c0.v, r.c1.v, a.c0.v, a.c1.v, b.c0.v, b.c1.v);
    return r;
}
template <class C>
BBS_F2_FUSED_ATTR Fp2<C> f2_sqr_fused(const Fp2<C> a) {
    Fp2<C> r;
    r28::f2sqr<FP>(r.c0.v, r.c1.v, a.c0.v, a.c1.v);
    return r;
}

// ---- Fp2 dot products with ONE pair of Montgomery reductions -----------------------------------
// acc += w * a * b for up to total weight 6 (w = 1 or 2), then finish(): Karatsuba on the unreduced column sums
//   t0 = sum a0 b0, t1 = sum a1 b1, ts = sum (a0 + a1)(b0 + b1);  re = t0 - t1 + K p^2,  im = ts - t0 - t1.
// 3 N^2 multiply-accumulates per product + 2 N^2 once, instead of 5 N^2 per product (f2_mul_fused).
// Operands normal.  Used by the lane-sliced Fp12 arithmetic (pairing_dist.hpp); host-testable (bbs_selftest_f2dot).
template <class C>
struct F2Acc {
    uint64_t t0[2 * C::FpP::N - 1], t1[2 * C::FpP::N - 1], ts[2 * C::FpP::N - 1];
#ifdef BBS_CHECK_BOUNDS
    int weight = 0;
#endif
};
template <class C>
BBS_HD void f2acc_zero(F2Acc<C>& acc) {
    r28::cols_zero<FP>(acc.t0); r28::cols_zero<FP>(acc.t1); r28::cols_zero<FP>(acc.ts);
}
template <class C, int W = 1>
BBS_HD void f2acc_mac(F2Acc<C>& acc, const Fp2<C>& a, const Fp2<C>& b) {
    constexpr int N = C::FpP::N;
    uint32_t a0[N], a1[N], sa[N], sb[N];
#pragma unroll
    for (int i = 0; i < N; i++) {
        a0[i] = a.c0.v[i] * (uint32_t)W; a1[i] = a.c1.v[i] * (uint32_t)W;      // W = 2: doubled limbs (< 2^29)
        sa[i] = a0[i] + a1[i];
        sb[i] = b.c0.v[i] + b.c1.v[i];          // exact limb-wise sums: Karatsuba needs the integers, not residues
    }
    // the ts columns may wrap modulo 2^64 (6 x 14 x 2^58 > 2^64); ts - t0 - t1 is computed modulo 2^64 as well and
    // its true value (sum a0 b1 + a1 b0 < 2^64) comes out exactly
    r28::cols_mac<FP>(acc.t0, a0, b.c0.v);
    r28::cols_mac<FP>(acc.t1, a1, b.c1.v);
    r28::cols_mac<FP>(acc.ts, sa, sb);
#ifdef BBS_CHECK_BOUNDS
    acc.weight += W;
    BBS_BOUND_ASSERT(acc.weight <= 6, "F2Acc total weight <= 6");
#endif
}
// runtime weight 2^sh (sh = 0 or 1), e.g. per-lane in the lane-sliced squaring
template <class C>
BBS_HD void f2acc_mac_sh(F2Acc<C>& acc, const Fp2<C>& a, const Fp2<C>& b, uint32_t sh) {
    constexpr int N = C::FpP::N;
    uint32_t a0[N], a1[N], sa[N], sb[N];
#pragma unroll
    for (int i = 0; i < N; i++) {
        a0[i] = a.c0.v[i] << sh; a1[i] = a.c1.v[i] << sh;
        sa[i] = a0[i] + a1[i];
        sb[i] = b.c0.v[i] + b.c1.v[i];
    }
    r28::cols_mac<FP>(acc.t0, a0, b.c0.v);
    r28::cols_mac<FP>(acc.t1, a1, b.c1.v);
    r28::cols_mac<FP>(acc.ts, sa, sb);
#ifdef BBS_CHECK_BOUNDS
    acc.weight += 1 << sh;
    BBS_BOUND_ASSERT(acc.weight <= 6, "F2Acc total weight <= 6");
#endif
}
template <class C>
BBS_HD void f2acc_mac_fp(F2Acc<C>& acc, const Fp2<C>& a, const Fp<C>& y) {       // b = y in Fp: t1 += 0
    constexpr int N = C::FpP::N;
    uint32_t sa[N];
#pragma unroll
    for (int i = 0; i < N; i++) sa[i] = a.c0.v[i] + a.c1.v[i];
    r28::cols_mac<FP>(acc.t0, a.c0.v, y.v);
    r28::cols_mac<FP>(acc.ts, sa, y.v);
#ifdef BBS_CHECK_BOUNDS
    acc.weight += 1;
    BBS_BOUND_ASSERT(acc.weight <= 6, "F2Acc total weight <= 6");
#endif
}
template <class C>
BBS_HD Fp2<C> f2acc_finish(F2Acc<C>& acc) {
    constexpr int N = C::FpP::N;
#pragma unroll
    for (int c = 0; c < 2 * N - 1; c++) {
        const uint64_t x = acc.t0[c], y = acc.t1[c];
        acc.ts[c] = acc.ts[c] - x - y;                       // sum a0 b1 + a1 b0 : non-negative column by column
        acc.t0[c] = x + C::FpP::WP2X[c] - y;
    }
    Fp2<C> r;
    r28::cols_reduce<FP>(r.c0.v, acc.t0);
    r28::cols_reduce<FP>(r.c1.v, acc.ts);
    return r;
}

// ---- the same dot products in two passes over the terms (r28::cols_*_lo / _hi): 3 x N (then 3 x (N-1)) accumulator
// columns live instead of 3 x (2N-1).  Usage: zero lo; mac_lo every term; st = finish_lo; zero hi; mac_hi every term
// AGAIN (the caller fetches the operands twice); finish_hi.  Same value as f2acc_finish, limb for limb.
template <class C>
struct F2AccLo { uint64_t t0[C::FpP::N], t1[C::FpP::N], ts[C::FpP::N]; };
template <class C>
struct F2AccHi { uint64_t t0[C::FpP::N - 1], t1[C::FpP::N - 1], ts[C::FpP::N - 1]; };
template <class C>
struct F2AccMid { uint32_t m_re[C::FpP::N], m_im[C::FpP::N]; uint64_t carry_re, carry_im; };   // between the passes

template <class C> BBS_HD void f2acc_lo_zero(F2AccLo<C>& a) { r28::cols_lo_zero<FP>(a.t0); r28::cols_lo_zero<FP>(a.t1); r28::cols_lo_zero<FP>(a.ts); }
template <class C> BBS_HD void f2acc_hi_zero(F2AccHi<C>& a) { r28::cols_hi_zero<FP>(a.t0); r28::cols_hi_zero<FP>(a.t1); r28::cols_hi_zero<FP>(a.ts); }

// HI = false: low pass, true: high pass.  sh: run-time weight 2^sh (0 or 1) on the a side.
template <class C, bool HI, class ACC>
BBS_HD void f2acc2_mac_sh(ACC& acc, const Fp2<C>& a, const Fp2<C>& b, uint32_t sh) {
    constexpr int N = C::FpP::N;
    uint32_t a0[N], a1[N];
#pragma unroll
    for (int i = 0; i < N; i++) { a0[i] = a.c0.v[i] << sh; a1[i] = a.c1.v[i] << sh; }
    if constexpr (HI) { r28::cols_mac_hi<FP>(acc.t0, a0, b.c0.v); r28::cols_mac_hi<FP>(acc.t1, a1, b.c1.v); }
    else { r28::cols_mac_lo<FP>(acc.t0, a0, b.c0.v); r28::cols_mac_lo<FP>(acc.t1, a1, b.c1.v); }
    uint32_t sb[N];
#pragma unroll
    for (int i = 0; i < N; i++) { a0[i] += a1[i]; sb[i] = b.c0.v[i] + b.c1.v[i]; }     // exact limb-wise sums (Karatsuba on the integers)
    if constexpr (HI) r28::cols_mac_hi<FP>(acc.ts, a0, sb); else r28::cols_mac_lo<FP>(acc.ts, a0, sb);
}
template <class C, bool HI, class ACC>
BBS_HD void f2acc2_mac_fp(ACC& acc, const Fp2<C>& a, const Fp<C>& y) {              // b = y in Fp: t1 += 0
    constexpr int N = C::FpP::N;
    uint32_t sa[N];
#pragma unroll
    for (int i = 0; i < N; i++) sa[i] = a.c0.v[i] + a.c1.v[i];
    if constexpr (HI) { r28::cols_mac_hi<FP>(acc.t0, a.c0.v, y.v); r28::cols_mac_hi<FP>(acc.ts, sa, y.v); }
    else { r28::cols_mac_lo<FP>(acc.t0, a.c0.v, y.v); r28::cols_mac_lo<FP>(acc.ts, sa, y.v); }
}
template <class C>
BBS_HD void f2acc_finish_lo(F2AccLo<C>& acc, F2AccMid<C>& st) {
    constexpr int N = C::FpP::N;
#pragma unroll
    for (int c = 0; c < N; c++) {
        const uint64_t x = acc.t0[c], y = acc.t1[c];
        acc.ts[c] = acc.ts[c] - x - y;
        acc.t0[c] = x + C::FpP::WP2X[c] - y;
    }
    r28::cols_reduce_lo<FP>(acc.t0, st.m_re, st.carry_re);
    r28::cols_reduce_lo<FP>(acc.ts, st.m_im, st.carry_im);
}
template <class C>
BBS_HD Fp2<C> f2acc_finish_hi(F2AccHi<C>& acc, const F2AccMid<C>& st) {
    constexpr int N = C::FpP::N;
#pragma unroll
    for (int k = 0; k < N - 1; k++) {
        const uint64_t x = acc.t0[k], y = acc.t1[k];
        acc.ts[k] = acc.ts[k] - x - y;
        acc.t0[k] = x + C::FpP::WP2X[N + k] - y;
    }
    Fp2<C> r;
    r28::cols_reduce_hi<FP>(r.c0.v, acc.t0, st.m_re, st.carry_re);
    r28::cols_reduce_hi<FP>(r.c1.v, acc.ts, st.m_im, st.carry_im);
    return r;
}

// One half of an Fp4 square (Granger-Scott cyclotomic squaring; Fp4 = Fp2[s]/(s^2 - xi), xi = c + u) as FOUR
// limb-column products and one reduction pair, the same instruction stream for both halves (operands selected by
// `hi`; the two lanes of a pair in pairing_dist.hpp run it side by side):
//   low  half (A = x0, B = x1):  x0^2 + xi x1^2 ;  high half (A = x1, B = x0):  2 x0 x1
//   slot      low                            high
//   X1        (A0 + A1)(A0 - A1)             2 A0 B0
//   X2        2 A0 A1                        2 A1 B1
//   X3        (B0 + B1)(B0 - B1)             2 A0 B1
//   X4        2 B0 B1                        2 A1 B0
//   re = X1 + K p^2 + (low ? c X3 - X4 : -X2) ;  im = X3 + (low ? c X4 + X2 : X4)
// 4 N^2 + 2 N^2 multiply-accumulates instead of two fused Fp2 squares (8 N^2) plus their linear chains.
// Column bounds: differences are taken reduced (normal limbs); for c = 9 (BN254) the sums as well, so that
// c X3 and c X4 stay below 2^63.5 (N = 10).  Host-testable: bbs_selftest_fp4sqr.
template <class C>
BBS_HD Fp2<C> fp4_sqr_part(bool hi, const Fp2<C>& A, const Fp2<C>& B) {
    using P = typename C::FpP;
    constexpr int N = P::N;
    constexpr uint32_t XC = C::K::XI_C0;
    const Fp<C> dA = fe_sub<P>(A.c0, A.c1), dB = fe_sub<P>(B.c0, B.c1);
    Fp<C> sA, sB;
    if constexpr (XC == 1) { sA = fe_add_nr<P>(A.c0, A.c1); sB = fe_add_nr<P>(B.c0, B.c1); }
    else { sA = fe_add<P>(A.c0, A.c1); sB = fe_add<P>(B.c0, B.c1); }
    uint32_t l1[N], r1[N], l2[N], r2[N], l3[N], r3[N], l4[N], r4[N];
#pragma unroll
    for (int i = 0; i < N; i++) {
        const uint32_t a0 = A.c0.v[i], a1 = A.c1.v[i], b0 = B.c0.v[i], b1 = B.c1.v[i];
        l1[i] = hi ? (a0 << 1) : sA.v[i];    r1[i] = hi ? b0 : dA.v[i];
        l2[i] = hi ? (a1 << 1) : (a0 << 1);  r2[i] = hi ? b1 : a1;
        l3[i] = hi ? (a0 << 1) : sB.v[i];    r3[i] = hi ? b1 : dB.v[i];
        l4[i] = hi ? (a1 << 1) : (b0 << 1);  r4[i] = hi ? b0 : b1;
    }
    uint64_t x1[2 * N - 1], x2[2 * N - 1], x3[2 * N - 1], x4[2 * N - 1];
    r28::cols_zero<P>(x1); r28::cols_zero<P>(x2); r28::cols_zero<P>(x3); r28::cols_zero<P>(x4);
    r28::cols_mac<P>(x1, l1, r1);
    r28::cols_mac<P>(x2, l2, r2);
    r28::cols_mac<P>(x3, l3, r3);
    r28::cols_mac<P>(x4, l4, r4);
#pragma unroll
    for (int c = 0; c < 2 * N - 1; c++) {
        const uint64_t cx3 = XC * x3[c], cx4 = XC * x4[c];
        const uint64_t re = x1[c] + P::WP2X[c] + (hi ? (0 - x2[c]) : (cx3 - x4[c]));
        const uint64_t im = x3[c] + (hi ? x4[c] : (cx4 + x2[c]));
        x1[c] = re;
        x2[c] = im;
    }
    Fp2<C> r;
    r28::cols_reduce<P>(r.c0.v, x1);
    r28::cols_reduce<P>(r.c1.v, x2);
    return r;
}

// fp4_sqr_part with the four limb-column products accumulated in two passes (columns 0 .. N-1, then N .. 2N-2): 4 x N
// accumulator columns live instead of 4 x (2N-1), the operand pair of each product formed when it is needed.  Same
// value, limb for limb (host-testable: bbs_selftest_fp4sqr with hi | 2).
template <class C>
BBS_HD Fp2<C> fp4_sqr_part2(bool hi, const Fp2<C>& A, const Fp2<C>& B) {
    using P = typename C::FpP;
    constexpr int N = P::N;
    constexpr uint32_t XC = C::K::XI_C0;
    const Fp<C> dA = fe_sub<P>(A.c0, A.c1), dB = fe_sub<P>(B.c0, B.c1);
    // operands of product k (see the table above fp4_sqr_part); sums formed here, lazily for xi = 1 + u
    auto operands = [&](int k, uint32_t* l, uint32_t* r) {
#pragma unroll
        for (int i = 0; i < N; i++) {
            const uint32_t a0 = A.c0.v[i], a1 = A.c1.v[i], b0 = B.c0.v[i], b1 = B.c1.v[i];
            if (k == 1) { l[i] = hi ? (a0 << 1) : (a0 + a1); r[i] = hi ? b0 : dA.v[i]; }
            if (k == 2) { l[i] = hi ? (a1 << 1) : (a0 << 1); r[i] = hi ? b1 : a1; }
            if (k == 3) { l[i] = hi ? (a0 << 1) : (b0 + b1); r[i] = hi ? b1 : dB.v[i]; }
            if (k == 4) { l[i] = hi ? (a1 << 1) : (b0 << 1); r[i] = hi ? b0 : b1; }
        }
    };
    static_assert(XC == 1, "two-pass Fp4 square: xi = 1 + u only (lazy sums; BN254 keeps the two-squarings form)");
    uint32_t l[N], r[N];
    uint64_t x1[N], x2[N], x3[N], x4[N];
    r28::cols_lo_zero<P>(x1); r28::cols_lo_zero<P>(x2); r28::cols_lo_zero<P>(x3); r28::cols_lo_zero<P>(x4);
    operands(1, l, r); r28::cols_mac_lo<P>(x1, l, r);
    operands(2, l, r); r28::cols_mac_lo<P>(x2, l, r);
    operands(3, l, r); r28::cols_mac_lo<P>(x3, l, r);
    operands(4, l, r); r28::cols_mac_lo<P>(x4, l, r);
#pragma unroll
    for (int c = 0; c < N; c++) {
        const uint64_t re = x1[c] + P::WP2X[c] + (hi ? (0 - x2[c]) : (x3[c] - x4[c]));
        const uint64_t im = x3[c] + (hi ? x4[c] : (x4[c] + x2[c]));
        x1[c] = re;
        x2[c] = im;
    }
    uint32_t m_re[N], m_im[N];
    uint64_t c_re, c_im;
    r28::cols_reduce_lo<P>(x1, m_re, c_re);
    r28::cols_reduce_lo<P>(x2, m_im, c_im);
    uint64_t y1[N - 1], y2[N - 1], y3[N - 1], y4[N - 1];
    r28::cols_hi_zero<P>(y1); r28::cols_hi_zero<P>(y2); r28::cols_hi_zero<P>(y3); r28::cols_hi_zero<P>(y4);
    operands(1, l, r); r28::cols_mac_hi<P>(y1, l, r);
    operands(2, l, r); r28::cols_mac_hi<P>(y2, l, r);
    operands(3, l, r); r28::cols_mac_hi<P>(y3, l, r);
    operands(4, l, r); r28::cols_mac_hi<P>(y4, l, r);
#pragma unroll
    for (int k = 0; k < N - 1; k++) {
        const uint64_t re = y1[k] + P::WP2X[N + k] + (hi ? (0 - y2[k]) : (y3[k] - y4[k]));
        const uint64_t im = y3[k] + (hi ? y4[k] : (y4[k] + y2[k]));
        y1[k] = re;
        y2[k] = im;
    }
    Fp2<C> out;
    r28::cols_reduce_hi<P>(out.c0.v, y1, m_re, c_re);
    r28::cols_reduce_hi<P>(out.c1.v, y2, m_im, c_im);
    return out;
}

template <class C>
BBS_HD Fp2<C> f2_mul(const Fp2<C>& a, const Fp2<C>& b) {
#ifndef BBS_NO_FUSED_F2
    if constexpr (C::FpP::W == 28) return f2_mul_fused<C>(a, b);
#endif
    // Karatsuba: 3 Fp multiplications
    Fp<C> t0 = fe_mul<FP>(a.c0, b.c0);
    Fp<C> t1 = fe_mul<FP>(a.c1, b.c1);
    Fp<C> s = fe_mul<FP>(fe_add_nr<FP>(a.c0, a.c1), fe_add_nr<FP>(b.c0, b.c1));   // lazy sums feed the multiplier only
    return {fe_sub<FP>(t0, t1), fe_sub<FP>(fe_sub<FP>(s, t0), t1)};
}

template <class C>
BBS_HD Fp2<C> f2_sqr(const Fp2<C>& a) {
#ifndef BBS_NO_FUSED_F2
    if constexpr (C::FpP::W == 28) return f2_sqr_fused<C>(a);
#endif
    // (a0+a1)(a0-a1) + 2 a0 a1 u : 2 Fp multiplications
    Fp<C> t = fe_mul<FP>(fe_add_nr<FP>(a.c0, a.c1), fe_sub<FP>(a.c0, a.c1));
    Fp<C> m = fe_mul<FP>(a.c0, a.c1);
    return {t, fe_dbl<FP>(m)};
}

template <class C>
BBS_HD Fp2<C> f2_mul_fp(const Fp2<C>& a, const Fp<C>& s) {
    return {fe_mul<FP>(a.c0, s), fe_mul<FP>(a.c1, s)};
}

// a * xi, xi = XI_C0 + u :  (c a0 - a1) + (c a1 + a0) u
template <class C>
BBS_HD Fp2<C> f2_mul_xi(const Fp2<C>& a) {
    static_assert(C::K::XI_C0 == 9 || C::K::XI_C0 == 1, "xi");
    if constexpr (C::K::XI_C0 == 1) {
        return {fe_sub<FP>(a.c0, a.c1), fe_add<FP>(a.c0, a.c1)};
    } else {
        // 9 a = 8 a + a : two chains per component (fe_lin weight limit is 8)
        const Fp<C> e0 = fe_scale<FP, 8>(a.c0), e1 = fe_scale<FP, 8>(a.c1);
        return {fe_lin<FP, 1, 1, -1>(e0, a.c0, a.c1), fe_lin<FP, 1, 1, 1>(e1, a.c1, a.c0)};
    }
}

// small linear combinations of Fp2 values, one reduction chain per component
template <class C, int C0, int C1>
BBS_HD Fp2<C> f2_lin(const Fp2<C>& x0, const Fp2<C>& x1) {
    return {fe_lin<FP, C0, C1>(x0.c0, x1.c0), fe_lin<FP, C0, C1>(x0.c1, x1.c1)};
}
// C0 x0 + C1 x1 (plus) or C0 x0 - C1 x1, the sign chosen at run time: one chain per component
template <class C, int C0, int C1>
BBS_HD Fp2<C> f2_lin_pm(const Fp2<C>& x0, const Fp2<C>& x1, bool plus) {
    return {fe_lin_pm<FP, C0, C1>(x0.c0, x1.c0, plus), fe_lin_pm<FP, C0, C1>(x0.c1, x1.c1, plus)};
}
template <class C, int C0, int C1, int C2>
BBS_HD Fp2<C> f2_lin(const Fp2<C>& x0, const Fp2<C>& x1, const Fp2<C>& x2) {
    return {fe_lin<FP, C0, C1, C2>(x0.c0, x1.c0, x2.c0), fe_lin<FP, C0, C1, C2>(x0.c1, x1.c1, x2.c1)};
}

template <class C>
BBS_HD_NOINLINE Fp2<C> f2_inv(const Fp2<C>& a) {
    Fp<C> n = fe_add<FP>(fe_sqr<FP>(a.c0), fe_sqr<FP>(a.c1));
    Fp<C> ni = fe_inv<FP>(n);
    return {fe_mul<FP>(a.c0, ni), fe_neg<FP>(fe_mul<FP>(a.c1, ni))};
}

// ------------------------------------------------------------------------- Fp6
template <class C>
struct Fp6 {
    Fp2<C> c0, c1, c2;
};

template <class C> BBS_HD Fp6<C> f6_zero() { return {f2_zero<C>(), f2_zero<C>(), f2_zero<C>()}; }
template <class C> BBS_HD Fp6<C> f6_one() { return {f2_one<C>(), f2_zero<C>(), f2_zero<C>()}; }
template <class C> BBS_HD Fp6<C> f6_add(const Fp6<C>& a, const Fp6<C>& b) { return {f2_add<C>(a.c0, b.c0), f2_add<C>(a.c1, b.c1), f2_add<C>(a.c2, b.c2)}; }
template <class C> BBS_HD Fp6<C> f6_sub(const Fp6<C>& a, const Fp6<C>& b) { return {f2_sub<C>(a.c0, b.c0), f2_sub<C>(a.c1, b.c1), f2_sub<C>(a.c2, b.c2)}; }
template <class C> BBS_HD Fp6<C> f6_neg(const Fp6<C>& a) { return {f2_neg<C>(a.c0), f2_neg<C>(a.c1), f2_neg<C>(a.c2)}; }
template <class C> BBS_HD bool f6_eq(const Fp6<C>& a, const Fp6<C>& b) { return f2_eq<C>(a.c0, b.c0) & f2_eq<C>(a.c1, b.c1) & f2_eq<C>(a.c2, b.c2); }

// a * v : (c0, c1, c2) -> (xi c2, c0, c1)
template <class C>
BBS_HD Fp6<C> f6_mul_v(const Fp6<C>& a) { return {f2_mul_xi<C>(a.c2), a.c0, a.c1}; }

template <class C>
BBS_HD_NOINLINE Fp6<C> f6_mul(const Fp6<C>& a, const Fp6<C>& b) {
    // Karatsuba / Toom-style: 6 Fp2 multiplications
    Fp2<C> v0 = f2_mul<C>(a.c0, b.c0);
    Fp2<C> v1 = f2_mul<C>(a.c1, b.c1);
    Fp2<C> v2 = f2_mul<C>(a.c2, b.c2);
    Fp2<C> t0 = f2_sub<C>(f2_sub<C>(f2_mul<C>(f2_add<C>(a.c1, a.c2), f2_add<C>(b.c1, b.c2)), v1), v2);
    Fp2<C> t1 = f2_sub<C>(f2_sub<C>(f2_mul<C>(f2_add<C>(a.c0, a.c1), f2_add<C>(b.c0, b.c1)), v0), v1);
    Fp2<C> t2 = f2_sub<C>(f2_sub<C>(f2_mul<C>(f2_add<C>(a.c0, a.c2), f2_add<C>(b.c0, b.c2)), v0), v2);
    return {f2_add<C>(v0, f2_mul_xi<C>(t0)), f2_add<C>(t1, f2_mul_xi<C>(v2)), f2_add<C>(t2, v1)};
}

// a * (b0 + b1 v)
template <class C>
BBS_HD_NOINLINE Fp6<C> f6_mul_by_01(const Fp6<C>& a, const Fp2<C>& b0, const Fp2<C>& b1) {
    Fp2<C> v0 = f2_mul<C>(a.c0, b0);
    Fp2<C> v1 = f2_mul<C>(a.c1, b1);
    // c0 = v0 + xi * a2 b1
    Fp2<C> c0 = f2_add<C>(v0, f2_mul_xi<C>(f2_mul<C>(a.c2, b1)));
    // c1 = (a0+a1)(b0+b1) - v0 - v1
    Fp2<C> c1 = f2_sub<C>(f2_sub<C>(f2_mul<C>(f2_add<C>(a.c0, a.c1), f2_add<C>(b0, b1)), v0), v1);
    // c2 = a2 b0 + v1
    Fp2<C> c2 = f2_add<C>(f2_mul<C>(a.c2, b0), v1);
    return {c0, c1, c2};
}

// a * (b1 v)
template <class C>
BBS_HD Fp6<C> f6_mul_by_1(const Fp6<C>& a, const Fp2<C>& b1) {
    return {f2_mul_xi<C>(f2_mul<C>(a.c2, b1)), f2_mul<C>(a.c0, b1), f2_mul<C>(a.c1, b1)};
}

// a * (b0), b0 in Fp2
template <class C>
BBS_HD Fp6<C> f6_mul_by_0(const Fp6<C>& a, const Fp2<C>& b0) {
    return {f2_mul<C>(a.c0, b0), f2_mul<C>(a.c1, b0), f2_mul<C>(a.c2, b0)};
}

template <class C>
BBS_HD Fp6<C> f6_mul_fp(const Fp6<C>& a, const Fp<C>& s) {
    return {f2_mul_fp<C>(a.c0, s), f2_mul_fp<C>(a.c1, s), f2_mul_fp<C>(a.c2, s)};
}

template <class C>
BBS_HD_NOINLINE Fp6<C> f6_inv(const Fp6<C>& a) {
    Fp2<C> t0 = f2_sub<C>(f2_sqr<C>(a.c0), f2_mul_xi<C>(f2_mul<C>(a.c1, a.c2)));
    Fp2<C> t1 = f2_sub<C>(f2_mul_xi<C>(f2_sqr<C>(a.c2)), f2_mul<C>(a.c0, a.c1));
    Fp2<C> t2 = f2_sub<C>(f2_sqr<C>(a.c1), f2_mul<C>(a.c0, a.c2));
    Fp2<C> d = f2_add<C>(f2_mul<C>(a.c0, t0),
                         f2_mul_xi<C>(f2_add<C>(f2_mul<C>(a.c2, t1), f2_mul<C>(a.c1, t2))));
    Fp2<C> di = f2_inv<C>(d);
    return {f2_mul<C>(t0, di), f2_mul<C>(t1, di), f2_mul<C>(t2, di)};
}

// ------------------------------------------------------------------------ Fp12
template <class C>
struct Fp12 {
    Fp6<C> c0, c1;
};

template <class C> BBS_HD Fp12<C> f12_one() { return {f6_one<C>(), f6_zero<C>()}; }
template <class C> BBS_HD bool f12_eq(const Fp12<C>& a, const Fp12<C>& b) { return f6_eq<C>(a.c0, b.c0) & f6_eq<C>(a.c1, b.c1); }
template <class C> BBS_HD bool f12_is_one(const Fp12<C>& a) { return f12_eq<C>(a, f12_one<C>()); }
template <class C> BBS_HD Fp12<C> f12_conj(const Fp12<C>& a) { return {a.c0, f6_neg<C>(a.c1)}; }

template <class C>
BBS_HD_NOINLINE Fp12<C> f12_mul(const Fp12<C>& a, const Fp12<C>& b) {
    Fp6<C> v0 = f6_mul<C>(a.c0, b.c0);
    Fp6<C> v1 = f6_mul<C>(a.c1, b.c1);
    Fp6<C> s = f6_mul<C>(f6_add<C>(a.c0, a.c1), f6_add<C>(b.c0, b.c1));
    return {f6_add<C>(v0, f6_mul_v<C>(v1)), f6_sub<C>(f6_sub<C>(s, v0), v1)};
}

template <class C>
BBS_HD_NOINLINE Fp12<C> f12_sqr(const Fp12<C>& a) {
    // complex squaring: 2 Fp6 multiplications
    Fp6<C> ab = f6_mul<C>(a.c0, a.c1);
    Fp6<C> t = f6_mul<C>(f6_add<C>(a.c0, a.c1), f6_add<C>(a.c0, f6_mul_v<C>(a.c1)));
    // c0 = t - ab - v*ab ; c1 = 2ab
    Fp6<C> c0 = f6_sub<C>(f6_sub<C>(t, ab), f6_mul_v<C>(ab));
    return {c0, f6_add<C>(ab, ab)};
}

template <class C>
BBS_HD_NOINLINE Fp12<C> f12_inv(const Fp12<C>& a) {
    // 1/(c0 + c1 w) = (c0 - c1 w) / (c0^2 - v c1^2)
    Fp6<C> d = f6_sub<C>(f6_mul<C>(a.c0, a.c0), f6_mul_v<C>(f6_mul<C>(a.c1, a.c1)));
    Fp6<C> di = f6_inv<C>(d);
    return {f6_mul<C>(a.c0, di), f6_neg<C>(f6_mul<C>(a.c1, di))};
}

// coefficient i (0..5) of the w-basis: f = sum g_i w^i ; g0,g2,g4 = c0.{c0,c1,c2} ; g1,g3,g5 = c1.{c0,c1,c2}
template <class C, int K>
BBS_HD_NOINLINE Fp12<C> f12_frob(const Fp12<C>& a) {
    static_assert(K >= 1 && K <= 3, "frobenius power");
    auto coef = [](int i) {
        Fp2<C> g;
#pragma unroll
        for (int j = 0; j < C::FpP::N; j++) {
            g.c0.v[j] = C::K::FROB[K - 1][i][0][j];
            g.c1.v[j] = C::K::FROB[K - 1][i][1][j];
        }
        return g;
    };
    auto cj = [](const Fp2<C>& x) { return (K & 1) ? f2_conj<C>(x) : x; };
    Fp12<C> r;
    r.c0.c0 = cj(a.c0.c0);
    r.c1.c0 = f2_mul<C>(cj(a.c1.c0), coef(1));
    r.c0.c1 = f2_mul<C>(cj(a.c0.c1), coef(2));
    r.c1.c1 = f2_mul<C>(cj(a.c1.c1), coef(3));
    r.c0.c2 = f2_mul<C>(cj(a.c0.c2), coef(4));
    r.c1.c2 = f2_mul<C>(cj(a.c1.c2), coef(5));
    return r;
}

// ---- sparse multiplications by Miller-loop line values --------------------------------------
// M-type twist (BLS12-381): line = l0 + l1 v + (yP) v w        (l0, l1 in Fp2, yP in Fp)
template <class C>
BBS_HD_NOINLINE Fp12<C> f12_mul_by_line_M(const Fp12<C>& f, const Fp2<C>& l0, const Fp2<C>& l1, const Fp<C>& yP) {
    // A = (l0, l1, 0), B = (0, yP, 0)
    Fp6<C> v0 = f6_mul_by_01<C>(f.c0, l0, l1);
    // f.c1 * (yP v) : (xi c2 yP, c0 yP, c1 yP)
    Fp6<C> v1 = {f2_mul_xi<C>(f2_mul_fp<C>(f.c1.c2, yP)), f2_mul_fp<C>(f.c1.c0, yP), f2_mul_fp<C>(f.c1.c1, yP)};
    Fp2<C> l1y = l1;
    l1y.c0 = fe_add<FP>(l1y.c0, yP);
    Fp6<C> s = f6_mul_by_01<C>(f6_add<C>(f.c0, f.c1), l0, l1y);
    return {f6_add<C>(v0, f6_mul_v<C>(v1)), f6_sub<C>(f6_sub<C>(s, v0), v1)};
}

// D-type twist (BN254): line = (yP) + (l0 + l1 v) w             (l0, l1 in Fp2, yP in Fp)
template <class C>
BBS_HD_NOINLINE Fp12<C> f12_mul_by_line_D(const Fp12<C>& f, const Fp2<C>& l0, const Fp2<C>& l1, const Fp<C>& yP) {
    // A = (yP, 0, 0), B = (l0, l1, 0)
    Fp6<C> v0 = f6_mul_fp<C>(f.c0, yP);
    Fp6<C> v1 = f6_mul_by_01<C>(f.c1, l0, l1);
    Fp2<C> l0y = l0;
    l0y.c0 = fe_add<FP>(l0y.c0, yP);
    Fp6<C> s = f6_mul_by_01<C>(f6_add<C>(f.c0, f.c1), l0y, l1);
    return {f6_add<C>(v0, f6_mul_v<C>(v1)), f6_sub<C>(f6_sub<C>(s, v0), v1)};
}

#undef FP
}  // namespace bbs
