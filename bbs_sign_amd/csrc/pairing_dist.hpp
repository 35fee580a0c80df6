// Wavefront-cooperative pairing check: ONE Fp12 value is spread over SIX lanes.
//
// A batch of 4096 pairing checks with one item per lane is 64 wavefronts on a chip with 1024
// SIMDs, and an Fp12 value is 144 VGPRs, so the one-lane form (pairing.hpp) runs at 6 % of the
// SIMDs and spills 8 KB per lane.  Here f = sum_{m<6} g_m w^m (w^6 = xi) and lane m of a
// six-lane group holds only g_m in Fp2 (24 VGPRs for BLS12-381); operands move between the lanes
// of a group with ds_bpermute (__shfl).  Ten groups share one wavefront (lanes 60..63 idle).
//
// Lane k computes output coefficient k of every operation as ONE Fp2 dot product whose limb products are summed
// in 64-bit columns and Montgomery-reduced once (tower.hpp F2Acc; N = limbs of Fp):
//   mul        : c_k = sum_j g_j h_(k-j) xi^[j>k]                      6 * 3 N^2 + 2 N^2 multiply-accumulates
//   square     : unordered pairs, four slots per lane                  4 * 3 N^2 + 2 N^2
//   line (M/D) : sparse, 3 non-zero coefficients                       8 N^2 + 2 N^2 (+ 4 N^2 for nl * xP)
//   cyclotomic square (Granger-Scott over Fp4 pairs (m, m+3))          four column products, 4 N^2 + 2 N^2
//   conj, frob : lane-local
//   inverse    : gathered on every lane (once per item), one safegcd inversion in Fp
// The kernel takes the whole register file (one wavefront per SIMD, Miller-loop operations inlined): DESIGN.md 5.
//
// Same group elements / booleans as pairing.hpp; selftest entry points compare the two on the GPU.
#pragma once
#include "pairing.hpp"

#if defined(__HIP_DEVICE_COMPILE__) || !defined(BBS_HOST_TWIN)
namespace bbs {

// hot lane-sliced operations: separate functions by default; -DBBS_DIST_INLINE=1 inlines the Miller-loop pair
// (square, line), =2 also the cyclotomic square (A/B knob)
#ifndef BBS_DIST_INLINE
#define BBS_DIST_INLINE 2
#endif
#if BBS_DIST_INLINE >= 1
#define BBS_DIST_MILLER __device__ __forceinline__
#else
#define BBS_DIST_MILLER __device__ __attribute__((noinline))
#endif
#if BBS_DIST_INLINE >= 2
#define BBS_DIST_CYCLO __device__ __forceinline__
#else
#define BBS_DIST_CYCLO __device__ __attribute__((noinline))
#endif

constexpr int GRP = 6;                 // lanes per item
constexpr int GRP_PER_WAVE = 10;

struct Lane6 {
    int base;     // first lane of the group inside the wavefront
    int m;        // coefficient index 0..5 held by this lane
};

template <class P>
__device__ __forceinline__ Fe<P> fe_shfl(const Fe<P>& v, int src_lane) {
    Fe<P> r;
#pragma unroll
    for (int i = 0; i < P::N; i++) r.v[i] = (uint32_t)__shfl((int)v.v[i], src_lane, 64);
    return r;
}
template <class C>
__device__ __forceinline__ Fp2<C> f2_shfl(const Fp2<C>& v, int src_lane) {
    return {fe_shfl<typename C::FpP>(v.c0, src_lane), fe_shfl<typename C::FpP>(v.c1, src_lane)};
}
template <class C>
__device__ __forceinline__ Fp2<C> f2_sel(bool c, const Fp2<C>& a, const Fp2<C>& b) {
    return {fe_select<typename C::FpP>(c, a.c0, b.c0), fe_select<typename C::FpP>(c, a.c1, b.c1)};
}

// value of coefficient `src` (0..5) of the group's element
template <class C>
__device__ __forceinline__ Fp2<C> d_coef(const Lane6& L, const Fp2<C>& g, int src) { return f2_shfl<C>(g, L.base + src); }

#ifndef BBS_DIST_LAZY
#define BBS_DIST_LAZY 1
#endif
// 1: the dot products accumulate their limb-product columns in two passes (tower.hpp F2AccLo / F2AccHi: half the
// accumulator registers, operands fetched twice) -- what lets the kernel keep two wavefronts per SIMD (BBS_PAIR_WAVES=2)
#ifndef BBS_DIST_TWOPASS
#define BBS_DIST_TWOPASS 0
#endif
// per phase: the Miller-loop operations (d_sqr, d_mul_line) and the final-exponentiation operations (d_mul, the
// cyclotomic square) can be switched separately -- with BBS_PAIR_SPLIT they live in different kernels
#ifndef BBS_DIST_TWOPASS_MILLER
#define BBS_DIST_TWOPASS_MILLER BBS_DIST_TWOPASS
#endif
#ifndef BBS_DIST_TWOPASS_FINAL
#define BBS_DIST_TWOPASS_FINAL BBS_DIST_TWOPASS
#endif
#ifndef BBS_DIST_UNROLL
#define BBS_DIST_UNROLL 1        // iterations of the dot-product loops kept rolled (1) or unrolled (6 / 4): A/B knob
#endif
// 1 (round 5): the factor xi of a wrapped term (w^6 = xi) is applied by the lane that PUBLISHES the operand, not by the lane
// that fetches it.  Every fetched operand used to be multiplied by xi on every lane (two limb chains) and then selected -- per
// term of every dot product; but in each step of each operation a lane's value is fetched by lanes that all want the same form
// of it (plain, or times xi), and which form depends only on the publishing lane's own index and the step.  So a lane
// computes xi * (its coefficient) ONCE per operation and publishes the form its readers want: one select per term instead of
// a multiplication by xi and a select.  Same field elements term by term (checked by the six-lane self-tests against the
// oracle's Fp12); A/B knob.
#ifndef BBS_DIST_XI_AT_SOURCE
#define BBS_DIST_XI_AT_SOURCE 1
#endif
#if BBS_DIST_LAZY
// f * h: lane k computes its own output coefficient  c_k = sum_j g_j h_{(k-j) mod 6} xi^[j > k]  as ONE Fp2 dot
// product (tower.hpp F2Acc): the six products are accumulated as unreduced column sums and reduced once --
// 6 * 3 N^2 + 2 N^2 multiply-accumulates instead of 6 * 5 N^2.  Operands come in by ds_bpermute.
template <class C, int V = 0>
__device__ __attribute__((noinline)) Fp2<C> d_mul(const Lane6& L, const Fp2<C>& g, const Fp2<C>& h) {
    const int k = L.m;
#if BBS_DIST_TWOPASS_FINAL
    F2AccLo<C> lo;
    f2acc_lo_zero<C>(lo);
#pragma unroll 1
    for (int j = 0; j < GRP; j++) {
        const Fp2<C> a = d_coef<C>(L, g, j);
        int src = k - j;
        if (src < 0) src += GRP;
        Fp2<C> b = d_coef<C>(L, h, src);
        b = f2_sel<C>(j > k, f2_mul_xi<C>(b), b);
        f2acc2_mac_sh<C, false>(lo, a, b, 0u);
    }
    F2AccMid<C> mid;
    f2acc_finish_lo<C>(lo, mid);
    F2AccHi<C> hi;
    f2acc_hi_zero<C>(hi);
#pragma unroll 1
    for (int j = 0; j < GRP; j++) {
        const Fp2<C> a = d_coef<C>(L, g, j);
        int src = k - j;
        if (src < 0) src += GRP;
        Fp2<C> b = d_coef<C>(L, h, src);
        b = f2_sel<C>(j > k, f2_mul_xi<C>(b), b);
        f2acc2_mac_sh<C, true>(hi, a, b, 0u);
    }
    return f2acc_finish_hi<C>(hi, mid);
#else
    F2Acc<C> acc;
    f2acc_zero<C>(acc);
#if BBS_DIST_XI_AT_SOURCE
    // the reader of this lane's h at step j is lane (k + j) mod 6, and it wrapped (wants xi h) iff k + j >= 6
    const Fp2<C> hx = f2_mul_xi<C>(h);
#endif
#pragma unroll BBS_DIST_UNROLL
    for (int j = 0; j < GRP; j++) {
        const Fp2<C> a = d_coef<C>(L, g, j);
        int src = k - j;
        if (src < 0) src += GRP;
#if BBS_DIST_XI_AT_SOURCE
        const Fp2<C> b = d_coef<C>(L, f2_sel<C>(k + j >= GRP, hx, h), src);
#else
        Fp2<C> b = d_coef<C>(L, h, src);
        b = f2_sel<C>(j > k, f2_mul_xi<C>(b), b);
#endif
        f2acc_mac<C, 1>(acc, a, b);
    }
    return f2acc_finish<C>(acc);
#endif
}

// f^2: c_k = sum over unordered pairs {i, j}, i + j = k mod 6, of w g_i g_j xi^[i + j >= 6] (w = 1 for squares,
// 2 otherwise): four slots per lane (odd lanes use three), total weight 6 -- 4 * 3 N^2 + 2 N^2.
// slot s of lane k: i = SQ_I[k][s], j = SQ_J[k][s]; j = 7 marks the unused slot.
template <class C>
BBS_DIST_MILLER Fp2<C> d_sqr(const Lane6& L, const Fp2<C>& g) {
    // packed per lane: 4 slots x (i:3 bits, j:3 bits) -- (0,0)(3,3)(1,5)(2,4) | (0,1)(2,5)(3,4)- | (1,1)(0,2)(4,4)(3,5) |
    //                  (0,3)(1,2)(4,5)- | (2,2)(0,4)(1,3)(5,5) | (0,5)(1,4)(2,3)-
    constexpr uint32_t PK[6] = {
        (0u | 0u << 3) | (3u | 3u << 3) << 6 | (1u | 5u << 3) << 12 | (2u | 4u << 3) << 18,
        (0u | 1u << 3) | (2u | 5u << 3) << 6 | (3u | 4u << 3) << 12 | (7u | 7u << 3) << 18,
        (1u | 1u << 3) | (0u | 2u << 3) << 6 | (4u | 4u << 3) << 12 | (3u | 5u << 3) << 18,
        (0u | 3u << 3) | (1u | 2u << 3) << 6 | (4u | 5u << 3) << 12 | (7u | 7u << 3) << 18,
        (2u | 2u << 3) | (0u | 4u << 3) << 6 | (1u | 3u << 3) << 12 | (5u | 5u << 3) << 18,
        (0u | 5u << 3) | (1u | 4u << 3) << 6 | (2u | 3u << 3) << 12 | (7u | 7u << 3) << 18};
    const int k = L.m;
    uint32_t pk = PK[0];
#pragma unroll
    for (int q = 1; q < 6; q++) pk = (k == q) ? PK[q] : pk;
#if BBS_DIST_TWOPASS_MILLER
    // slot s: operands and weight (fetched in both passes)
    auto slot = [&](int s, Fp2<C>& a, Fp2<C>& b, uint32_t& sh) {
        const uint32_t i = (pk >> (6 * s)) & 7u, j = (pk >> (6 * s + 3)) & 7u;
        const bool used = i != 7u;
        const uint32_t ii = used ? i : 0u, jj = used ? j : 0u;
        a = d_coef<C>(L, g, (int)ii);
        b = d_coef<C>(L, g, (int)jj);
        a = f2_sel<C>(used, a, f2_zero<C>());
        b = f2_sel<C>(ii + jj >= (uint32_t)GRP, f2_mul_xi<C>(b), b);
        sh = (used && ii != jj) ? 1u : 0u;
    };
    F2AccLo<C> lo;
    f2acc_lo_zero<C>(lo);
#pragma unroll 1
    for (int s = 0; s < 4; s++) {
        Fp2<C> a, b;
        uint32_t sh;
        slot(s, a, b, sh);
        f2acc2_mac_sh<C, false>(lo, a, b, sh);
    }
    F2AccMid<C> mid;
    f2acc_finish_lo<C>(lo, mid);
    F2AccHi<C> hi;
    f2acc_hi_zero<C>(hi);
#pragma unroll 1
    for (int s = 0; s < 4; s++) {
        Fp2<C> a, b;
        uint32_t sh;
        slot(s, a, b, sh);
        f2acc2_mac_sh<C, true>(hi, a, b, sh);
    }
    return f2acc_finish_hi<C>(hi, mid);
#else
    F2Acc<C> acc;
    f2acc_zero<C>(acc);
#if BBS_DIST_XI_AT_SOURCE
    // who reads this lane's g as the SECOND factor of slot s, and does the pair wrap (i + j >= 6)?  From the table above: slot 0
    // never wraps; slot 1: lanes 3 and 5 are read by wrapping pairs only ((3,3), (2,5)), lanes 2 and 4 by plain ones; slots 2
    // and 3: lanes 4 and 5 by wrapping pairs only ((3,4), (4,4), (1,5), (4,5) / (2,4), (3,5), (5,5)), lane 3 by plain ones.
    const Fp2<C> gx = f2_mul_xi<C>(g);
    // bit s of the mask: this lane publishes xi g in slot s
    const uint32_t xmask = (k == 3) ? 0x2u : ((k == 4) ? 0xCu : ((k == 5) ? 0xEu : 0u));
#endif
#pragma unroll BBS_DIST_UNROLL
    for (int s = 0; s < 4; s++) {
        const uint32_t i = (pk >> (6 * s)) & 7u, j = (pk >> (6 * s + 3)) & 7u;
        const bool used = i != 7u;
        const uint32_t ii = used ? i : 0u, jj = used ? j : 0u;
        Fp2<C> a = d_coef<C>(L, g, (int)ii);
#if BBS_DIST_XI_AT_SOURCE
        const Fp2<C> b = d_coef<C>(L, f2_sel<C>(((xmask >> s) & 1u) != 0, gx, g), (int)jj);
        a = f2_sel<C>(used, a, f2_zero<C>());
#else
        Fp2<C> b = d_coef<C>(L, g, (int)jj);
        a = f2_sel<C>(used, a, f2_zero<C>());
        b = f2_sel<C>(ii + jj >= (uint32_t)GRP, f2_mul_xi<C>(b), b);
#endif
        f2acc_mac_sh<C>(acc, a, b, (used && ii != jj) ? 1u : 0u);
    }
    return f2acc_finish<C>(acc);
#endif
}
#else
// f * h.  Lane m multiplies g_m by every h_j; product j belongs to w^(m+j): it is rotated to lane
// (m + j) mod 6 and, on the RECEIVING lane k, lands in the plain sum (k >= j) or in the sum that
// still has to be multiplied by xi (k < j, i.e. m + j >= 6).  Both sums are accumulated lazily
// (limb-wise, no reduction) and reduced once: 2 chains per product instead of 24.
template <class C, int V = 0>
__device__ __attribute__((noinline)) Fp2<C> d_mul(const Lane6& L, const Fp2<C>& g, const Fp2<C>& h) {
    using P = typename C::FpP;
    constexpr int N = P::N;
    uint32_t A0[N], A1[N], B0[N], B1[N];
#pragma unroll
    for (int i = 0; i < N; i++) { A0[i] = 0; A1[i] = 0; B0[i] = 0; B1[i] = 0; }
    for (int j = 0; j < GRP; j++) {
        Fp2<C> hj = d_coef<C>(L, h, j);
        Fp2<C> p = f2_mul<C>(g, hj);                       // g_m h_j -> w^(m+j)
        int src = L.m - j;                                  // lane k receives from lane (k - j) mod 6
        if (src < 0) src += GRP;
        Fp2<C> q = f2_shfl<C>(p, L.base + src);
        const bool wrap = L.m < j;
#pragma unroll
        for (int i = 0; i < N; i++) {
            A0[i] += wrap ? 0u : q.c0.v[i]; A1[i] += wrap ? 0u : q.c1.v[i];
            B0[i] += wrap ? q.c0.v[i] : 0u; B1[i] += wrap ? q.c1.v[i] : 0u;
        }
    }
    if constexpr (C::K::XI_C0 == 1) {
        // A + (1 + u) B = (A0 + B0 - B1) + (A1 + B1 + B0) u ; 6 negative terms compensated by 6 * BOUND * p
        Fp2<C> r;
        r.c0 = r28::reduce_fn<P>([&](int i) { return (int64_t)A0[i] + (int64_t)B0[i] - (int64_t)B1[i] + 6 * (int64_t)P::MODB[i]; }, 24);
        r.c1 = r28::reduce_fn<P>([&](int i) { return (int64_t)A1[i] + (int64_t)B1[i] + (int64_t)B0[i]; }, 12);
        return r;
    } else {
        Fp2<C> a, b;
        a.c0 = r28::reduce_fn<P>([&](int i) { return (int64_t)A0[i]; }, 6);
        a.c1 = r28::reduce_fn<P>([&](int i) { return (int64_t)A1[i]; }, 6);
        b.c0 = r28::reduce_fn<P>([&](int i) { return (int64_t)B0[i]; }, 6);
        b.c1 = r28::reduce_fn<P>([&](int i) { return (int64_t)B1[i]; }, 6);
        return f2_add<C>(a, f2_mul_xi<C>(b));
    }
}

// f^2 with the symmetry of squaring: lane m computes g_m^2 and g_m g_{m+d} for d = 1, 2, 3 (the d = 3
// products are needed from lanes 0..2 only): 1 Fp2 square + 3 Fp2 products instead of 6 products.
// Product (m, d) belongs to w^(m + (m+d)%6), i.e. to lane k = (2m + d) % 6 -- two senders (m0 and
// m0 + 3) per receiver and per d of the right parity -- with factor 2 for d > 0 and xi when the
// exponent wraps.  Lazy accumulation and final reduction as in d_mul.
template <class C>
BBS_DIST_MILLER Fp2<C> d_sqr(const Lane6& L, const Fp2<C>& g) {
    using P = typename C::FpP;
    constexpr int N = P::N;
    uint32_t A0[N], A1[N], B0[N], B1[N];
#pragma unroll
    for (int i = 0; i < N; i++) { A0[i] = 0; A1[i] = 0; B0[i] = 0; B1[i] = 0; }
    const int k = L.m;
    for (int d = 0; d < 4; d++) {
        Fp2<C> X;
        if (d == 0) {
            X = f2_sqr<C>(g);
        } else {
            int pj = L.m + d;
            if (pj >= GRP) pj -= GRP;
            X = f2_mul<C>(g, d_coef<C>(L, g, pj));
        }
        int kd = k - d;
        if (kd < 0) kd += GRP;
        const bool parity = (kd & 1) == 0;
        const int m0 = kd >> 1;                               // senders m0 and m0 + 3
        const Fp2<C> q0 = f2_shfl<C>(X, L.base + m0);
        const Fp2<C> q1 = f2_shfl<C>(X, L.base + m0 + 3);
        auto wraps = [&](int m) { int j = m + d; if (j >= GRP) j -= GRP; return m + j >= GRP; };
        const bool v0 = parity, v1 = parity && d < 3;
        const bool w0 = wraps(m0), w1 = wraps(m0 + 3);
        const uint32_t sh = d ? 1u : 0u;                      // factor 2 for the off-diagonal products
#pragma unroll
        for (int i = 0; i < N; i++) {
            const uint32_t a0 = q0.c0.v[i] << sh, a1 = q0.c1.v[i] << sh, b0 = q1.c0.v[i] << sh, b1 = q1.c1.v[i] << sh;
            A0[i] += ((v0 && !w0) ? a0 : 0u) + ((v1 && !w1) ? b0 : 0u);
            A1[i] += ((v0 && !w0) ? a1 : 0u) + ((v1 && !w1) ? b1 : 0u);
            B0[i] += ((v0 && w0) ? a0 : 0u) + ((v1 && w1) ? b0 : 0u);
            B1[i] += ((v0 && w0) ? a1 : 0u) + ((v1 && w1) ? b1 : 0u);
        }
    }
    if constexpr (C::K::XI_C0 == 1) {
        Fp2<C> r;
        r.c0 = r28::reduce_fn<P>([&](int i) { return (int64_t)A0[i] + (int64_t)B0[i] - (int64_t)B1[i] + 6 * (int64_t)P::MODB[i]; }, 24);
        r.c1 = r28::reduce_fn<P>([&](int i) { return (int64_t)A1[i] + (int64_t)B1[i] + (int64_t)B0[i]; }, 12);
        return r;
    } else {
        Fp2<C> a, b;
        a.c0 = r28::reduce_fn<P>([&](int i) { return (int64_t)A0[i]; }, 6);
        a.c1 = r28::reduce_fn<P>([&](int i) { return (int64_t)A1[i]; }, 6);
        b.c0 = r28::reduce_fn<P>([&](int i) { return (int64_t)B0[i]; }, 6);
        b.c1 = r28::reduce_fn<P>([&](int i) { return (int64_t)B1[i]; }, 6);
        return f2_add<C>(a, f2_mul_xi<C>(b));
    }
}

#endif

template <class C>
__device__ __forceinline__ Fp2<C> d_conj(const Lane6& L, const Fp2<C>& g) {   // w -> -w
    return f2_sel<C>((L.m & 1) != 0, f2_neg<C>(g), g);
}

// multiply by the line through the twist point evaluated at P (see pairing.hpp for the forms)
#if BBS_DIST_LAZY
// three sparse terms as ONE dot product: 3 + 3 + 2 N^2 multiply-accumulates + one reduction pair
template <class C>
BBS_DIST_MILLER Fp2<C> d_mul_line(const Lane6& L, const Fp2<C>& g, const LineEntry<C>& le, const G1Aff<C>& P) {
    // lx = nl * xP is the same on the six lanes of an item: each lane of a pair (m, m ^ 1) computes one of its two
    // components and fetches the other (one Fp product + 14 ds_bpermute instead of two products)
    const bool odd = (L.m & 1) != 0;
    // (inlined multiplier: the called one takes half of its operands on the stack -- a scratch round trip per line)
    const Fp<C> mine = fe_mul_i<typename C::FpP>(fe_select<typename C::FpP>(odd, le.nl.c1, le.nl.c0), P.x);
    const Fp<C> other = fe_shfl<typename C::FpP>(mine, L.base + (L.m ^ 1));
    const Fp2<C> lx = {fe_select<typename C::FpP>(odd, other, mine), fe_select<typename C::FpP>(odd, mine, other)};
    const int k = L.m;
    // l = c + lx w^2 + yP w^3 (M twist) :  c_k = g_k c + xi^[k<2] g_{k-2} lx + xi^[k<3] g_{k-3} yP
    // l = yP + lx w + c w^3   (D twist) :  c_k = g_k yP + xi^[k<1] g_{k-1} lx + xi^[k<3] g_{k-3} c
    constexpr int SA = C::K::TWIST_M ? 2 : 1;
#if BBS_DIST_XI_AT_SOURCE
    // g_(k - SA) is read from lane k - SA + 6 exactly when it wraps: lanes >= 6 - SA publish xi g for the first fetch, lanes
    // >= 3 for the second
    const Fp2<C> gx = f2_mul_xi<C>(g);
    const Fp2<C> a = d_coef<C>(L, f2_sel<C>(k >= GRP - SA, gx, g), k < SA ? k + GRP - SA : k - SA);
    const Fp2<C> b = d_coef<C>(L, f2_sel<C>(k >= 3, gx, g), k < 3 ? k + 3 : k - 3);
#else
    Fp2<C> a = d_coef<C>(L, g, k < SA ? k + GRP - SA : k - SA);
    Fp2<C> b = d_coef<C>(L, g, k < 3 ? k + 3 : k - 3);
    a = f2_sel<C>(k < SA, f2_mul_xi<C>(a), a);
    b = f2_sel<C>(k < 3, f2_mul_xi<C>(b), b);
#endif
#if BBS_DIST_TWOPASS_MILLER
    F2AccLo<C> lo;
    f2acc_lo_zero<C>(lo);
    if constexpr (C::K::TWIST_M) {
        f2acc2_mac_sh<C, false>(lo, g, le.c, 0u); f2acc2_mac_sh<C, false>(lo, a, lx, 0u); f2acc2_mac_fp<C, false>(lo, b, P.y);
    } else {
        f2acc2_mac_fp<C, false>(lo, g, P.y); f2acc2_mac_sh<C, false>(lo, a, lx, 0u); f2acc2_mac_sh<C, false>(lo, b, le.c, 0u);
    }
    F2AccMid<C> mid;
    f2acc_finish_lo<C>(lo, mid);
    F2AccHi<C> hi;
    f2acc_hi_zero<C>(hi);
    if constexpr (C::K::TWIST_M) {
        f2acc2_mac_sh<C, true>(hi, g, le.c, 0u); f2acc2_mac_sh<C, true>(hi, a, lx, 0u); f2acc2_mac_fp<C, true>(hi, b, P.y);
    } else {
        f2acc2_mac_fp<C, true>(hi, g, P.y); f2acc2_mac_sh<C, true>(hi, a, lx, 0u); f2acc2_mac_sh<C, true>(hi, b, le.c, 0u);
    }
    return f2acc_finish_hi<C>(hi, mid);
#else
    F2Acc<C> acc;
    f2acc_zero<C>(acc);
    if constexpr (C::K::TWIST_M) {
        f2acc_mac<C, 1>(acc, g, le.c);
        f2acc_mac<C, 1>(acc, a, lx);
        f2acc_mac_fp<C>(acc, b, P.y);
    } else {
        f2acc_mac_fp<C>(acc, g, P.y);
        f2acc_mac<C, 1>(acc, a, lx);
        f2acc_mac<C, 1>(acc, b, le.c);
    }
    return f2acc_finish<C>(acc);
#endif
}
#else
template <class C>
BBS_DIST_MILLER Fp2<C> d_mul_line(const Lane6& L, const Fp2<C>& g, const LineEntry<C>& le, const G1Aff<C>& P) {
    using FPp = typename C::FpP;
    Fp2<C> lx = f2_mul_fp<C>(le.nl, P.x);
    int s1, s2;
    Fp2<C> t0, t1, t2;
    if constexpr (C::K::TWIST_M) {
        // l = c + lx w^2 + yP w^3 :  c_k = g_k c + xi^[k<2] g_{k-2} lx + xi^[k<3] g_{k-3} yP
        s1 = L.m - 2; s2 = L.m - 3;
        t0 = f2_mul<C>(g, le.c);
        Fp2<C> a = d_coef<C>(L, g, s1 < 0 ? s1 + GRP : s1);
        Fp2<C> b = d_coef<C>(L, g, s2 < 0 ? s2 + GRP : s2);
        t1 = f2_mul<C>(a, lx);
        t2 = f2_mul_fp<C>(b, P.y);
    } else {
        // l = yP + lx w + c w^3    :  c_k = g_k yP + xi^[k<1] g_{k-1} lx + xi^[k<3] g_{k-3} c
        s1 = L.m - 1; s2 = L.m - 3;
        t0 = f2_mul_fp<C>(g, P.y);
        Fp2<C> a = d_coef<C>(L, g, s1 < 0 ? s1 + GRP : s1);
        Fp2<C> b = d_coef<C>(L, g, s2 < 0 ? s2 + GRP : s2);
        t1 = f2_mul<C>(a, lx);
        t2 = f2_mul<C>(b, le.c);
    }
    t1 = f2_sel<C>(s1 < 0, f2_mul_xi<C>(t1), t1);
    t2 = f2_sel<C>(s2 < 0, f2_mul_xi<C>(t2), t2);
    (void)sizeof(FPp);
    return f2_add<C>(f2_add<C>(t0, t1), t2);
}

#endif

// Granger-Scott squaring of a cyclotomic element: Fp4 pairs (g_m, g_{m+3}), m = 0,1,2
//   X = (x0, x1):  X^2 = (x0^2 + xi x1^2, 2 x0 x1)
//   g0' = 3 A2[0] - 2 g0   g3' = 3 A2[1] + 2 g3
//   g1' = 3 xi C2[1] + 2 g1   g4' = 3 C2[0] - 2 g4
//   g2' = 3 B2[0] - 2 g2   g5' = 3 B2[1] + 2 g5
template <class C>
BBS_DIST_CYCLO Fp2<C> d_cyclo_sqr(const Lane6& L, const Fp2<C>& g) {
    const bool hi = L.m >= 3;
    const int partner = hi ? L.m - 3 : L.m + 3;
    Fp2<C> px = d_coef<C>(L, g, partner);
    Fp2<C> sq;                                              // lane m<3: X2[0] of pair m ; m>=3: X2[1] of pair m-3
#if BBS_DIST_LAZY
    if constexpr (C::K::XI_C0 == 1) {                       // BN254 (xi = 9 + u, 10 limbs): measured no faster, old form kept
#if BBS_DIST_TWOPASS_FINAL
        sq = fp4_sqr_part2<C>(hi, g, px);                   // the same in two passes over the columns
#else
        sq = fp4_sqr_part<C>(hi, g, px);                    // tower.hpp: four column products, one reduction pair
#endif
    } else
#endif
    {
        Fp2<C> u = f2_sqr<C>(g);                            // x0^2 on the low lane, x1^2 on the high lane
        Fp2<C> v = f2_sqr<C>(f2_add<C>(g, px));             // (x0 + x1)^2 on both
        Fp2<C> pu = d_coef<C>(L, u, partner);
        // low lane: x0^2 + xi x1^2 ; high lane: 2 x0 x1 = (x0+x1)^2 - x0^2 - x1^2   (one chain per component)
        Fp2<C> lo_val, hi_val;
        if constexpr (C::K::XI_C0 == 1) {
            lo_val = {fe_lin<typename C::FpP, 1, 1, -1>(u.c0, pu.c0, pu.c1), fe_lin<typename C::FpP, 1, 1, 1>(u.c1, pu.c0, pu.c1)};
        } else {
            lo_val = f2_add<C>(u, f2_mul_xi<C>(pu));
        }
        hi_val = f2_lin<C, 1, -1, -1>(v, u, pu);
        sq = f2_sel<C>(hi, hi_val, lo_val);
    }
    // pull pattern: 0<-0, 3<-3, 1<-5, 4<-2, 2<-1, 5<-4
    const int src = (L.m == 0) ? 0 : (L.m == 3) ? 3 : (L.m == 1) ? 5 : (L.m == 4) ? 2 : (L.m == 2) ? 1 : 4;
    Fp2<C> t = d_coef<C>(L, sq, src);
    t = f2_sel<C>(L.m == 1, f2_mul_xi<C>(t), t);
    // odd lanes: 3 t + 2 g ; even lanes: 3 t - 2 g
    return f2_lin_pm<C, 3, 2>(t, g, (L.m & 1) != 0);
}

template <class C, int K, int V = 0>
__device__ __attribute__((noinline)) Fp2<C> d_frob(const Lane6& L, const Fp2<C>& g, const uint32_t* frob_tab) {
    // frob_tab: [3][6][2][N] Montgomery constants xi^(m (p^K - 1)/6)
    constexpr int N = C::FpP::N;
    Fp2<C> co;
    const uint32_t* t = frob_tab + ((size_t)(K - 1) * 6 + L.m) * 2 * N;
#pragma unroll
    for (int j = 0; j < N; j++) { co.c0.v[j] = t[j]; co.c1.v[j] = t[N + j]; }
    Fp2<C> x = (K & 1) ? f2_conj<C>(g) : g;
    return f2_mul<C>(x, co);
}

// gather the whole element on every lane (tower layout: c0 = (g0, g2, g4), c1 = (g1, g3, g5))
template <class C>
__device__ __attribute__((noinline)) Fp12<C> d_gather(const Lane6& L, const Fp2<C>& g) {
    Fp12<C> f;
    f.c0.c0 = d_coef<C>(L, g, 0); f.c1.c0 = d_coef<C>(L, g, 1);
    f.c0.c1 = d_coef<C>(L, g, 2); f.c1.c1 = d_coef<C>(L, g, 3);
    f.c0.c2 = d_coef<C>(L, g, 4); f.c1.c2 = d_coef<C>(L, g, 5);
    return f;
}
template <class C>
__device__ __forceinline__ Fp2<C> d_scatter(const Lane6& L, const Fp12<C>& f) {
    Fp2<C> r = f.c0.c0;
    r = f2_sel<C>(L.m == 1, f.c1.c0, r);
    r = f2_sel<C>(L.m == 2, f.c0.c1, r);
    r = f2_sel<C>(L.m == 3, f.c1.c1, r);
    r = f2_sel<C>(L.m == 4, f.c0.c2, r);
    r = f2_sel<C>(L.m == 5, f.c1.c2, r);
    return r;
}

// 1/f = conj6(f) * N^-1 with N = f * conj6(f) in Fp6 = Fp2[v]/(v^3 - xi), v = w^2: N has only the even
// coefficients (lanes 0, 2, 4).  Every lane fetches (n0, n1, n2), inverts the cubic extension
// element redundantly (one Fp inversion, the only long chain), then multiplies its rotated
// coefficients of conj6(f) by the three coefficients of N^-1.  No lane ever holds a whole Fp12.
template <class C, int V = 0>
__device__ __attribute__((noinline)) Fp2<C> d_inv(const Lane6& L, const Fp2<C>& g) {
    const Fp2<C> gc = d_conj<C>(L, g);
    const Fp2<C> nn = d_mul<C, V>(L, g, gc);
    const Fp2<C> n0 = d_coef<C>(L, nn, 0), n1 = d_coef<C>(L, nn, 2), n2 = d_coef<C>(L, nn, 4);
    Fp2<C> t0 = f2_sub<C>(f2_sqr<C>(n0), f2_mul_xi<C>(f2_mul<C>(n1, n2)));
    Fp2<C> t1 = f2_sub<C>(f2_mul_xi<C>(f2_sqr<C>(n2)), f2_mul<C>(n0, n1));
    Fp2<C> t2 = f2_sub<C>(f2_sqr<C>(n1), f2_mul<C>(n0, n2));
    Fp2<C> d = f2_add<C>(f2_mul<C>(n0, t0), f2_mul_xi<C>(f2_add<C>(f2_mul<C>(n2, t1), f2_mul<C>(n1, t2))));
    const Fp2<C> di = f2_inv<C>(d);
    t0 = f2_mul<C>(t0, di);      // N^-1 = t0 + t1 v + t2 v^2 = t0 + t1 w^2 + t2 w^4
    t1 = f2_mul<C>(t1, di);
    t2 = f2_mul<C>(t2, di);
    // (gc * N^-1)_k = gc_k t0 + xi^[k<2] gc_{k-2} t1 + xi^[k<4] gc_{k-4} t2
    const int s1 = L.m - 2, s2 = L.m - 4;
    Fp2<C> a = d_coef<C>(L, gc, s1 < 0 ? s1 + GRP : s1);
    Fp2<C> b = d_coef<C>(L, gc, s2 < 0 ? s2 + GRP : s2);
    Fp2<C> r = f2_mul<C>(gc, t0);
    Fp2<C> u = f2_mul<C>(a, t1);
    u = f2_sel<C>(s1 < 0, f2_mul_xi<C>(u), u);
    Fp2<C> w = f2_mul<C>(b, t2);
    w = f2_sel<C>(s2 < 0, f2_mul_xi<C>(w), w);
    return f2_add<C>(f2_add<C>(r, u), w);
}

template <class C>
__device__ __forceinline__ Fp2<C> d_one(const Lane6& L) { return f2_sel<C>(L.m == 0, f2_one<C>(), f2_zero<C>()); }

template <class C>
__device__ __forceinline__ bool d_is_one(const Lane6& L, const Fp2<C>& g) {
    int ok = f2_eq<C>(g, d_one<C>(L)) ? 1 : 0;
    int all = 1;
    for (int j = 0; j < GRP; j++) all &= __shfl(ok, L.base + j, 64);
    return all != 0;
}

// f^|x| (f cyclotomic).  The running value must stay in registers across the 63 squarings.  Two things used to make
// it a memory object for the whole loop, so that every squaring began with a scratch round trip (7 loads, wait, ...,
// 7 stores): being the function's return slot, and being handed by reference to the non-inlined multiplication.
// Hence the explicit output parameter and the copy `t`.
template <class C, int V = 0>
__device__ __attribute__((noinline)) void d_pow_xabs_to(const Lane6& L, const Fp2<C>& f_in, Fp2<C>& out) {
    const Fp2<C> f = f_in;
    Fp2<C> r = f;
    const uint64_t x = C::K::X_ABS;
    int top = 63;
    while (!((x >> top) & 1)) top--;
    for (int i = top - 1; i >= 0; i--) {
        r = d_cyclo_sqr<C>(L, r);
        if ((x >> i) & 1) {
            const Fp2<C> t = r, h = f;
            r = d_mul<C, V>(L, t, h);
        }
    }
    out = r;
}
template <class C, int V = 0>
__device__ __forceinline__ Fp2<C> d_pow_xabs(const Lane6& L, const Fp2<C>& f) {
    Fp2<C> r;
    d_pow_xabs_to<C, V>(L, f, r);
    return r;
}
template <class C, int V = 0>
__device__ __forceinline__ Fp2<C> d_pow_x(const Lane6& L, const Fp2<C>& f) {
    Fp2<C> r = d_pow_xabs<C, V>(L, f);
    if constexpr (C::K::X_NEG) r = d_conj<C>(L, r);
    return r;
}

// same exponent as final_exponentiation() in pairing.hpp
// V: instance tag.  A device function is compiled for the LOOSEST register budget among the kernels that reach it; the
// final-exponentiation kernel that runs two wavefronts per SIMD (PairFinalDist, <= 256 registers) therefore has its own
// instances (V = 1) of every non-inlined function below it, apart from the ones the one-wavefront kernels call.
template <class C, int V = 0>
__device__ __attribute__((noinline)) Fp2<C> d_final_exp(const Lane6& L, const Fp2<C>& f_in, const uint32_t* frob_tab) {
    Fp2<C> f = d_mul<C, V>(L, d_conj<C>(L, f_in), d_inv<C, V>(L, f_in));
    f = d_mul<C, V>(L, d_frob<C, 2, V>(L, f, frob_tab), f);
    if constexpr (C::ID == 0) {
        Fp2<C> a = d_mul<C, V>(L, d_pow_x<C, V>(L, f), d_conj<C>(L, f));
        a = d_mul<C, V>(L, d_pow_x<C, V>(L, a), d_conj<C>(L, a));
        Fp2<C> b = d_mul<C, V>(L, d_pow_x<C, V>(L, a), d_frob<C, 1, V>(L, a, frob_tab));
        Fp2<C> c = d_pow_x<C, V>(L, d_pow_x<C, V>(L, b));
        c = d_mul<C, V>(L, c, d_frob<C, 2, V>(L, b, frob_tab));
        c = d_mul<C, V>(L, c, d_conj<C>(L, b));
        Fp2<C> f3 = d_mul<C, V>(L, d_cyclo_sqr<C>(L, f), f);
        return d_mul<C, V>(L, c, f3);
    } else {
        Fp2<C> fu = d_pow_x<C, V>(L, f);
        Fp2<C> fu2 = d_pow_x<C, V>(L, fu);
        Fp2<C> fu3 = d_pow_x<C, V>(L, fu2);
        Fp2<C> y0 = d_mul<C, V>(L, d_mul<C, V>(L, d_frob<C, 1, V>(L, f, frob_tab), d_frob<C, 2, V>(L, f, frob_tab)), d_frob<C, 3, V>(L, f, frob_tab));
        Fp2<C> y1 = d_conj<C>(L, f);
        Fp2<C> y2 = d_frob<C, 2, V>(L, fu2, frob_tab);
        Fp2<C> y3 = d_conj<C>(L, d_frob<C, 1, V>(L, fu, frob_tab));
        Fp2<C> y4 = d_conj<C>(L, d_mul<C, V>(L, fu, d_frob<C, 1, V>(L, fu2, frob_tab)));
        Fp2<C> y5 = d_conj<C>(L, fu2);
        Fp2<C> y6 = d_conj<C>(L, d_mul<C, V>(L, fu3, d_frob<C, 1, V>(L, fu3, frob_tab)));
        Fp2<C> t0 = d_cyclo_sqr<C>(L, y6);
        t0 = d_mul<C, V>(L, t0, y4);
        t0 = d_mul<C, V>(L, t0, y5);
        Fp2<C> t1 = d_mul<C, V>(L, y3, y5);
        t1 = d_mul<C, V>(L, t1, t0);
        t0 = d_mul<C, V>(L, t0, y2);
        t1 = d_cyclo_sqr<C>(L, t1);
        t1 = d_mul<C, V>(L, t1, t0);
        t1 = d_cyclo_sqr<C>(L, t1);
        t0 = d_mul<C, V>(L, t1, y1);
        t1 = d_mul<C, V>(L, t1, y0);
        t0 = d_cyclo_sqr<C>(L, t0);
        return d_mul<C, V>(L, t0, t1);
    }
}

}  // namespace bbs
#endif
