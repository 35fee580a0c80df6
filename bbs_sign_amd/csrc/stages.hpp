// Per-item device stages of the four batched BBS+ core operations.
//
// Every stage is a __host__ __device__ "item function" run by one GPU lane; the __global__
// wrappers live in capi.hip.  Batch data is SoA in HBM: word w of item i of an array is at
// base[w * n + i], so the 64 lanes of a wavefront (64 consecutive items) read 256 contiguous
// bytes per word -- coalesced.
//
// Work split (MI355X: 1024 SIMDs, a batch of 4096 items is only 64 wavefronts, so each item is
// split over several lanes wherever the algebra allows):
//   * a multi-scalar multiplication is cut into PARTS -- a variable-base scalar multiplication (or the
//     joint multiplication of T1's three terms), plus NFIX chunks of the fixed-base (windowed,
//     precomputed-table) sum -- every part on its own lane (part-major thread index: a wavefront runs
//     one kind of part);
//   * a pairing product is sliced over six lanes per item (pairing_dist.hpp, stage PairDist); the
//     one-lane PairMiller / PairFinal stages below serve the CPU-side test build only.
//
// Algebraic restructurings (bit-identical group elements / booleans):
//   proof_verify: T2 = Bv*c + D*r3^ + sum H_j m^_j  with  Bv = P1 + Q1*domain + sum H_i m_i
//                 (src/proof_verify.rs:165-182) is evaluated as ONE fixed-base sum over
//                 {P1, Q1, H_*} with scalars {c, domain*c, m_i*c | m^_j} plus D*r3^.
//   verify      : e(A, W + e*BP2) * e(B, -BP2) == 1  (src/verify.rs:88-92)
//                 <=>  e(A, W) * e(e*A - B, BP2) == 1 : both G2 arguments fixed.
//   sign        : A = B * (sk+e)^-1 (src/sign.rs:128-130) = fixed-base sum with scalars * inv.
//   proof_gen   : D, Abar, Bbar, T1, T2 (src/proof_gen.rs:249-263) as fixed-base sums over B's
//                 terms plus scalar multiples of the signature point A.
#pragma once
#include "pairing.hpp"
#include "pairing_dist.hpp"
#include "sha256.hpp"

namespace bbs {

constexpr int NFIX = 8;          // fixed-base chunks per MSM (one lane each)
constexpr int MAX_DST = 255;
// Internal per-item states.  Neither is a value the C ABI may return: an item that no kernel has decided stays at one of
// them and bbs_job_fetch_status / bbs_job_wait then fail with BBS_E_STATE instead of reporting Ok(true) (fail closed).
constexpr int8_t ST_PENDING = -128;   // accepted by validation, nothing computed yet
constexpr int8_t ST_PAIRING = -127;   // every check before the pairing passed: the pairing product decides
#ifndef BBS_PAIR_WAVES
#define BBS_PAIR_WAVES 1
#endif
// A/B knobs (round 4): the throughput form of a per-item pairing check as TWO kernels -- PairMillerBoth (both Miller loops of
// an item on one six-lane group, shared squarings: the first half of PairDist) and PairFinalDist (the final exponentiation)
// -- so that the final exponentiation, which alone fits 256 registers with 6 spilled and the same instruction count
// (profiles/r04_c_occupancy_resource_usage.txt), can run TWO wavefronts per SIMD (BBS_PAIRFINAL_WAVES = 2) while the Miller
// loop (277 spills under that cap) keeps the whole register file.  Measured (profiles/r04_d_ab.log, two alternating repeats,
// 96 steps): fused 1.511 / 1.513 M proof_verify/s, split with one wavefront per SIMD 1.511 / 1.513, split with two 1.490 /
// 1.483 -- and a batch that is alone gets SLOWER (final exponentiation 2.45 -> 3.1 - 4.0 ms: the dispatcher packs two
// wavefronts onto one SIMD while others idle).  Two co-resident wavefronts do not issue faster here: in time units the
// kernels already run at 2.28 ns per wave-instruction per SIMD against 2.18 ns for this opcode mix at two wavefronts per
// SIMD (tools/ubench, profiles/r03_p_ubench_valu_int.csv; the chip clocks down as more wavefronts issue).  Hence: fused.
#ifndef BBS_PAIR_SPLIT2
#define BBS_PAIR_SPLIT2 0
#endif
#ifndef BBS_PAIRFINAL_WAVES
#define BBS_PAIRFINAL_WAVES 1
#endif
#ifndef BBS_MSM_WAVES
#define BBS_MSM_WAVES 1          // multi-scalar-multiplication stages (2 and 3 measured: no gain, spills)
#endif
// The doubling-chain kernels of proof_verify capped at 256 registers, so that two of their wavefronts -- or one and a
// wavefront of a fixed-base chunk kernel (246) -- share a SIMD.  They are 64 wavefronts of 3 - 5 ms each per batch: alone on a
// SIMD they keep the other 212 registers of it idle for that long.  Round 5, measured on the headline loop (profiles/r05_i_*,
// three alternating repeats): BLS12-381 T1 chain 300 -> 256 registers (97 spilled) long_region 1.581 -> 1.606 M/s; the
// single multiplication 354 -> 256 as well (251 spilled) no further gain (1.60 M) -- it stays at one.  BN254's three kernels
// need 244 / 264 / 266: capped, 0 / 10 / 26 spilled; likewise BN254's PvChallenge(Bv), VfVarMul, MsmVarMul (264 - 268).
#ifndef BBS_T1_WAVES
#define BBS_T1_WAVES 2
#endif
#ifndef BBS_VARMUL_WAVES
#define BBS_VARMUL_WAVES 1       // BLS12-381; BN254: 2 (chain_waves below)
#endif
#ifndef BBS_BN_CHAIN_WAVES
#define BBS_BN_CHAIN_WAVES 2
#endif
template <class C> constexpr int chain_waves(int bls_default) { return C::FpP::N <= 10 ? BBS_BN_CHAIN_WAVES : bls_default; }

// ---- context constants resident in HBM ------------------------------------------------------
struct HashCtx {
    uint32_t dom_mid[8];         // SHA-256 state after Z_pad || domain prefix, at a block boundary
    uint64_t dom_mid_total;
    uint8_t dom_tail[64];
    uint32_t dom_tail_len;
    uint8_t dst_h2s[256];        // api_id || "H2S_"
    uint32_t dst_h2s_len;
};

template <class C>
struct CtxConsts {
    HashCtx hash;
    G1Aff<C> p1;                 // Montgomery form
    int L;                       // number of message generators
    int n_bases;                 // L + 2 : P1, Q1, H_1..H_L
    int win_bits;                // c
    int n_windows;               // W = ceil(256 / c)
    uint32_t fix_bias[8];        // K = sum over w < W - 1 of 2^(c w + c - 1): signed-digit recoding of the fixed-base scalars
    const uint32_t* tables;      // [base][window][|digit| - 1][fix_tab_stride] affine Montgomery, |digit| in 1 .. 2^(c-1)
    uint32_t frob[3][6][2][C::FpP::N];   // xi^(m (p^k - 1)/6), Montgomery (for the lane-sliced Fp12)
    MillerSchedule sched;
    LineTable<C> tab_pk;         // lines of W = pk
    LineTable<C> tab_bp2;        // lines of BP2
};

// Words from one entry of the fixed-base window tables to the next.  An entry is 2N words (x, y); BLS12-381's 112 bytes are
// padded to 128 (round 5): the tables are read at random, one entry per mixed addition, and an unaligned 112-byte entry
// straddles two 128-byte lines in 7 cases of 8 -- the counters showed 2 x 205 MB fetched per 4096-item batch for 203 MB of
// entries (profiles/r05_p_pmc.csv before the change).  Aligned, an entry is one line and seven 16-byte loads.  BN254's 80
// bytes stay packed (16-byte aligned; padding them to 128 would cost 60 % more table memory).
#ifndef BBS_FIX_TAB_PACKED
#define BBS_FIX_TAB_PACKED 0     // A/B knob: 1 = entries packed at 2N words as before round 5
#endif
template <class C>
constexpr int fix_tab_stride() { return (!BBS_FIX_TAB_PACKED && 2 * C::FpP::N == 28) ? 32 : 2 * C::FpP::N; }
// one entry (16-byte loads: every entry starts on a 16-byte boundary)
template <class C>
BBS_HD void fix_tab_load(const uint32_t* e, G1Aff<C>& q) {
    constexpr int N = C::FpP::N;
    static_assert((2 * N) % 4 == 0 && fix_tab_stride<C>() % 4 == 0, "table entries are whole 16-byte groups");
    uint32_t w[2 * N];
#pragma unroll
    for (int g = 0; g < 2 * N / 4; g++) {
        const uint4 v = reinterpret_cast<const uint4*>(e)[g];
        w[4 * g] = v.x; w[4 * g + 1] = v.y; w[4 * g + 2] = v.z; w[4 * g + 3] = v.w;
    }
#pragma unroll
    for (int j = 0; j < N; j++) { q.x.v[j] = w[j]; q.y.v[j] = w[N + j]; }
}

// ---- SoA helpers ----------------------------------------------------------------------------
template <int NW>
BBS_HD void soa_ld(const uint32_t* base, size_t n, size_t i, uint32_t* out) {
#pragma unroll
    for (int w = 0; w < NW; w++) out[w] = base[(size_t)w * n + i];
}
template <int NW>
BBS_HD void soa_st(uint32_t* base, size_t n, size_t i, const uint32_t* v) {
#pragma unroll
    for (int w = 0; w < NW; w++) base[(size_t)w * n + i] = v[w];
}

template <class C>
BBS_HD Fr<C> fr_load_canon(const uint32_t* base, size_t n, size_t i) {   // canonical limbs, no conversion
    Fr<C> r;
    soa_ld<8>(base, n, i, r.v);
    return r;
}
template <class C>
BBS_HD Fr<C> fr_to_mont(const Fr<C>& canon) { return fe_from_limbs<typename C::FrP>(canon.v); }

// canonical affine point: 2 * NC 32-bit words per item (x then y)
template <class C>
BBS_HD G1Aff<C> g1a_load_canon_to_mont(const uint32_t* base, size_t n, size_t i) {
    constexpr int NC = C::FpP::NC;
    uint32_t w[2 * NC];
    soa_ld<2 * NC>(base, n, i, w);
    G1Aff<C> p;
    p.x = fe_from_words<typename C::FpP>(w);
    p.y = fe_from_words<typename C::FpP>(w + NC);
    return p;
}
template <class C>
BBS_HD G1Aff<C> g1a_load_mont(const uint32_t* base, size_t n, size_t i) {
    constexpr int N = C::FpP::N;
    G1Aff<C> p;
    soa_ld<N>(base, n, i, p.x.v);
    soa_ld<N>(base + (size_t)N * n, n, i, p.y.v);
    return p;
}
template <class C>
BBS_HD void g1a_store_mont(uint32_t* base, size_t n, size_t i, const G1Aff<C>& p) {
    constexpr int N = C::FpP::N;
    soa_st<N>(base, n, i, p.x.v);
    soa_st<N>(base + (size_t)N * n, n, i, p.y.v);
}
template <class C>
BBS_HD void g1a_store_canon(uint32_t* base, size_t n, size_t i, const G1Aff<C>& p) {
    constexpr int NC = C::FpP::NC;
    uint32_t x[NC], y[NC];
    fe_to_words<typename C::FpP>(p.x, x);
    fe_to_words<typename C::FpP>(p.y, y);
    soa_st<NC>(base, n, i, x);
    soa_st<NC>(base + (size_t)NC * n, n, i, y);
}
template <class C>
BBS_HD G1Jac<C> g1j_load(const uint32_t* base, size_t n, size_t i) {
    constexpr int N = C::FpP::N;
    G1Jac<C> p;
    soa_ld<N>(base, n, i, p.x.v);
    soa_ld<N>(base + (size_t)N * n, n, i, p.y.v);
    soa_ld<N>(base + (size_t)2 * N * n, n, i, p.z.v);
    return p;
}
template <class C>
BBS_HD void g1j_store(uint32_t* base, size_t n, size_t i, const G1Jac<C>& p) {
    constexpr int N = C::FpP::N;
    soa_st<N>(base, n, i, p.x.v);
    soa_st<N>(base + (size_t)N * n, n, i, p.y.v);
    soa_st<N>(base + (size_t)2 * N * n, n, i, p.z.v);
}

// ---- hashing helpers ------------------------------------------------------------------------
// ark-serialize compressed G1 absorbed into a hash (core_utilities.rs:39-47, proof_gen.rs:304-311)
template <class C>
__host__ __device__ inline void sha256_g1_compressed(Sha256& s, const G1Aff<C>& p) {
    using P = typename C::FpP;
    constexpr int N = P::NC;                     // canonical words
    const bool inf = g1a_is_inf<C>(p);
    struct { uint32_t v[P::NC]; } x, yw;
    fe_to_words<P>(p.x, x.v);
    fe_to_words<P>(p.y, yw.v);
    const bool ybig = words_gt_half<P>(yw.v);
    if constexpr (C::ID == 0) {
        // 48 bytes big-endian, flags in the first byte
        uint32_t flags = inf ? 0xC0000000u : (0x80000000u | (ybig ? 0x20000000u : 0u));
#pragma unroll
        for (int i = N - 1; i >= 0; i--) {
            uint32_t w = inf ? 0u : x.v[i];
            if (i == N - 1) w |= flags;
            sha256_word(s, w);
        }
    } else {
        // 32 bytes little-endian, flags in the last byte
        uint32_t flags = inf ? 0x40u : (ybig ? 0x80u : 0u);
#pragma unroll
        for (int i = 0; i < N; i++) {
            uint32_t l = inf ? 0u : x.v[i];
            uint32_t w = (l << 24) | ((l & 0xff00u) << 8) | ((l >> 8) & 0xff00u) | (l >> 24);   // bswap
            if (i == N - 1) w |= flags;
            sha256_word(s, w);
        }
    }
}

// calculate_domain (core_utilities.rs:24-63) from the cached prefix midstate
template <class C>
__host__ __device__ inline Fr<C> domain_from_header(const HashCtx& h, const uint8_t* hdr, uint32_t hdr_len) {
    Sha256 s;
    sha256_init_mid(s, h.dom_mid, h.dom_mid_total);
    sha256_bytes(s, h.dom_tail, h.dom_tail_len);
    sha256_u64be(s, hdr_len);
    sha256_bytes(s, hdr, hdr_len);
    uint32_t okm[12];
    xmd48_finish(s, h.dst_h2s, h.dst_h2s_len, okm);
    return fr_from_okm<C>(okm);
}

// ---- multi-scalar multiplication parts --------------------------------------------------------
// one chunk of the fixed-base sum: terms are (base k, window w) pairs, flattened index t = k*W + w,
// chunk f of NFIX handles t in [f*T/NFIX, (f+1)*T/NFIX).
// (result through `out`, the accumulator a plain local: where this function is not inlined, a named return value is
// the caller's memory and every addition of the loop would start with a scratch round trip -- DESIGN.md 5 rule 7b)
// SIGNED digits (round 3): a table holds 2^(c-1) entries per (base, window) instead of 2^c - 1 -- half the memory, half the
// build time, the same number of additions.  The scalar s < r < 2^255 is biased once, sb = s + K with
// K = sum_{w < W-1} 2^(c w + c - 1) (no carry chain between windows: one 256-bit addition per scalar); window w < W - 1 then
// contributes the digit  ((sb >> c w) mod 2^c) - 2^(c-1)  in [-2^(c-1), 2^(c-1) - 1], the top window  sb >> c (W - 1)  in
// [0, 2^(c-1)] (it holds at most c - 1 bits of s plus the carry), and  sum_w digit_w 2^(c w) = sb - K = s.  A negative
// digit adds the NEGATED table entry (y -> -y).
template <class C>
BBS_HD void fixed_bias_scalar(const CtxConsts<C>& cc, uint32_t* sc) {
    uint64_t cy = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) { cy += (uint64_t)sc[j] + cc.fix_bias[j]; sc[j] = (uint32_t)cy; cy >>= 32; }
}
// window w of a biased scalar: |digit| (0 = nothing to add) and its sign
BBS_HD uint32_t fixed_digit(const uint32_t* sb, int w, int c, int W, bool& neg) {
    const int bit = w * c;
    const int li = bit >> 5, sh = bit & 31;
    uint64_t two = sb[li];
    if (li + 1 < 8) two |= (uint64_t)sb[li + 1] << 32;
    const uint32_t half = 1u << (c - 1);
    uint32_t raw = (uint32_t)(two >> sh);
    if (w == W - 1) {
        // top window: the 256 - c (W - 1) <= c remaining bits (nothing is loaded from beyond bit 256).  A canonical scalar
        // gives raw <= 2^(c-1); clamped so that a non-canonical one could never index past the table
        neg = false;
        return raw > half ? half : raw;
    }
    raw &= (half << 1) - 1u;
    neg = raw < half;
    return neg ? half - raw : raw - half;
}

template <class C>
__host__ __device__ inline void fixed_msm_chunk_to(const CtxConsts<C>& cc, const uint32_t* fscal, size_t n, size_t i,
                                                   int n_terms, int chunk, G1Jac<C>& out) {
    constexpr int N = C::FpP::N;
    const int W = cc.n_windows, c = cc.win_bits;
    const int T = n_terms * W;
    const int t0 = (int)(((long long)T * chunk) / NFIX), t1 = (int)(((long long)T * (chunk + 1)) / NFIX);
    const size_t per_win = (size_t)1 << (c - 1);
    G1Jac<C> acc = g1j_inf<C>();
    int k_cur = -1;
    uint32_t sc[8];
    // table entry of term t (false: digit 0, nothing to add)
    auto fetch = [&](int t, G1Aff<C>& q) -> bool {
        const int k = t / W, w = t - k * W;
        if (k != k_cur) { soa_ld<8>(fscal + (size_t)k * 8 * n, n, i, sc); fixed_bias_scalar<C>(cc, sc); k_cur = k; }
        bool neg;
        const uint32_t d = fixed_digit(sc, w, c, W, neg);
        if (d == 0) return false;
        fix_tab_load<C>(cc.tables + (((size_t)k * W + w) * per_win + (d - 1)) * fix_tab_stride<C>(), q);
        q.y = fe_select<typename C::FpP>(neg, fe_neg<typename C::FpP>(q.y), q.y);
        return true;
    };
    // the entry of term t + 1 is requested before the addition of term t: the (random, HBM) table read of one
    // term overlaps the ~11 multiplications of the previous one -- with one wavefront per SIMD nothing else hides it
    G1Aff<C> qn = g1a_inf<C>();
    bool hn = t0 < t1 ? fetch(t0, qn) : false;
    for (int t = t0; t < t1; t++) {
        const G1Aff<C> q = qn;
        const bool h = hn;
        hn = t + 1 < t1 ? fetch(t + 1, qn) : false;
        if (h) acc = g1j_add_aff<C>(acc, q);
    }
    out = acc;
}
template <class C>
BBS_HD G1Jac<C> fixed_msm_chunk(const CtxConsts<C>& cc, const uint32_t* fscal, size_t n, size_t i, int n_terms, int chunk) {
    G1Jac<C> r;
    fixed_msm_chunk_to<C>(cc, fscal, n, i, n_terms, chunk, r);
    return r;
}

// ---- the fixed-base sum as a tree of AFFINE additions (bbs_ctx_set_fixed_base_tree) -------------------------------------
// All T = n_terms * W table entries of an item are summed by ONE lane, pairwise, level by level.  The slopes of a level
// share one inversion (Montgomery's trick: prefix products on the way up, one fe_inv, back-substitution on the way
// down): 5M + 1S per addition instead of the 7M + 4S of a mixed Jacobian addition, ceil(log2 T) inversions per item.
// The points of a level live in HBM work arrays of the job ([slot][2N words][n items]: coalesced over the items of a
// wavefront); (0, 0) is the identity.  Every exceptional case of affine addition is resolved per pair: an identity
// operand (digit 0), equal points (doubling, slope 3x^2 / 2y -- caller-supplied generators may repeat), opposite points
// (identity).  The result is the same group element as the sum of the NFIX chunks of fixed_msm_chunk.
template <class C>
struct FixTreeWork {
    uint32_t* pts0;     // [T][2N][n]
    uint32_t* pts1;     // [ceil(T/2)][2N][n]
    uint32_t* pre;      // [floor(T/2)][N][n]   prefix products of a level
};

template <class C>
__host__ __device__ inline void fixed_msm_tree_to(const CtxConsts<C>& cc, const uint32_t* fscal, size_t n, size_t i, int n_terms,
                                                  const FixTreeWork<C>& wk, G1Jac<C>& out) {
    using P = typename C::FpP;
    constexpr int N = P::N;
    const int W = cc.n_windows, c = cc.win_bits;
    const int T = n_terms * W;
    const size_t per_win = (size_t)1 << (c - 1);
    auto ld = [&](const uint32_t* a, int slot) {
        G1Aff<C> q;
        const uint32_t* b = a + (size_t)slot * 2 * N * n + i;
#pragma unroll
        for (int j = 0; j < N; j++) { q.x.v[j] = b[(size_t)j * n]; q.y.v[j] = b[(size_t)(N + j) * n]; }
        return q;
    };
    auto st = [&](uint32_t* a, int slot, const G1Aff<C>& q) {
        uint32_t* b = a + (size_t)slot * 2 * N * n + i;
#pragma unroll
        for (int j = 0; j < N; j++) { b[(size_t)j * n] = q.x.v[j]; b[(size_t)(N + j) * n] = q.y.v[j]; }
    };
    // level 0: the table entries themselves (digit 0 -> identity).  The reads are random 112-byte HBM accesses and nothing
    // depends on them but the store behind them: four are in flight at a time (a load -> store chain per entry would pay
    // the full memory latency 442 times per item)
    {
        auto entry = [&](int t, bool& neg) -> const uint32_t* {
            const int k = t / W, w = t - k * W;
            uint32_t sc[8];
            soa_ld<8>(fscal + (size_t)k * 8 * n, n, i, sc);
            fixed_bias_scalar<C>(cc, sc);
            const uint32_t d = fixed_digit(sc, w, c, W, neg);
            return d ? cc.tables + (((size_t)k * W + w) * per_win + (d - 1)) * fix_tab_stride<C>() : nullptr;
        };
        constexpr int G = 4;
        for (int t0 = 0; t0 < T; t0 += G) {
            G1Aff<C> q[G];
#pragma unroll
            for (int g = 0; g < G; g++) {
                q[g] = g1a_inf<C>();
                bool neg = false;
                const uint32_t* e = t0 + g < T ? entry(t0 + g, neg) : nullptr;
                if (e) {
                    fix_tab_load<C>(e, q[g]);
                    q[g].y = fe_select<P>(neg, fe_neg<P>(q[g].y), q[g].y);
                }
            }
#pragma unroll
            for (int g = 0; g < G; g++) if (t0 + g < T) st(wk.pts0, t0 + g, q[g]);
        }
    }
    // what a pair needs: the denominator of its slope (1 when the result needs none), and how to finish it
    struct Pair { Fp<C> den, dy; bool trivial, dbl, pinf, qinf; };
    auto classify = [&](const G1Aff<C>& p, const G1Aff<C>& q) {
        Pair r;
        r.pinf = g1a_is_inf<C>(p); r.qinf = g1a_is_inf<C>(q);
        const Fp<C> dx = fe_sub<P>(q.x, p.x);
        r.dy = fe_sub<P>(q.y, p.y);
        const bool same_x = fe_is_zero<P>(dx), same_y = fe_is_zero<P>(r.dy);
        r.trivial = r.pinf | r.qinf | (same_x & !same_y);            // the other operand, or P + (-P) = identity
        r.dbl = !r.pinf & !r.qinf & same_x & same_y;                 // P + P (y != 0: no point of order two on these curves)
        r.den = fe_select<P>(r.trivial, fe_one<P>(), fe_select<P>(r.dbl, fe_dbl<P>(p.y), dx));
        return r;
    };
    uint32_t* src = wk.pts0;
    uint32_t* dst = wk.pts1;
    int m = T;
    while (m > 1) {
        const int h = m >> 1;
        // up: prefix products of the denominators (pre[j] = product of den_0 .. den_{j-1})
        Fp<C> acc = fe_one<P>();
        {
            G1Aff<C> pn = ld(src, 0), qn = ld(src, 1);
            for (int j = 0; j < h; j++) {
                const G1Aff<C> p = pn, q = qn;
                if (j + 1 < h) { pn = ld(src, 2 * j + 2); qn = ld(src, 2 * j + 3); }     // requested one pair ahead
                const Pair pr = classify(p, q);
                uint32_t* b = wk.pre + (size_t)j * N * n + i;
#pragma unroll
                for (int l = 0; l < N; l++) b[(size_t)l * n] = acc.v[l];
                acc = fe_mul_i<P>(acc, pr.den);
            }
        }
        Fp<C> inv = fe_inv<P>(acc);                                   // never zero: every den is non-zero by construction
        // down: slope of pair j = num_j * inv(den_j), inv(den_j) = pre[j] * inv(den_0 .. den_j)
        {
            G1Aff<C> pn = ld(src, 2 * h - 2), qn = ld(src, 2 * h - 1);
            Fp<C> pren;
            { const uint32_t* b = wk.pre + (size_t)(h - 1) * N * n + i;
#pragma unroll
              for (int l = 0; l < N; l++) pren.v[l] = b[(size_t)l * n]; }
            for (int j = h - 1; j >= 0; j--) {
                const G1Aff<C> p = pn, q = qn;
                const Fp<C> pre_j = pren;
                if (j > 0) {
                    pn = ld(src, 2 * j - 2); qn = ld(src, 2 * j - 1);
                    const uint32_t* b = wk.pre + (size_t)(j - 1) * N * n + i;
#pragma unroll
                    for (int l = 0; l < N; l++) pren.v[l] = b[(size_t)l * n];
                }
                const Pair pr = classify(p, q);
                const Fp<C> inv_j = fe_mul_i<P>(inv, pre_j);
                inv = fe_mul_i<P>(inv, pr.den);
                Fp<C> num = pr.dy;
                if (pr.dbl) num = fe_scale<P, 3>(fe_sqr_i<P>(p.x));
                const Fp<C> lam = fe_mul_i<P>(num, inv_j);
                G1Aff<C> r;
                r.x = fe_lin<P, 1, -1, -1>(fe_sqr_i<P>(lam), p.x, q.x);
                r.y = fe_sub<P>(fe_mul_i<P>(lam, fe_sub<P>(p.x, r.x)), p.y);
                const G1Aff<C> other = pr.pinf ? q : (pr.qinf ? p : g1a_inf<C>());
                r.x = fe_select<P>(pr.trivial, other.x, r.x);
                r.y = fe_select<P>(pr.trivial, other.y, r.y);
                st(dst, j, r);
            }
        }
        if (m & 1) st(dst, h, ld(src, m - 1));
        m = h + (m & 1);
        uint32_t* t = src; src = dst; dst = t;
    }
    out = T > 0 ? g1j_from_aff<C>(ld(src, 0)) : g1j_inf<C>();
}

// shared inversion for two Jacobian points -> affine (Montgomery trick), identities preserved
template <class C>
__host__ __device__ inline void g1j_to_aff2(const G1Jac<C>& a, const G1Jac<C>& b, G1Aff<C>& oa, G1Aff<C>& ob) {
    using P = typename C::FpP;
    const bool ia = g1j_is_inf<C>(a), ib = g1j_is_inf<C>(b);
    Fp<C> za = ia ? fe_one<P>() : a.z, zb = ib ? fe_one<P>() : b.z;
    Fp<C> inv = fe_inv<P>(fe_mul<P>(za, zb));
    Fp<C> zai = fe_mul<P>(inv, zb), zbi = fe_mul<P>(inv, za);
    Fp<C> zai2 = fe_sqr<P>(zai), zbi2 = fe_sqr<P>(zbi);
    oa = ia ? g1a_inf<C>() : G1Aff<C>{fe_mul<P>(a.x, zai2), fe_mul<P>(fe_mul<P>(a.y, zai2), zai)};
    ob = ib ? g1a_inf<C>() : G1Aff<C>{fe_mul<P>(b.x, zbi2), fe_mul<P>(fe_mul<P>(b.y, zbi2), zbi)};
}

// =============================================================================================
// proof_verify
// =============================================================================================
constexpr int PV_NVAR = 2;                    // (c*Bbar + e^*Abar + r1^*D) jointly, r3^*D
constexpr int PV_NVAR_SPLIT = 4;              // latency mode: c*Bbar, e^*Abar, r1^*D, r3^*D each on its own lane
constexpr int PV_NPARTS = PV_NVAR + NFIX;
constexpr int PV_NPARTS_MAX = PV_NVAR_SPLIT + NFIX;
// throughput form only (nvar = PV_NVAR): two more terms of T1, behind the fixed-base chunks.  Identity unless the joint chain
// of an item could not be used (a proof point that is the identity or of small order): then T1's three products are computed
// one by one and land in slots 0, PV_T1_EXTRA, PV_T1_EXTRA + 1; PvChallenge adds them up in either case.
constexpr int PV_T1_EXTRA = PV_NVAR + NFIX;
static_assert(PV_T1_EXTRA + 2 <= PV_NPARTS_MAX, "partial-sum slots");

template <class C>
struct PvArgs {
    size_t n;
    int L, Rmax;
    const CtxConsts<C>* cc;
    int glv;                  // inputs vouched to be in G1: GLV split for the variable-base terms (BLS12-381)
    int nvar;                 // PV_NVAR (throughput: T1 as one joint chain) or PV_NVAR_SPLIT (latency: bbs_ctx_set_latency_mode)
    // batch verification, throughput form: PvChallenge also prepares the combination (RlcPrep of pippenger.hpp) -- null otherwise
    uint8_t* bv_dig; uint32_t* bv_ppts; size_t bv_n_pad; uint32_t bv_seed[8];
    // inputs (canonical limbs, SoA)
    const uint32_t* pts;      // [3][2NC][n] a_bar, b_bar, d (canonical words)
    const uint32_t* sc;       // [4][8][n]   e_cap, r1_cap, r3_cap, challenge
    const uint32_t* slots;    // [L][8][n]   slot j: disclosed message m_j or commitment m^_j
    const uint32_t* dmask;    // [ceil(L/32)][n]
    const uint32_t* didx;     // [Rmax][n]   disclosed indexes in caller order
    const uint32_t* rcount;   // [n]
    const uint32_t* hdr_off; const uint32_t* hdr_len; const uint8_t* hdr_bytes;
    const uint32_t* ph_off;  const uint32_t* ph_len;  const uint8_t* ph_bytes;
    int8_t* status;           // [n]; ST_PENDING = to compute, ST_PAIRING = pairing pending, else final
    // intermediates
    uint32_t* dom;            // [8][n] domain, Montgomery
    uint32_t* fscal;          // [L+2][8][n] canonical fixed-base scalars
    uint32_t* partials;       // [nvar + NFIX][3N][n] Jacobian
    uint32_t* aff;            // [5][2N][n] Montgomery affine: a_bar, b_bar, d, T1, T2
    uint32_t* fmiller;        // [2][12N][n]
    uint32_t* vtab;           // [4][G1_TAB][2N][n] window tables: three of the joint multiplication, one of D * r3^ (g1.hpp)
    FixTreeWork<C> fixwk;     // pts0 != nullptr: the fixed-base sum as one tree of affine additions per item (chunk 0's lane)
};

// stage 0 (lane per item, once per upload): the work the reference does before any arithmetic, on the raw batch as the
// C ABI receives it (item-major records, ragged arrays with 64-bit offsets) -- proof_verify_init's checks in the
// reference's order (src/proof_verify.rs:139-150), the duplicate test behind its out-of-bounds panic (:177-179), range
// checks of every scalar and coordinate (arkworks' types cannot hold a non-canonical value), and the transposition
// into the SoA arrays the later stages read.  Reads are strided by the record size, writes are coalesced.
template <class C>
struct PvIngestArgs {
    size_t n;
    int L, dst_too_long;
    const uint32_t* rec;                  // n records a_bar || b_bar || d || e^ || r1^ || r3^ || c, little-endian words
    const uint64_t *cm_off, *dm_off, *di_off, *hdr_off64, *ph_off64;   // n + 1 entries each, rebased to start at 0
    const uint32_t* cm;                   // commitments, 8 words each
    const uint32_t* dm;                   // disclosed messages, 8 words each
    const uint64_t* di;                   // disclosed indexes
    uint32_t *pts, *sc, *slots, *dmask, *didx, *rcount, *hdr_off, *hdr_len, *ph_off, *ph_len;   // PvArgs arrays
    int8_t* status0;                      // ST_PENDING or the reference's Err / panic / a non-canonical input
};

template <class C>
struct PvIngest {
    static __host__ __device__ void run(const PvIngestArgs<C>& a, size_t i) {
        using P = typename C::FpP;
        using R = typename C::FrP;
        constexpr int NC = P::NC;
        constexpr int RECW = 6 * NC + 32;
        const size_t n = a.n;
        a.hdr_off[i] = (uint32_t)a.hdr_off64[i];
        a.hdr_len[i] = (uint32_t)(a.hdr_off64[i + 1] - a.hdr_off64[i]);
        a.ph_off[i] = (uint32_t)a.ph_off64[i];
        a.ph_len[i] = (uint32_t)(a.ph_off64[i + 1] - a.ph_off64[i]);
        const uint64_t u = a.cm_off[i + 1] - a.cm_off[i], r = a.di_off[i + 1] - a.di_off[i], rm = a.dm_off[i + 1] - a.dm_off[i];
        const uint64_t l = u + r;
        const uint64_t* idx = a.di + a.di_off[i];
        const int MW = ((a.L > 1 ? a.L : 1) + 31) / 32;
        for (int w = 0; w < MW; w++) a.dmask[(size_t)w * n + i] = 0;
        a.rcount[i] = 0;
        int8_t st = ST_PENDING;
        bool bad = false;
        for (uint64_t k = 0; k < r; k++) bad |= idx[k] >= l;
        if (bad) st = -3;                                      // InvalidDisclosedIndex
        else if (rm != r) st = -6;                             // InvalidIndicesAndMessagesLength
        else if (l != (uint64_t)a.L) st = -1;                  // InvalidMessageAndGeneratorsLength
        else if (a.dst_too_long) st = -23;
        else {
            // duplicates make the undisclosed set larger than `commitments`: the reference indexes
            // proof.commitments[i] out of bounds (proof_verify.rs:177-179) and panics
            uint64_t distinct = 0;
            for (uint64_t k = 0; k < r; k++) {
                const size_t j = (size_t)idx[k];
                uint32_t* wp = a.dmask + (j >> 5) * n + i;
                const uint32_t w = *wp, bit = 1u << (j & 31);
                if (!(w & bit)) { *wp = w | bit; distinct++; }
            }
            if (distinct != r) st = -22;
        }
        if (st != ST_PENDING) { a.status0[i] = st; return; }
        bool ok = true;
        const uint32_t* pf = a.rec + i * (size_t)RECW;
        for (int c = 0; c < 6; c++) {                          // six coordinates
            uint32_t w[NC];
#pragma unroll
            for (int k = 0; k < NC; k++) w[k] = pf[c * NC + k];
            ok &= limbs_lt_mod<P>(w);
            soa_st<NC>(a.pts + (size_t)c * NC * n, n, i, w);
        }
        for (int c = 0; c < 4; c++) {
            uint32_t w[8];
            soa_ld<8>(pf + 6 * NC + 8 * c, 1, 0, w);
            ok &= limbs_lt_mod<R>(w);
            soa_st<8>(a.sc + (size_t)c * 8 * n, n, i, w);
        }
        // slots: disclosed messages at their index, commitments at the sorted undisclosed indexes
        for (uint64_t k = 0; k < r; k++) {
            const size_t j = (size_t)idx[k];
            uint32_t w[8];
            soa_ld<8>(a.dm + (a.dm_off[i] + k) * 8, 1, 0, w);
            ok &= limbs_lt_mod<R>(w);
            soa_st<8>(a.slots + j * 8 * n, n, i, w);
            a.didx[(size_t)k * n + i] = (uint32_t)j;
        }
        uint64_t cu = 0;
        for (size_t j = 0; j < (size_t)l; j++) {
            if ((a.dmask[(j >> 5) * n + i] >> (j & 31)) & 1u) continue;
            uint32_t w[8];
            soa_ld<8>(a.cm + (a.cm_off[i] + cu) * 8, 1, 0, w);
            ok &= limbs_lt_mod<R>(w);
            soa_st<8>(a.slots + j * 8 * n, n, i, w);
            cu++;
        }
        a.rcount[i] = (uint32_t)r;
        a.status0[i] = ok ? ST_PENDING : (int8_t)-40;
    }
};

// stage 1 (lane per item): domain, fixed-base scalars
template <class C>
struct PvScalars {
    static __host__ __device__ void run(const PvArgs<C>& a, size_t i) {
        using R = typename C::FrP;
        if (a.status[i] != ST_PENDING) return;
        const size_t n = a.n;
        Fr<C> dom = domain_from_header<C>(a.cc->hash, a.hdr_bytes + a.hdr_off[i], a.hdr_len[i]);
        soa_st<8>(a.dom, n, i, dom.v);
        Fr<C> c_canon = fr_load_canon<C>(a.sc + (size_t)3 * 8 * n, n, i);
        Fr<C> c_m = fr_to_mont<C>(c_canon);
        // P1 * c
        soa_st<8>(a.fscal, n, i, c_canon.v);
        // Q1 * (domain * c) : mont_mul(dom_mont, c_canon) = dom*c canonical... dom is Montgomery:
        Fr<C> dc = fe_mul<R>(dom, c_canon);                 // (dom*R)*c/R = dom*c canonical
        soa_st<8>(a.fscal + (size_t)1 * 8 * n, n, i, dc.v);
        for (int j = 0; j < a.L; j++) {
            Fr<C> s = fr_load_canon<C>(a.slots + (size_t)j * 8 * n, n, i);
            const uint32_t m = a.dmask[(size_t)(j >> 5) * n + i];
            if ((m >> (j & 31)) & 1u) s = fe_mul<R>(c_m, s);    // (c*R)*m/R = c*m canonical
            soa_st<8>(a.fscal + (size_t)(2 + j) * 8 * n, n, i, s.v);
        }
    }
};

// stage 2: the multi-scalar multiplication, as THREE kernels with their own register and scratch budgets (round 5; one
// kernel with four branch bodies -- 428 registers, 2.7 KB of scratch per lane -- charged that budget to the 512 of its 640
// wavefronts that only look up table entries and add them).  Every part writes its Jacobian partial sum to
// partials[part], part = 0: T1's chain, 1 .. nvar-1: single variable-base multiplications, nvar + f: fixed-base chunk f.
//
// stage 2a (lane per item): the on-curve checks of the proof's three points, their Montgomery copies, and -- throughput form
// -- T1 = c*Bbar + e^*Abar + r1^*D (proof_verify.rs:163-164) on one shared doubling chain.  (Latency form: T1's three terms are
// parts 0 .. 2 of PvVarMul, summed by PvChallenge, and this stage only checks and converts.)  The only stage that can decide
// -41 (a point off the curve); the other two may run beside it on other streams and need not see that verdict: what they
// compute for such an item is never read (PvChallenge runs behind all three and skips it).
template <class C>
struct PvT1Chain {
    static constexpr int WAVES_PER_EU = chain_waves<C>(BBS_T1_WAVES);
    static BBS_HD void run(const PvArgs<C>& a, size_t i) {
        constexpr int N = C::FpP::N;
        constexpr int NC = C::FpP::NC;
        constexpr size_t TW = (size_t)G1_TAB * 2 * N;
        const size_t n = a.n;
        if (a.status[i] != ST_PENDING) return;
        const bool joint = a.nvar == PV_NVAR;
        {
            // Abar, Bbar, D: on the curve?  Montgomery copies for the challenge stage (and batch verification); in the
            // throughput form also entry 0 of their window tables (tables 1, 0, 2: the chain's order is Bbar, Abar, D)
            bool on = true;
#pragma unroll 1
            for (int k = 0; k < 3; k++) {
                const G1Aff<C> p = g1a_load_canon_to_mont<C>(a.pts + (size_t)k * 2 * NC * n, n, i);
                on = g1a_on_curve<C>(p) && on;
                g1a_store_mont<C>(a.aff + (size_t)k * 2 * N * n, n, i, p);
                if (joint) TabHbm<C>{a.vtab + (size_t)(k == 0 ? 1 : (k == 1 ? 0 : 2)) * TW * n + i, n}.st(0, p);
            }
            if (!on) { a.status[i] = -41; return; }
        }
        if (!joint) return;
        uint32_t kc[8], ke[8], k1[8];
        soa_ld<8>(a.sc + (size_t)3 * 8 * n, n, i, kc);
        soa_ld<8>(a.sc, n, i, ke);
        soa_ld<8>(a.sc + (size_t)1 * 8 * n, n, i, k1);
        uint32_t* const x0 = a.partials + (size_t)PV_T1_EXTRA * 3 * N * n;
        uint32_t* const x1 = a.partials + (size_t)(PV_T1_EXTRA + 1) * 3 * N * n;
        G1Jac<C> r;
        bool done = false;
        if constexpr (C::K::HAS_GLV) {
            if (a.glv) done = g1_mul3_tabs_fast<C, true>(kc, ke, k1, a.vtab + i, n, r);
        }
        if (!a.glv) done = g1_mul3_tabs_fast<C, false>(kc, ke, k1, a.vtab + i, n, r);
        if (done) {
            g1j_store<C>(a.partials, n, i, r);
            r = g1j_inf<C>();
            g1j_store<C>(x0, n, i, r);
            g1j_store<C>(x1, n, i, r);
        } else {
            // a table hit an exceptional case (a proof point that is the identity or of small order, i.e. outside the
            // prime-order subgroup): the three products one by one on the generic double-and-add chain, which is right for
            // any on-curve point; PvChallenge sums the three slots.  Rare by construction, so its speed is of no concern,
            // but its frame is: a windowed multiplication here would add 0.9 KB to the kernel's scratch.
#pragma unroll 1
            for (int k = 0; k < 3; k++) {
                const G1Aff<C> p = g1a_load_mont<C>(a.aff + (size_t)(k == 0 ? 1 : (k == 1 ? 0 : 2)) * 2 * N * n, n, i);
                const uint32_t* kk = k == 0 ? kc : (k == 1 ? ke : k1);
                g1j_store<C>(k == 0 ? a.partials : (k == 1 ? x0 : x1), n, i, g1_mul_aff_naf<C>(p, kk));
            }
        }
    }
};
// stage 2b (lane per (part, item)): single variable-base multiplications -- r3^*D, the variable-base term of T2
// (proof_verify.rs:175-182), always (part nvar - 1); in the latency form also T1's three terms c*Bbar, e^*Abar, r1^*D (parts
// 0, 1, 2).  Window tables in HBM (a private table is 0.9 KB of scratch per lane of the whole kernel, and scratch x hardware
// queues is a budget: DESIGN.md 5 rule 6).  Reads the points in canonical form, as the ingest stage left them.
template <class C>
struct PvVarMul {
    static constexpr int WAVES_PER_EU = chain_waves<C>(BBS_VARMUL_WAVES);
    static __host__ __device__ int first_part(const PvArgs<C>& a) { return a.nvar == PV_NVAR ? 1 : 0; }
    static BBS_HD void run(const PvArgs<C>& a, size_t t) {
        constexpr int N = C::FpP::N;
        constexpr int NC = C::FpP::NC;
        const size_t n = a.n;
        const int rel = (int)(t / n);
        const int part = first_part(a) + rel;
        const size_t i = t - (size_t)rel * n;
        if (a.status[i] != ST_PENDING) return;
        const bool last = part == a.nvar - 1;
        const int pt = last ? 2 : (part == 0 ? 1 : (part == 1 ? 0 : 2));   // point: D | Bbar, Abar, D
        const int sc = last ? 2 : (part == 0 ? 3 : (part == 1 ? 0 : 1));   // scalar: r3^ | c, e^, r1^
        G1Aff<C> p = g1a_load_canon_to_mont<C>(a.pts + (size_t)pt * 2 * NC * n, n, i);
        uint32_t k[8];
        soa_ld<8>(a.sc + (size_t)sc * 8 * n, n, i, k);
        const int slot = last ? 3 : part;                                   // vtab slot 3 is D * r3^ in both forms
        G1Jac<C> r;
        g1_mul_aff_sel_hbm_inl<C>(p, k, a.glv != 0, a.vtab + (size_t)slot * G1_TAB * 2 * N * n + i, n, r);
        g1j_store<C>(a.partials + (size_t)part * 3 * N * n, n, i, r);
    }
};
// stages 2a + 2b as ONE launch (lane per (unit, item); unit 0 = stage 2a, the others stage 2b): for a job that keeps
// everything on one stream (batch verification's throughput form), where two launches would run one after the other
template <class C>
struct PvChains {
    static constexpr int WAVES_PER_EU = chain_waves<C>(BBS_T1_WAVES < BBS_VARMUL_WAVES ? BBS_T1_WAVES : BBS_VARMUL_WAVES);
    static __host__ __device__ size_t units(const PvArgs<C>& a) { return (size_t)1 + (size_t)(a.nvar - PvVarMul<C>::first_part(a)); }
    static BBS_HD void run(const PvArgs<C>& a, size_t t) {
        if (t < a.n) PvT1Chain<C>::run(a, t);
        else PvVarMul<C>::run(a, t - a.n);
    }
};
// stage 2c (lane per (chunk, item)): the NFIX chunks of the fixed-base sum over {P1, Q1, H_*}: table look-ups and mixed
// additions with inlined multipliers, nothing else -- no scratch, two wavefronts per SIMD.
template <class C>
struct PvFixedChunk {
    static __host__ __device__ void run(const PvArgs<C>& a, size_t t) {
        constexpr int N = C::FpP::N;
        const size_t n = a.n;
        const int chunk = (int)(t / n);
        const size_t i = t - (size_t)chunk * n;
        if (a.status[i] != ST_PENDING) return;
        G1Jac<C> r;
        fixed_msm_chunk_to<C>(*a.cc, a.fscal, n, i, a.L + 2, chunk, r);
        g1j_store<C>(a.partials + (size_t)(a.nvar + chunk) * 3 * N * n, n, i, r);
    }
};
// the same sum as ONE tree of affine additions per item (bbs_ctx_set_fixed_base_tree; lane per item): the result is chunk 0's
// partial sum, the other chunks are the identity
template <class C>
struct PvFixedTree {
    static __host__ __device__ void run(const PvArgs<C>& a, size_t i) {
        constexpr int N = C::FpP::N;
        const size_t n = a.n;
        if (a.status[i] != ST_PENDING) return;
        G1Jac<C> r = g1j_inf<C>();
        for (int f = 1; f < NFIX; f++) g1j_store<C>(a.partials + (size_t)(a.nvar + f) * 3 * N * n, n, i, r);
        fixed_msm_tree_to<C>(*a.cc, a.fscal, n, i, a.L + 2, a.fixwk, r);
        g1j_store<C>(a.partials + (size_t)a.nvar * 3 * N * n, n, i, r);
    }
};

// stage 3 (lane per item): combine parts, normalise, challenge hash, compare
template <class C>
struct PvChallenge {
    static constexpr int WAVES_PER_EU = chain_waves<C>(1);      // BN254: 264 - 268 registers -> 256
    static __host__ __device__ void run(const PvArgs<C>& a, size_t i) {
        constexpr int N = C::FpP::N;
        const size_t n = a.n;
        if (a.status[i] != ST_PENDING) return;
        auto part = [&](int p) { return g1j_load<C>(a.partials + (size_t)p * 3 * N * n, n, i); };
        G1Jac<C> t1 = part(0);
        for (int p = 1; p < a.nvar - 1; p++) t1 = g1j_add_i<C>(t1, part(p));       // latency mode: the three terms of T1
        if (a.nvar == PV_NVAR) {                                                  // throughput form: identities unless the joint chain was not usable
            t1 = g1j_add_i<C>(t1, part(PV_T1_EXTRA));
            t1 = g1j_add_i<C>(t1, part(PV_T1_EXTRA + 1));
        }
        G1Jac<C> t2 = part(a.nvar - 1);
        for (int f = 0; f < NFIX; f++) t2 = g1j_add_i<C>(t2, part(a.nvar + f));
        G1Aff<C> T1, T2;
        g1j_to_aff2<C>(t1, t2, T1, T2);
        // challenge (proof_gen.rs:272-328)
        Sha256 s;
        xmd48_begin(s);
        const uint32_t R = a.rcount[i];
        sha256_u64be(s, R);
        for (uint32_t k = 0; k < R; k++) {
            const uint32_t idx = a.didx[(size_t)k * n + i];
            sha256_u64be(s, idx);
            uint32_t m[8];
            soa_ld<8>(a.slots + (size_t)idx * 8 * n, n, i, m);
            sha256_limbs_be8(s, m);
        }
        for (int p = 0; p < 3; p++) sha256_g1_compressed<C>(s, g1a_load_mont<C>(a.aff + (size_t)p * 2 * N * n, n, i));
        sha256_g1_compressed<C>(s, T1);
        sha256_g1_compressed<C>(s, T2);
        Fr<C> dom;
        soa_ld<8>(a.dom, n, i, dom.v);
        sha256_fr_be<C>(s, dom);
        sha256_u64be(s, a.ph_len[i]);
        sha256_bytes(s, a.ph_bytes + a.ph_off[i], a.ph_len[i]);
        uint32_t okm[12];
        xmd48_finish(s, a.cc->hash.dst_h2s, a.cc->hash.dst_h2s_len, okm);
        Fr<C> chal = fe_to_canonical<typename C::FrP>(fr_from_okm<C>(okm));
        Fr<C> c = fr_load_canon<C>(a.sc + (size_t)3 * 8 * n, n, i);
        // proof_verify.rs:108-110: mismatch -> Ok(false) before any pairing
        a.status[i] = fe_eq<typename C::FrP>(chal, c) ? ST_PAIRING : (int8_t)0;
    }
    // batch verification, throughput form: every lane -- also those that left run() early -- then prepares its item's part
    // of the combination (digits of rho_i, the two points item-major); see RlcPrep in pippenger.hpp
    static __host__ __device__ void run_with_bv_prep(const PvArgs<C>& a, size_t i) {
        constexpr int N = C::FpP::N;
        run(a, i);
        const size_t n = a.n;
        uint32_t h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        G1Aff<C> p = g1a_inf<C>(), q = g1a_inf<C>();
        if (a.status[i] == ST_PAIRING) {
            p = g1a_load_mont<C>(a.aff, n, i);
            q = g1a_load_mont<C>(a.aff + (size_t)2 * N * n, n, i);
            Sha256 s;
            sha256_init(s);
            for (int k = 0; k < 8; k++) sha256_word(s, a.bv_seed[k]);
            sha256_u64be(s, (uint64_t)i);
            sha256_final(s, h);
        }
        for (int w = 0; w < 16; w++) a.bv_dig[(size_t)w * a.bv_n_pad + i] = (uint8_t)(h[w >> 2] >> (8 * (w & 3)));
        g1a_store_mont<C>(a.bv_ppts + i * 2 * N, 1, 0, p);
        g1a_store_mont<C>(a.bv_ppts + (n + i) * 2 * N, 1, 0, q);
    }
};
template <class C>
struct PvChallengeBv {
    static constexpr int WAVES_PER_EU = chain_waves<C>(1);      // BN254: 264 - 268 registers -> 256
    static __host__ __device__ void run(const PvArgs<C>& a, size_t i) { PvChallenge<C>::run_with_bv_prep(a, i); }
};

// generic pairing stages: e(Pa, pk) * e(Pb, BP2) == 1 for items whose status is 2
template <class C>
struct PairArgs {
    size_t n;
    const CtxConsts<C>* cc;
    const uint32_t* pa;       // [2N][n] Montgomery affine
    const uint32_t* pb;       // [2N][n]
    int negate_b;             // use -Pb (e(P, -Q) = e(-P, Q))
    int canonical;            // pa/pb hold canonical limbs (converted here) instead of Montgomery
    const int8_t* gate_arr;   // item i is processed iff gate_arr[i] == gate
    int gate;
    int8_t* out;              // result 1 / 0 per item (may alias the status array)
    uint32_t* fmiller;        // [2][12N][n]
    int single;               // fmiller holds ONE value per item (PairMillerBoth), not one per pair (PairMillerHalf)
    // batch verification: this launch is the per-item FALLBACK behind the combined checks -- if all n_checks of them passed
    // (batch_ok[k] == 1), every gated item's product is 1 (error 2^-128) and the lane only writes that; null otherwise
    const int8_t* batch_ok;
    int n_checks;
};
template <class C>
BBS_HD bool pair_batch_passed(const PairArgs<C>& a) {
    if (!a.batch_ok) return false;
    int ok = 1;
    for (int k = 0; k < a.n_checks; k++) ok &= (a.batch_ok[k] == 1);
    return ok != 0;
}

template <class C>
BBS_HD G1Aff<C> pair_load_point(const PairArgs<C>& a, const uint32_t* base, size_t i) {
    return a.canonical ? g1a_load_canon_to_mont<C>(base, a.n, i) : g1a_load_mont<C>(base, a.n, i);
}

// proof_verify runs the pairing concurrently with the MSM/challenge stages (the pairing needs only
// the proof's own points); this joins the two results.  proof_verify.rs:108-115: challenge
// mismatch -> Ok(false), otherwise the pairing boolean.
struct PvFinishArgs { size_t n; int8_t* status; const int8_t* pair_ok; };
struct PvFinish {
    static __host__ __device__ void run(const PvFinishArgs& a, size_t i) {
        if (a.status[i] == ST_PAIRING) a.status[i] = a.pair_ok[i] == 1 ? 1 : 0;
    }
};

// Fp12 <-> 12 Fp in tower order (c0.c0.c0, c0.c0.c1, c0.c1.c0, .., c1.c2.c1), no pointer casts
template <class C>
BBS_HD void f12_to_array(const Fp12<C>& f, Fp<C>* e) {
    e[0] = f.c0.c0.c0; e[1] = f.c0.c0.c1; e[2] = f.c0.c1.c0; e[3] = f.c0.c1.c1; e[4] = f.c0.c2.c0; e[5] = f.c0.c2.c1;
    e[6] = f.c1.c0.c0; e[7] = f.c1.c0.c1; e[8] = f.c1.c1.c0; e[9] = f.c1.c1.c1; e[10] = f.c1.c2.c0; e[11] = f.c1.c2.c1;
}
template <class C>
BBS_HD Fp12<C> f12_from_array(const Fp<C>* e) {
    Fp12<C> f;
    f.c0.c0.c0 = e[0]; f.c0.c0.c1 = e[1]; f.c0.c1.c0 = e[2]; f.c0.c1.c1 = e[3]; f.c0.c2.c0 = e[4]; f.c0.c2.c1 = e[5];
    f.c1.c0.c0 = e[6]; f.c1.c0.c1 = e[7]; f.c1.c1.c0 = e[8]; f.c1.c1.c1 = e[9]; f.c1.c2.c0 = e[10]; f.c1.c2.c1 = e[11];
    return f;
}
template <class C>
BBS_HD void f12_store(uint32_t* base, size_t n, size_t i, const Fp12<C>& f) {
    constexpr int N = C::FpP::N;
    Fp<C> e[12];
    f12_to_array<C>(f, e);
#pragma unroll
    for (int k = 0; k < 12; k++) soa_st<N>(base + (size_t)k * N * n, n, i, e[k].v);
}
template <class C>
BBS_HD Fp12<C> f12_load(const uint32_t* base, size_t n, size_t i) {
    constexpr int N = C::FpP::N;
    Fp<C> e[12];
#pragma unroll
    for (int k = 0; k < 12; k++) soa_ld<N>(base + (size_t)k * N * n, n, i, e[k].v);
    return f12_from_array<C>(e);
}

// lane per (pair, item)
template <class C>
struct PairMiller {
    static __host__ __device__ void run(const PairArgs<C>& a, size_t t) {
        constexpr int N = C::FpP::N;
        const size_t n = a.n;
        const int pair = (int)(t / n);
        const size_t i = t - (size_t)pair * n;
        if (a.gate_arr[i] != a.gate) return;
        if (pair_batch_passed<C>(a)) return;
        G1Aff<C> P = pair_load_point<C>(a, pair == 0 ? a.pa : a.pb, i);
        if (pair == 1 && a.negate_b) P = g1a_neg<C>(P);
        const LineTable<C>* tab = pair == 0 ? &a.cc->tab_pk : &a.cc->tab_bp2;
        Fp12<C> f = f12_one<C>();
        const bool skip = g1a_is_inf<C>(P) | (tab->q_is_identity != 0);
        if (!skip) {
            int li = 0;
            const int nops = a.cc->sched.n_ops;
            for (int k = 0; k < nops; k++) {
                if (a.cc->sched.op[k] == 0) f = f12_sqr<C>(f);
                else f = f12_mul_line<C>(f, tab->e[li++], P);
            }
            if constexpr (C::K::X_NEG) f = f12_conj<C>(f);
        }
        f12_store<C>(a.fmiller + (size_t)pair * 12 * N * n, n, i, f);
    }
};

// lane per item
template <class C>
struct PairFinal {
    static __host__ __device__ void run(const PairArgs<C>& a, size_t i) {
        constexpr int N = C::FpP::N;
        const size_t n = a.n;
        if (a.gate_arr[i] != a.gate) return;
        if (pair_batch_passed<C>(a)) { a.out[i] = 1; return; }
        Fp12<C> f = f12_mul<C>(f12_load<C>(a.fmiller, n, i), f12_load<C>(a.fmiller + (size_t)12 * N * n, n, i));
        a.out[i] = f12_is_one<C>(final_exponentiation<C>(f)) ? 1 : 0;
    }
};

// n-point batch normalisation (one inversion), identities preserved
// emit(k, affine point k), k = K-1 .. 0 (one shared inversion; a caller that stores the points elsewhere needs no array of them)
template <class C, int K, class Emit>
__host__ __device__ inline void g1j_batch_to_aff_emit(const G1Jac<C>* in, Emit emit) {
    using P = typename C::FpP;
    Fp<C> pre[K];
    Fp<C> acc = fe_one<P>();
    for (int k = 0; k < K; k++) {
        pre[k] = acc;
        if (!g1j_is_inf<C>(in[k])) acc = fe_mul<P>(acc, in[k].z);
    }
    Fp<C> inv = fe_inv<P>(acc);
    for (int k = K - 1; k >= 0; k--) {
        if (g1j_is_inf<C>(in[k])) { emit(k, g1a_inf<C>()); continue; }
        Fp<C> zi = fe_mul<P>(inv, pre[k]);
        inv = fe_mul<P>(inv, in[k].z);
        Fp<C> zi2 = fe_sqr<P>(zi);
        emit(k, G1Aff<C>{fe_mul<P>(in[k].x, zi2), fe_mul<P>(fe_mul<P>(in[k].y, zi2), zi)});
    }
}
template <class C, int K>
__host__ __device__ inline void g1j_batch_to_aff(const G1Jac<C>* in, G1Aff<C>* out) {
    g1j_batch_to_aff_emit<C, K>(in, [&](int k, const G1Aff<C>& p) { out[k] = p; });
}

// =============================================================================================
// fixed-base table construction (once per generator set)
// =============================================================================================
template <class C>
struct TabArgs {
    int n_bases, win_bits, n_windows;
    const uint32_t* bases;    // [n_bases][2N] Montgomery affine (AoS)
    uint32_t* winbase;        // [n_bases][W][2N] : 2^(c*w) * G_k
    uint32_t* tables;         // [n_bases][W][2^(c-1)][fix_tab_stride]
};

// lane per base: the W window bases by repeated doubling
template <class C>
struct TabWinBase {
    static __host__ __device__ void run(const TabArgs<C>& a, size_t k) {
        constexpr int N = C::FpP::N;
        G1Aff<C> b;
        for (int j = 0; j < N; j++) { b.x.v[j] = a.bases[k * 2 * N + j]; b.y.v[j] = a.bases[k * 2 * N + N + j]; }
        for (int w = 0; w < a.n_windows; w++) {
            uint32_t* o = a.winbase + ((size_t)k * a.n_windows + w) * 2 * N;
            for (int j = 0; j < N; j++) { o[j] = b.x.v[j]; o[N + j] = b.y.v[j]; }
            G1Jac<C> t = g1j_from_aff<C>(b);
            for (int d = 0; d < a.win_bits; d++) t = g1j_dbl<C>(t);
            b = g1j_to_aff<C>(t);
        }
    }
};

// lane per table entry (k, w, d): d * winbase[k][w], affine
template <class C>
struct TabEntry {
    static __host__ __device__ void run(const TabArgs<C>& a, size_t t) {
        constexpr int N = C::FpP::N;
        const size_t per_win = (size_t)1 << (a.win_bits - 1);        // |digit| = 1 .. 2^(c-1) (signed digits)
        const size_t kw = t / per_win;
        const uint32_t d = (uint32_t)(t - kw * per_win) + 1;
        const uint32_t* bsrc = a.winbase + kw * 2 * N;
        G1Aff<C> b;
        for (int j = 0; j < N; j++) { b.x.v[j] = bsrc[j]; b.y.v[j] = bsrc[N + j]; }
        G1Jac<C> r = g1j_inf<C>();
        for (int i = a.win_bits - 1; i >= 0; i--) {
            r = g1j_dbl<C>(r);
            if ((d >> i) & 1u) r = g1j_add_aff<C>(r, b);
        }
        G1Aff<C> o = g1j_to_aff<C>(r);
        uint32_t* dst = a.tables + t * fix_tab_stride<C>();
        for (int j = 0; j < N; j++) { dst[j] = o.x.v[j]; dst[N + j] = o.y.v[j]; }
        for (int j = 2 * N; j < fix_tab_stride<C>(); j++) dst[j] = 0;
    }
};

// =============================================================================================
// verify
// =============================================================================================
constexpr int VF_NPARTS = 1 + NFIX;

template <class C>
struct VfArgs {
    size_t n;
    int L;
    const CtxConsts<C>* cc;
    int glv;
    const uint32_t* sig_a;    // [2NC][n] canonical
    const uint32_t* sig_e;    // [8][n]
    const uint32_t* msgs;     // [L][8][n]
    const uint32_t* hdr_off; const uint32_t* hdr_len; const uint8_t* hdr_bytes;
    int8_t* status;
    uint32_t* fscal;          // [L+2][8][n]
    uint32_t* partials;       // [VF_NPARTS][3N][n]
    uint32_t* aff;            // [2][2N][n] : A, e*A - B  (Montgomery)
    uint32_t* fmiller;
    uint32_t* vtab;           // [G1_TAB][2N][n] window table of e * A (g1.hpp TabHbm)
};

// 32 big-endian bytes (any alignment) -> 8 little-endian words
BBS_HD void be32_words(const uint8_t* b, uint32_t* w) {
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const uint8_t* q = b + 28 - 4 * k;
        w[k] = ((uint32_t)q[0] << 24) | ((uint32_t)q[1] << 16) | ((uint32_t)q[2] << 8) | (uint32_t)q[3];
    }
}

// 8 little-endian words -> 32 big-endian bytes (I2OSP(x, 32))
BBS_HD void words_be32(const uint32_t* w, uint8_t* b) {
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const uint32_t v = w[7 - k];
        b[4 * k] = (uint8_t)(v >> 24); b[4 * k + 1] = (uint8_t)(v >> 16); b[4 * k + 2] = (uint8_t)(v >> 8); b[4 * k + 3] = (uint8_t)v;
    }
}
// compressed G1 octets from CANONICAL affine words (x: NC words, y: NC words; all zero = identity), the formats of
// codec_dev.hpp: BLS12-381 48 bytes big-endian with flags 0x80 / 0x40 / 0x20 in byte 0, BN254 32 bytes little-endian with
// flags 0x80 (y is the larger root) / 0x40 (identity) in the last byte
template <class C>
BBS_HD void g1_words_to_octets(const uint32_t* xw, const uint32_t* yw, uint8_t* out) {
    using P = typename C::FpP;
    constexpr int NC = P::NC, NB = 4 * NC;
    uint32_t any = 0;
#pragma unroll
    for (int k = 0; k < NC; k++) any |= xw[k] | yw[k];
    const bool inf = any == 0;
    const bool ybig = !inf && words_gt_half<P>(yw);
    if constexpr (C::ID == 0) {
#pragma unroll
        for (int k = 0; k < NC; k++) {
            const uint32_t v = xw[NC - 1 - k];
            out[4 * k] = (uint8_t)(v >> 24); out[4 * k + 1] = (uint8_t)(v >> 16); out[4 * k + 2] = (uint8_t)(v >> 8); out[4 * k + 3] = (uint8_t)v;
        }
        out[0] |= (uint8_t)(0x80u | (inf ? 0x40u : 0u) | (ybig ? 0x20u : 0u));
    } else {
#pragma unroll
        for (int k = 0; k < NC; k++) {
            const uint32_t v = xw[k];
            out[4 * k] = (uint8_t)v; out[4 * k + 1] = (uint8_t)(v >> 8); out[4 * k + 2] = (uint8_t)(v >> 16); out[4 * k + 3] = (uint8_t)(v >> 24);
        }
        out[NB - 1] |= (uint8_t)((inf ? 0x40u : 0u) | (ybig ? 0x80u : 0u));
    }
}

// stage 0 of verify (lane per item, once per upload): verify.rs:69-71's length check, range checks of the signature
// and the messages, transposition of the item-major staging image into the SoA arrays (see PvIngest)
template <class C>
struct VfIngestArgs {
    size_t n;
    int L, dst_too_long, has_sig;         // has_sig = 0: core_sign (no signature record, only messages)
    const uint32_t* rec;                  // n records A || e, little-endian words (has_sig, record form)
    // wire form (has_sig, oct != nullptr): n octet strings compress(A) || e big-endian; A has been decoded into sig_a by
    // VfOctDecode (codec_dev.hpp), its verdict is pcode[i]
    const uint8_t* oct;
    const int8_t* pcode;
    int msg_dst_too_long;                 // raw-message form: the reference's msg_to_scalars panics (DST > 255 bytes)
    const uint64_t *m_off, *hdr_off64;    // n + 1 entries each, rebased to 0
    const uint32_t* m;                    // messages, 8 words each
    uint32_t *sig_a, *sig_e, *msgs, *hdr_off, *hdr_len;
    int8_t* status0;
};
template <class C>
struct VfIngest {
    static __host__ __device__ void run(const VfIngestArgs<C>& a, size_t i) {
        using P = typename C::FpP;
        using R = typename C::FrP;
        constexpr int NC = P::NC;
        const size_t n = a.n;
        a.hdr_off[i] = (uint32_t)a.hdr_off64[i];
        a.hdr_len[i] = (uint32_t)(a.hdr_off64[i + 1] - a.hdr_off64[i]);
        if (a.has_sig && a.oct) {
            // the verdicts of bbs_signature_from_octets come first, in its order: the point's code, the identity, e >= r,
            // e = 0; only a decodable signature reaches core_verify's own checks
            constexpr size_t NB = 4 * NC;
            uint32_t e[8];
            be32_words(a.oct + i * (NB + 32) + NB, e);
            uint32_t any = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) any |= e[k];
            const int8_t c = a.pcode[i];
            int8_t pre = ST_PENDING;
            if (c < 0) pre = c;
            else if (c == 1) pre = -42;
            else if (!limbs_lt_mod<R>(e)) pre = -40;
            else if (!any) pre = -42;
            if (pre != ST_PENDING) {
#pragma unroll
                for (int k = 0; k < 8; k++) e[k] = 0;
            }
            soa_st<8>(a.sig_e, n, i, e);
            if (pre != ST_PENDING) { a.status0[i] = pre; return; }
        }
        const uint64_t l = a.m_off[i + 1] - a.m_off[i];
        // raw-message form: msg_to_scalars runs first in the reference's public functions (sign.rs:45, verify.rs:32)
        if (a.msg_dst_too_long && l > 0) { a.status0[i] = -23; return; }
        if (l != (uint64_t)a.L) { a.status0[i] = -1; return; }            // InvalidMessageAndGeneratorsLength
        if (a.dst_too_long) { a.status0[i] = -23; return; }
        bool ok = true;
        if (a.has_sig && !a.oct) {
            const uint32_t* sg = a.rec + i * (size_t)(2 * NC + 8);
            for (int c = 0; c < 2; c++) {
                uint32_t w[NC];
#pragma unroll
                for (int k = 0; k < NC; k++) w[k] = sg[c * NC + k];
                ok &= limbs_lt_mod<P>(w);
                soa_st<NC>(a.sig_a + (size_t)c * NC * n, n, i, w);
            }
            uint32_t e[8];
            soa_ld<8>(sg + 2 * NC, 1, 0, e);
            ok &= limbs_lt_mod<R>(e);
            soa_st<8>(a.sig_e, n, i, e);
        }
        for (uint64_t j = 0; j < l; j++) {
            uint32_t w[8];
            soa_ld<8>(a.m + (a.m_off[i] + j) * 8, 1, 0, w);
            ok &= limbs_lt_mod<R>(w);
            soa_st<8>(a.msgs + (size_t)j * 8 * n, n, i, w);
        }
        a.status0[i] = ok ? ST_PENDING : (int8_t)-40;
    }
};

template <class C>
struct VfScalars {
    static __host__ __device__ void run(const VfArgs<C>& a, size_t i) {
        using R = typename C::FrP;
        if (a.status[i] != ST_PENDING) return;
        const size_t n = a.n;
        Fr<C> dom = fe_to_canonical<R>(domain_from_header<C>(a.cc->hash, a.hdr_bytes + a.hdr_off[i], a.hdr_len[i]));
        Fr<C> one = fe_zero<R>();
        one.v[0] = 1;
        soa_st<8>(a.fscal, n, i, one.v);
        soa_st<8>(a.fscal + (size_t)8 * n, n, i, dom.v);
        for (int j = 0; j < a.L; j++) {
            uint32_t m[8];
            soa_ld<8>(a.msgs + (size_t)j * 8 * n, n, i, m);
            soa_st<8>(a.fscal + (size_t)(2 + j) * 8 * n, n, i, m);
        }
    }
};

// the multi-scalar multiplication as two kernels with their own budgets (round 5, as for proof_verify: PvVarMul / PvFixedChunk)
// lane per item: A on the curve?, its Montgomery copy, e * A (window table in HBM) -> partials[0]
template <class C>
struct VfVarMul {
    static constexpr int WAVES_PER_EU = chain_waves<C>(1);      // BN254: 264 - 268 registers -> 256
    static BBS_HD void run(const VfArgs<C>& a, size_t i) {
        const size_t n = a.n;
        if (a.status[i] != ST_PENDING) return;
        G1Aff<C> A = g1a_load_canon_to_mont<C>(a.sig_a, n, i);
        if (!g1a_on_curve<C>(A)) { a.status[i] = -41; return; }
        g1a_store_mont<C>(a.aff, n, i, A);
        uint32_t k[8];
        soa_ld<8>(a.sig_e, n, i, k);
        G1Jac<C> r;
        g1_mul_aff_sel_hbm_inl<C>(A, k, a.glv != 0, a.vtab + i, n, r);
        g1j_store<C>(a.partials, n, i, r);
    }
};
// lane per (chunk, item): the fixed-base sum B over {P1, Q1, H_*} -> partials[1 + chunk]
template <class C>
struct VfFixedChunk {
    static __host__ __device__ void run(const VfArgs<C>& a, size_t t) {
        constexpr int N = C::FpP::N;
        const size_t n = a.n;
        const int chunk = (int)(t / n);
        const size_t i = t - (size_t)chunk * n;
        if (a.status[i] != ST_PENDING) return;
        G1Jac<C> r;
        fixed_msm_chunk_to<C>(*a.cc, a.fscal, n, i, a.L + 2, chunk, r);
        g1j_store<C>(a.partials + (size_t)(1 + chunk) * 3 * N * n, n, i, r);
    }
};

template <class C>
struct VfCombine {
    static __host__ __device__ void run(const VfArgs<C>& a, size_t i) {
        constexpr int N = C::FpP::N;
        const size_t n = a.n;
        if (a.status[i] != ST_PENDING) return;
        auto part = [&](int p) { return g1j_load<C>(a.partials + (size_t)p * 3 * N * n, n, i); };
        G1Jac<C> b = part(1);
        for (int f = 1; f < NFIX; f++) b = g1j_add_i<C>(b, part(1 + f));
        G1Jac<C> x = g1j_add_i<C>(part(0), g1j_neg<C>(b));          // e*A - B
        g1a_store_mont<C>(a.aff + (size_t)2 * N * n, n, i, g1j_to_aff<C>(x));
        a.status[i] = ST_PAIRING;
    }
};

// =============================================================================================
// sign
// =============================================================================================
template <class C>
struct SgArgs {
    size_t n;
    int L;
    const CtxConsts<C>* cc;
    uint32_t sk[8];           // canonical
    const uint32_t* msgs;     // [L][8][n]
    const uint32_t* hdr_off; const uint32_t* hdr_len; const uint8_t* hdr_bytes;
    int8_t* status;
    uint32_t* fscal;          // [L+2][8][n]
    uint32_t* partials;       // [NFIX][3N][n]
    uint32_t* out_a;          // [2NC][n] canonical
    uint32_t* out_e;          // [8][n] canonical
    uint32_t* out_rec;        // [n][2NC + 8]: the records A || e as the caller receives them (SgEmit)
    int oct_form;             // 1: out_rec holds n octet strings compress(A) || I2OSP(e, 32) instead (fp_bytes + 32 each)
};

template <class C>
struct SgScalars {
    static __host__ __device__ void run(const SgArgs<C>& a, size_t i) {
        using R = typename C::FrP;
        if (a.status[i] != ST_PENDING) return;
        const size_t n = a.n;
        Fr<C> dom = domain_from_header<C>(a.cc->hash, a.hdr_bytes + a.hdr_off[i], a.hdr_len[i]);
        // e = hash_to_scalar(sk || m_1 .. m_L || domain)   (sign.rs:90-118)
        Sha256 s;
        xmd48_begin(s);
        sha256_limbs_be8(s, a.sk);
        for (int j = 0; j < a.L; j++) {
            uint32_t m[8];
            soa_ld<8>(a.msgs + (size_t)j * 8 * n, n, i, m);
            sha256_limbs_be8(s, m);
        }
        sha256_fr_be<C>(s, dom);
        uint32_t okm[12];
        xmd48_finish(s, a.cc->hash.dst_h2s, a.cc->hash.dst_h2s_len, okm);
        Fr<C> e = fr_from_okm<C>(okm);
        Fr<C> ec = fe_to_canonical<R>(e);
        soa_st<8>(a.out_e, n, i, ec.v);
        Fr<C> skm = fe_from_limbs<R>(a.sk);
        Fr<C> spe = fe_add<R>(skm, e);
        if (fe_is_zero<R>(spe)) { a.status[i] = -20; return; }     // sign.rs:129 unwrap
        Fr<C> inv = fe_inv<R>(spe);                                 // Montgomery
        Fr<C> invc = fe_to_canonical<R>(inv);
        soa_st<8>(a.fscal, n, i, invc.v);
        Fr<C> di = fe_mul<R>(dom, invc);                            // (dom R) inv / R = dom*inv canonical
        soa_st<8>(a.fscal + (size_t)8 * n, n, i, di.v);
        for (int j = 0; j < a.L; j++) {
            Fr<C> m = fr_load_canon<C>(a.msgs + (size_t)j * 8 * n, n, i);
            Fr<C> mi = fe_mul<R>(inv, m);
            soa_st<8>(a.fscal + (size_t)(2 + j) * 8 * n, n, i, mi.v);
        }
    }
};

template <class C>
struct SgMsmPart {
    static constexpr int WAVES_PER_EU = BBS_MSM_WAVES;
    static __host__ __device__ void run(const SgArgs<C>& a, size_t t) {
        constexpr int N = C::FpP::N;
        const size_t n = a.n;
        const int part = (int)(t / n);
        const size_t i = t - (size_t)part * n;
        if (a.status[i] != ST_PENDING) return;
        g1j_store<C>(a.partials + (size_t)part * 3 * N * n, n, i,
                     fixed_msm_chunk<C>(*a.cc, a.fscal, n, i, a.L + 2, part));
    }
};

template <class C>
struct SgCombine {
    static __host__ __device__ void run(const SgArgs<C>& a, size_t i) {
        constexpr int N = C::FpP::N;
        const size_t n = a.n;
        if (a.status[i] != ST_PENDING) return;
        G1Jac<C> acc = g1j_load<C>(a.partials, n, i);
        for (int f = 1; f < NFIX; f++) acc = g1j_add_i<C>(acc, g1j_load<C>(a.partials + (size_t)f * 3 * N * n, n, i));
        g1a_store_canon<C>(a.out_a, n, i, g1j_to_aff<C>(acc));
        a.status[i] = 1;
    }
};

// last stage of sign (lane per item): the signature record in the caller's layout (A affine || e, little-endian words,
// zeros unless the status is 1), so that delivery is one contiguous copy
template <class C>
struct SgEmit {
    static __host__ __device__ void run(const SgArgs<C>& a, size_t i) {
        constexpr int NC = C::FpP::NC, W = 2 * NC + 8;
        const size_t n = a.n;
        const bool ok = a.status[i] == 1;
        if (a.oct_form) {
            constexpr size_t NB = 4 * NC;
            uint8_t* o = reinterpret_cast<uint8_t*>(a.out_rec) + i * (NB + 32);
            if (!ok) { for (size_t k = 0; k < NB + 32; k++) o[k] = 0; return; }
            uint32_t pw[2 * NC], e[8];
            for (int k = 0; k < 2 * NC; k++) pw[k] = a.out_a[(size_t)k * n + i];
            for (int k = 0; k < 8; k++) e[k] = a.out_e[(size_t)k * n + i];
            g1_words_to_octets<C>(pw, pw + NC, o);
            words_be32(e, o + NB);
            return;
        }
        uint32_t* r = a.out_rec + i * (size_t)W;
        for (int k = 0; k < 2 * NC; k++) r[k] = ok ? a.out_a[(size_t)k * n + i] : 0u;
        for (int k = 0; k < 8; k++) r[2 * NC + k] = ok ? a.out_e[(size_t)k * n + i] : 0u;
    }
};

// =============================================================================================
// proof_gen
// =============================================================================================
constexpr int PG_NVAR = 7;                  // 4 multiples of B, 3 multiples of A: the seven scalars of PgArgs::vscal
constexpr int PG_NPARTS = PG_NVAR + NFIX;   // + chunks of sum m~_j H_j (the split form: one lane per multiplication)
// Throughput form (round 4): Bbar = (r1 r2) B - (e r1 r2) A and T1 = (r1~ r2) B + (e~ r1 r2) A each on ONE shared doubling
// chain (g1_mul2_aff): five lanes and five chains of ~252 doublings per item instead of seven -- 14 % fewer instructions per
// proof, a 29 % longer longest lane.  The split form stays the layout of a job that is alone (bbs_ctx_set_latency_mode).
constexpr int PG_NVAR_JOINT = 5;            // D, Abar, Bbar (joint), T1 (joint), T2's multiple of B

template <class C>
struct PgArgs {
    size_t n;
    int L, Rmax;
    const CtxConsts<C>* cc;
    int glv;                  // see PvArgs
    const uint32_t* sig_a;    // [2NC][n] canonical
    const uint32_t* sig_e;    // [8][n]
    const uint32_t* msgs;     // [L][8][n]
    const uint32_t* dmask;    // [ceil(L/32)][n] disclosed slots
    const uint32_t* didx;     // [Rmax][n] sorted distinct disclosed indexes
    const uint32_t* rcount;   // [n] number of distinct disclosed indexes
    const uint32_t* rnd5;     // [5][8][n]  r1, r2, e~, r1~, r3~
    const uint32_t* mtilde;   // [L][8][n]  m~_j at undisclosed slots, 0 elsewhere
    const uint32_t* hdr_off; const uint32_t* hdr_len; const uint8_t* hdr_bytes;
    const uint32_t* ph_off;  const uint32_t* ph_len;  const uint8_t* ph_bytes;
    int8_t* status;
    // intermediates
    uint32_t* dom;            // [8][n] Montgomery
    uint32_t* fscal;          // [L+2][8][n]  B's scalars (1, domain, m_j)
    uint32_t* fscal2;         // [L+2][8][n]  (0, 0, m~_j)
    uint32_t* vscal;          // [PG_NVAR][8][n] canonical scalars of the variable-base parts
    int nvar;                 // PG_NVAR (split form) or PG_NVAR_JOINT
    uint32_t* vtab;           // [PG_NVAR][G1_TAB][2N][n] window tables of the variable-base parts: split form table k = part k;
                              // joint form tables 0, 1 = Bbar's chain (B, -A), 2, 3 = T1's (B, A), 4, 5, 6 = parts 0, 1, 4
    // comb form of the joint layout (g1.hpp g1_comb_sum_to): stage PgTables writes, per item, the tables of the 2^(64 j)
    // multiples of B and A -- [base 2][piece 4][entry 8][2N][n] -- and comb_ok[base * n + i] = 1; null: not used
    uint32_t* ctab;
    int8_t* comb_ok;
    uint32_t* bpart;          // [NFIX][3N][n]
    uint32_t* baff;           // [2][2N][n]  B, A (Montgomery affine)
    uint32_t* partials;       // [PG_NPARTS][3N][n]
    // outputs (canonical)
    uint32_t* out_pts;        // [3][2NC][n] a_bar, b_bar, d (canonical)
    uint32_t* out_sc;         // [4][8][n]   e^, r1^, r3^, c
    uint32_t* out_mhat;       // [L][8][n]   m^_j at undisclosed slots
    // what the caller receives (PgEmit): records [n][6NC + 32] (Abar, Bbar, D, e^, r1^, r3^, c), the m^ of the undisclosed
    // messages in ascending index order [n][L][8], and their number per item
    uint32_t* out_rec;
    uint32_t* out_mh;
    uint32_t* ucount;
    // 1: the wire form instead -- out_rec holds, at a stride of 3 fp_bytes + 32 (4 + max(L, 1)) bytes per item, the octet
    // string compress(Abar) || compress(Bbar) || compress(D) || e^ || r1^ || r3^ || m^_1 .. m^_U || c (scalars big-endian),
    // 3 fp_bytes + 32 (4 + U) bytes of it used; out_mh is not written
    int oct_form;
};

// stage 0 of proof_gen (lane per item, once per upload): the checks of proof_gen.rs:133-143 and :229-239 in the
// reference's order (the count of random scalars is a contract of this ABI and checked on the host), deduplication and
// sorting of the disclosed indexes (:151-161) through the bit mask, range checks, transposition (see PvIngest)
template <class C>
struct PgIngestArgs {
    size_t n;
    int L, dst_too_long;
    const uint32_t* rec;                  // n signature records A || e
    // wire form (oct != nullptr): n signature octet strings compress(A) || e big-endian; A has been decoded into sig_a by
    // VfOctDecode, its verdict is pcode[i] (as VfIngestArgs)
    const uint8_t* oct;
    const int8_t* pcode;
    int msg_dst_too_long;                 // raw-message form: the reference's msg_to_scalars panics (DST > 255 bytes)
    const uint64_t *m_off, *di_off, *rnd_off, *hdr_off64, *ph_off64;
    const uint32_t* m;                    // messages
    const uint64_t* di;                   // disclosed indexes, caller order, duplicates possible
    const uint32_t* rnd;                  // random scalars: r1, r2, e~, r1~, r3~, then m~_j for the undisclosed j ascending
    uint32_t *sig_a, *sig_e, *msgs, *dmask, *didx, *rcount, *rnd5, *mtilde, *hdr_off, *hdr_len, *ph_off, *ph_len;
    int8_t* status0;
};
template <class C>
struct PgIngest {
    static __host__ __device__ void run(const PgIngestArgs<C>& a, size_t i) {
        using P = typename C::FpP;
        using R = typename C::FrP;
        constexpr int NC = P::NC;
        const size_t n = a.n;
        a.hdr_off[i] = (uint32_t)a.hdr_off64[i];
        a.hdr_len[i] = (uint32_t)(a.hdr_off64[i + 1] - a.hdr_off64[i]);
        a.ph_off[i] = (uint32_t)a.ph_off64[i];
        a.ph_len[i] = (uint32_t)(a.ph_off64[i + 1] - a.ph_off64[i]);
        const uint64_t l = a.m_off[i + 1] - a.m_off[i], r = a.di_off[i + 1] - a.di_off[i];
        const uint64_t* idx = a.di + a.di_off[i];
        const int MW = ((a.L > 1 ? a.L : 1) + 31) / 32;
        for (int w = 0; w < MW; w++) a.dmask[(size_t)w * n + i] = 0;
        a.rcount[i] = 0;
        if (a.oct) {
            // the verdicts of bbs_signature_from_octets first, in its order (see VfIngest)
            constexpr size_t NB = 4 * NC;
            uint32_t e[8];
            be32_words(a.oct + i * (NB + 32) + NB, e);
            uint32_t any = 0;
#pragma unroll
            for (int k = 0; k < 8; k++) any |= e[k];
            const int8_t c = a.pcode[i];
            int8_t pre = ST_PENDING;
            if (c < 0) pre = c;
            else if (c == 1) pre = -42;
            else if (!limbs_lt_mod<R>(e)) pre = -40;
            else if (!any) pre = -42;
            if (pre != ST_PENDING) { a.status0[i] = pre; return; }
            soa_st<8>(a.sig_e, n, i, e);
        }
        // raw-message form: msg_to_scalars runs first in the reference's public proof_gen (proof_gen.rs:95)
        if (a.msg_dst_too_long && l > 0) { a.status0[i] = -23; return; }
        if (r > l) { a.status0[i] = -2; return; }                          // InvalidDisclosedIndicesLength
        bool bad = false;
        for (uint64_t k = 0; k < r; k++) bad |= idx[k] >= l;
        if (bad) { a.status0[i] = -3; return; }                            // InvalidDisclosedIndex
        if (l != (uint64_t)a.L) { a.status0[i] = -1; return; }             // proof_init: InvalidMessageAndGeneratorsLength
        uint64_t distinct = 0;
        for (uint64_t k = 0; k < r; k++) {
            const size_t j = (size_t)idx[k];
            uint32_t* wp = a.dmask + (j >> 5) * n + i;
            const uint32_t w = *wp, bit = 1u << (j & 31);
            if (!(w & bit)) { *wp = w | bit; distinct++; }
        }
        // the random scalars were sized from the un-deduplicated length: a duplicate leaves fewer than 5 + undisclosed
        if (distinct != r) { a.status0[i] = -4; return; }
        if (a.dst_too_long) { a.status0[i] = -23; return; }
        bool ok = true;
        if (!a.oct) {
            const uint32_t* sg = a.rec + i * (size_t)(2 * NC + 8);
            for (int c = 0; c < 2; c++) {
                uint32_t w[NC];
#pragma unroll
                for (int k = 0; k < NC; k++) w[k] = sg[c * NC + k];
                ok &= limbs_lt_mod<P>(w);
                soa_st<NC>(a.sig_a + (size_t)c * NC * n, n, i, w);
            }
            uint32_t e[8];
            soa_ld<8>(sg + 2 * NC, 1, 0, e);
            ok &= limbs_lt_mod<R>(e);
            soa_st<8>(a.sig_e, n, i, e);
        }
        const uint32_t* rs = a.rnd + a.rnd_off[i] * 8;
        for (int k = 0; k < 5; k++) {
            uint32_t w[8];
            soa_ld<8>(rs + 8 * k, 1, 0, w);
            ok &= limbs_lt_mod<R>(w);
            soa_st<8>(a.rnd5 + (size_t)k * 8 * n, n, i, w);
        }
        uint32_t ku = 0, kd = 0;
        const uint32_t zero[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (size_t j = 0; j < (size_t)l; j++) {
            uint32_t w[8];
            soa_ld<8>(a.m + (a.m_off[i] + j) * 8, 1, 0, w);
            ok &= limbs_lt_mod<R>(w);
            soa_st<8>(a.msgs + j * 8 * n, n, i, w);
            if ((a.dmask[(j >> 5) * n + i] >> (j & 31)) & 1u) {
                a.didx[(size_t)kd * n + i] = (uint32_t)j;
                kd++;
                soa_st<8>(a.mtilde + j * 8 * n, n, i, zero);
            } else {
                soa_ld<8>(rs + 8 * (5 + ku), 1, 0, w);
                ok &= limbs_lt_mod<R>(w);
                soa_st<8>(a.mtilde + j * 8 * n, n, i, w);
                ku++;
            }
        }
        a.rcount[i] = kd;
        a.status0[i] = ok ? ST_PENDING : (int8_t)-40;
    }
};

template <class C>
struct PgScalars {
    static __host__ __device__ void run(const PgArgs<C>& a, size_t i) {
        using R = typename C::FrP;
        if (a.status[i] != ST_PENDING) return;
        const size_t n = a.n;
        Fr<C> r2c = fr_load_canon<C>(a.rnd5 + (size_t)1 * 8 * n, n, i);
        Fr<C> dom = domain_from_header<C>(a.cc->hash, a.hdr_bytes + a.hdr_off[i], a.hdr_len[i]);
        soa_st<8>(a.dom, n, i, dom.v);
        Fr<C> domc = fe_to_canonical<R>(dom);
        Fr<C> one = fe_zero<R>();
        one.v[0] = 1;
        Fr<C> zero = fe_zero<R>();
        soa_st<8>(a.fscal, n, i, one.v);
        soa_st<8>(a.fscal + (size_t)8 * n, n, i, domc.v);
        soa_st<8>(a.fscal2, n, i, zero.v);
        soa_st<8>(a.fscal2 + (size_t)8 * n, n, i, zero.v);
        for (int j = 0; j < a.L; j++) {
            uint32_t m[8];
            soa_ld<8>(a.msgs + (size_t)j * 8 * n, n, i, m);
            soa_st<8>(a.fscal + (size_t)(2 + j) * 8 * n, n, i, m);
            soa_ld<8>(a.mtilde + (size_t)j * 8 * n, n, i, m);
            soa_st<8>(a.fscal2 + (size_t)(2 + j) * 8 * n, n, i, m);
        }
        // variable-base scalars (proof_gen.rs:254-258, restructured over B and A)
        Fr<C> r1 = fr_to_mont<C>(fr_load_canon<C>(a.rnd5, n, i));
        Fr<C> r2 = fr_to_mont<C>(r2c);
        Fr<C> et = fr_load_canon<C>(a.rnd5 + (size_t)2 * 8 * n, n, i);
        Fr<C> r1t = fr_load_canon<C>(a.rnd5 + (size_t)3 * 8 * n, n, i);
        Fr<C> r3t = fr_load_canon<C>(a.rnd5 + (size_t)4 * 8 * n, n, i);
        Fr<C> e = fr_load_canon<C>(a.sig_e, n, i);
        Fr<C> r1r2 = fe_mul<R>(r1, r2);                               // Montgomery
        Fr<C> v[PG_NVAR];
        v[0] = r2c;                                                   // D      = r2 * B
        v[1] = fe_to_canonical<R>(r1r2);                              // r1r2 * B
        v[2] = fe_mul<R>(r2, r1t);                                    // T1 part: (r1~ r2) * B
        v[3] = fe_mul<R>(r2, r3t);                                    // T2 part: (r3~ r2) * B
        v[4] = v[1];                                                  // Abar   = (r1 r2) * A
        v[5] = fe_mul<R>(r1r2, e);                                    // (e r1 r2) * A
        v[6] = fe_mul<R>(r1r2, et);                                   // (e~ r1 r2) * A
        for (int k = 0; k < PG_NVAR; k++) soa_st<8>(a.vscal + (size_t)k * 8 * n, n, i, v[k].v);
    }
};

// lane per (chunk, item): B = P1 + Q1*domain + sum H_j m_j
// lane per (sum, chunk, item): BOTH fixed-base sums of an item over {P1, Q1, H_*} -- B = P1 + Q1 domain + sum H_j m_j
// (scalars fscal; chunks -> bpart, summed by PgBCombine) and T2's sum H_j m~_j (scalars fscal2; chunks -> partials[nvar + f],
// summed by PgFinalize).  Both depend on the scalar stage only, so the second sum no longer rides in the kernel of the
// doubling chains (round 5): table look-ups and mixed additions, 246 registers, no scratch, two wavefronts per SIMD.
template <class C>
struct PgBPart {
    static __host__ __device__ void run(const PgArgs<C>& a, size_t t) {
        constexpr int N = C::FpP::N;
        const size_t n = a.n;
        const int part = (int)(t / n);                        // 0 .. NFIX-1: B;  NFIX .. 2 NFIX-1: T2's sum
        const size_t i = t - (size_t)part * n;
        if (a.status[i] != ST_PENDING) return;
        const bool second = part >= NFIX;
        const int chunk = second ? part - NFIX : part;
        G1Jac<C> r;
        fixed_msm_chunk_to<C>(*a.cc, second ? a.fscal2 : a.fscal, n, i, a.L + 2, chunk, r);
        g1j_store<C>((second ? a.partials + (size_t)a.nvar * 3 * N * n : a.bpart) + (size_t)chunk * 3 * N * n, n, i, r);
    }
};

template <class C>
struct PgBCombine {
    static __host__ __device__ void run(const PgArgs<C>& a, size_t i) {
        constexpr int N = C::FpP::N;
        const size_t n = a.n;
        if (a.status[i] != ST_PENDING) return;
        G1Jac<C> acc = g1j_load<C>(a.bpart, n, i);
        for (int f = 1; f < NFIX; f++) acc = g1j_add_i<C>(acc, g1j_load<C>(a.bpart + (size_t)f * 3 * N * n, n, i));
        g1a_store_mont<C>(a.baff, n, i, g1j_to_aff<C>(acc));
        G1Aff<C> A = g1a_load_canon_to_mont<C>(a.sig_a, n, i);
        if (!g1a_on_curve<C>(A)) { a.status[i] = -41; return; }
        g1a_store_mont<C>(a.baff + (size_t)2 * N * n, n, i, A);
    }
};

// lane per (base, item), base 0 = B, 1 = A: the sub-bases 2^(64 j) P and their tables of odd multiples, true affine, for the comb
// (g1.hpp).  Anything unusual -- the identity, a point of small order (they exist only outside the prime-order subgroup), a
// degenerate step -- leaves comb_ok = 0 and the item's lanes take the joint chains instead: same group elements either way.
template <class C>
struct PgTables {
    static constexpr int WAVES_PER_EU = BBS_MSM_WAVES;
    static __host__ __device__ void run(const PgArgs<C>& a, size_t t) {
        using P = typename C::FpP;
        constexpr int N = P::N;
        const size_t n = a.n;
        const int base = (int)(t / n);
        const size_t i = t - (size_t)base * n;
        a.comb_ok[(size_t)base * n + i] = 0;
        if (a.status[i] != ST_PENDING) return;
        const G1Aff<C> p0 = g1a_load_mont<C>(a.baff + (size_t)base * 2 * N * n, n, i);
        if (g1a_is_inf<C>(p0)) return;
        // sub-bases: 2^(64 j) P for j < 4 -- or, under the GLV split (points known to be in the subgroup), P and 2^64 P only:
        // the other two are their images under the endomorphism
        const bool glv = C::K::HAS_GLV && a.glv != 0;
        const int n_sub = glv ? 2 : COMB_PIECES;
        G1Jac<C> q[COMB_PIECES - 1];
        G1Jac<C> cur = g1j_from_aff<C>(p0);
#pragma unroll 1
        for (int j = 0; j < n_sub - 1; j++) {
#pragma unroll 1
            for (int d = 0; d < 64; d++) cur = g1j_dbl<C>(cur);
            if (g1j_is_inf<C>(cur)) return;
            q[j] = cur;
        }
        for (int j = n_sub - 1; j < COMB_PIECES - 1; j++) q[j] = g1j_inf<C>();
        // the sub-bases in affine form go straight to entry 0 of their tables (HBM), where the table builder picks them up:
        // no array of them in this lane's frame (scratch x hardware queues is a budget, DESIGN.md 5 rule 6)
        uint32_t* tb = a.ctab + (size_t)base * comb_table_words(N) * n + i;
        TabHbm<C>{tb, n}.st(0, p0);
        g1j_batch_to_aff_emit<C, COMB_PIECES - 1>(q, [&](int k, const G1Aff<C>& s) { TabHbm<C>{tb + (size_t)(k + 1) * G1_TAB * 2 * N * n, n}.st(0, s); });
        Fp<C> zc[COMB_PIECES];
        bool ok = true;
#pragma unroll 1
        for (int j = 0; j < n_sub; j++) {
            TabHbm<C> tab{tb + (size_t)j * G1_TAB * 2 * N * n, n};
            ok = g1_odd_table<C>(tab.ld(0), tab, zc[j]) && ok;
        }
        if (!ok) return;
        // entries (x', y') of table j are the Jacobian points (x', y', zc_j): to true affine with ONE inversion for the scales
        Fp<C> pre[COMB_PIECES];
        Fp<C> acc = fe_one<P>();
#pragma unroll 1
        for (int j = 0; j < n_sub; j++) { pre[j] = acc; acc = fe_mul<P>(acc, zc[j]); }
        Fp<C> inv = fe_inv<P>(acc);
#pragma unroll 1
        for (int j = n_sub - 1; j >= 0; j--) {
            const Fp<C> zi = fe_mul<P>(inv, pre[j]);
            inv = fe_mul<P>(inv, zc[j]);
            const Fp<C> zi2 = fe_sqr<P>(zi), zi3 = fe_mul<P>(zi2, zi);
            TabHbm<C> tab{tb + (size_t)j * G1_TAB * 2 * N * n, n};
#pragma unroll 1
            for (int e = 0; e < G1_TAB; e++) {
                const G1Aff<C> v = tab.ld(e);
                tab.st(e, G1Aff<C>{fe_mul<P>(v.x, zi2), fe_mul<P>(v.y, zi3)});
            }
        }
        if constexpr (C::K::HAS_GLV) {
            if (glv) {                               // tables 2, 3 = phi of tables 0, 1: (beta x, y)
                const Fp<C> beta = glv_beta<C>();
#pragma unroll 1
                for (int j = 0; j < 2; j++) {
                    TabHbm<C> src{tb + (size_t)j * G1_TAB * 2 * N * n, n}, dst{tb + (size_t)(2 + j) * G1_TAB * 2 * N * n, n};
#pragma unroll 1
                    for (int e = 0; e < G1_TAB; e++) {
                        const G1Aff<C> v = src.ld(e);
                        dst.st(e, G1Aff<C>{fe_mul<P>(v.x, beta), v.y});
                    }
                }
            }
        }
        a.comb_ok[(size_t)base * n + i] = 1;
    }
};

// lane per (part < nvar, item): the variable-base parts -- multiples of B and of the signature point A.  A kernel of its own
// (round 5: the fixed-base chunks are in PgBPart), window tables in HBM, the multiplication routines inlined.
template <class C>
struct PgVarPart {
    static BBS_HD void run(const PgArgs<C>& a, size_t t) {
        constexpr int N = C::FpP::N;
        constexpr size_t TW = (size_t)G1_TAB * 2 * N;
        const size_t n = a.n;
        const int part = (int)(t / n);
        const size_t i = t - (size_t)part * n;
        if (a.status[i] != ST_PENDING) return;
        uint32_t* out = a.partials + (size_t)part * 3 * N * n;
        auto scalar = [&](int k, uint32_t* dst) { soa_ld<8>(a.vscal + (size_t)k * 8 * n, n, i, dst); };
        G1Jac<C> r;
        if (a.nvar == PG_NVAR) {
            // split form: part k multiplies B (k < 4) or A by scalar k
            const G1Aff<C> p = g1a_load_mont<C>(a.baff + (size_t)(part < 4 ? 0 : 1) * 2 * N * n, n, i);
            uint32_t k[8];
            scalar(part, k);
            g1_mul_aff_sel_hbm_inl<C>(p, k, a.glv != 0, a.vtab + (size_t)part * TW * n + i, n, r);
        } else if (a.ctab && a.comb_ok[i] && a.comb_ok[n + i]) {
            // comb form: every multiple of B and A from the tables of their 2^(64 j) multiples, 60 doublings per chain.
            // part 0: D = v0 B; 1: Abar = v4 A; 2: Bbar = v1 B - v5 A; 3: T1 = v2 B + v6 A; 4: T2's v3 B
            const uint32_t* tB = a.ctab + i;
            const uint32_t* tA = a.ctab + comb_table_words(N) * n + i;
            uint32_t k0[8], k1[8];
            if (part == 2 || part == 3) {
                scalar(part == 2 ? 1 : 2, k0);
                scalar(part == 2 ? 5 : 6, k1);
                CombTerm tm[2];
                bool done = false;
                if constexpr (C::K::HAS_GLV) {
                    if (a.glv) { comb_recode_glv<C>(k0, false, tB, tm[0]); comb_recode_glv<C>(k1, part == 2, tA, tm[1]); done = true; }
                }
                if (!done) { comb_recode(k0, false, tB, tm[0]); comb_recode(k1, part == 2, tA, tm[1]); }
                g1_comb_sum_to<C, 2>(tm, n, r);
            } else {
                scalar(part == 0 ? 0 : (part == 1 ? 4 : 3), k0);
                CombTerm tm[1];
                bool done = false;
                if constexpr (C::K::HAS_GLV) {
                    if (a.glv) { comb_recode_glv<C>(k0, false, part == 1 ? tA : tB, tm[0]); done = true; }
                }
                if (!done) comb_recode(k0, false, part == 1 ? tA : tB, tm[0]);
                g1_comb_sum_to<C, 1>(tm, n, r);
            }
        } else if (part == 2 || part == 3) {
            // joint form: Bbar = v1 B + v5 (-A) (part 2), T1 = v2 B + v6 A (part 3) -- one doubling chain each; if a table hits an
            // exceptional case (B or A the identity or of small order) the two products one by one on the generic chain
            const G1Aff<C> B = g1a_load_mont<C>(a.baff, n, i);
            G1Aff<C> A = g1a_load_mont<C>(a.baff + (size_t)2 * N * n, n, i);
            if (part == 2) A = g1a_neg<C>(A);
            uint32_t kb[8], ka[8];
            scalar(part == 2 ? 1 : 2, kb);
            scalar(part == 2 ? 5 : 6, ka);
            uint32_t* tabs = a.vtab + (size_t)(part - 2) * 2 * TW * n + i;
            TabHbm<C>{tabs, n}.st(0, B);
            TabHbm<C>{tabs + TW * n, n}.st(0, A);
            bool done = false;
            if constexpr (C::K::HAS_GLV) {
                if (a.glv) done = g1_mul2_tabs_fast<C, true>(kb, ka, tabs, n, r);
            }
            if (!a.glv) done = g1_mul2_tabs_fast<C, false>(kb, ka, tabs, n, r);
            if (!done) {
                const G1Jac<C> x = g1_mul_aff_naf<C>(B, kb);
                r = g1j_add_i<C>(x, g1_mul_aff_naf<C>(A, ka));
            }
        } else {
            // joint form, single multiplications: D = v0 B (part 0), Abar = v4 A (part 1), T2's v3 B (part 4)
            const G1Aff<C> p = g1a_load_mont<C>(a.baff + (size_t)(part == 1 ? 1 : 0) * 2 * N * n, n, i);
            uint32_t k[8];
            scalar(part == 0 ? 0 : (part == 1 ? 4 : 3), k);
            const int slot = part == 0 ? 4 : (part == 1 ? 5 : 6);
            g1_mul_aff_sel_hbm_inl<C>(p, k, a.glv != 0, a.vtab + (size_t)slot * TW * n + i, n, r);
        }
        g1j_store<C>(out, n, i, r);
    }
};

template <class C>
struct PgFinalize {
    static __host__ __device__ void run(const PgArgs<C>& a, size_t i) {
        using R = typename C::FrP;
        constexpr int N = C::FpP::N;
        const size_t n = a.n;
        if (a.status[i] != ST_PENDING) return;
        auto part = [&](int p) { return g1j_load<C>(a.partials + (size_t)p * 3 * N * n, n, i); };
        G1Jac<C> pj[5];
        if (a.nvar == PG_NVAR) {
            pj[0] = part(4);                                              // Abar
            pj[1] = g1j_add_i<C>(part(1), g1j_neg<C>(part(5)));             // Bbar = r1r2 B - e r1r2 A
            pj[2] = part(0);                                              // D
            pj[3] = g1j_add_i<C>(part(6), part(2));                         // T1
            pj[4] = part(3);                                              // T2
        } else {
            pj[0] = part(1); pj[1] = part(2); pj[2] = part(0); pj[3] = part(3); pj[4] = part(4);      // the joint chains' own sums
        }
        for (int f = 0; f < NFIX; f++) pj[4] = g1j_add_i<C>(pj[4], part(a.nvar + f));
        G1Aff<C> pa[5];
        g1j_batch_to_aff<C, 5>(pj, pa);
        // challenge (proof_gen.rs:272-328), disclosed indexes sorted + deduplicated (:151-161)
        Sha256 s;
        xmd48_begin(s);
        const uint32_t Rn = a.rcount[i];
        sha256_u64be(s, Rn);
        for (uint32_t k = 0; k < Rn; k++) {
            const uint32_t idx = a.didx[(size_t)k * n + i];
            sha256_u64be(s, idx);
            uint32_t m[8];
            soa_ld<8>(a.msgs + (size_t)idx * 8 * n, n, i, m);
            sha256_limbs_be8(s, m);
        }
        for (int p = 0; p < 5; p++) sha256_g1_compressed<C>(s, pa[p]);
        Fr<C> dom;
        soa_ld<8>(a.dom, n, i, dom.v);
        sha256_fr_be<C>(s, dom);
        sha256_u64be(s, a.ph_len[i]);
        sha256_bytes(s, a.ph_bytes + a.ph_off[i], a.ph_len[i]);
        uint32_t okm[12];
        xmd48_finish(s, a.cc->hash.dst_h2s, a.cc->hash.dst_h2s_len, okm);
        Fr<C> c = fr_from_okm<C>(okm);                                // Montgomery
        // proof_finalize (proof_gen.rs:331-365)
        Fr<C> r2 = fr_to_mont<C>(fr_load_canon<C>(a.rnd5 + (size_t)1 * 8 * n, n, i));
        if (fe_is_zero<R>(r2)) { a.status[i] = -21; return; }         // :346 unwrap
        Fr<C> r3 = fe_inv<R>(r2);
        Fr<C> r1 = fr_to_mont<C>(fr_load_canon<C>(a.rnd5, n, i));
        Fr<C> et = fr_to_mont<C>(fr_load_canon<C>(a.rnd5 + (size_t)2 * 8 * n, n, i));
        Fr<C> r1t = fr_to_mont<C>(fr_load_canon<C>(a.rnd5 + (size_t)3 * 8 * n, n, i));
        Fr<C> r3t = fr_to_mont<C>(fr_load_canon<C>(a.rnd5 + (size_t)4 * 8 * n, n, i));
        Fr<C> e = fr_to_mont<C>(fr_load_canon<C>(a.sig_e, n, i));
        Fr<C> o[4];
        o[0] = fe_to_canonical<R>(fe_add<R>(et, fe_mul<R>(e, c)));
        o[1] = fe_to_canonical<R>(fe_sub<R>(r1t, fe_mul<R>(r1, c)));
        o[2] = fe_to_canonical<R>(fe_sub<R>(r3t, fe_mul<R>(r3, c)));
        o[3] = fe_to_canonical<R>(c);
        for (int k = 0; k < 4; k++) soa_st<8>(a.out_sc + (size_t)k * 8 * n, n, i, o[k].v);
        for (int j = 0; j < a.L; j++) {
            const uint32_t dm = a.dmask[(size_t)(j >> 5) * n + i];
            if ((dm >> (j & 31)) & 1u) continue;
            Fr<C> m = fr_load_canon<C>(a.msgs + (size_t)j * 8 * n, n, i);       // canonical
            Fr<C> mt = fr_load_canon<C>(a.mtilde + (size_t)j * 8 * n, n, i);   // canonical
            // m~ + m*c : mont_mul(c_mont, m_canon) = m*c canonical ; add canonical values mod r
            Fr<C> mh = fe_add<R>(mt, fe_mul<R>(c, m));
            soa_st<8>(a.out_mhat + (size_t)j * 8 * n, n, i, mh.v);
        }
        for (int p = 0; p < 3; p++) g1a_store_canon<C>(a.out_pts + (size_t)p * 2 * C::FpP::NC * n, n, i, pa[p]);
        a.status[i] = 1;
    }
};

// last stage of proof_gen (lane per item): the proof in the caller's layout, zeros / no commitments unless the status is 1
template <class C>
struct PgEmit {
    static __host__ __device__ void run(const PgArgs<C>& a, size_t i) {
        constexpr int NC = C::FpP::NC, W = 6 * NC + 32;
        const size_t n = a.n;
        const bool ok = a.status[i] == 1;
        if (a.oct_form) {
            constexpr size_t NB = 4 * NC;
            const size_t stride = 3 * NB + 32 * (size_t)(4 + (a.L > 1 ? a.L : 1));
            uint8_t* o = reinterpret_cast<uint8_t*>(a.out_rec) + i * stride;
            uint32_t u = 0;
            if (ok) {
                for (int p = 0; p < 3; p++) {
                    uint32_t pw[2 * NC];
                    for (int k = 0; k < 2 * NC; k++) pw[k] = a.out_pts[((size_t)p * 2 * NC + k) * n + i];
                    g1_words_to_octets<C>(pw, pw + NC, o + (size_t)p * NB);
                }
                uint32_t w[8];
                for (int q = 0; q < 3; q++) {
                    for (int k = 0; k < 8; k++) w[k] = a.out_sc[((size_t)q * 8 + k) * n + i];
                    words_be32(w, o + 3 * NB + 32 * (size_t)q);
                }
                for (int j = 0; j < a.L; j++) {
                    const uint32_t dm = a.dmask[(size_t)(j >> 5) * n + i];
                    if ((dm >> (j & 31)) & 1u) continue;
                    for (int k = 0; k < 8; k++) w[k] = a.out_mhat[((size_t)j * 8 + k) * n + i];
                    words_be32(w, o + 3 * NB + 96 + 32 * (size_t)u);
                    u++;
                }
                for (int k = 0; k < 8; k++) w[k] = a.out_sc[((size_t)3 * 8 + k) * n + i];
                words_be32(w, o + 3 * NB + 96 + 32 * (size_t)u);
            }
            a.ucount[i] = ok ? u : 0xFFFFFFFFu;          // no string at all for a failed item
            return;
        }
        uint32_t* r = a.out_rec + i * (size_t)W;
        for (int k = 0; k < 6 * NC; k++) r[k] = ok ? a.out_pts[(size_t)k * n + i] : 0u;
        for (int k = 0; k < 32; k++) r[6 * NC + k] = ok ? a.out_sc[(size_t)k * n + i] : 0u;
        uint32_t u = 0;
        if (ok) {
            uint32_t* m = a.out_mh + i * (size_t)(a.L > 1 ? a.L : 1) * 8;
            for (int j = 0; j < a.L; j++) {
                const uint32_t dm = a.dmask[(size_t)(j >> 5) * n + i];
                if ((dm >> (j & 31)) & 1u) continue;
                for (int k = 0; k < 8; k++) m[(size_t)u * 8 + k] = a.out_mhat[((size_t)j * 8 + k) * n + i];
                u++;
            }
        }
        a.ucount[i] = u;
    }
};

// =============================================================================================
// unit-parity primitives
// =============================================================================================
struct H2sArgs {
    size_t n;
    const uint32_t* off; const uint32_t* len; const uint8_t* bytes;
    uint8_t dst[256];
    uint32_t dst_len;
    uint32_t* out;            // [8][n] canonical
};

template <class C>
struct H2sItem {
    static __host__ __device__ void run(const H2sArgs& a, size_t i) {
        Sha256 s;
        xmd48_begin(s);
        sha256_bytes(s, a.bytes + a.off[i], a.len[i]);
        uint32_t okm[12];
        xmd48_finish(s, a.dst, a.dst_len, okm);
        Fr<C> r = fe_to_canonical<typename C::FrP>(fr_from_okm<C>(okm));
        soa_st<8>(a.out, a.n, i, r.v);
    }
};

template <class C>
struct MsmArgs {
    size_t n;
    int n_fixed, n_var;
    int glv;                  // see PvArgs
    const CtxConsts<C>* cc;
    const uint32_t* fscal;    // [n_fixed][8][n]
    const uint32_t* vpts;     // [n_var][2NC][n] canonical
    const uint32_t* vscal;    // [n_var][8][n]
    int8_t* status;
    uint32_t* partials;       // [n_var + NFIX][3N][n]
    uint32_t* out;            // [2NC][n] canonical
    FixTreeWork<C> fixwk;     // see PvArgs
    uint32_t* vtab;           // [n_var][G1_TAB][2N][n] window tables of the variable-base terms
};

// lane per (variable-base term, item)
template <class C>
struct MsmVarMul {
    static constexpr int WAVES_PER_EU = chain_waves<C>(1);      // BN254: 264 - 268 registers -> 256
    static BBS_HD void run(const MsmArgs<C>& a, size_t t) {
        constexpr int N = C::FpP::N;
        const size_t n = a.n;
        const int part = (int)(t / n);
        const size_t i = t - (size_t)part * n;
        if (a.status[i] != ST_PENDING) return;
        uint32_t* out = a.partials + (size_t)part * 3 * N * n;
        G1Aff<C> p = g1a_load_canon_to_mont<C>(a.vpts + (size_t)part * 2 * C::FpP::NC * n, n, i);
        G1Jac<C> r = g1j_inf<C>();
        if (!g1a_on_curve<C>(p)) { a.status[i] = -41; g1j_store<C>(out, n, i, r); return; }
        uint32_t k[8];
        soa_ld<8>(a.vscal + (size_t)part * 8 * n, n, i, k);
        g1_mul_aff_sel_hbm_inl<C>(p, k, a.glv != 0, a.vtab + (size_t)part * G1_TAB * 2 * N * n + i, n, r);
        g1j_store<C>(out, n, i, r);
    }
};
// lane per (chunk, item)
template <class C>
struct MsmFixedChunk {
    static __host__ __device__ void run(const MsmArgs<C>& a, size_t t) {
        constexpr int N = C::FpP::N;
        const size_t n = a.n;
        const int chunk = (int)(t / n);
        const size_t i = t - (size_t)chunk * n;
        if (a.status[i] != ST_PENDING) return;
        G1Jac<C> r;
        fixed_msm_chunk_to<C>(*a.cc, a.fscal, n, i, a.n_fixed, chunk, r);
        g1j_store<C>(a.partials + (size_t)(a.n_var + chunk) * 3 * N * n, n, i, r);
    }
};
// the fixed-base sum as one tree of affine additions per item (bbs_ctx_set_fixed_base_tree; lane per item)
template <class C>
struct MsmFixedTree {
    static __host__ __device__ void run(const MsmArgs<C>& a, size_t i) {
        constexpr int N = C::FpP::N;
        const size_t n = a.n;
        if (a.status[i] != ST_PENDING) return;
        G1Jac<C> r = g1j_inf<C>();
        for (int f = 1; f < NFIX; f++) g1j_store<C>(a.partials + (size_t)(a.n_var + f) * 3 * N * n, n, i, r);
        fixed_msm_tree_to<C>(*a.cc, a.fscal, n, i, a.n_fixed, a.fixwk, r);
        g1j_store<C>(a.partials + (size_t)a.n_var * 3 * N * n, n, i, r);
    }
};

template <class C>
struct MsmCombine {
    static __host__ __device__ void run(const MsmArgs<C>& a, size_t i) {
        constexpr int N = C::FpP::N;
        const size_t n = a.n;
        if (a.status[i] != ST_PENDING) return;
        G1Jac<C> acc = g1j_inf<C>();
        for (int p = 0; p < a.n_var + NFIX; p++) acc = g1j_add_i<C>(acc, g1j_load<C>(a.partials + (size_t)p * 3 * N * n, n, i));
        g1a_store_canon<C>(a.out, n, i, g1j_to_aff<C>(acc));
        a.status[i] = 1;
    }
};

// canonical affine inputs -> Montgomery, on-curve check, status 2 (pairing pending)
template <class C>
struct PairPrep {
    const uint32_t* pa_c; const uint32_t* pb_c; uint32_t* pa; uint32_t* pb; int8_t* status; size_t n;
    static __host__ __device__ void run(const PairPrep<C>& a, size_t i) {
        if (a.status[i] != ST_PENDING) return;     // flagged by validation
        G1Aff<C> p = g1a_load_canon_to_mont<C>(a.pa_c, a.n, i), q = g1a_load_canon_to_mont<C>(a.pb_c, a.n, i);
        if (!g1a_on_curve<C>(p) || !g1a_on_curve<C>(q)) { a.status[i] = -41; return; }
        g1a_store_mont<C>(a.pa, a.n, i, p);
        g1a_store_mont<C>(a.pb, a.n, i, q);
        a.status[i] = ST_PAIRING;
    }
};

#if !defined(BBS_HOST_TWIN)
// =============================================================================================
// wavefront-cooperative pairing check (pairing_dist.hpp): six lanes per item, Miller loop of both
// pairs (shared squarings) and the final exponentiation fused in one kernel, nothing spilled to HBM.
// Thread index: wave = t / 64 ; group = (t % 64) / 6 ; item = wave * 10 + group.
// =============================================================================================
template <class C>
struct PairDist {
    static constexpr int WAVES_PER_EU = BBS_PAIR_WAVES;      // 1: the whole register file for one wavefront (measured best)
    static __device__ void run(const PairArgs<C>& a, size_t t) {
        const int lane = (int)(t & 63);
        const int grp = lane / GRP;
        if (grp >= GRP_PER_WAVE) return;
        const size_t i = (t >> 6) * GRP_PER_WAVE + grp;
        if (i >= a.n) return;
        if (a.gate_arr[i] != a.gate) return;
        Lane6 L{grp * GRP, lane - grp * GRP};
        if (pair_batch_passed<C>(a)) { if (L.m == 0) a.out[i] = 1; return; }
        G1Aff<C> Pa = pair_load_point<C>(a, a.pa, i);
        G1Aff<C> Pb = pair_load_point<C>(a, a.pb, i);
        if (a.negate_b) Pb = g1a_neg<C>(Pb);
        const CtxConsts<C>* cc = a.cc;
        const bool skipA = g1a_is_inf<C>(Pa) | (cc->tab_pk.q_is_identity != 0);
        const bool skipB = g1a_is_inf<C>(Pb) | (cc->tab_bp2.q_is_identity != 0);
        Fp2<C> f = d_one<C>(L);
        if (!(skipA & skipB)) {
            // the Miller accumulator is its own variable, never handed by reference to a non-inlined function: that
            // would make it a memory object and put a scratch store / load of it around every step of the loop
            Fp2<C> m = d_one<C>(L);
            int li = 0;
            const int nops = cc->sched.n_ops;
            for (int k = 0; k < nops; k++) {
                if (cc->sched.op[k] == 0) {
                    m = d_sqr<C>(L, m);
                } else {
                    if (!skipA) m = d_mul_line<C>(L, m, cc->tab_pk.e[li], Pa);
                    if (!skipB) m = d_mul_line<C>(L, m, cc->tab_bp2.e[li], Pb);
                    li++;
                }
            }
            if constexpr (C::K::X_NEG) m = d_conj<C>(L, m);
            const Fp2<C> mf = m;
            f = d_final_exp<C>(L, mf, &cc->frob[0][0][0][0]);
        }
        const bool one = d_is_one<C>(L, f);
        if (L.m == 0) a.out[i] = one ? 1 : 0;
    }
};

// ---- throughput form in two kernels (round 4, BBS_PAIR_SPLIT2): PairDist up to the end of the Miller loops; the value is
// handed to PairFinalDist in HBM ([coefficient][2N][n], coalesced over the items' lanes m: 1.3 KB per item each way)
template <class C>
struct PairMillerBoth {
    static constexpr int WAVES_PER_EU = BBS_PAIR_WAVES;
    static __device__ void run(const PairArgs<C>& a, size_t t) {
        constexpr int N = C::FpP::N;
        const int lane = (int)(t & 63);
        const int grp = lane / GRP;
        if (grp >= GRP_PER_WAVE) return;
        const size_t i = (t >> 6) * GRP_PER_WAVE + grp;
        if (i >= a.n) return;
        if (a.gate_arr[i] != a.gate) return;
        if (pair_batch_passed<C>(a)) return;                   // PairFinalDist writes the verdict
        Lane6 L{grp * GRP, lane - grp * GRP};
        G1Aff<C> Pa = pair_load_point<C>(a, a.pa, i);
        G1Aff<C> Pb = pair_load_point<C>(a, a.pb, i);
        if (a.negate_b) Pb = g1a_neg<C>(Pb);
        const CtxConsts<C>* cc = a.cc;
        const bool skipA = g1a_is_inf<C>(Pa) | (cc->tab_pk.q_is_identity != 0);
        const bool skipB = g1a_is_inf<C>(Pb) | (cc->tab_bp2.q_is_identity != 0);
        Fp2<C> m = d_one<C>(L);
        if (!(skipA & skipB)) {
            int li = 0;
            const int nops = cc->sched.n_ops;
            for (int k = 0; k < nops; k++) {
                if (cc->sched.op[k] == 0) {
                    m = d_sqr<C>(L, m);
                } else {
                    if (!skipA) m = d_mul_line<C>(L, m, cc->tab_pk.e[li], Pa);
                    if (!skipB) m = d_mul_line<C>(L, m, cc->tab_bp2.e[li], Pb);
                    li++;
                }
            }
            if constexpr (C::K::X_NEG) m = d_conj<C>(L, m);
        }
        uint32_t* o = a.fmiller + (size_t)L.m * 2 * N * a.n + i;
#pragma unroll
        for (int j = 0; j < N; j++) { o[(size_t)j * a.n] = m.c0.v[j]; o[(size_t)(N + j) * a.n] = m.c1.v[j]; }
    }
};

// ---- latency form (round 3): the two Miller loops of an item on SEPARATE six-lane groups ------------------------------
// PairDist runs both pairs of an item on one group (63 shared squarings + 2 x 68 line products) and then the final
// exponentiation: 410 wavefronts of ~4.9 ms for a 4096-item batch on a chip of 1024 SIMDs.  When a batch has the chip to
// itself that is the critical path.  Here wavefront 2 v runs the loop of pair 0 = (Pa, pk) and wavefront 2 v + 1 the loop
// of pair 1 = (+-Pb, BP2) of the same ten items (the line table is uniform per wavefront: scalar loads), each 63
// squarings + 68 line products (0.66 of the joint loop), the two values are handed over in HBM ([pair][coefficient]
// [2N][n], coalesced over the items' lanes m) and PairFinalDist multiplies them and runs the final exponentiation:
// 820 wavefronts x 0.66 + 410 wavefronts x 1 instead of 410 x 2: 17 % more wave-time, a critical path ~0.8 ms shorter.
template <class C>
struct PairMillerHalf {
    static constexpr int WAVES_PER_EU = BBS_PAIR_WAVES;
    static __device__ void run(const PairArgs<C>& a, size_t t) {
        constexpr int N = C::FpP::N;
        const int lane = (int)(t & 63);
        const int grp = lane / GRP;
        if (grp >= GRP_PER_WAVE) return;
        const size_t wave = t >> 6;
        const int pair = (int)(wave & 1);
        const size_t i = (wave >> 1) * GRP_PER_WAVE + grp;
        if (i >= a.n) return;
        if (a.gate_arr[i] != a.gate) return;
        if (pair_batch_passed<C>(a)) return;                   // PairFinalDist writes the verdict
        Lane6 L{grp * GRP, lane - grp * GRP};
        G1Aff<C> P = pair_load_point<C>(a, pair ? a.pb : a.pa, i);
        if (pair && a.negate_b) P = g1a_neg<C>(P);
        const CtxConsts<C>* cc = a.cc;
        const LineTable<C>& tab = pair ? cc->tab_bp2 : cc->tab_pk;
        const bool skip = g1a_is_inf<C>(P) | (tab.q_is_identity != 0);
        Fp2<C> m = d_one<C>(L);
        if (!skip) {
            int li = 0;
            const int nops = cc->sched.n_ops;
            for (int k = 0; k < nops; k++) {
                if (cc->sched.op[k] == 0) m = d_sqr<C>(L, m);
                else m = d_mul_line<C>(L, m, tab.e[li++], P);
            }
            if constexpr (C::K::X_NEG) m = d_conj<C>(L, m);
        }
        uint32_t* o = a.fmiller + ((size_t)pair * GRP + L.m) * 2 * N * a.n + i;
#pragma unroll
        for (int j = 0; j < N; j++) { o[(size_t)j * a.n] = m.c0.v[j]; o[(size_t)(N + j) * a.n] = m.c1.v[j]; }
    }
};
template <class C>
struct PairFinalDist {
    static constexpr int WAVES_PER_EU = BBS_PAIRFINAL_WAVES;      // 2: fits 256 registers, see BBS_PAIR_SPLIT2
    static constexpr int V = BBS_PAIRFINAL_WAVES > 1 ? 1 : 0;     // its own instances of the non-inlined functions (pairing_dist.hpp d_final_exp)
    static __device__ void run(const PairArgs<C>& a, size_t t) {
        constexpr int N = C::FpP::N;
        const int lane = (int)(t & 63);
        const int grp = lane / GRP;
        if (grp >= GRP_PER_WAVE) return;
        const size_t i = (t >> 6) * GRP_PER_WAVE + grp;
        if (i >= a.n) return;
        if (a.gate_arr[i] != a.gate) return;
        Lane6 L{grp * GRP, lane - grp * GRP};
        if (pair_batch_passed<C>(a)) { if (L.m == 0) a.out[i] = 1; return; }
        Fp2<C> g0;
        const uint32_t* p0 = a.fmiller + (size_t)L.m * 2 * N * a.n + i;
#pragma unroll
        for (int j = 0; j < N; j++) { g0.c0.v[j] = p0[(size_t)j * a.n]; g0.c1.v[j] = p0[(size_t)(N + j) * a.n]; }
        Fp2<C> mf = g0;
        if (!a.single) {                                        // (uniform over the launch)
            Fp2<C> g1;
            const uint32_t* p1 = a.fmiller + ((size_t)GRP + L.m) * 2 * N * a.n + i;
#pragma unroll
            for (int j = 0; j < N; j++) { g1.c0.v[j] = p1[(size_t)j * a.n]; g1.c1.v[j] = p1[(size_t)(N + j) * a.n]; }
            mf = d_mul<C, V>(L, g0, g1);
        }
        const Fp2<C> mfc = mf;
        const Fp2<C> f = d_final_exp<C, V>(L, mfc, &a.cc->frob[0][0][0][0]);
        const bool one = d_is_one<C>(L, f);
        if (L.m == 0) a.out[i] = one ? 1 : 0;
    }
};

// self-test: one Fp12 operation computed by the one-lane code and by the six-lane code
template <class C>
struct SelfTestArgs {
    int op;
    const CtxConsts<C>* cc;
    const uint32_t* a;      // 12 Fp (tower order c0.c0.c0, c0.c0.c1, c0.c1.c0 ... ), Montgomery
    const uint32_t* b;
    uint32_t* out_single;   // 12 Fp
    uint32_t* out_dist;     // 12 Fp
};
// six-lane version; input x = out_dist as prepared by the host (already cyclotomic for OP >= 10)
template <class C, int OP>
struct SelfTestDist {
    static __device__ void run(const SelfTestArgs<C>& a, size_t t) {
        constexpr int N = C::FpP::N;
        const int lane = (int)(t & 63);
        const int grp = lane / GRP;
        if (grp >= 1) return;
        Lane6 L{grp * GRP, lane - grp * GRP};
        // w-basis coefficient m of a tower-ordered array: g_m = (e[2k], e[2k+1]) with k = (m & 1) * 3 + (m >> 1)
        const int k = (L.m & 1) * 3 + (L.m >> 1);
        Fp2<C> gx, gy;
        for (int j = 0; j < N; j++) {
            gx.c0.v[j] = a.out_dist[(2 * k) * N + j]; gx.c1.v[j] = a.out_dist[(2 * k + 1) * N + j];
            gy.c0.v[j] = a.b[(2 * k) * N + j]; gy.c1.v[j] = a.b[(2 * k + 1) * N + j];
        }
        G1Aff<C> P;
        for (int j = 0; j < N; j++) { P.x.v[j] = a.b[j]; P.y.v[j] = a.b[N + j]; }
        const uint32_t* ft = &a.cc->frob[0][0][0][0];
        Fp2<C> rd;
        if constexpr (OP == 0) rd = d_mul<C>(L, gx, gy);
        else if constexpr (OP == 1) rd = d_frob<C, 1>(L, gx, ft);
        else if constexpr (OP == 2) rd = d_frob<C, 2>(L, gx, ft);
        else if constexpr (OP == 3) rd = d_frob<C, 3>(L, gx, ft);
        else if constexpr (OP == 4) rd = d_inv<C>(L, gx);
        else if constexpr (OP == 5) rd = d_conj<C>(L, gx);
        else if constexpr (OP == 6) rd = d_mul_line<C>(L, gx, a.cc->tab_bp2.e[3], P);
        else if constexpr (OP == 7) rd = d_final_exp<C>(L, gx, ft);
        else if constexpr (OP == 10) rd = d_cyclo_sqr<C>(L, gx);
        else if constexpr (OP == 8) rd = d_sqr<C>(L, gx);
        else if constexpr (OP == 11) rd = d_pow_x<C>(L, gx);
        else rd = gx;
        for (int j = 0; j < N; j++) { a.out_dist[(2 * k) * N + j] = rd.c0.v[j]; a.out_dist[(2 * k + 1) * N + j] = rd.c1.v[j]; }
    }
};
#endif

}  // namespace bbs
