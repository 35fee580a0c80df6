// Large-n variable-base multi-scalar multiplication  S = sum_i k_i * P_i  by the bucket (Pippenger) method,
// and the random-linear-combination batch check built on it.
//
// The per-item multi-scalar multiplications of the four core operations (stages.hpp) have at most 38 terms and
// fixed bases: they use window tables.  THIS file is the other shape: thousands of per-item points, one sum.
// Used by
//   * bbs_g1_msm_pippenger (unit parity against the oracle's sum), and
//   * batch verification: instead of n two-pairing products e(Abar_i, W) e(Bbar_i, -BP2) == 1
//     (src/proof_verify.rs:112-115) one checks  e(sum rho_i Abar_i, W) e(sum rho_i Bbar_i, -BP2) == 1  for secret
//     random 128-bit rho_i; the left side equals prod_i t_i^rho_i where t_i is item i's own pairing product, an
//     r-th root of unity after the final exponentiation, so a batch that passes has every t_i = 1 except with
//     probability 2^-128 (cofactor components of a point never reach GT: the final exponentiation removes them).
//     A batch that fails is re-checked item by item with the exact kernel, so the booleans are the reference's.
//
// Stages (all lane-per-work-unit functors like the rest of the engine; window c = 8 bits, digits in HBM as bytes):
//   PipBuckets : lane per (point set, window, bucket b): scans the window's n digits (every lane of a wavefront
//                reads the same words -> one broadcast load per 4 items), claims its slice of the window's index
//                list (offset = number of smaller non-zero digits), fills it, then sums its points with mixed
//                additions -- no atomics, no conflicts, every lane of a wavefront in the same loop.
//   PipSegments: lane per 16 buckets: running sums  sum_b (b - 16 s) B_b  and  sum_b B_b, then  + 16 s * (sum B_b).
//   PipWindows : lane per window: adds its 16 segments and shifts by 2^(8 w).
//   PipFinal   : lane per point set: adds the windows, one inversion, affine Montgomery out.
// Batch verification does not need the one 128-bit combination: the 16 windows are 16 INDEPENDENT 8-bit
// combinations (digit bytes of a hash are independent and uniform), each checked by its own pairing product; a bad
// item survives one of them with probability <= 1/256 and all of them with 2^-128.  So PipWindowSums (no shifts)
// writes 16 affine sums per point set and the 16 checks run side by side in two wavefronts of the pairing kernel --
// no 120 doublings, no final addition chain.
#pragma once
#include "stages.hpp"

namespace bbs {

constexpr int PIP_C = 8;             // window bits
constexpr int PIP_NB = 1 << PIP_C;   // buckets per window (bucket 0 unused)
constexpr int PIP_SEG = 16;          // buckets per segment

template <class C>
struct PipArgs {
    size_t n;                 // points per set
    size_t n_pad;             // n rounded up to a multiple of 4: length of a digit row
    int M;                    // point sets sharing the digits (1 or 2)
    int NW;                   // windows = scalar bits / 8
    const uint32_t* pts0;     // [2N][n] Montgomery affine, (0,0) = identity
    const uint32_t* pts1;
    const uint8_t* dig;       // [NW][n_pad] window digits, 0 = no contribution
    uint32_t* list;           // [M][NW][n] item indexes grouped by bucket
    uint32_t* buckets;        // [3N][M*NW*256] Jacobian
    uint32_t* segs;           // [3N][M*NW*16]
    uint32_t* wins;           // [3N][M*NW]
    uint32_t* out;            // [M][2N] Montgomery affine sums
};

template <class C>
struct PipBuckets {
    static __host__ __device__ void run(const PipArgs<C>& a, size_t t) {
        const size_t T = (size_t)a.M * a.NW * PIP_NB;
        const uint32_t b = (uint32_t)(t & (PIP_NB - 1));
        const size_t mw = t >> PIP_C;
        const int m = (int)(mw / (size_t)a.NW);
        const int w = (int)(mw - (size_t)m * a.NW);
        if (b == 0) { g1j_store<C>(a.buckets, T, t, g1j_inf<C>()); return; }
        const uint32_t* dw = reinterpret_cast<const uint32_t*>(a.dig + (size_t)w * a.n_pad);
        const size_t nq = a.n_pad >> 2;
        uint32_t lower = 0, cnt = 0;
        for (size_t q = 0; q < nq; q++) {
            const uint32_t word = dw[q];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t d = (word >> (8 * k)) & 0xffu;
                lower += ((d - 1u) < (b - 1u)) ? 1u : 0u;        // 0 < d < b
                cnt += (d == b) ? 1u : 0u;
            }
        }
        uint32_t* lst = a.list + mw * a.n + lower;
        uint32_t pos = 0;
        for (size_t q = 0; q < nq && pos < cnt; q++) {
            const uint32_t word = dw[q];
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (((word >> (8 * k)) & 0xffu) == b) lst[pos++] = (uint32_t)(4 * q + k);
        }
        const uint32_t* pts = m ? a.pts1 : a.pts0;
        G1Jac<C> acc = g1j_inf<C>();
        for (uint32_t k = 0; k < cnt; k++) acc = g1j_add_aff<C>(acc, g1a_load_mont<C>(pts, a.n, lst[k]));
        g1j_store<C>(a.buckets, T, t, acc);
    }
};

template <class C>
struct PipSegments {
    static __host__ __device__ void run(const PipArgs<C>& a, size_t t) {
        const size_t T = (size_t)a.M * a.NW * PIP_NB, TS = (size_t)a.M * a.NW * (PIP_NB / PIP_SEG);
        const uint32_t s = (uint32_t)(t % (PIP_NB / PIP_SEG));
        const size_t base = (t / (PIP_NB / PIP_SEG)) * PIP_NB + (size_t)s * PIP_SEG;
        G1Jac<C> run = g1j_inf<C>(), acc = g1j_inf<C>();
        for (int b = PIP_SEG - 1; b >= 1; b--) {
            run = g1j_add<C>(run, g1j_load<C>(a.buckets, T, base + b));
            acc = g1j_add<C>(acc, run);                      // acc = sum_b (b - 16 s) B_b
        }
        run = g1j_add<C>(run, g1j_load<C>(a.buckets, T, base));
        if (s) {                                             // + 16 s * sum_b B_b
            G1Jac<C> sr = g1j_inf<C>();
            for (int bit = 3; bit >= 0; bit--) {
                sr = g1j_dbl<C>(sr);
                if ((s >> bit) & 1u) sr = g1j_add<C>(sr, run);
            }
            for (int k = 0; k < 4; k++) sr = g1j_dbl<C>(sr);
            acc = g1j_add<C>(acc, sr);
        }
        g1j_store<C>(a.segs, TS, t, acc);
    }
};

template <class C>
struct PipWindows {
    static __host__ __device__ void run(const PipArgs<C>& a, size_t t) {
        const size_t TS = (size_t)a.M * a.NW * (PIP_NB / PIP_SEG), TW = (size_t)a.M * a.NW;
        const int w = (int)(t % (size_t)a.NW);
        G1Jac<C> acc = g1j_inf<C>();
        for (int s = 0; s < PIP_NB / PIP_SEG; s++) acc = g1j_add<C>(acc, g1j_load<C>(a.segs, TS, t * (PIP_NB / PIP_SEG) + s));
        for (int k = 0; k < PIP_C * w; k++) acc = g1j_dbl<C>(acc);
        g1j_store<C>(a.wins, TW, t, acc);
    }
};

template <class C>
struct PipFinal {
    static __host__ __device__ void run(const PipArgs<C>& a, size_t m) {
        constexpr int N = C::FpP::N;
        const size_t TW = (size_t)a.M * a.NW;
        G1Jac<C> acc = g1j_inf<C>();
        for (int w = 0; w < a.NW; w++) acc = g1j_add<C>(acc, g1j_load<C>(a.wins, TW, m * a.NW + w));
        g1a_store_mont<C>(a.out + m * 2 * N, 1, 0, g1j_to_aff<C>(acc));
    }
};

// batch verification: window sums without the 2^(8 w) shift, normalised: out[m] is an SoA array of NW affine points
// ([2N][NW], Montgomery), i.e. NW "items" for the pairing stage
template <class C>
struct PipWindowSums {
    static __host__ __device__ void run(const PipArgs<C>& a, size_t t) {
        constexpr int N = C::FpP::N;
        const size_t TS = (size_t)a.M * a.NW * (PIP_NB / PIP_SEG);
        const size_t m = t / (size_t)a.NW, w = t - m * (size_t)a.NW;
        G1Jac<C> acc = g1j_inf<C>();
        for (int s = 0; s < PIP_NB / PIP_SEG; s++) acc = g1j_add<C>(acc, g1j_load<C>(a.segs, TS, t * (PIP_NB / PIP_SEG) + s));
        g1a_store_mont<C>(a.out + m * 2 * N * a.NW, (size_t)a.NW, w, g1j_to_aff<C>(acc));
    }
};

// ---- batch verification glue ------------------------------------------------------------------
struct RlcArgs {
    size_t n, n_pad;
    int8_t* status;           // ST_PAIRING = challenge matched, pairing pending
    uint32_t seed[8];         // secret per-batch seed
    uint8_t* dig;             // [16][n_pad]
    const int8_t* batch_ok;   // [n_checks] results of the combined pairing checks
    int n_checks;
};

// rho_i = first 128 bits of SHA-256(seed || I2OSP(i, 8)) for pending items, 0 otherwise
struct RlcScalars {
    static __host__ __device__ void run(const RlcArgs& a, size_t i) {
        uint32_t h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        if (a.status[i] == ST_PAIRING) {
            Sha256 s;
            sha256_init(s);
            for (int k = 0; k < 8; k++) sha256_word(s, a.seed[k]);
            sha256_u64be(s, (uint64_t)i);
            sha256_final(s, h);
        }
        for (int w = 0; w < 16; w++) a.dig[(size_t)w * a.n_pad + i] = (uint8_t)(h[w >> 2] >> (8 * (w & 3)));
    }
};

// all combined checks passed: every pending item's pairing product is 1
struct RlcApply {
    static __host__ __device__ void run(const RlcArgs& a, size_t i) {
        if (a.status[i] != ST_PAIRING) return;
        int ok = 1;
        for (int k = 0; k < a.n_checks; k++) ok &= (a.batch_ok[k] == 1);
        if (ok) a.status[i] = 1;
    }
};

// generic digits for the unit-parity primitive: 256-bit canonical scalars -> 32 byte digits
struct PipDigitArgs { size_t n, n_pad; const uint32_t* scal; /* [8][n] */ const int8_t* status; uint8_t* dig; /* [32][n_pad] */ };
struct PipDigits {
    static __host__ __device__ void run(const PipDigitArgs& a, size_t i) {
        for (int w = 0; w < 32; w++) {
            const uint32_t l = a.scal[(size_t)(w >> 2) * a.n + i];
            a.dig[(size_t)w * a.n_pad + i] = a.status[i] != 1 ? 0 : (uint8_t)(l >> (8 * (w & 3)));
        }
    }
};

// canonical affine points -> Montgomery, on-curve check (status -41, the item then contributes nothing)
template <class C>
struct PipPrep {
    const uint32_t* pts_c; uint32_t* pts; int8_t* status; size_t n;
    static __host__ __device__ void run(const PipPrep<C>& a, size_t i) {
        G1Aff<C> p = g1a_inf<C>();
        if (a.status[i] == ST_PENDING) {
            p = g1a_load_canon_to_mont<C>(a.pts_c, a.n, i);
            if (!g1a_on_curve<C>(p)) { a.status[i] = -41; p = g1a_inf<C>(); }
            else a.status[i] = 1;
        }
        g1a_store_mont<C>(a.pts, a.n, i, p);
    }
};

}  // namespace bbs
