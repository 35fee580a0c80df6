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
// Host-twin stages (lane-per-work-unit functors; window c = 8 bits, digits in HBM as bytes) -- the device runs the
// workgroup-cooperative kernel k_pip_window further down instead of the first three:
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
    const uint32_t* ppts;     // [M][n][2N] Montgomery affine, item-major, (0,0) = identity
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
        const uint32_t* pts = a.ppts + (size_t)m * a.n * 2 * C::FpP::N;
        G1Jac<C> acc = g1j_inf<C>();
        for (uint32_t k = 0; k < cnt; k++) acc = g1j_add_aff<C>(acc, g1a_load_mont<C>(pts + (size_t)lst[k] * 2 * C::FpP::N, 1, 0));
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

// =============================================================================================
// Workgroup-cooperative bucket accumulation (round 3; device only -- the lane functors above remain the host twin's path)
// =============================================================================================
// One workgroup of 256 threads (four wavefronts = one per SIMD of a compute unit) per (point set m, window w, tile of
// PIP_TILE items); thread b OWNS bucket b.
//   1. the tile's digit bytes of window w are staged ONCE into LDS (coalesced 16-byte loads);
//   2. bucket membership by wavefront ballots: for every 64 digits the eight ballots B_j of their bits are formed, and
//      thread b's members are the lanes in  AND_j (b_j ? B_j : ~B_j)  -- 16 mask operations per 64 items instead of 64
//      byte compares; a first pass counts (popcount), an exclusive scan over the 256 counts (wavefront shuffles + one
//      LDS hop between the four wavefronts) gives every bucket its slice of an index list in LDS, a second pass fills
//      it in item order (deterministic: no atomics);
//   3. thread b sums the points of its slice (mixed additions; points read as 16-byte vectors from an item-major copy);
//   4. sum_b b * B_b  =  sum_{b >= 1} T_b  with the suffix sums T_b = sum_{j >= b} B_j: a Hillis-Steele scan and a tree
//      reduction over the 256 threads, the points moved between lanes with wavefront shuffles and between the four
//      wavefronts through LDS -- 19 Jacobian additions deep instead of the 48 of the segment / window-sum stages;
//   5. thread 0 stores the tile's window sum (Jacobian).  Window sums are linear in the bucket contents, so tiles simply
//      add up: PipTileSums (lane per window) adds the tiles and normalises (or shifts, for the plain MSM).
// HBM traffic per workgroup: PIP_TILE digit bytes + its points once (112 B each on BLS12-381) -- the algorithmic minimum.
constexpr int PIP_TILE = 4096;
constexpr int PIP_WG = 256;

template <class C>
struct PipCoopArgs {
    size_t n, n_pad;
    int M, NW, n_tiles;
    const uint32_t* ppts;     // [M][n][2N] Montgomery affine, ITEM-major (one point = 2N consecutive words), (0,0) = identity
    const uint8_t* dig;       // [NW][n_pad]
    uint32_t* tile_sums;      // [3N][M*NW*n_tiles] Jacobian (SoA over the work units)
    uint32_t* out_aff;        // n_tiles == 1 only, or null: the window sum normalised straight into [M][2N][NW] (no PipTileSums launch)
};

#if !defined(BBS_HOST_TWIN)
template <class C>
__device__ __forceinline__ G1Jac<C> g1j_shfl_down(const G1Jac<C>& p, int delta) {
    G1Jac<C> r;
#pragma unroll
    for (int j = 0; j < C::FpP::N; j++) {
        r.x.v[j] = (uint32_t)__shfl_down((int)p.x.v[j], delta, 64);
        r.y.v[j] = (uint32_t)__shfl_down((int)p.y.v[j], delta, 64);
        r.z.v[j] = (uint32_t)__shfl_down((int)p.z.v[j], delta, 64);
    }
    return r;
}
template <class C>
__device__ __forceinline__ void g1j_to_lds(uint32_t* s, const G1Jac<C>& p) {
#pragma unroll
    for (int j = 0; j < C::FpP::N; j++) { s[j] = p.x.v[j]; s[C::FpP::N + j] = p.y.v[j]; s[2 * C::FpP::N + j] = p.z.v[j]; }
}
template <class C>
__device__ __forceinline__ G1Jac<C> g1j_from_lds(const uint32_t* s) {
    G1Jac<C> p;
#pragma unroll
    for (int j = 0; j < C::FpP::N; j++) { p.x.v[j] = s[j]; p.y.v[j] = s[C::FpP::N + j]; p.z.v[j] = s[2 * C::FpP::N + j]; }
    return p;
}

template <class C>
__global__ void __launch_bounds__(PIP_WG) k_pip_window(PipCoopArgs<C> a) {
    constexpr int N = C::FpP::N;
    __shared__ __attribute__((aligned(16))) uint8_t s_dig[PIP_TILE];
    __shared__ uint16_t s_list[PIP_TILE];
    __shared__ uint32_t s_wave[4];
    __shared__ uint32_t s_pt[4][3 * N];
    const int b = (int)threadIdx.x, lane = b & 63, wv = b >> 6;
    const size_t unit = blockIdx.x;                          // (m * NW + w) * n_tiles + tile
    const int tile = (int)(unit % (size_t)a.n_tiles);
    const size_t mw = unit / (size_t)a.n_tiles;
    const int w = (int)(mw % (size_t)a.NW), m = (int)(mw / (size_t)a.NW);
    const size_t i0 = (size_t)tile * PIP_TILE;
    const int n_t = (int)((a.n - i0) < (size_t)PIP_TILE ? (a.n - i0) : (size_t)PIP_TILE);
    // 1. digits -> LDS (the row is padded to a multiple of 4 and a tile starts at a multiple of 4096: 4-byte loads are aligned)
    {
        const uint8_t* row = a.dig + (size_t)w * a.n_pad + i0;
        uint32_t* s32 = reinterpret_cast<uint32_t*>(s_dig);
        for (int q = b; q < PIP_TILE / 4; q += PIP_WG) {
            uint32_t v = 0;
            if (4 * q < n_t) {
                v = *reinterpret_cast<const uint32_t*>(row + 4 * q);
                const int rem = n_t - 4 * q;
                if (rem < 4) v &= (1u << (8 * rem)) - 1u;
            }
            s32[q] = v;
        }
    }
    __syncthreads();
    const int n_chunks = (n_t + 63) >> 6;
    // membership mask of bucket b among the 64 digits of chunk q (every wavefront forms the same eight ballots)
    auto members = [&](int q) -> uint64_t {
        const uint32_t d = s_dig[q * 64 + lane];
        uint64_t mk = ~(uint64_t)0;
#pragma unroll
        for (int j = 0; j < PIP_C; j++) {
            const uint64_t bj = __ballot((d >> j) & 1u);
            mk &= ((b >> j) & 1) ? bj : ~bj;
        }
        return b ? mk : (uint64_t)0;                         // digit 0 contributes nothing
    };
    // 2a. counts
    uint32_t cnt = 0;
    for (int q = 0; q < n_chunks; q++) cnt += (uint32_t)__popcll(members(q));
    // 2b. exclusive scan of the 256 counts: inclusive scan inside the wavefront, wavefront totals through LDS
    uint32_t inc = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = (uint32_t)__shfl_up((int)inc, off, 64);
        if (lane >= off) inc += v;
    }
    if (lane == 63) s_wave[wv] = inc;
    __syncthreads();
    uint32_t start = inc - cnt;
    for (int k = 0; k < wv; k++) start += s_wave[k];
    // 2c. fill (item order)
    {
        uint32_t pos = start;
        for (int q = 0; q < n_chunks; q++) {
            uint64_t mk = members(q);
            while (mk) {
                const int l = __ffsll((unsigned long long)mk) - 1;
                mk &= mk - 1;
                s_list[pos++] = (uint16_t)(q * 64 + l);
            }
        }
    }
    __syncthreads();
    // 3. bucket sum
    G1Jac<C> acc = g1j_inf<C>();
    {
        const uint32_t* pts = a.ppts + ((size_t)m * a.n + i0) * (2 * N);
        for (uint32_t k = 0; k < cnt; k++) {
            const uint4* e = reinterpret_cast<const uint4*>(pts + (size_t)s_list[start + k] * (2 * N));
            uint32_t wds[2 * N];
            static_assert((2 * N) % 4 == 0, "a point is a whole number of 16-byte vectors");
#pragma unroll
            for (int v = 0; v < (2 * N) / 4; v++) { const uint4 t = e[v]; wds[4 * v] = t.x; wds[4 * v + 1] = t.y; wds[4 * v + 2] = t.z; wds[4 * v + 3] = t.w; }
            G1Aff<C> pnt;
#pragma unroll
            for (int j = 0; j < N; j++) { pnt.x.v[j] = wds[j]; pnt.y.v[j] = wds[N + j]; }
            acc = g1j_add_aff<C>(acc, pnt);
        }
    }
    // 4a. suffix sums T_b = sum_{j >= b} B_j: inside the wavefront by shuffles ...
#pragma unroll 1
    for (int off = 1; off < 64; off <<= 1) {
        const G1Jac<C> o = g1j_shfl_down<C>(acc, off);
        if (lane + off < 64) acc = g1j_add<C>(acc, o);
    }
    // ... then + everything held by the wavefronts above (their totals are in their lane 0)
    if (lane == 0) g1j_to_lds<C>(s_pt[wv], acc);
    __syncthreads();
    {
        G1Jac<C> above = g1j_inf<C>();
        for (int k = 3; k > wv; k--) above = g1j_add<C>(above, g1j_from_lds<C>(s_pt[k]));
        acc = g1j_add<C>(acc, above);
    }
    __syncthreads();
    // 4b. sum_{b >= 1} T_b: bucket 0 holds nothing of its own, its suffix sum is not a term
    if (b == 0) acc = g1j_inf<C>();
#pragma unroll 1
    for (int off = 32; off >= 1; off >>= 1) {
        const G1Jac<C> o = g1j_shfl_down<C>(acc, off);
        if (lane < off) acc = g1j_add<C>(acc, o);
    }
    if (lane == 0) g1j_to_lds<C>(s_pt[wv], acc);
    __syncthreads();
    // 5. thread 0: the tile's window sum
    if (b == 0) {
        for (int k = 1; k < 4; k++) acc = g1j_add<C>(acc, g1j_from_lds<C>(s_pt[k]));
        if (a.out_aff && a.n_tiles == 1) g1a_store_mont<C>(a.out_aff + (size_t)m * 2 * N * a.NW, (size_t)a.NW, (size_t)w, g1j_to_aff<C>(acc));
        else g1j_store<C>(a.tile_sums, (size_t)a.M * a.NW * a.n_tiles, unit, acc);
    }
}
#endif

// lane per (m, w): add the tiles' window sums.  shift = 0: normalise into the SoA array of NW affine points per set the
// combined pairing checks read (batch verification: the windows are independent 8-bit combinations); shift = 1: the plain
// MSM -- multiply window w by 2^(8 w) and leave it Jacobian in `wins` for PipFinal.
template <class C>
struct PipTileSumArgs {
    int M, NW, n_tiles, shift;
    const uint32_t* tile_sums;   // [3N][M*NW*n_tiles]
    uint32_t* out;               // [M][2N][NW] Montgomery affine (shift = 0)
    uint32_t* wins;              // [3N][M*NW] (shift = 1)
};
template <class C>
struct PipTileSums {
    static __host__ __device__ void run(const PipTileSumArgs<C>& a, size_t t) {
        constexpr int N = C::FpP::N;
        const size_t TU = (size_t)a.M * a.NW * a.n_tiles, TW = (size_t)a.M * a.NW;
        const size_t m = t / (size_t)a.NW, w = t - m * (size_t)a.NW;
        G1Jac<C> acc = g1j_inf<C>();
        for (int k = 0; k < a.n_tiles; k++) acc = g1j_add<C>(acc, g1j_load<C>(a.tile_sums, TU, t * a.n_tiles + k));
        if (a.shift) {
            for (size_t k = 0; k < (size_t)PIP_C * w; k++) acc = g1j_dbl<C>(acc);
            g1j_store<C>(a.wins, TW, t, acc);
        } else {
            g1a_store_mont<C>(a.out + m * 2 * N * a.NW, (size_t)a.NW, w, g1j_to_aff<C>(acc));
        }
    }
};

// ---- batch verification glue ------------------------------------------------------------------
// lane per item, in front of the bucket stage: the item's two points in Montgomery form, item-major, and its sixteen
// digit bytes  rho_i = first 128 bits of SHA-256(seed || I2OSP(i, 8)).  An item takes part iff gate_arr[i] == gate and
// (canonical inputs) both points are on the curve; otherwise its digits are 0 and it contributes nothing.
//   verify      : gate = the status after VfCombine (ST_PAIRING), points = A and e A - B (Montgomery, computed).
//   proof_verify: gate = the status the ingest stage left (ST_PENDING = structurally valid), points = the proof's own
//                 Abar, Bbar (canonical) -- the combination does not wait for the challenge stage: it runs on the job's
//                 second stream beside the MSM chain, as the per-item pairing does.  An item whose challenge does not
//                 match is then part of the combination although its status is already Ok(false): harmless when its
//                 pairing product is 1 (a tampered commitment / scalar), and when it is not, the combined check fails
//                 and the per-item kernel decides every item still pending -- the booleans are the reference's either way.
template <class C>
struct RlcPrepArgs {
    size_t n, n_pad;
    const uint32_t* pa;       // [2NC][n] canonical words or [2N][n] Montgomery limbs
    const uint32_t* pb;
    int canonical;
    const int8_t* gate_arr;
    int gate;
    uint32_t seed[8];         // secret per-batch seed
    uint8_t* dig;             // [16][n_pad]
    uint32_t* ppts;           // [2][n][2N]
};
template <class C>
struct RlcPrep {
    static __host__ __device__ void run(const RlcPrepArgs<C>& a, size_t i) {
        constexpr int N = C::FpP::N;
        uint32_t h[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        G1Aff<C> p = g1a_inf<C>(), q = g1a_inf<C>();
        bool in = a.gate_arr[i] == (int8_t)a.gate;
        if (in) {
            if (a.canonical) {
                p = g1a_load_canon_to_mont<C>(a.pa, a.n, i);
                q = g1a_load_canon_to_mont<C>(a.pb, a.n, i);
                in = g1a_on_curve<C>(p) && g1a_on_curve<C>(q);
            } else {
                p = g1a_load_mont<C>(a.pa, a.n, i);
                q = g1a_load_mont<C>(a.pb, a.n, i);
            }
        }
        if (in) {
            Sha256 s;
            sha256_init(s);
            for (int k = 0; k < 8; k++) sha256_word(s, a.seed[k]);
            sha256_u64be(s, (uint64_t)i);
            sha256_final(s, h);
        } else {
            p = g1a_inf<C>(); q = g1a_inf<C>();
        }
        for (int w = 0; w < 16; w++) a.dig[(size_t)w * a.n_pad + i] = (uint8_t)(h[w >> 2] >> (8 * (w & 3)));
        g1a_store_mont<C>(a.ppts + i * 2 * N, 1, 0, p);
        g1a_store_mont<C>(a.ppts + (a.n + i) * 2 * N, 1, 0, q);
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
    const uint32_t* pts_c; uint32_t* pts; int8_t* status; size_t n;     // pts: [n][2N] item-major
    static __host__ __device__ void run(const PipPrep<C>& a, size_t i) {
        G1Aff<C> p = g1a_inf<C>();
        if (a.status[i] == ST_PENDING) {
            p = g1a_load_canon_to_mont<C>(a.pts_c, a.n, i);
            if (!g1a_on_curve<C>(p)) { a.status[i] = -41; p = g1a_inf<C>(); }
            else a.status[i] = 1;
        }
        g1a_store_mont<C>(a.pts + i * 2 * C::FpP::N, 1, 0, p);
    }
};

}  // namespace bbs
