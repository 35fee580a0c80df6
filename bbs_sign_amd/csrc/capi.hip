// C ABI entry points (include/bbs_sign_amd.h) dispatching on the curve.
#include "ops_decl.hpp"
#include "host_h2c.hpp"
#include "host_codec.hpp"
#include "issuer.hpp"
#include "pool.hpp"

// Many independent batches are kept in flight, every job on streams of its own (proof_verify: three -- the fixed-base chain,
// T1's doubling chain, the pairing).  The HIP runtime maps the streams of a process onto 4 hardware queues unless told
// otherwise, and a long narrow kernel then blocks the streams sharing its queue (measured on MI355X: 645k -> 780k
// proof_verify/s in round 1).  20 (round 5; 14 before): six jobs in flight are 18 streams, and with the largest kernel frame of
// the library at 1.8 KB the scratch budget allows 25 hardware queues (runtime.hpp queue_budget; DESIGN.md 5 rule 6) --
// 1.48 / 1.53 / 1.55 / 1.53 M proof_verify/s at 14 / 18 / 20 / 24 (profiles/r05_a_ab_split_msm_layouts.log).  Takes effect only if
// this library is loaded before the process makes its first HIP call; an explicit setting in the environment wins.
__attribute__((constructor)) static void bbs_default_hw_queues() { setenv("GPU_MAX_HW_QUEUES", "20", 0); }
// what the process environment says now (0 = unset): a service can log it at start-up and, when it reads 4 or was set
// after the first HIP call, prefer larger batches (two 16384-item batches in flight fill the chip: DESIGN.md 6a)
extern "C" __attribute__((visibility("default"))) int bbs_runtime_hw_queues(void) {
    const char* v = getenv("GPU_MAX_HW_QUEUES");
    return v ? atoi(v) : 0;
}

// Job streams with hardware queues of their own (runtime.hpp stream_create): k = 0 off, k > 0 at most k per device (at most
// 16); takes effect for streams created afterwards (streams are pooled: call it before the first context is created).
extern "C" __attribute__((visibility("default"))) int bbs_runtime_set_dedicated_queues(int k) {
    if (k < 0 || k > 16) return BBS_E_ARG;
#ifndef BBS_HOST_TWIN
    rt::dedicated_queues().store(k);      // a wish: runtime.hpp stream_create grants min(k, what the scratch budget leaves beside the pool)
#endif
    return BBS_OK;
}
// The hardware-queue budget of a device (runtime.hpp): scratch bytes per lane of the library's largest kernel frame, the
// number of hardware queues (pooled + dedicated) that frame allows, the runtime's pool as the environment has it, and the
// dedicated queues that fit beside the pool.  Any pointer may be null.
extern "C" __attribute__((visibility("default"))) int bbs_runtime_queue_budget(int device_id, int* total, int* pool, int* dedicated_cap, size_t* scratch_bytes_per_lane) {
    if (device_id < 0 || device_id >= rt::device_count()) return BBS_E_NO_DEVICE;
    if (rt::set_device(device_id)) return BBS_E_HIP;
    const rt::QueueBudget b = rt::queue_budget(device_id);
    if (total) *total = b.total;
    if (pool) *pool = b.pool;
    if (dedicated_cap) *dedicated_cap = b.dedicated_cap;
    if (scratch_bytes_per_lane) *scratch_bytes_per_lane = b.scratch;
    return BBS_OK;
}

// =============================================================================================
// C ABI
// =============================================================================================
#define DISPATCH(ctx, expr_bls, expr_bn) ((ctx)->curve == BBS_CURVE_BLS12_381 ? (expr_bls) : (expr_bn))
#define AS_BLS(ctx) static_cast<Ctx<BlsCurve>*>(ctx)
#define AS_BN(ctx) static_cast<Ctx<BnCurve>*>(ctx)

// ---- host-side helpers shared by both curves --------------------------------------------------
template <class C>
static int create_generators_out(size_t count, const uint8_t* api_id, size_t api_id_len, uint8_t* out_affine) {
    constexpr size_t FPB = 4 * C::FpP::NC;
    std::vector<G1Aff<C>> g;
    if (create_generators_host<C>(count, api_id, api_id_len, g)) return BBS_ST_PANIC_DST_TOO_LONG;
    for (size_t k = 0; k < count; k++) {
        if (g1a_is_inf<C>(g[k])) { std::memset(out_affine + k * 2 * FPB, 0, 2 * FPB); continue; }
        fe_to_le_bytes<typename C::FpP>(g[k].x, out_affine + k * 2 * FPB);
        fe_to_le_bytes<typename C::FpP>(g[k].y, out_affine + k * 2 * FPB + FPB);
    }
    return BBS_OK;
}
template <class C>
static int hash_to_g1_out(const uint8_t* msg, size_t msg_len, const uint8_t* dst, size_t dst_len, uint8_t* out_affine) {
    constexpr size_t FPB = 4 * C::FpP::NC;
    bool ok = true;
    G1Aff<C> p = H2cOf<C>::run(msg, msg_len, dst, dst_len, ok);
    if (!ok) return BBS_ST_PANIC_DST_TOO_LONG;
    if (g1a_is_inf<C>(p)) { std::memset(out_affine, 0, 2 * FPB); return BBS_OK; }
    fe_to_le_bytes<typename C::FpP>(p.x, out_affine);
    fe_to_le_bytes<typename C::FpP>(p.y, out_affine + FPB);
    return BBS_OK;
}
// host arithmetic self-test: sum_k w_k a_k b_k in Fp2 through the lazily reduced column accumulators (tower.hpp F2Acc)
template <class C>
static int selftest_f2dot(size_t n_terms, const uint8_t* a, const uint8_t* b, const uint8_t* w, uint8_t* out) {
    using P = typename C::FpP;
    constexpr size_t FPB = 4 * P::NC;
    F2Acc<C> acc;
    f2acc_zero<C>(acc);
    int weight = 0;
    for (size_t k = 0; k < n_terms; k++) {
        Fp2<C> x, y;
        if (!fe_from_le_bytes<P>(a + k * 2 * FPB, x.c0) || !fe_from_le_bytes<P>(a + k * 2 * FPB + FPB, x.c1) ||
            !fe_from_le_bytes<P>(b + k * 2 * FPB, y.c0) || !fe_from_le_bytes<P>(b + k * 2 * FPB + FPB, y.c1)) return BBS_E_ARG;
        weight += w[k] ? w[k] : 1;
        if (w[k] > 2 || weight > 6) return BBS_E_ARG;
        if (w[k] == 0) f2acc_mac_fp<C>(acc, x, y.c0);
        else if (w[k] == 1) f2acc_mac<C, 1>(acc, x, y);
        else f2acc_mac<C, 2>(acc, x, y);
    }
    const Fp2<C> r = f2acc_finish<C>(acc);
    fe_to_le_bytes<P>(r.c0, out);
    fe_to_le_bytes<P>(r.c1, out + FPB);
    return BBS_OK;
}

// the same dot product by the two-pass accumulators (low columns, quotients, high columns: tower.hpp F2AccLo / F2AccHi)
template <class C>
static int selftest_f2dot2(size_t n_terms, const uint8_t* a, const uint8_t* b, const uint8_t* w, uint8_t* out) {
    using P = typename C::FpP;
    constexpr size_t FPB = 4 * P::NC;
    std::vector<Fp2<C>> X(n_terms), Y(n_terms);
    int weight = 0;
    for (size_t k = 0; k < n_terms; k++) {
        if (!fe_from_le_bytes<P>(a + k * 2 * FPB, X[k].c0) || !fe_from_le_bytes<P>(a + k * 2 * FPB + FPB, X[k].c1) ||
            !fe_from_le_bytes<P>(b + k * 2 * FPB, Y[k].c0) || !fe_from_le_bytes<P>(b + k * 2 * FPB + FPB, Y[k].c1)) return BBS_E_ARG;
        weight += w[k] ? w[k] : 1;
        if (w[k] > 2 || weight > 6) return BBS_E_ARG;
    }
    F2AccLo<C> lo;
    f2acc_lo_zero<C>(lo);
    for (size_t k = 0; k < n_terms; k++) {
        if (w[k] == 0) f2acc2_mac_fp<C, false>(lo, X[k], Y[k].c0); else f2acc2_mac_sh<C, false>(lo, X[k], Y[k], w[k] - 1u);
    }
    F2AccMid<C> mid;
    f2acc_finish_lo<C>(lo, mid);
    F2AccHi<C> hi;
    f2acc_hi_zero<C>(hi);
    for (size_t k = 0; k < n_terms; k++) {
        if (w[k] == 0) f2acc2_mac_fp<C, true>(hi, X[k], Y[k].c0); else f2acc2_mac_sh<C, true>(hi, X[k], Y[k], w[k] - 1u);
    }
    const Fp2<C> r = f2acc_finish_hi<C>(hi, mid);
    fe_to_le_bytes<P>(r.c0, out);
    fe_to_le_bytes<P>(r.c1, out + FPB);
    return BBS_OK;
}

// host arithmetic self-test: x^-1 by the safegcd inversion and by the Fermat power (independent code paths)
template <class P>
static int selftest_inv(const uint8_t* x, uint8_t* out_safegcd, uint8_t* out_fermat) {
    Fe<P> a;
    if (!fe_from_le_bytes<P>(x, a)) return BBS_E_ARG;
    fe_to_le_bytes<P>(fe_inv<P>(a), out_safegcd);
    fe_to_le_bytes<P>(fe_inv_fermat<P>(a), out_fermat);
    return BBS_OK;
}

// host arithmetic self-test: 3 x0 +- 2 x1 in the base field by the run-time-sign chain of the cyclotomic square
template <class P>
static int selftest_lin_pm(int plus, const uint8_t* x0, const uint8_t* x1, uint8_t* out) {
    Fe<P> a, b;
    if (!fe_from_le_bytes<P>(x0, a) || !fe_from_le_bytes<P>(x1, b)) return BBS_E_ARG;
    fe_to_le_bytes<P>(fe_lin_pm<P, 3, 2>(a, b, plus != 0), out);
    return BBS_OK;
}

// host arithmetic self-test: one half of an Fp4 square by the four-column form (tower.hpp fp4_sqr_part)
template <class C>
static int selftest_fp4sqr(int hi, const uint8_t* a, const uint8_t* b, uint8_t* out) {
    using P = typename C::FpP;
    constexpr size_t FPB = 4 * P::NC;
    Fp2<C> A, B;
    if (!fe_from_le_bytes<P>(a, A.c0) || !fe_from_le_bytes<P>(a + FPB, A.c1) || !fe_from_le_bytes<P>(b, B.c0) ||
        !fe_from_le_bytes<P>(b + FPB, B.c1)) return BBS_E_ARG;
    Fp2<C> r;
    if constexpr (C::K::XI_C0 == 1) r = (hi & 2) ? fp4_sqr_part2<C>((hi & 1) != 0, A, B) : fp4_sqr_part<C>((hi & 1) != 0, A, B);
    else r = fp4_sqr_part<C>((hi & 1) != 0, A, B);
    fe_to_le_bytes<P>(r.c0, out);
    fe_to_le_bytes<P>(r.c1, out + FPB);
    return BBS_OK;
}

// host arithmetic self-test: k0 P0 + k1 P1 + k2 P2 by the joint chain of proof_verify's T1 (g1.hpp g1_mul3_aff), plain or GLV
template <class C>
static int selftest_mul3(int glv, const uint8_t* pts, const uint8_t* scal, uint8_t* out) {
    constexpr int N = C::FpP::N;
    constexpr size_t FPB = 4 * C::FpP::NC;
    if (glv && !C::K::HAS_GLV) return BBS_E_ARG;
    G1Aff<C> p[3];
    uint32_t k[3][8];
    for (int j = 0; j < 3; j++) {
        if (!codec::g1_record_to_aff<C>(pts + j * 2 * FPB, p[j]) || !g1a_on_curve<C>(p[j])) return BBS_E_ARG;
        for (int w = 0; w < 8; w++) k[j][w] = le32(scal + 32 * j + 4 * w);
        if (!limbs_lt_mod<typename C::FrP>(k[j])) return BBS_E_ARG;
    }
    std::vector<uint32_t> tabs((size_t)3 * G1_TAB * 2 * N);
    const G1Jac<C> r = g1_mul3_aff<C>(p[0], k[0], p[1], k[1], p[2], k[2], tabs.data(), 1, glv != 0);
    codec::g1_aff_to_record<C>(g1j_to_aff<C>(r), out);
    return BBS_OK;
}

// FromOkm (src/utils/utilities_helper.rs:15-40): 48 big-endian bytes -> scalar mod r, canonical LE out
template <class C>
static void scalar_from_okm(const uint8_t* okm48, uint8_t* out32) {
    uint32_t be[12];
    for (int k = 0; k < 12; k++) be[k] = ((uint32_t)okm48[4 * k] << 24) | ((uint32_t)okm48[4 * k + 1] << 16) | ((uint32_t)okm48[4 * k + 2] << 8) | okm48[4 * k + 3];
    const Fr<C> r = fe_to_canonical<typename C::FrP>(fr_from_okm<C>(be));
    for (int k = 0; k < 8; k++) put_le32(out32 + 4 * k, r.v[k]);
}

// compress a G1 record (x || y, canonical LE; zeros = identity) without any field arithmetic: the flag needs only
// the comparison y > (p - 1) / 2 on the canonical words.  false: a coordinate is not canonical.
template <class C>
static bool g1_record_compress(const uint8_t* rec, uint8_t* out) {
    using P = typename C::FpP;
    constexpr int NC = P::NC, NB = 4 * NC;
    uint32_t xw[NC], yw[NC], any = 0;
    for (int k = 0; k < NC; k++) { xw[k] = le32(rec + 4 * k); yw[k] = le32(rec + NB + 4 * k); any |= xw[k] | yw[k]; }
    if (!limbs_lt_mod<P>(xw) || !limbs_lt_mod<P>(yw)) return false;
    if (C::ID == 0) {
        if (!any) { std::memset(out, 0, NB); out[0] = 0xC0; return true; }
        for (int i = 0; i < NB; i++) out[i] = rec[NB - 1 - i];
        out[0] |= 0x80;
        if (words_gt_half<P>(yw)) out[0] |= 0x20;
    } else {
        if (!any) { std::memset(out, 0, NB); out[NB - 1] = 0x40; return true; }
        std::memcpy(out, rec, NB);
        if (words_gt_half<P>(yw)) out[NB - 1] |= 0x80;
    }
    return true;
}
// n proofs -> octets (same bytes as proof_to_octets per item); status[i] = 1 or BBS_ST_NONCANONICAL
template <class C>
static int proofs_to_octets_batch(size_t n, const uint8_t* pf, const uint8_t* cm, const uint64_t* cm_off, uint8_t* out, uint64_t* out_off,
                                  int8_t* status) {
    constexpr size_t NB = 4 * C::FpP::NC, rec = 6 * NB + 128;
    out_off[0] = 0;
    for (size_t i = 0; i < n; i++) {
        const size_t u = (size_t)(cm_off[i + 1] - cm_off[i]);
        uint8_t* o = out + out_off[i];
        const uint8_t* r = pf + i * rec;
        out_off[i + 1] = out_off[i] + 3 * NB + 32 * (4 + u);
        bool ok = true;
        for (int p = 0; p < 3; p++) ok &= g1_record_compress<C>(r + (size_t)p * 2 * NB, o + (size_t)p * NB);
        o += 3 * NB;
        for (int k = 0; k < 3; k++) codec::scalar_le_to_be(r + 6 * NB + 32 * k, o + 32 * k);
        for (size_t k = 0; k < u; k++) codec::scalar_le_to_be(cm + (cm_off[i] + k) * 32, o + 96 + 32 * k);
        codec::scalar_le_to_be(r + 6 * NB + 96, o + 96 + 32 * u);
        status[i] = ok ? 1 : BBS_ST_NONCANONICAL;
        if (!ok) std::memset(out + out_off[i], 0, (size_t)(out_off[i + 1] - out_off[i]));
    }
    return BBS_OK;
}

#pragma GCC visibility push(default)
extern "C" {

size_t bbs_fp_bytes(int curve) { return curve == BBS_CURVE_BLS12_381 ? 48 : 32; }
const char* bbs_version(void) {
#ifdef BBS_HOST_TWIN
    return "bbs_sign_amd 0.1 (HOST TWIN - TEST ONLY)";
#else
    return "bbs_sign_amd 0.1 (gfx950)";
#endif
}
#ifndef BBS_SRC_HASH
#define BBS_SRC_HASH "unknown"
#endif
static const char BBS_SRC_HASH_MARK[] = "BBS_SRC_HASH=" BBS_SRC_HASH;      // found by bbs_sign_amd/build.py in the file
const char* bbs_source_hash(void) { return BBS_SRC_HASH_MARK + 13; }
int bbs_device_count(void) { return rt::device_count(); }

int bbs_ctx_create(int curve, int device_id, bbs_ctx** out) {
    if (!out || (curve != BBS_CURVE_BLS12_381 && curve != BBS_CURVE_BN254)) return BBS_E_ARG;
    if (device_id < 0 || device_id >= rt::device_count()) return BBS_E_NO_DEVICE;
    int rc;
    if (curve == BBS_CURVE_BLS12_381) {
        auto* c = new Ctx<BlsCurve>();
        c->curve = curve;
        rc = c->init(device_id);
        if (rc) { delete c; return rc; }
        *out = c;
    } else {
        auto* c = new Ctx<BnCurve>();
        c->curve = curve;
        rc = c->init(device_id);
        if (rc) { delete c; return rc; }
        *out = c;
    }
    return BBS_OK;
}
void bbs_ctx_destroy(bbs_ctx* ctx) { delete ctx; }

int bbs_ctx_set_window_bits(bbs_ctx* ctx, int bits) {
    if (!ctx || (bits != 0 && (bits < 4 || bits > 22))) return BBS_E_ARG;      // 0: chosen from the free device memory at set_generators
    // (takes effect at the next bbs_ctx_set_generators; the width of tables already built does not change)
    if (ctx->curve == BBS_CURVE_BLS12_381) AS_BLS(ctx)->win_bits_requested = bits; else AS_BN(ctx)->win_bits_requested = bits;
    return BBS_OK;
}
int bbs_ctx_set_batch_verification(bbs_ctx* ctx, int enabled, const uint8_t* seed32) {
    if (!ctx) return BBS_E_ARG;
    return DISPATCH(ctx, AS_BLS(ctx)->set_batch_verification(enabled, seed32), AS_BN(ctx)->set_batch_verification(enabled, seed32));
}
int bbs_ctx_set_points_in_subgroup(bbs_ctx* ctx, int vouched) {
    if (!ctx) return BBS_E_ARG;
    if (ctx->curve == BBS_CURVE_BLS12_381) AS_BLS(ctx)->points_in_subgroup = vouched != 0; else AS_BN(ctx)->points_in_subgroup = vouched != 0;
    return BBS_OK;
}
int bbs_ctx_set_latency_mode(bbs_ctx* ctx, int enabled) {
    if (!ctx) return BBS_E_ARG;
    if (enabled < 0 || enabled > 2) return BBS_E_ARG;
    if (ctx->curve == BBS_CURVE_BLS12_381) AS_BLS(ctx)->latency_mode = enabled; else AS_BN(ctx)->latency_mode = enabled;
    return BBS_OK;
}
int bbs_ctx_set_fixed_base_tree(bbs_ctx* ctx, int enabled) {
    if (!ctx) return BBS_E_ARG;
    if (ctx->curve == BBS_CURVE_BLS12_381) AS_BLS(ctx)->fix_tree = enabled != 0; else AS_BN(ctx)->fix_tree = enabled != 0;
    return BBS_OK;
}
int bbs_selftest_glv_split(int curve, const uint8_t* k32, uint8_t* k1_16, uint8_t* k2_16, int* neg1, int* neg2) {
    if ((curve != BBS_CURVE_BLS12_381 && curve != BBS_CURVE_BN254) || !k32 || !k1_16 || !k2_16 || !neg1 || !neg2) return BBS_E_ARG;
    uint32_t k[8], k1[4], k2[4];
    bool n1 = false, n2 = false;
    for (int j = 0; j < 8; j++) k[j] = le32(k32 + 4 * j);
    if (curve == BBS_CURVE_BLS12_381) {
        if (!limbs_lt_mod<BlsCurve::FrP>(k)) return BBS_E_ARG;
        glv_split<BlsCurve>(k, k1, k2, n1, n2);
    } else {
        if (!limbs_lt_mod<BnCurve::FrP>(k)) return BBS_E_ARG;
        glv_split<BnCurve>(k, k1, k2, n1, n2);
    }
    for (int j = 0; j < 4; j++) { put_le32(k1_16 + 4 * j, k1[j]); put_le32(k2_16 + 4 * j, k2[j]); }
    *neg1 = n1 ? 1 : 0;
    *neg2 = n2 ? 1 : 0;
    return BBS_OK;
}
int bbs_selftest_mul3(int curve, int glv, const uint8_t* points, const uint8_t* scalars, uint8_t* out_affine) {
    if (!points || !scalars || !out_affine) return BBS_E_ARG;
    if (curve == BBS_CURVE_BLS12_381) return selftest_mul3<BlsCurve>(glv, points, scalars, out_affine);
    if (curve == BBS_CURVE_BN254) return selftest_mul3<BnCurve>(glv, points, scalars, out_affine);
    return BBS_E_ARG;
}
int bbs_ctx_set_generators(bbs_ctx* ctx, const uint8_t* g, size_t count, const uint8_t* api_id, size_t api_id_len) {
    if (!ctx) return BBS_E_ARG;
    return DISPATCH(ctx, AS_BLS(ctx)->set_generators(g, count, api_id, api_id_len), AS_BN(ctx)->set_generators(g, count, api_id, api_id_len));
}
int bbs_ctx_set_public_key(bbs_ctx* ctx, const uint8_t* pk, int is_identity) {
    if (!ctx) return BBS_E_ARG;
    return DISPATCH(ctx, AS_BLS(ctx)->set_public_key(pk, is_identity), AS_BN(ctx)->set_public_key(pk, is_identity));
}
int bbs_ctx_set_secret_key(bbs_ctx* ctx, const uint8_t* sk32) {
    if (!ctx) return BBS_E_ARG;
    return DISPATCH(ctx, AS_BLS(ctx)->set_secret_key(sk32), AS_BN(ctx)->set_secret_key(sk32));
}
int bbs_ctx_get_public_key(bbs_ctx* ctx, uint8_t* out, int* inf) {
    if (!ctx) return BBS_E_ARG;
    return DISPATCH(ctx, AS_BLS(ctx)->get_public_key(out, inf), AS_BN(ctx)->get_public_key(out, inf));
}
int bbs_ctx_get_public_key_compressed(bbs_ctx* ctx, uint8_t* out, size_t cap, size_t* len_out) {
    if (!ctx || !out) return BBS_E_ARG;
    const size_t need = 2 * bbs_fp_bytes(ctx->curve);
    if (cap < need) return BBS_E_ARG;
    if (ctx->curve == BBS_CURVE_BLS12_381) {
        if (!AS_BLS(ctx)->pk_set) return BBS_E_STATE;
        g2_compress<BlsCurve>(AS_BLS(ctx)->pk, out);
    } else {
        if (!AS_BN(ctx)->pk_set) return BBS_E_STATE;
        g2_compress<BnCurve>(AS_BN(ctx)->pk, out);
    }
    if (len_out) *len_out = need;
    return BBS_OK;
}

int bbs_core_proof_verify_upload(bbs_ctx* ctx, size_t n, const uint8_t* pf, const uint8_t* cm, const uint64_t* cmo,
                                 const uint8_t* dm, const uint64_t* dmo, const uint64_t* di, const uint64_t* dio,
                                 const uint8_t* h, const uint64_t* ho, const uint8_t* ph, const uint64_t* pho, bbs_job** job) {
    if (!ctx) return BBS_E_ARG;
    return DISPATCH(ctx, pv_upload<BlsCurve>(AS_BLS(ctx), n, pf, cm, cmo, dm, dmo, di, dio, h, ho, ph, pho, job, nullptr, nullptr, nullptr, nullptr),
                    pv_upload<BnCurve>(AS_BN(ctx), n, pf, cm, cmo, dm, dmo, di, dio, h, ho, ph, pho, job, nullptr, nullptr, nullptr, nullptr));
}
int bbs_core_verify_upload(bbs_ctx* ctx, size_t n, const uint8_t* sigs, const uint8_t* m, const uint64_t* mo,
                           const uint8_t* h, const uint64_t* ho, bbs_job** job) {
    if (!ctx) return BBS_E_ARG;
    return DISPATCH(ctx, vf_upload<BlsCurve>(AS_BLS(ctx), n, sigs, m, mo, h, ho, job, nullptr, nullptr, nullptr), vf_upload<BnCurve>(AS_BN(ctx), n, sigs, m, mo, h, ho, job, nullptr, nullptr, nullptr));
}
int bbs_core_sign_upload(bbs_ctx* ctx, size_t n, const uint8_t* m, const uint64_t* mo, const uint8_t* h, const uint64_t* ho, bbs_job** job) {
    if (!ctx) return BBS_E_ARG;
    return DISPATCH(ctx, sg_upload<BlsCurve>(AS_BLS(ctx), n, m, mo, h, ho, job, nullptr, nullptr), sg_upload<BnCurve>(AS_BN(ctx), n, m, mo, h, ho, job, nullptr, nullptr));
}
int bbs_core_proof_gen_upload(bbs_ctx* ctx, size_t n, const uint8_t* sigs, const uint8_t* m, const uint64_t* mo,
                              const uint64_t* di, const uint64_t* dio, const uint8_t* rnd, const uint64_t* rno,
                              const uint8_t* h, const uint64_t* ho, const uint8_t* ph, const uint64_t* pho, bbs_job** job) {
    if (!ctx) return BBS_E_ARG;
    return DISPATCH(ctx, pg_upload<BlsCurve>(AS_BLS(ctx), n, sigs, m, mo, di, dio, rnd, rno, h, ho, ph, pho, job, nullptr, nullptr, nullptr),
                    pg_upload<BnCurve>(AS_BN(ctx), n, sigs, m, mo, di, dio, rnd, rno, h, ho, ph, pho, job, nullptr, nullptr, nullptr));
}

size_t bbs_device_free_bytes(int device_id) {
    if (device_id < 0 || device_id >= rt::device_count() || rt::set_device(device_id)) return 0;
    return rt::mem_free_bytes();
}
size_t bbs_ctx_table_bytes(const bbs_ctx* ctx) {
    if (!ctx) return 0;
    return ctx->curve == BBS_CURVE_BLS12_381 ? static_cast<const Ctx<BlsCurve>*>(ctx)->table_bytes() : static_cast<const Ctx<BnCurve>*>(ctx)->table_bytes();
}
int bbs_job_run(bbs_job* job) { return job ? job->run() : BBS_E_ARG; }
int bbs_job_wait(bbs_job* job) { return job ? job->wait() : BBS_E_ARG; }
int bbs_job_poll(const bbs_job* job) { return !job ? BBS_E_ARG : (job->completed() ? 1 : 0); }
// Completion-order retire: sleeps (condition variable, woken by the host functions behind the jobs' last operations)
// until one of the jobs that have been run has completed, delivers it exactly as bbs_job_wait does and reports its
// position.  Among several completed jobs the one that completed FIRST is taken.
int bbs_jobs_wait_any(bbs_job* const* jobs, size_t n, size_t* index_out) {
    if (!jobs || !index_out) return BBS_E_ARG;
    CompletionHub& hub = CompletionHub::get();
    size_t best = n;
    {
        std::unique_lock<std::mutex> lk(hub.mu);
        for (;;) {
            bool any_running = false;
            uint64_t best_seq = 0;
            for (size_t i = 0; i < n; i++) {
                const bbs_job* j = jobs[i];
                if (!j || !j->ever_run()) continue;
                if (j->completed()) {
                    const uint64_t q = j->done_seq.load(std::memory_order_relaxed);
                    if (best == n || q < best_seq) { best = i; best_seq = q; }
                } else any_running = true;
            }
            if (best != n) break;
            if (!any_running) return BBS_E_STATE;         // nothing in the set has been run: there is nothing to wait for
            hub.cv.wait(lk);
        }
    }
    *index_out = best;
    return jobs[best]->wait();
}
size_t bbs_job_size(const bbs_job* job) { return job ? job->n : 0; }
int bbs_job_fetch_status(bbs_job* job, int8_t* st) { return (job && st) ? job->fetch_status(st) : BBS_E_ARG; }
size_t bbs_job_device_bytes(const bbs_job* job) { return job ? job->device_bytes() : 0; }
int bbs_job_fetch_signatures(bbs_job* job, uint8_t* out) { return (job && out) ? job->fetch_signatures(out) : BBS_E_ARG; }
int bbs_job_fetch_proofs(bbs_job* job, uint8_t* pf, uint8_t* cm, uint64_t* cmo) { return job ? job->fetch_proofs(pf, cm, cmo) : BBS_E_ARG; }
void bbs_job_free(bbs_job* job) { delete job; }
const char* bbs_job_stage_name(const bbs_job* job, int k) {
    return (job && k >= 0 && (size_t)k < job->stages.size()) ? job->stages[k].name : nullptr;
}
int bbs_job_run_timed(bbs_job* job, int reps, float* total_ms, float* kernel_ms, int cap, int* n_stages) {
    if (!job || reps < 1) return BBS_E_ARG;
    if (job->use()) return BBS_E_HIP;
    const int ns = (int)job->stages.size();
    if (n_stages) *n_stages = ns;
    const size_t per_rep = 1 + 2 * (size_t)ns;
    rt::EventList ev((size_t)reps * per_rep);
    if (rt::sync(job->stream())) return BBS_E_HIP;
    for (int r = 0; r < reps; r++) if (job->run_recorded(&ev)) return BBS_E_HIP;
    if (ev.finish(job->stream()) || job->sync_aux()) return BBS_E_HIP;
    // the last stage of a rep is on the main stream: total = first event .. last stop
    if (total_ms) *total_ms = ev.ms(0, (size_t)reps * per_rep - 1);
    if (kernel_ms) {
        for (int k = 0; k < ns && k < cap; k++) {
            float acc = 0.f;
            for (int r = 0; r < reps; r++) acc += ev.ms((size_t)r * per_rep + 1 + 2 * k, (size_t)r * per_rep + 2 + 2 * k);
            kernel_ms[k] = acc;
        }
    }
    return BBS_OK;
}

// Throughput run over several device-resident jobs (batches in flight): step k runs on job
// k % njobs, every job on its own stream pair, HIP events around every stage.  Outputs: wall time
// from the first to the last event (ms), and per stage index the SUM of the stage durations over
// all steps (stage lists of all jobs must be identical).
int bbs_jobs_run_timed(bbs_job** jobs, int njobs, int steps, float* total_ms, float* kernel_ms, int cap, int* n_stages) {
    if (!jobs || njobs < 1 || steps < 1) return BBS_E_ARG;
    for (int j = 0; j < njobs; j++) if (!jobs[j]) return BBS_E_ARG;
    const int ns = (int)jobs[0]->stages.size();
    for (int j = 1; j < njobs; j++) if ((int)jobs[j]->stages.size() != ns) return BBS_E_ARG;
    if (n_stages) *n_stages = ns;
    if (jobs[0]->use()) return BBS_E_HIP;
    const size_t per_step = 1 + 2 * (size_t)ns;
    rt::EventList ev((size_t)steps * per_step);
    for (int j = 0; j < njobs; j++) if (rt::sync(jobs[j]->stream())) return BBS_E_HIP;
    for (int k = 0; k < steps; k++) if (jobs[k % njobs]->run_recorded(&ev)) return BBS_E_HIP;
    for (int j = 0; j < njobs; j++) {
        if (rt::sync(jobs[j]->stream())) return BBS_E_HIP;
        if (jobs[j]->sync_aux()) return BBS_E_HIP;
    }
    if (total_ms) {
        // the steps end in different orders on different streams: take the latest last-stage stop
        float best = 0.f;
        for (int k = 0; k < steps; k++) {
            const float m = ev.ms(0, (size_t)k * per_step + per_step - 1);
            if (m > best) best = m;
        }
        *total_ms = best;
    }
    if (kernel_ms) {
        for (int s = 0; s < ns && s < cap; s++) {
            float acc = 0.f;
            for (int k = 0; k < steps; k++) acc += ev.ms((size_t)k * per_step + 1 + 2 * s, (size_t)k * per_step + 2 + 2 * s);
            kernel_ms[s] = acc;
        }
    }
    return BBS_OK;
}
int bbs_ctx_set_stage_timing(bbs_ctx* ctx, int enabled) {
    if (!ctx) return BBS_E_ARG;
    ctx->stage_timing = enabled != 0;
    return BBS_OK;
}
int bbs_job_stage_times(bbs_job* job, float* total_ms, float* kernel_ms, int cap, int* n_stages) {
    return job ? job->stage_times(total_ms, kernel_ms, cap, n_stages) : BBS_E_ARG;
}

// upload (one asynchronous H2D copy + the ingest kernel) -> kernels -> asynchronous copy of the statuses to page-locked
// memory; nothing waits for the device.  bbs_job_wait delivers the statuses to `status`.
// The ONE submit epilogue of every *_submit entry point: run the stages, enqueue the copy of the statuses (sign / proof_gen:
// and of the produced records) to page-locked memory behind them, arm the completion notification behind that, and only
// then let the job know where bbs_job_wait has to deliver.  On any failure the job is freed and nothing is handed out.
// (bbs_job_wait refuses with BBS_E_STATE if an item was left undecided: fail closed, tested for every entry point.)
static int submit_with_results(bbs_job* job, int8_t* status, uint8_t* o1, uint8_t* o2, uint64_t* o3, bbs_job** job_out) {
    job->hold_arm = true;
    int rc = job->run();
    job->hold_arm = false;
    if (!rc) rc = job->enqueue_status_fetch();
    if (!rc && (o1 || o2 || o3)) rc = job->enqueue_result_fetch();
    if (!rc) rc = job->arm_completion();
    if (rc) { delete job; return rc; }
    job->deliver_to = status;
    job->results_wanted = o1 || o2 || o3;
    job->set_result_targets(o1, o2, o3);
    *job_out = job;
    return BBS_OK;
}
// the *_batch forms: submit, wait, free
static int wait_and_free(int rc_submit, bbs_job* const& job) {      // by reference: read after the submit call has set it
    if (rc_submit) return rc_submit;
    const int rc = job->wait();
    delete job;
    return rc;
}
int bbs_core_proof_verify_submit(bbs_ctx* ctx, size_t n, const uint8_t* pf, const uint8_t* cm, const uint64_t* cmo,
                                 const uint8_t* dm, const uint64_t* dmo, const uint64_t* di, const uint64_t* dio,
                                 const uint8_t* h, const uint64_t* ho, const uint8_t* ph, const uint64_t* pho, int8_t* status,
                                 bbs_job** job_out) {
    if (!status || !job_out) return BBS_E_ARG;
    bbs_job* job = nullptr;
    int rc = bbs_core_proof_verify_upload(ctx, n, pf, cm, cmo, dm, dmo, di, dio, h, ho, ph, pho, &job);
    if (rc) return rc;
    return submit_with_results(job, status, nullptr, nullptr, nullptr, job_out);
}
int bbs_core_proof_verify_batch(bbs_ctx* ctx, size_t n, const uint8_t* pf, const uint8_t* cm, const uint64_t* cmo,
                                const uint8_t* dm, const uint64_t* dmo, const uint64_t* di, const uint64_t* dio,
                                const uint8_t* h, const uint64_t* ho, const uint8_t* ph, const uint64_t* pho, int8_t* status) {
    bbs_job* job = nullptr;
    return wait_and_free(bbs_core_proof_verify_submit(ctx, n, pf, cm, cmo, dm, dmo, di, dio, h, ho, ph, pho, status, &job), job);
}
// proof_verify from proof OCTET strings: decoding (square roots, subgroup checks) on the device, then the same pipeline
int bbs_proof_verify_octets_submit(bbs_ctx* ctx, size_t n, const uint8_t* oct, const uint64_t* oct_off,
                                   const uint8_t* dm, const uint64_t* dmo, const uint64_t* di, const uint64_t* dio,
                                   const uint8_t* h, const uint64_t* ho, const uint8_t* ph, const uint64_t* pho, int8_t* status,
                                   bbs_job** job_out) {
    if (!ctx || !status || !job_out || (n && !oct_off)) return BBS_E_ARG;
    bbs_job* job = nullptr;
    int rc = DISPATCH(ctx, pv_upload<BlsCurve>(AS_BLS(ctx), n, nullptr, nullptr, nullptr, dm, dmo, di, dio, h, ho, ph, pho, &job, oct, oct_off, nullptr, nullptr),
                      pv_upload<BnCurve>(AS_BN(ctx), n, nullptr, nullptr, nullptr, dm, dmo, di, dio, h, ho, ph, pho, &job, oct, oct_off, nullptr, nullptr));
    if (rc) return rc;
    return submit_with_results(job, status, nullptr, nullptr, nullptr, job_out);
}
// the public proof_verify of the reference for a fixed number of messages, in one call: proof octets AND the disclosed
// messages as raw bytes (msg_to_scalars on the device in front of the ingest stage)
int bbs_proof_verify_wire_submit(bbs_ctx* ctx, size_t n, const uint8_t* oct, const uint64_t* oct_off,
                                 const uint8_t* msg_bytes, const uint64_t* msg_byte_off, const uint64_t* msg_item_off,
                                 const uint64_t* di, const uint64_t* dio,
                                 const uint8_t* h, const uint64_t* ho, const uint8_t* ph, const uint64_t* pho, int8_t* status,
                                 bbs_job** job_out) {
    if (!ctx || !status || !job_out || (n && (!oct_off || !msg_item_off))) return BBS_E_ARG;
    static const uint64_t zero_off[1] = {0};
    if (!msg_byte_off) {                                  // no disclosed message in the whole batch
        if (n && msg_item_off[n] != msg_item_off[0]) return BBS_E_ARG;
        msg_byte_off = zero_off;
    }
    bbs_job* job = nullptr;
    int rc = DISPATCH(ctx, pv_upload<BlsCurve>(AS_BLS(ctx), n, nullptr, nullptr, nullptr, nullptr, msg_item_off, di, dio, h, ho, ph, pho, &job, oct, oct_off, msg_bytes, msg_byte_off),
                      pv_upload<BnCurve>(AS_BN(ctx), n, nullptr, nullptr, nullptr, nullptr, msg_item_off, di, dio, h, ho, ph, pho, &job, oct, oct_off, msg_bytes, msg_byte_off));
    if (rc) return rc;
    return submit_with_results(job, status, nullptr, nullptr, nullptr, job_out);
}
int bbs_proof_verify_wire_batch(bbs_ctx* ctx, size_t n, const uint8_t* oct, const uint64_t* oct_off,
                                const uint8_t* msg_bytes, const uint64_t* msg_byte_off, const uint64_t* msg_item_off,
                                const uint64_t* di, const uint64_t* dio,
                                const uint8_t* h, const uint64_t* ho, const uint8_t* ph, const uint64_t* pho, int8_t* status) {
    bbs_job* job = nullptr;
    return wait_and_free(bbs_proof_verify_wire_submit(ctx, n, oct, oct_off, msg_bytes, msg_byte_off, msg_item_off, di, dio, h, ho, ph, pho, status, &job), job);
}
int bbs_proof_verify_octets_batch(bbs_ctx* ctx, size_t n, const uint8_t* oct, const uint64_t* oct_off,
                                  const uint8_t* dm, const uint64_t* dmo, const uint64_t* di, const uint64_t* dio,
                                  const uint8_t* h, const uint64_t* ho, const uint8_t* ph, const uint64_t* pho, int8_t* status) {
    bbs_job* job = nullptr;
    return wait_and_free(bbs_proof_verify_octets_submit(ctx, n, oct, oct_off, dm, dmo, di, dio, h, ho, ph, pho, status, &job), job);
}
int bbs_core_verify_submit(bbs_ctx* ctx, size_t n, const uint8_t* sigs, const uint8_t* m, const uint64_t* mo,
                           const uint8_t* h, const uint64_t* ho, int8_t* status, bbs_job** job_out) {
    if (!status || !job_out) return BBS_E_ARG;
    bbs_job* job = nullptr;
    int rc = bbs_core_verify_upload(ctx, n, sigs, m, mo, h, ho, &job);
    if (rc) return rc;
    return submit_with_results(job, status, nullptr, nullptr, nullptr, job_out);
}
// verify from the wire: signature octet strings (compress(A) || e) decoded and checked on the device in front of
// core_verify; statuses as bbs_signature_from_octets followed by core_verify would give them
int bbs_verify_octets_submit(bbs_ctx* ctx, size_t n, const uint8_t* sig_octets, const uint8_t* m, const uint64_t* mo,
                             const uint8_t* h, const uint64_t* ho, int8_t* status, bbs_job** job_out) {
    if (!ctx || !status || !job_out || (n && !sig_octets)) return BBS_E_ARG;
    bbs_job* job = nullptr;
    int rc = DISPATCH(ctx, vf_upload<BlsCurve>(AS_BLS(ctx), n, nullptr, m, mo, h, ho, &job, sig_octets, nullptr, nullptr),
                      vf_upload<BnCurve>(AS_BN(ctx), n, nullptr, m, mo, h, ho, &job, sig_octets, nullptr, nullptr));
    if (rc) return rc;
    return submit_with_results(job, status, nullptr, nullptr, nullptr, job_out);
}
int bbs_verify_octets_batch(bbs_ctx* ctx, size_t n, const uint8_t* sig_octets, const uint8_t* m, const uint64_t* mo,
                            const uint8_t* h, const uint64_t* ho, int8_t* status) {
    bbs_job* job = nullptr;
    return wait_and_free(bbs_verify_octets_submit(ctx, n, sig_octets, m, mo, h, ho, status, &job), job);
}
// the reference's PUBLIC verify / sign for a context's number of messages, in one call: raw messages in (msg_to_scalars on
// the device), signatures as octet strings in (verify) / out (sign)
static const uint64_t BBS_ZERO_OFF[1] = {0};
int bbs_verify_wire_submit(bbs_ctx* ctx, size_t n, const uint8_t* sig_octets, const uint8_t* msg_bytes, const uint64_t* msg_byte_off,
                           const uint64_t* msg_item_off, const uint8_t* h, const uint64_t* ho, int8_t* status, bbs_job** job_out) {
    if (!ctx || !status || !job_out || (n && (!sig_octets || !msg_item_off))) return BBS_E_ARG;
    if (!msg_byte_off) {
        if (n && msg_item_off[n] != msg_item_off[0]) return BBS_E_ARG;
        msg_byte_off = BBS_ZERO_OFF;
    }
    bbs_job* job = nullptr;
    int rc = DISPATCH(ctx, vf_upload<BlsCurve>(AS_BLS(ctx), n, nullptr, nullptr, msg_item_off, h, ho, &job, sig_octets, msg_bytes, msg_byte_off),
                      vf_upload<BnCurve>(AS_BN(ctx), n, nullptr, nullptr, msg_item_off, h, ho, &job, sig_octets, msg_bytes, msg_byte_off));
    if (rc) return rc;
    return submit_with_results(job, status, nullptr, nullptr, nullptr, job_out);
}
int bbs_verify_wire_batch(bbs_ctx* ctx, size_t n, const uint8_t* sig_octets, const uint8_t* msg_bytes, const uint64_t* msg_byte_off,
                          const uint64_t* msg_item_off, const uint8_t* h, const uint64_t* ho, int8_t* status) {
    bbs_job* job = nullptr;
    return wait_and_free(bbs_verify_wire_submit(ctx, n, sig_octets, msg_bytes, msg_byte_off, msg_item_off, h, ho, status, &job), job);
}
int bbs_sign_wire_submit(bbs_ctx* ctx, size_t n, const uint8_t* msg_bytes, const uint64_t* msg_byte_off, const uint64_t* msg_item_off,
                         const uint8_t* h, const uint64_t* ho, uint8_t* sig_octets_out, int8_t* status, bbs_job** job_out) {
    if (!ctx || !status || !job_out || (n && (!sig_octets_out || !msg_item_off))) return BBS_E_ARG;
    if (!msg_byte_off) {
        if (n && msg_item_off[n] != msg_item_off[0]) return BBS_E_ARG;
        msg_byte_off = BBS_ZERO_OFF;
    }
    bbs_job* job = nullptr;
    int rc = DISPATCH(ctx, sg_upload<BlsCurve>(AS_BLS(ctx), n, nullptr, msg_item_off, h, ho, &job, msg_bytes, msg_byte_off),
                      sg_upload<BnCurve>(AS_BN(ctx), n, nullptr, msg_item_off, h, ho, &job, msg_bytes, msg_byte_off));
    if (rc) return rc;
    if ((rc = job->set_octet_form())) { delete job; return rc; }
    return submit_with_results(job, status, sig_octets_out, nullptr, nullptr, job_out);
}
int bbs_sign_wire_batch(bbs_ctx* ctx, size_t n, const uint8_t* msg_bytes, const uint64_t* msg_byte_off, const uint64_t* msg_item_off,
                        const uint8_t* h, const uint64_t* ho, uint8_t* sig_octets_out, int8_t* status) {
    bbs_job* job = nullptr;
    return wait_and_free(bbs_sign_wire_submit(ctx, n, msg_bytes, msg_byte_off, msg_item_off, h, ho, sig_octets_out, status, &job), job);
}
int bbs_core_verify_batch(bbs_ctx* ctx, size_t n, const uint8_t* sigs, const uint8_t* m, const uint64_t* mo,
                          const uint8_t* h, const uint64_t* ho, int8_t* status) {
    bbs_job* job = nullptr;
    return wait_and_free(bbs_core_verify_submit(ctx, n, sigs, m, mo, h, ho, status, &job), job);
}
int bbs_core_sign_submit(bbs_ctx* ctx, size_t n, const uint8_t* m, const uint64_t* mo, const uint8_t* h, const uint64_t* ho,
                         uint8_t* sigs_out, int8_t* status, bbs_job** job_out) {
    if (!status || !job_out) return BBS_E_ARG;
    bbs_job* job = nullptr;
    int rc = bbs_core_sign_upload(ctx, n, m, mo, h, ho, &job);
    if (rc) return rc;
    return submit_with_results(job, status, sigs_out, nullptr, nullptr, job_out);
}
int bbs_core_proof_gen_submit(bbs_ctx* ctx, size_t n, const uint8_t* sigs, const uint8_t* m, const uint64_t* mo,
                              const uint64_t* di, const uint64_t* dio, const uint8_t* rnd, const uint64_t* rno,
                              const uint8_t* h, const uint64_t* ho, const uint8_t* ph, const uint64_t* pho,
                              uint8_t* pf_out, uint8_t* cm_out, uint64_t* cmo_out, int8_t* status, bbs_job** job_out) {
    if (!status || !job_out) return BBS_E_ARG;
    bbs_job* job = nullptr;
    int rc = bbs_core_proof_gen_upload(ctx, n, sigs, m, mo, di, dio, rnd, rno, h, ho, ph, pho, &job);
    if (rc) return rc;
    return submit_with_results(job, status, pf_out, cm_out, cmo_out, job_out);
}
// sign / proof_gen to the WIRE: the results leave the device as octet strings (compression on the device)
int bbs_sign_octets_submit(bbs_ctx* ctx, size_t n, const uint8_t* m, const uint64_t* mo, const uint8_t* h, const uint64_t* ho,
                           uint8_t* sig_octets_out, int8_t* status, bbs_job** job_out) {
    if (!status || !job_out || (n && !sig_octets_out)) return BBS_E_ARG;
    bbs_job* job = nullptr;
    int rc = bbs_core_sign_upload(ctx, n, m, mo, h, ho, &job);
    if (rc) return rc;
    if ((rc = job->set_octet_form())) { delete job; return rc; }
    return submit_with_results(job, status, sig_octets_out, nullptr, nullptr, job_out);
}
int bbs_sign_octets_batch(bbs_ctx* ctx, size_t n, const uint8_t* m, const uint64_t* mo, const uint8_t* h, const uint64_t* ho,
                          uint8_t* sig_octets_out, int8_t* status) {
    bbs_job* job = nullptr;
    return wait_and_free(bbs_sign_octets_submit(ctx, n, m, mo, h, ho, sig_octets_out, status, &job), job);
}
int bbs_proof_gen_octets_submit(bbs_ctx* ctx, size_t n, const uint8_t* sigs, const uint8_t* m, const uint64_t* mo,
                                const uint64_t* di, const uint64_t* dio, const uint8_t* rnd, const uint64_t* rno,
                                const uint8_t* h, const uint64_t* ho, const uint8_t* ph, const uint64_t* pho,
                                uint8_t* octets_out, uint64_t* oct_off_out, int8_t* status, bbs_job** job_out) {
    if (!status || !job_out || !oct_off_out || (n && !octets_out)) return BBS_E_ARG;
    bbs_job* job = nullptr;
    int rc = bbs_core_proof_gen_upload(ctx, n, sigs, m, mo, di, dio, rnd, rno, h, ho, ph, pho, &job);
    if (rc) return rc;
    if ((rc = job->set_octet_form())) { delete job; return rc; }
    return submit_with_results(job, status, octets_out, nullptr, oct_off_out, job_out);
}
int bbs_proof_gen_octets_batch(bbs_ctx* ctx, size_t n, const uint8_t* sigs, const uint8_t* m, const uint64_t* mo,
                               const uint64_t* di, const uint64_t* dio, const uint8_t* rnd, const uint64_t* rno,
                               const uint8_t* h, const uint64_t* ho, const uint8_t* ph, const uint64_t* pho,
                               uint8_t* octets_out, uint64_t* oct_off_out, int8_t* status) {
    bbs_job* job = nullptr;
    return wait_and_free(bbs_proof_gen_octets_submit(ctx, n, sigs, m, mo, di, dio, rnd, rno, h, ho, ph, pho, octets_out, oct_off_out, status, &job), job);
}
// the reference's PUBLIC proof_gen (src/proof_gen.rs:78-113) in one call: signature octets and raw messages in, proof octets out
int bbs_proof_gen_wire_submit(bbs_ctx* ctx, size_t n, const uint8_t* sig_octets, const uint8_t* msg_bytes, const uint64_t* msg_byte_off,
                              const uint64_t* msg_item_off, const uint64_t* di, const uint64_t* dio, const uint8_t* rnd, const uint64_t* rno,
                              const uint8_t* h, const uint64_t* ho, const uint8_t* ph, const uint64_t* pho,
                              uint8_t* octets_out, uint64_t* oct_off_out, int8_t* status, bbs_job** job_out) {
    if (!ctx || !status || !job_out || !oct_off_out || (n && (!octets_out || !sig_octets || !msg_item_off))) return BBS_E_ARG;
    if (!msg_byte_off) {
        if (n && msg_item_off[n] != msg_item_off[0]) return BBS_E_ARG;
        msg_byte_off = BBS_ZERO_OFF;
    }
    bbs_job* job = nullptr;
    int rc = DISPATCH(ctx, pg_upload<BlsCurve>(AS_BLS(ctx), n, nullptr, nullptr, msg_item_off, di, dio, rnd, rno, h, ho, ph, pho, &job, sig_octets, msg_bytes, msg_byte_off),
                      pg_upload<BnCurve>(AS_BN(ctx), n, nullptr, nullptr, msg_item_off, di, dio, rnd, rno, h, ho, ph, pho, &job, sig_octets, msg_bytes, msg_byte_off));
    if (rc) return rc;
    if ((rc = job->set_octet_form())) { delete job; return rc; }
    return submit_with_results(job, status, octets_out, nullptr, oct_off_out, job_out);
}
int bbs_proof_gen_wire_batch(bbs_ctx* ctx, size_t n, const uint8_t* sig_octets, const uint8_t* msg_bytes, const uint64_t* msg_byte_off,
                             const uint64_t* msg_item_off, const uint64_t* di, const uint64_t* dio, const uint8_t* rnd, const uint64_t* rno,
                             const uint8_t* h, const uint64_t* ho, const uint8_t* ph, const uint64_t* pho,
                             uint8_t* octets_out, uint64_t* oct_off_out, int8_t* status) {
    bbs_job* job = nullptr;
    return wait_and_free(bbs_proof_gen_wire_submit(ctx, n, sig_octets, msg_bytes, msg_byte_off, msg_item_off, di, dio, rnd, rno, h, ho, ph, pho,
                                       octets_out, oct_off_out, status, &job), job);
}
int bbs_core_sign_batch(bbs_ctx* ctx, size_t n, const uint8_t* m, const uint64_t* mo, const uint8_t* h, const uint64_t* ho,
                        uint8_t* sigs_out, int8_t* status) {
    bbs_job* job = nullptr;
    int rc;
    if (status) {
        if ((rc = bbs_core_sign_submit(ctx, n, m, mo, h, ho, sigs_out, status, &job))) return rc;
        rc = job->wait();
    } else {                                       // records without statuses: failed items read as zero records
        if ((rc = bbs_core_sign_upload(ctx, n, m, mo, h, ho, &job))) return rc;
        rc = job->run();
        if (!rc && sigs_out) rc = job->fetch_signatures(sigs_out);
    }
    delete job;
    return rc;
}
int bbs_core_proof_gen_batch(bbs_ctx* ctx, size_t n, const uint8_t* sigs, const uint8_t* m, const uint64_t* mo,
                             const uint64_t* di, const uint64_t* dio, const uint8_t* rnd, const uint64_t* rno,
                             const uint8_t* h, const uint64_t* ho, const uint8_t* ph, const uint64_t* pho,
                             uint8_t* pf_out, uint8_t* cm_out, uint64_t* cmo_out, int8_t* status) {
    bbs_job* job = nullptr;
    int rc;
    if (status) {
        if ((rc = bbs_core_proof_gen_submit(ctx, n, sigs, m, mo, di, dio, rnd, rno, h, ho, ph, pho, pf_out, cm_out, cmo_out, status, &job))) return rc;
        rc = job->wait();
    } else {
        if ((rc = bbs_core_proof_gen_upload(ctx, n, sigs, m, mo, di, dio, rnd, rno, h, ho, ph, pho, &job))) return rc;
        rc = job->run();
        if (!rc) rc = job->fetch_proofs(pf_out, cm_out, cmo_out);
    }
    delete job;
    return rc;
}

// ---- unit-parity primitives -------------------------------------------------------------------
int bbs_hash_to_scalar_batch(bbs_ctx* ctx, size_t n, const uint8_t* msgs, const uint64_t* off, const uint8_t* dst, size_t dst_len, uint8_t* out) {
    if (!ctx) return BBS_E_ARG;
    return DISPATCH(ctx, h2s_batch<BlsCurve>(AS_BLS(ctx), n, msgs, off, dst, dst_len, out), h2s_batch<BnCurve>(AS_BN(ctx), n, msgs, off, dst, dst_len, out));
}

int bbs_g1_msm_batch(bbs_ctx* ctx, size_t n, const uint8_t* fs, size_t nf, const uint8_t* vp, const uint8_t* vs, size_t nv, uint8_t* out, int8_t* status) {
    if (!ctx) return BBS_E_ARG;
    return DISPATCH(ctx, msm_batch<BlsCurve>(AS_BLS(ctx), n, fs, nf, vp, vs, nv, out, status), msm_batch<BnCurve>(AS_BN(ctx), n, fs, nf, vp, vs, nv, out, status));
}

int bbs_g1_msm_pippenger(bbs_ctx* ctx, size_t n, const uint8_t* pts, const uint8_t* scal, uint8_t* out, int* out_inf, int8_t* status) {
    if (!ctx) return BBS_E_ARG;
    return DISPATCH(ctx, msm_pippenger<BlsCurve>(AS_BLS(ctx), n, pts, scal, out, out_inf, status), msm_pippenger<BnCurve>(AS_BN(ctx), n, pts, scal, out, out_inf, status));
}

// ---- host-side setup helpers (once per ciphersuite / key; no GPU involved) ---------------------
int bbs_create_generators(int curve, size_t count, const uint8_t* api_id, size_t api_id_len, uint8_t* out_affine) {
    if ((api_id_len && !api_id) || (count && !out_affine)) return BBS_E_ARG;
    if (curve == BBS_CURVE_BLS12_381) return create_generators_out<BlsCurve>(count, api_id, api_id_len, out_affine);
    if (curve == BBS_CURVE_BN254) return create_generators_out<BnCurve>(count, api_id, api_id_len, out_affine);
    return BBS_E_ARG;
}

int bbs_hash_to_g1(int curve, const uint8_t* msg, size_t msg_len, const uint8_t* dst, size_t dst_len, uint8_t* out_affine) {
    if ((msg_len && !msg) || (dst_len && !dst) || !out_affine) return BBS_E_ARG;
    if (curve == BBS_CURVE_BLS12_381) return hash_to_g1_out<BlsCurve>(msg, msg_len, dst, dst_len, out_affine);
    if (curve == BBS_CURVE_BN254) return hash_to_g1_out<BnCurve>(msg, msg_len, dst, dst_len, out_affine);
    return BBS_E_ARG;
}

int bbs_scalar_from_okm(int curve, const uint8_t* okm48, uint8_t* scalar_out) {
    if (!okm48 || !scalar_out) return BBS_E_ARG;
    if (curve == BBS_CURVE_BLS12_381) scalar_from_okm<BlsCurve>(okm48, scalar_out);
    else if (curve == BBS_CURVE_BN254) scalar_from_okm<BnCurve>(okm48, scalar_out);
    else return BBS_E_ARG;
    return BBS_OK;
}

// SecretKey::key_gen (src/key_gen.rs:46-81)
int bbs_key_gen(int curve, const uint8_t* key_material, size_t km_len, const uint8_t* key_info, size_t ki_len,
                const uint8_t* key_dst, size_t dst_len, uint8_t* sk32_out) {
    if (!sk32_out || (km_len && !key_material) || (ki_len && !key_info) || (dst_len && !key_dst)) return BBS_E_ARG;
    if (curve != BBS_CURVE_BLS12_381 && curve != BBS_CURVE_BN254) return BBS_E_ARG;
    if (km_len < 32) return BBS_ST_INVALID_KEY_MATERIAL_LENGTH;
    if (ki_len > 65535) return BBS_ST_INVALID_KEY_INFO_LENGTH;
    std::vector<uint8_t> in(key_material, key_material + km_len);
    in.push_back((uint8_t)(ki_len >> 8));
    in.push_back((uint8_t)(ki_len & 0xff));
    in.insert(in.end(), key_info, key_info + ki_len);
    uint32_t w[8];
    const bool ok = curve == BBS_CURVE_BLS12_381 ? hash_to_scalar_host<BlsCurve>(in.data(), in.size(), key_dst, dst_len, w)
                                                 : hash_to_scalar_host<BnCurve>(in.data(), in.size(), key_dst, dst_len, w);
    if (!ok) return BBS_ST_PANIC_DST_TOO_LONG;
    uint32_t any = 0;
    for (int i = 0; i < 8; i++) any |= w[i];
    if (!any) return BBS_ST_INVALID_SECRET_KEY;
    for (int i = 0; i < 8; i++) put_le32(sk32_out + 4 * i, w[i]);
    return BBS_OK;
}

// ---- wire codec (host) ---------------------------------------------------------------------------
}  // extern "C"
#pragma GCC visibility pop
template <class C>
static int sig_to_octets(const uint8_t* rec, uint8_t* out) {
    constexpr int NB = 4 * C::FpP::NC;
    G1Aff<C> a;
    if (!codec::g1_record_to_aff<C>(rec, a)) return BBS_ST_NONCANONICAL;
    g1_compress_host<C>(a, out);
    codec::scalar_le_to_be(rec + 2 * NB, out + NB);
    return BBS_OK;
}
template <class C>
static int sig_from_octets(const uint8_t* oct, uint8_t* rec) {
    constexpr int NB = 4 * C::FpP::NC;
    G1Aff<C> a;
    bool inf;
    const int rc = codec::g1_decompress<C>(oct, a, inf);
    if (rc == -1) return BBS_ST_NONCANONICAL;
    if (rc) return BBS_ST_NOT_ON_CURVE;
    if (inf) return BBS_ST_INVALID_ENCODING;                       // octets_to_signature: A must not be the identity
    if (!codec::scalar_be_to_le<C>(oct + NB, rec + 2 * NB)) return BBS_ST_NONCANONICAL;
    bool zero = true;
    for (int i = 0; i < 32; i++) zero &= rec[2 * NB + i] == 0;
    if (zero) return BBS_ST_INVALID_ENCODING;                      // e = 0 is rejected
    codec::g1_aff_to_record<C>(a, rec);
    return BBS_OK;
}
template <class C>
static int proof_to_octets(const uint8_t* pf, const uint8_t* cm, size_t n_cm, uint8_t* out) {
    constexpr int NB = 4 * C::FpP::NC;
    for (int p = 0; p < 3; p++) {
        G1Aff<C> a;
        if (!codec::g1_record_to_aff<C>(pf + (size_t)p * 2 * NB, a)) return BBS_ST_NONCANONICAL;
        g1_compress_host<C>(a, out + (size_t)p * NB);
    }
    uint8_t* o = out + 3 * NB;
    for (int k = 0; k < 3; k++) codec::scalar_le_to_be(pf + 6 * NB + 32 * k, o + 32 * k);          // e^, r1^, r3^
    for (size_t k = 0; k < n_cm; k++) codec::scalar_le_to_be(cm + 32 * k, o + 96 + 32 * k);       // m^_j
    codec::scalar_le_to_be(pf + 6 * NB + 96, o + 96 + 32 * n_cm);                                  // challenge
    return BBS_OK;
}
template <class C>
static int proof_from_octets(const uint8_t* oct, size_t len, uint8_t* pf, uint8_t* cm, size_t cm_cap, size_t* n_cm) {
    constexpr int NB = 4 * C::FpP::NC;
    const size_t fixed = 3 * NB + 4 * 32;
    if (len < fixed || (len - fixed) % 32) return BBS_ST_INVALID_ENCODING;
    const size_t u = (len - fixed) / 32;
    if (u > cm_cap) return BBS_E_ARG;
    for (int p = 0; p < 3; p++) {
        G1Aff<C> a;
        bool inf;
        const int rc = codec::g1_decompress<C>(oct + (size_t)p * NB, a, inf);
        if (rc == -1) return BBS_ST_NONCANONICAL;
        if (rc) return BBS_ST_NOT_ON_CURVE;
        if (inf) return BBS_ST_INVALID_ENCODING;                   // octets_to_proof rejects identity points
        codec::g1_aff_to_record<C>(a, pf + (size_t)p * 2 * NB);
    }
    const uint8_t* o = oct + 3 * NB;
    for (int k = 0; k < 3; k++) if (!codec::scalar_be_to_le<C>(o + 32 * k, pf + 6 * NB + 32 * k)) return BBS_ST_NONCANONICAL;
    for (size_t k = 0; k < u; k++) if (!codec::scalar_be_to_le<C>(o + 96 + 32 * k, cm + 32 * k)) return BBS_ST_NONCANONICAL;
    if (!codec::scalar_be_to_le<C>(o + 96 + 32 * u, pf + 6 * NB + 96)) return BBS_ST_NONCANONICAL;
    if (n_cm) *n_cm = u;
    return BBS_OK;
}
template <class C>
static int pk_from_octets(const uint8_t* oct, uint8_t* rec, int* is_identity) {
    using P = typename C::FpP;
    constexpr int NB = 4 * P::NC;
    G2Aff<C> q;
    const int rc = codec::g2_decompress<C>(oct, q);
    if (rc == -1) return BBS_ST_NONCANONICAL;
    if (rc) return BBS_ST_NOT_ON_CURVE;
    if (is_identity) *is_identity = q.inf ? 1 : 0;
    if (q.inf) { std::memset(rec, 0, 4 * NB); return BBS_OK; }
    fe_to_le_bytes<P>(q.x.c0, rec); fe_to_le_bytes<P>(q.x.c1, rec + NB);
    fe_to_le_bytes<P>(q.y.c0, rec + 2 * NB); fe_to_le_bytes<P>(q.y.c1, rec + 3 * NB);
    return BBS_OK;
}
template <class C>
static int pk_to_octets(const uint8_t* rec, int is_identity, uint8_t* out) {
    using P = typename C::FpP;
    constexpr int NB = 4 * P::NC;
    G2Aff<C> q{};
    q.inf = is_identity != 0;
    if (!q.inf && (!fe_from_le_bytes<P>(rec, q.x.c0) || !fe_from_le_bytes<P>(rec + NB, q.x.c1) ||
                   !fe_from_le_bytes<P>(rec + 2 * NB, q.y.c0) || !fe_from_le_bytes<P>(rec + 3 * NB, q.y.c1))) return BBS_ST_NONCANONICAL;
    g2_compress<C>(q, out);
    return BBS_OK;
}
// what bbs_ctx_set_public_key checks, without a context (bbs_issuer validates a key when it is set, not when it is first used)
template <class C>
static int pk_validate(const uint8_t* rec, int is_identity) {
    using P = typename C::FpP;
    constexpr int NB = 4 * P::NC;
    if (is_identity) return BBS_OK;
    G2Aff<C> q{};
    if (!fe_from_le_bytes<P>(rec, q.x.c0) || !fe_from_le_bytes<P>(rec + NB, q.x.c1) ||
        !fe_from_le_bytes<P>(rec + 2 * NB, q.y.c0) || !fe_from_le_bytes<P>(rec + 3 * NB, q.y.c1)) return BBS_E_PUBLIC_KEY;
    return (g2_on_curve<C>(q) && g2_in_subgroup<C>(q)) ? BBS_OK : BBS_E_PUBLIC_KEY;
}
template <class C>
static bool sk_in_range(const uint8_t* sk32) {
    uint32_t l[8];
    for (int k = 0; k < 8; k++) l[k] = le32(sk32 + 4 * k);
    return limbs_lt_mod<typename C::FrP>(l);
}
#pragma GCC visibility push(default)
extern "C" {
#define CURVE_OK(c) ((c) == BBS_CURVE_BLS12_381 || (c) == BBS_CURVE_BN254)
int bbs_signature_to_octets(int curve, const uint8_t* sig_record, uint8_t* out) {
    if (!CURVE_OK(curve) || !sig_record || !out) return BBS_E_ARG;
    return curve == BBS_CURVE_BLS12_381 ? sig_to_octets<BlsCurve>(sig_record, out) : sig_to_octets<BnCurve>(sig_record, out);
}
int bbs_signature_from_octets(int curve, const uint8_t* octets, uint8_t* sig_record_out) {
    if (!CURVE_OK(curve) || !octets || !sig_record_out) return BBS_E_ARG;
    return curve == BBS_CURVE_BLS12_381 ? sig_from_octets<BlsCurve>(octets, sig_record_out) : sig_from_octets<BnCurve>(octets, sig_record_out);
}
int bbs_proof_to_octets(int curve, const uint8_t* pf, const uint8_t* cm, size_t n_cm, uint8_t* out) {
    if (!CURVE_OK(curve) || !pf || !out || (n_cm && !cm)) return BBS_E_ARG;
    return curve == BBS_CURVE_BLS12_381 ? proof_to_octets<BlsCurve>(pf, cm, n_cm, out) : proof_to_octets<BnCurve>(pf, cm, n_cm, out);
}
int bbs_proof_from_octets(int curve, const uint8_t* octets, size_t len, uint8_t* pf_out, uint8_t* cm_out, size_t cm_cap, size_t* n_cm_out) {
    if (!CURVE_OK(curve) || !octets || !pf_out || (cm_cap && !cm_out)) return BBS_E_ARG;
    return curve == BBS_CURVE_BLS12_381 ? proof_from_octets<BlsCurve>(octets, len, pf_out, cm_out, cm_cap, n_cm_out)
                                        : proof_from_octets<BnCurve>(octets, len, pf_out, cm_out, cm_cap, n_cm_out);
}
int bbs_g1_decompress_batch(bbs_ctx* ctx, size_t n, const uint8_t* compressed, uint8_t* out_affine, int8_t* code) {
    if (!ctx) return BBS_E_ARG;
    return DISPATCH(ctx, g1_decompress_batch<BlsCurve>(AS_BLS(ctx), n, compressed, out_affine, code),
                    g1_decompress_batch<BnCurve>(AS_BN(ctx), n, compressed, out_affine, code));
}
int bbs_signatures_from_octets_batch(bbs_ctx* ctx, size_t n, const uint8_t* octets, uint8_t* sig_records_out, int8_t* status) {
    if (!ctx) return BBS_E_ARG;
    return DISPATCH(ctx, signatures_from_octets_batch<BlsCurve>(AS_BLS(ctx), n, octets, sig_records_out, status),
                    signatures_from_octets_batch<BnCurve>(AS_BN(ctx), n, octets, sig_records_out, status));
}
int bbs_proofs_to_octets_batch(int curve, size_t n, const uint8_t* pf, const uint8_t* cm, const uint64_t* cm_off, uint8_t* out, uint64_t* out_off,
                               int8_t* status) {
    if (!CURVE_OK(curve) || !cm_off || !out_off || !status || (n && (!pf || !out))) return BBS_E_ARG;
    return curve == BBS_CURVE_BLS12_381 ? proofs_to_octets_batch<BlsCurve>(n, pf, cm, cm_off, out, out_off, status)
                                        : proofs_to_octets_batch<BnCurve>(n, pf, cm, cm_off, out, out_off, status);
}
int bbs_proofs_from_octets_batch(bbs_ctx* ctx, size_t n, const uint8_t* octets, const uint64_t* oct_off, uint8_t* pf_out, uint8_t* cm_out,
                                 uint64_t* cm_off_out, int8_t* status) {
    if (!ctx) return BBS_E_ARG;
    return DISPATCH(ctx, proofs_from_octets_batch<BlsCurve>(AS_BLS(ctx), n, octets, oct_off, pf_out, cm_out, cm_off_out, status),
                    proofs_from_octets_batch<BnCurve>(AS_BN(ctx), n, octets, oct_off, pf_out, cm_out, cm_off_out, status));
}
int bbs_public_key_to_octets(int curve, const uint8_t* pk_affine, int is_identity, uint8_t* out) {
    if (!CURVE_OK(curve) || !out || (!is_identity && !pk_affine)) return BBS_E_ARG;
    return curve == BBS_CURVE_BLS12_381 ? pk_to_octets<BlsCurve>(pk_affine, is_identity, out) : pk_to_octets<BnCurve>(pk_affine, is_identity, out);
}
int bbs_public_key_from_octets(int curve, const uint8_t* octets, uint8_t* pk_affine_out, int* is_identity_out) {
    if (!CURVE_OK(curve) || !octets || !pk_affine_out) return BBS_E_ARG;
    return curve == BBS_CURVE_BLS12_381 ? pk_from_octets<BlsCurve>(octets, pk_affine_out, is_identity_out) : pk_from_octets<BnCurve>(octets, pk_affine_out, is_identity_out);
}

int bbs_selftest_f12(bbs_ctx* ctx, int op, const uint8_t* a, const uint8_t* b, uint8_t* out_single, uint8_t* out_dist) {
    if (!ctx || !a || !b || !out_single || !out_dist) return BBS_E_ARG;
    return DISPATCH(ctx, selftest_f12<BlsCurve>(AS_BLS(ctx), op, a, b, out_single, out_dist), selftest_f12<BnCurve>(AS_BN(ctx), op, a, b, out_single, out_dist));
}

int bbs_selftest_inv(int curve, int scalar_field, const uint8_t* x, uint8_t* out_safegcd, uint8_t* out_fermat) {
    if (!x || !out_safegcd || !out_fermat) return BBS_E_ARG;
    if (curve == BBS_CURVE_BLS12_381) return scalar_field ? selftest_inv<BlsFrParams>(x, out_safegcd, out_fermat) : selftest_inv<BlsFpParams>(x, out_safegcd, out_fermat);
    if (curve == BBS_CURVE_BN254) return scalar_field ? selftest_inv<BnFrParams>(x, out_safegcd, out_fermat) : selftest_inv<BnFpParams>(x, out_safegcd, out_fermat);
    return BBS_E_ARG;
}

int bbs_selftest_lin_pm(int curve, int plus, const uint8_t* x0, const uint8_t* x1, uint8_t* out) {
    if (!x0 || !x1 || !out) return BBS_E_ARG;
    if (curve == BBS_CURVE_BLS12_381) return selftest_lin_pm<BlsFpParams>(plus, x0, x1, out);
    if (curve == BBS_CURVE_BN254) return selftest_lin_pm<BnFpParams>(plus, x0, x1, out);
    return BBS_E_ARG;
}

int bbs_selftest_fp4sqr(int curve, int hi, const uint8_t* a, const uint8_t* b, uint8_t* out) {
    if (!a || !b || !out) return BBS_E_ARG;
    if (curve == BBS_CURVE_BLS12_381) return selftest_fp4sqr<BlsCurve>(hi, a, b, out);
    if (curve == BBS_CURVE_BN254) return selftest_fp4sqr<BnCurve>(hi, a, b, out);
    return BBS_E_ARG;
}

int bbs_selftest_f2dot(int curve, size_t n_terms, const uint8_t* a, const uint8_t* b, const uint8_t* weights, uint8_t* out) {
    if (!a || !b || !weights || !out) return BBS_E_ARG;
    if (curve == BBS_CURVE_BLS12_381) return selftest_f2dot<BlsCurve>(n_terms, a, b, weights, out);
    if (curve == BBS_CURVE_BN254) return selftest_f2dot<BnCurve>(n_terms, a, b, weights, out);
    return BBS_E_ARG;
}
int bbs_selftest_f2dot2(int curve, size_t n_terms, const uint8_t* a, const uint8_t* b, const uint8_t* weights, uint8_t* out) {
    if (!a || !b || !weights || !out) return BBS_E_ARG;
    if (curve == BBS_CURVE_BLS12_381) return selftest_f2dot2<BlsCurve>(n_terms, a, b, weights, out);
    if (curve == BBS_CURVE_BN254) return selftest_f2dot2<BnCurve>(n_terms, a, b, weights, out);
    return BBS_E_ARG;
}

// ---- bbs_issuer: items of any message count in one call (issuer.hpp) --------------------------------------------------
int bbs_issuer_create(int curve, int device_id, const uint8_t* api_id, size_t api_id_len, bbs_issuer** out) {
    if (!out || (curve != BBS_CURVE_BLS12_381 && curve != BBS_CURVE_BN254) || (api_id_len && !api_id)) return BBS_E_ARG;
    if (device_id < 0 || device_id >= bbs_device_count()) return BBS_E_NO_DEVICE;
    auto* is = new bbs_issuer();
    is->curve = curve; is->device = device_id;
    is->api_id.assign(api_id, api_id + api_id_len);
    *out = is;
    return BBS_OK;
}
void bbs_issuer_destroy(bbs_issuer* is) { delete is; }
int bbs_issuer_set_public_key(bbs_issuer* is, const uint8_t* pk_affine, int is_identity) {
    if (!is || (!is_identity && !pk_affine)) return BBS_E_ARG;
    if (int rc = is->curve == BBS_CURVE_BLS12_381 ? pk_validate<BlsCurve>(pk_affine, is_identity) : pk_validate<BnCurve>(pk_affine, is_identity)) return rc;
    {
        std::lock_guard<std::mutex> g(is->mu);
        if (is->configuration_locked()) return BBS_E_STATE;   // a routed list is in flight: its jobs read the contexts' keys
        const size_t fpb = bbs_fp_bytes(is->curve);
        is->pk.assign(4 * fpb, 0);
        if (!is_identity) std::memcpy(is->pk.data(), pk_affine, 4 * fpb);
        is->pk_inf = is_identity ? 1 : 0;
        is->pk_set = true; is->sk_set = false;
        volatile uint8_t* s = is->sk;
        for (int k = 0; k < 32; k++) s[k] = 0;
        is->epoch++;
    }
    return issuer_refresh_handed_out(is);                     // contexts the caller drives directly get the key now
}
int bbs_issuer_set_secret_key(bbs_issuer* is, const uint8_t* sk32) {
    if (!is || !sk32) return BBS_E_ARG;
    if (!(is->curve == BBS_CURVE_BLS12_381 ? sk_in_range<BlsCurve>(sk32) : sk_in_range<BnCurve>(sk32))) return BBS_E_ARG;
    {
        std::lock_guard<std::mutex> g(is->mu);
        if (is->configuration_locked()) return BBS_E_STATE;
        std::memcpy(is->sk, sk32, 32);
        is->sk_set = true; is->pk_set = true;
        is->epoch++;
    }
    return issuer_refresh_handed_out(is);
}
int bbs_issuer_set_limits(bbs_issuer* is, size_t max_messages, int window_bits) {
    if (!is || (window_bits != 0 && (window_bits < 4 || window_bits > 22))) return BBS_E_ARG;
    std::lock_guard<std::mutex> g(is->mu);
    is->max_messages = max_messages; is->window_bits = window_bits;
    return BBS_OK;
}
int bbs_issuer_set_budget(bbs_issuer* is, size_t max_contexts, size_t max_table_bytes) {
    if (!is || max_contexts < 1) return BBS_E_ARG;
    std::vector<std::shared_ptr<bbs_issuer_entry>> victims;
    {
        std::lock_guard<std::mutex> g(is->mu);
        is->max_contexts = max_contexts; is->max_table_bytes = max_table_bytes;
        is->evict(0, victims, nullptr);
    }
    return BBS_OK;                                             // (the victims' tables are released here, outside the lock)
}
int bbs_issuer_set_modes(bbs_issuer* is, int latency_mode, int batch_verification, int points_in_subgroup) {
    if (!is || latency_mode < 0 || latency_mode > 2) return BBS_E_ARG;
    {
        std::lock_guard<std::mutex> g(is->mu);
        if (is->configuration_locked()) return BBS_E_STATE;
        is->latency_mode = latency_mode; is->batch_verify = batch_verification != 0; is->in_subgroup = points_in_subgroup != 0;
        is->epoch++;
    }
    return issuer_refresh_handed_out(is);
}
int bbs_issuer_context(bbs_issuer* is, size_t message_count, bbs_ctx** out) {
    if (!is || !out) return BBS_E_ARG;
    return is->context(message_count, out);
}
size_t bbs_issuer_context_count(bbs_issuer* is) {
    if (!is) return 0;
    std::lock_guard<std::mutex> g(is->mu);
    return is->by_count.size();
}
size_t bbs_issuer_table_bytes(bbs_issuer* is) {
    if (!is) return 0;
    std::lock_guard<std::mutex> g(is->mu);
    return is->table_bytes;
}

using issuer_detail::Group;
using issuer_detail::Ragged;
// items -> groups by message count; items whose count exceeds the issuer's limit get `too_many` and join no group
static void issuer_group(bbs_issuer* is, size_t n, const std::vector<uint64_t>& count, const std::vector<int8_t>& pre, int8_t* status,
                         std::map<size_t, Group>& groups) {
    size_t max_messages;
    { std::lock_guard<std::mutex> g(is->mu); max_messages = is->max_messages; }
    for (size_t i = 0; i < n; i++) {
        if (pre[i] != 1) { status[i] = pre[i]; continue; }
        if (count[i] > max_messages) { status[i] = BBS_ST_INVALID_MESSAGE_AND_GENERATORS_LENGTH; continue; }
        Group& g = groups[(size_t)count[i]];
        g.L = (size_t)count[i];
        g.items.push_back(i);
    }
}

// the reference's PUBLIC proof_verify (src/proof_verify.rs:19-61), every item with the generators of ITS OWN length
// L_i = U_i + R_i (:40-43): U_i from the length of the proof's octet string, R_i = number of disclosed indexes
// the asynchronous form of every bbs_issuer_* call: *_submit packs and submits the groups and returns; bbs_issuer_job_wait
// waits for every group and scatters statuses / outputs into the caller's buffers (which stay valid until then; the inputs may
// be released when submit returns)
static bool issuer_has(bbs_issuer* is, bool need_sk) {
    std::lock_guard<std::mutex> g(is->mu);
    return need_sk ? is->sk_set : is->pk_set;
}
static int issuer_finish_submit(int rc, std::unique_ptr<bbs_issuer_job>& job, bbs_issuer_job** out) {
    if (rc) { (void)issuer_detail::wait_all(job->issuer, job->groups); return rc; }
    *out = job.release();
    return BBS_OK;
}
int bbs_issuer_job_wait(bbs_issuer_job* job) {
    if (!job) return BBS_E_ARG;
    if (job->delivered) return BBS_OK;
    const int rc = issuer_detail::wait_all(job->issuer, job->groups);
    if (rc) return rc;
    if (job->scatter) job->scatter(job->groups);
    job->delivered = true;
    return BBS_OK;
}
void bbs_issuer_job_free(bbs_issuer_job* job) {
    if (!job) return;
    (void)issuer_detail::wait_all(job->issuer, job->groups);       // the groups' buffers must outlive their jobs; releases the contexts
    delete job;
}
static int issuer_sync(int rc, bbs_issuer_job* job) {
    if (rc) return rc;
    rc = bbs_issuer_job_wait(job);
    bbs_issuer_job_free(job);
    return rc;
}

int bbs_issuer_proof_verify_submit(bbs_issuer* is, size_t n, const uint8_t* oct, const uint64_t* oct_off, const uint8_t* msg_bytes,
                                   const uint64_t* msg_byte_off, const uint64_t* msg_item_off, const uint64_t* di, const uint64_t* dio,
                                   const uint8_t* h, const uint64_t* ho, const uint8_t* ph, const uint64_t* pho, int8_t* status,
                                   bbs_issuer_job** job_out) {
    if (!is || !status || !job_out || (n && (!oct_off || !msg_item_off || !dio))) return BBS_E_ARG;
    if (!issuer_has(is, false)) return BBS_E_STATE;
    const Ragged ro{oct, oct_off, 1}, rdi{reinterpret_cast<const uint8_t*>(di), dio, 8}, rh{h, ho, 1}, rp{ph, pho, 1};
    if (!ro.sane(n) || !rdi.sane(n) || !rh.sane(n) || !rp.sane(n) || !issuer_detail::msgs_sane(n, msg_bytes, msg_byte_off, msg_item_off)) return BBS_E_ARG;
    const size_t fpb = bbs_fp_bytes(is->curve);
    std::vector<uint64_t> count(n, 0);
    std::vector<int8_t> pre(n, 1);
    for (size_t i = 0; i < n; i++) {
        const uint64_t len = ro.count(i), floor_ = 3 * fpb + 4 * 32;
        // shape of the octet string (what bbs_proof_from_octets checks first): 3 points, 4 + U scalars
        if (len < floor_ || (len - floor_) % 32) { pre[i] = BBS_ST_INVALID_ENCODING; continue; }
        count[i] = (len - floor_) / 32 + rdi.count(i);
    }
    std::unique_ptr<bbs_issuer_job> job(new bbs_issuer_job());
    job->issuer = is;
    std::map<size_t, Group>& groups = job->groups;
    issuer_group(is, n, count, pre, status, groups);
    int rc = BBS_OK;
    for (auto& kv : groups) {
        Group& g = kv.second;
        g.oct.gather(ro, g.items); g.di.gather(rdi, g.items); g.hdr.gather(rh, g.items); g.ph.gather(rp, g.items);
        g.msgs.gather(msg_bytes, msg_byte_off, msg_item_off, g.items);
        g.status.assign(g.items.size(), ST_PENDING);
        rc = issuer_detail::submit_group(is, g, [&g](bbs_ctx* c) {
            return bbs_proof_verify_wire_submit(c, g.items.size(), g.oct.data.data(), g.oct.off.data(), g.msgs.bytes.data(), g.msgs.byte_off.data(),
                                                g.msgs.item_off.data(), reinterpret_cast<const uint64_t*>(g.di.data.data()), g.di.off.data(),
                                                g.hdr.data.data(), g.hdr.off.data(), g.ph.data.data(), g.ph.off.data(), g.status.data(), &g.job);
        });
        if (rc) break;
    }
    job->scatter = [status](std::map<size_t, Group>& gs) {
        for (auto& kv : gs) for (size_t k = 0; k < kv.second.items.size(); k++) status[kv.second.items[k]] = kv.second.status[k];
    };
    return issuer_finish_submit(rc, job, job_out);
}
int bbs_issuer_proof_verify(bbs_issuer* is, size_t n, const uint8_t* oct, const uint64_t* oct_off, const uint8_t* msg_bytes,
                            const uint64_t* msg_byte_off, const uint64_t* msg_item_off, const uint64_t* di, const uint64_t* dio,
                            const uint8_t* h, const uint64_t* ho, const uint8_t* ph, const uint64_t* pho, int8_t* status) {
    bbs_issuer_job* job = nullptr;
    return issuer_sync(bbs_issuer_proof_verify_submit(is, n, oct, oct_off, msg_bytes, msg_byte_off, msg_item_off, di, dio, h, ho, ph, pho, status, &job), job);
}

// the reference's PUBLIC verify (src/verify.rs:18-50): generators by the item's number of messages (:30-35)
int bbs_issuer_verify_submit(bbs_issuer* is, size_t n, const uint8_t* sig_octets, const uint8_t* msg_bytes, const uint64_t* msg_byte_off,
                             const uint64_t* msg_item_off, const uint8_t* h, const uint64_t* ho, int8_t* status, bbs_issuer_job** job_out) {
    if (!is || !status || !job_out || (n && (!sig_octets || !msg_item_off))) return BBS_E_ARG;
    if (!issuer_has(is, false)) return BBS_E_STATE;
    const Ragged rh{h, ho, 1};
    if (!rh.sane(n) || !issuer_detail::msgs_sane(n, msg_bytes, msg_byte_off, msg_item_off)) return BBS_E_ARG;
    const size_t so = bbs_fp_bytes(is->curve) + 32;
    std::vector<uint64_t> count(n);
    for (size_t i = 0; i < n; i++) count[i] = msg_item_off[i + 1] - msg_item_off[i];
    std::unique_ptr<bbs_issuer_job> job(new bbs_issuer_job());
    job->issuer = is;
    std::map<size_t, Group>& groups = job->groups;
    issuer_group(is, n, count, std::vector<int8_t>(n, 1), status, groups);
    int rc = BBS_OK;
    for (auto& kv : groups) {
        Group& g = kv.second;
        g.oct.data.resize(g.items.size() * so + 8);
        for (size_t k = 0; k < g.items.size(); k++) std::memcpy(g.oct.data.data() + k * so, sig_octets + g.items[k] * so, so);
        g.hdr.gather(rh, g.items);
        g.msgs.gather(msg_bytes, msg_byte_off, msg_item_off, g.items);
        g.status.assign(g.items.size(), ST_PENDING);
        rc = issuer_detail::submit_group(is, g, [&g](bbs_ctx* c) {
            return bbs_verify_wire_submit(c, g.items.size(), g.oct.data.data(), g.msgs.bytes.data(), g.msgs.byte_off.data(), g.msgs.item_off.data(),
                                          g.hdr.data.data(), g.hdr.off.data(), g.status.data(), &g.job);
        });
        if (rc) break;
    }
    job->scatter = [status](std::map<size_t, Group>& gs) {
        for (auto& kv : gs) for (size_t k = 0; k < kv.second.items.size(); k++) status[kv.second.items[k]] = kv.second.status[k];
    };
    return issuer_finish_submit(rc, job, job_out);
}
int bbs_issuer_verify(bbs_issuer* is, size_t n, const uint8_t* sig_octets, const uint8_t* msg_bytes, const uint64_t* msg_byte_off,
                      const uint64_t* msg_item_off, const uint8_t* h, const uint64_t* ho, int8_t* status) {
    bbs_issuer_job* job = nullptr;
    return issuer_sync(bbs_issuer_verify_submit(is, n, sig_octets, msg_bytes, msg_byte_off, msg_item_off, h, ho, status, &job), job);
}

// the reference's PUBLIC sign (src/sign.rs:32-60): generators by the item's number of messages (:44-49); signature octet
// strings (fp_bytes + 32 each, zeros where status != 1) in the caller's order
int bbs_issuer_sign_submit(bbs_issuer* is, size_t n, const uint8_t* msg_bytes, const uint64_t* msg_byte_off, const uint64_t* msg_item_off,
                           const uint8_t* h, const uint64_t* ho, uint8_t* sig_octets_out, int8_t* status, bbs_issuer_job** job_out) {
    if (!is || !status || !job_out || (n && (!sig_octets_out || !msg_item_off))) return BBS_E_ARG;
    if (!issuer_has(is, true)) return BBS_E_STATE;
    const Ragged rh{h, ho, 1};
    if (!rh.sane(n) || !issuer_detail::msgs_sane(n, msg_bytes, msg_byte_off, msg_item_off)) return BBS_E_ARG;
    const size_t so = bbs_fp_bytes(is->curve) + 32;
    std::vector<uint64_t> count(n);
    for (size_t i = 0; i < n; i++) count[i] = msg_item_off[i + 1] - msg_item_off[i];
    std::unique_ptr<bbs_issuer_job> job(new bbs_issuer_job());
    job->issuer = is;
    std::map<size_t, Group>& groups = job->groups;
    issuer_group(is, n, count, std::vector<int8_t>(n, 1), status, groups);
    if (n) std::memset(sig_octets_out, 0, n * so);
    int rc = BBS_OK;
    for (auto& kv : groups) {
        Group& g = kv.second;
        g.hdr.gather(rh, g.items);
        g.msgs.gather(msg_bytes, msg_byte_off, msg_item_off, g.items);
        g.status.assign(g.items.size(), ST_PENDING);
        g.out.assign(g.items.size() * so + 8, 0);
        rc = issuer_detail::submit_group(is, g, [&g](bbs_ctx* c) {
            return bbs_sign_wire_submit(c, g.items.size(), g.msgs.bytes.data(), g.msgs.byte_off.data(), g.msgs.item_off.data(), g.hdr.data.data(),
                                        g.hdr.off.data(), g.out.data(), g.status.data(), &g.job);
        });
        if (rc) break;
    }
    job->scatter = [status, sig_octets_out, so](std::map<size_t, Group>& gs) {
        for (auto& kv : gs) for (size_t k = 0; k < kv.second.items.size(); k++) {
            status[kv.second.items[k]] = kv.second.status[k];
            std::memcpy(sig_octets_out + kv.second.items[k] * so, kv.second.out.data() + k * so, so);
        }
    };
    return issuer_finish_submit(rc, job, job_out);
}
int bbs_issuer_sign(bbs_issuer* is, size_t n, const uint8_t* msg_bytes, const uint64_t* msg_byte_off, const uint64_t* msg_item_off,
                    const uint8_t* h, const uint64_t* ho, uint8_t* sig_octets_out, int8_t* status) {
    bbs_issuer_job* job = nullptr;
    return issuer_sync(bbs_issuer_sign_submit(is, n, msg_bytes, msg_byte_off, msg_item_off, h, ho, sig_octets_out, status, &job), job);
}

// the reference's PUBLIC proof_gen (src/proof_gen.rs:78-113): generators by the item's number of messages (:91-96); proof
// octet strings packed in the caller's order (oct_off_out: n + 1 byte offsets; a failed item is empty).  octets_out needs
// sum_i (3 * fp_bytes + 32 * (4 + messages_i)) bytes.
int bbs_issuer_proof_gen_submit(bbs_issuer* is, size_t n, const uint8_t* sig_octets, const uint8_t* msg_bytes, const uint64_t* msg_byte_off,
                                const uint64_t* msg_item_off, const uint64_t* di, const uint64_t* dio, const uint8_t* rnd, const uint64_t* rno,
                                const uint8_t* h, const uint64_t* ho, const uint8_t* ph, const uint64_t* pho,
                                uint8_t* octets_out, uint64_t* oct_off_out, int8_t* status, bbs_issuer_job** job_out) {
    if (!is || !status || !oct_off_out || !job_out || (n && (!sig_octets || !octets_out || !msg_item_off || !dio || !rno))) return BBS_E_ARG;
    if (!issuer_has(is, false)) return BBS_E_STATE;
    const Ragged rdi{reinterpret_cast<const uint8_t*>(di), dio, 8}, rr{rnd, rno, 32}, rh{h, ho, 1}, rp{ph, pho, 1};
    if (!rdi.sane(n) || !rr.sane(n) || !rh.sane(n) || !rp.sane(n) || !issuer_detail::msgs_sane(n, msg_bytes, msg_byte_off, msg_item_off)) return BBS_E_ARG;
    const size_t fpb = bbs_fp_bytes(is->curve), so = fpb + 32;
    std::vector<uint64_t> count(n);
    for (size_t i = 0; i < n; i++) count[i] = msg_item_off[i + 1] - msg_item_off[i];
    std::unique_ptr<bbs_issuer_job> job(new bbs_issuer_job());
    job->issuer = is;
    std::map<size_t, Group>& groups = job->groups;
    issuer_group(is, n, count, std::vector<int8_t>(n, 1), status, groups);
    int rc = BBS_OK;
    for (auto& kv : groups) {
        Group& g = kv.second;
        g.oct.data.resize(g.items.size() * so + 8);
        for (size_t k = 0; k < g.items.size(); k++) std::memcpy(g.oct.data.data() + k * so, sig_octets + g.items[k] * so, so);
        g.di.gather(rdi, g.items); g.rnd.gather(rr, g.items); g.hdr.gather(rh, g.items); g.ph.gather(rp, g.items);
        g.msgs.gather(msg_bytes, msg_byte_off, msg_item_off, g.items);
        g.status.assign(g.items.size(), ST_PENDING);
        g.out.assign(g.items.size() * (3 * fpb + 32 * (4 + g.L)) + 8, 0);
        g.out_off.assign(g.items.size() + 1, 0);
        rc = issuer_detail::submit_group(is, g, [&g](bbs_ctx* c) {
            return bbs_proof_gen_wire_submit(c, g.items.size(), g.oct.data.data(), g.msgs.bytes.data(), g.msgs.byte_off.data(), g.msgs.item_off.data(),
                                             reinterpret_cast<const uint64_t*>(g.di.data.data()), g.di.off.data(), g.rnd.data.data(), g.rnd.off.data(),
                                             g.hdr.data.data(), g.hdr.off.data(), g.ph.data.data(), g.ph.off.data(), g.out.data(), g.out_off.data(),
                                             g.status.data(), &g.job);
        });
        if (rc) break;
    }
    job->scatter = [status, octets_out, oct_off_out, n](std::map<size_t, Group>& gs) {
        // lengths in the caller's order, then the bytes
        std::vector<uint64_t> len(n, 0);
        std::vector<const uint8_t*> src(n, nullptr);
        for (auto& kv : gs) for (size_t k = 0; k < kv.second.items.size(); k++) {
            const size_t i = kv.second.items[k];
            status[i] = kv.second.status[k];
            len[i] = kv.second.out_off[k + 1] - kv.second.out_off[k];
            src[i] = kv.second.out.data() + kv.second.out_off[k];
        }
        oct_off_out[0] = 0;
        for (size_t i = 0; i < n; i++) {
            if (len[i]) std::memcpy(octets_out + oct_off_out[i], src[i], (size_t)len[i]);
            oct_off_out[i + 1] = oct_off_out[i] + len[i];
        }
    };
    return issuer_finish_submit(rc, job, job_out);
}
int bbs_issuer_proof_gen(bbs_issuer* is, size_t n, const uint8_t* sig_octets, const uint8_t* msg_bytes, const uint64_t* msg_byte_off,
                         const uint64_t* msg_item_off, const uint64_t* di, const uint64_t* dio, const uint8_t* rnd, const uint64_t* rno,
                         const uint8_t* h, const uint64_t* ho, const uint8_t* ph, const uint64_t* pho,
                         uint8_t* octets_out, uint64_t* oct_off_out, int8_t* status) {
    bbs_issuer_job* job = nullptr;
    return issuer_sync(bbs_issuer_proof_gen_submit(is, n, sig_octets, msg_bytes, msg_byte_off, msg_item_off, di, dio, rnd, rno, h, ho, ph, pho,
                                                   octets_out, oct_off_out, status, &job), job);
}

int bbs_pairing_product2_is_one_batch(bbs_ctx* ctx, size_t n, const uint8_t* pa, const uint8_t* pb, int8_t* status) {
    if (!ctx) return BBS_E_ARG;
    return DISPATCH(ctx, pairing_batch<BlsCurve>(AS_BLS(ctx), n, pa, pb, status), pairing_batch<BnCurve>(AS_BN(ctx), n, pa, pb, status));
}

}  // extern "C"
// ---- pool: a list of proofs over several GPUs (pool.hpp) ----------------------------------------
int bbs_pool_create(const int* device_ids, size_t n_devices, bbs_pool** out) {
    if (!device_ids || !n_devices || n_devices > 64 || !out) return BBS_E_ARG;
    const int have = rt::device_count();
    for (size_t d = 0; d < n_devices; d++) if (device_ids[d] < 0 || device_ids[d] >= have) return BBS_E_NO_DEVICE;
    auto p = std::unique_ptr<bbs_pool>(new bbs_pool());
    p->devices.assign(device_ids, device_ids + n_devices);
    *out = p.release();
    return BBS_OK;
}
void bbs_pool_destroy(bbs_pool* pool) { delete pool; }
size_t bbs_pool_device_count(const bbs_pool* pool) { return pool ? pool->devices.size() : 0; }
int bbs_pool_set_window_bits(bbs_pool* pool, int curve, int bits) {
    if (!pool) return BBS_E_ARG;
    return pool->each(curve, [&](bbs_ctx* c) { return bbs_ctx_set_window_bits(c, bits); });
}
int bbs_pool_set_generators(bbs_pool* pool, int curve, const uint8_t* g, size_t count, const uint8_t* api_id, size_t api_id_len) {
    if (!pool) return BBS_E_ARG;
    return pool->each(curve, [&](bbs_ctx* c) { return bbs_ctx_set_generators(c, g, count, api_id, api_id_len); });
}
int bbs_pool_set_public_key(bbs_pool* pool, int curve, const uint8_t* pk, int is_identity) {
    if (!pool) return BBS_E_ARG;
    return pool->each(curve, [&](bbs_ctx* c) { return bbs_ctx_set_public_key(c, pk, is_identity); });
}
int bbs_pool_set_inflight(bbs_pool* pool, int jobs_per_member) {
    if (!pool || jobs_per_member < 1 || jobs_per_member > 64) return BBS_E_ARG;
    pool->inflight.store(jobs_per_member);
    return BBS_OK;
}
int bbs_pool_context(bbs_pool* pool, int curve, size_t member, bbs_ctx** out) {
    if (!pool || !out || member >= pool->devices.size()) return BBS_E_ARG;
    std::lock_guard<std::mutex> g(pool->mu);
    if (int rc = pool->ensure(curve)) return rc;
    *out = pool->ctx[bbs_pool::curve_slot(curve)][member];
    return BBS_OK;
}
int bbs_pool_proof_verify_submit(bbs_pool* pool, const bbs_pv_list* lists, size_t n_lists, size_t max_batch, bbs_pool_job** job_out) {
    return pool_proof_verify_submit(pool, lists, n_lists, max_batch, job_out);
}
int bbs_pool_job_wait(bbs_pool_job* job) { return pool_job_wait(job); }
void bbs_pool_job_free(bbs_pool_job* job) { pool_job_free(job); }
int bbs_pool_proof_verify(bbs_pool* pool, const bbs_pv_list* lists, size_t n_lists, size_t max_batch) {
    bbs_pool_job* job = nullptr;
    int rc = pool_proof_verify_submit(pool, lists, n_lists, max_batch, &job);
    if (rc) return rc;
    rc = pool_job_wait(job);
    pool_job_free(job);
    return rc;
}
#pragma GCC visibility pop
