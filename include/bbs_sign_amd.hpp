// C++ host mirror of the reference's PUBLIC interface (README.md:43-128) over the C ABI of bbs_sign_amd.h:
//
//   SecretKey::key_gen / sk_to_pk / sign      src/key_gen.rs:46-90, src/sign.rs:32-60
//   PublicKey::verify                          src/verify.rs:18-50
//   proof_gen / proof_verify                   src/proof_gen.rs:78-113, src/proof_verify.rs:19-61
//
// Same names, argument meaning and error behaviour: messages are byte strings, results are Result<T> holding the
// value or the reference's error variant (the BBS_ST_* code); what the reference panics on (dst too long, ...)
// throws.  What the reference recomputes on every call -- create_generators and the api_id strings -- is computed
// once per (curve, message count, key) and kept in an engine context (bbs_ctx).  Header-only; link libbbs_sign_amd.so.
#pragma once
#include <array>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>

#include "bbs_sign_amd.h"

namespace bbs_plus {

enum class Curve : int { Bls12_381 = BBS_CURVE_BLS12_381, Bn254 = BBS_CURVE_BN254 };
using Bytes = std::vector<uint8_t>;

// the reference's Result<T, SignatureError | ProofGenError | KeyGenError>
template <class T>
struct Result {
    T value{};
    int error = 0;                       // 0 = Ok, else the BBS_ST_* code of the Err variant
    bool is_ok() const { return error == 0; }
    bool is_err() const { return error != 0; }
    const T& unwrap() const { if (error) throw std::runtime_error("unwrap on Err(" + std::to_string(error) + ")"); return value; }
    static Result ok(T v) { Result r; r.value = std::move(v); return r; }
    static Result err(int code) { Result r; r.error = code; return r; }
};

inline const Bytes& ciphersuite_id(Curve c) {        // src/constants.rs
    static const Bytes bls = {'B','B','S','_','B','L','S','1','2','3','8','1','G','1','_','X','M','D',':','S','H','A','-','2','5','6','_','S','S','W','U','_','R','O','_'};
    static const std::string bn_s = "BBS_QUUX-V01-CS02-with-BN254G1_XMD:SHA-256_SVDW_RO_";
    static const Bytes bn(bn_s.begin(), bn_s.end());
    return c == Curve::Bls12_381 ? bls : bn;
}
inline Bytes api_id(Curve c) {                        // sign.rs:44, verify.rs:31, proof_gen.rs:94, proof_verify.rs:35
    Bytes a = ciphersuite_id(c);
    const char* suf = "H2G_HM2S_";
    a.insert(a.end(), suf, suf + 9);
    return a;
}

struct Signature {                                    // src/sign.rs:18-22 : A (G1 affine, x || y LE, identity = zeros) || e
    Bytes record;
};
struct Proof {                                        // src/proof_gen.rs:29-39
    Bytes fixed;                                      // a_bar || b_bar || d || e_cap || r1_cap || r3_cap || challenge
    Bytes commitments;                                // 32 B LE each
    size_t n_commitments() const { return commitments.size() / 32; }
};

namespace detail {

inline void check(int rc, const char* where) {
    if (rc != BBS_OK) throw std::runtime_error(std::string(where) + ": " + std::to_string(rc));
}
// a reference panic or an input arkworks' types cannot hold is not an Err variant
inline void raise_if_not_variant(int8_t st) {
    if (st <= BBS_ST_PANIC_SK_PLUS_E_ZERO) throw std::runtime_error("bbs_sign_amd status " + std::to_string((int)st));
}

struct CtxDeleter { void operator()(bbs_ctx* c) const { bbs_ctx_destroy(c); } };
using CtxPtr = std::shared_ptr<bbs_ctx>;

// one engine context per (curve, L, key role, key bytes); generators from the library's create_generators
inline CtxPtr context(Curve c, size_t L, bool secret, const Bytes& key, bool pk_identity, int device = 0) {
    static std::mutex mu;
    static std::map<std::string, CtxPtr> cache;
    std::string k = std::to_string((int)c) + ":" + std::to_string(L) + ":" + (secret ? "s" : pk_identity ? "i" : "p") + ":" +
                    std::string(key.begin(), key.end()) + ":" + std::to_string(device);
    std::lock_guard<std::mutex> g(mu);
    auto it = cache.find(k);
    if (it != cache.end()) return it->second;
    const size_t fpb = bbs_fp_bytes((int)c);
    const Bytes aid = api_id(c);
    Bytes gens((L + 1) * 2 * fpb);
    check(bbs_create_generators((int)c, L + 1, aid.data(), aid.size(), gens.data()), "bbs_create_generators");
    bbs_ctx* raw = nullptr;
    check(bbs_ctx_create((int)c, device, &raw), "bbs_ctx_create");
    CtxPtr ctx(raw, CtxDeleter());
    check(bbs_ctx_set_generators(raw, gens.data(), L + 1, aid.data(), aid.size()), "bbs_ctx_set_generators");
    if (secret) check(bbs_ctx_set_secret_key(raw, key.data()), "bbs_ctx_set_secret_key");
    else check(bbs_ctx_set_public_key(raw, key.data(), pk_identity ? 1 : 0), "bbs_ctx_set_public_key");
    cache[k] = ctx;
    return ctx;
}

// msg_to_scalars (src/utils/interface_utilities.rs:76-88) on the device
inline Bytes msg_to_scalars(bbs_ctx* ctx, Curve c, const std::vector<Bytes>& msgs) {
    Bytes out(32 * msgs.size());
    if (msgs.empty()) return out;
    Bytes flat;
    std::vector<uint64_t> off{0};
    for (const auto& m : msgs) { flat.insert(flat.end(), m.begin(), m.end()); off.push_back(flat.size()); }
    Bytes dst = api_id(c);
    const char* suf = "MAP_MSG_TO_SCALAR_AS_HASH_";
    dst.insert(dst.end(), suf, suf + 26);
    flat.push_back(0);
    check(bbs_hash_to_scalar_batch(ctx, msgs.size(), flat.data(), off.data(), dst.data(), dst.size(), out.data()), "bbs_hash_to_scalar_batch");
    return out;
}
inline const uint8_t* ptr(const Bytes& b) { static const uint8_t z = 0; return b.empty() ? &z : b.data(); }

}  // namespace detail

class PublicKey {                                     // src/key_gen.rs:12-15
public:
    Curve curve = Curve::Bls12_381;
    Bytes pk;                                         // x.c0 || x.c1 || y.c0 || y.c1, canonical LE
    bool identity = false;                            // PublicKey::default()

    // PublicKey::verify (src/verify.rs:18-50)
    Result<bool> verify(const Signature& sig, const Bytes& header, const std::vector<Bytes>& msgs) const {
        auto ctx = detail::context(curve, msgs.size(), false, pk, identity);
        const Bytes sc = detail::msg_to_scalars(ctx.get(), curve, msgs);
        const uint64_t moff[2] = {0, (uint64_t)msgs.size()}, hoff[2] = {0, (uint64_t)header.size()};
        int8_t st = 0;
        detail::check(bbs_core_verify_batch(ctx.get(), 1, sig.record.data(), detail::ptr(sc), moff, detail::ptr(header), hoff, &st), "bbs_core_verify_batch");
        detail::raise_if_not_variant(st);
        return st >= 0 ? Result<bool>::ok(st == 1) : Result<bool>::err(st);
    }
};

class SecretKey {
public:
    Curve curve = Curve::Bls12_381;
    std::array<uint8_t, 32> sk{};                     // canonical LE

    // SecretKey::key_gen (src/key_gen.rs:46-81)
    static Result<SecretKey> key_gen(Curve c, const Bytes& key_material, const Bytes& key_info, const Bytes& key_dst) {
        SecretKey s;
        s.curve = c;
        const int rc = bbs_key_gen((int)c, detail::ptr(key_material), key_material.size(), detail::ptr(key_info), key_info.size(),
                                   detail::ptr(key_dst), key_dst.size(), s.sk.data());
        if (rc == BBS_ST_INVALID_KEY_MATERIAL_LENGTH || rc == BBS_ST_INVALID_KEY_INFO_LENGTH || rc == BBS_ST_INVALID_SECRET_KEY)
            return Result<SecretKey>::err(rc);
        detail::check(rc, "bbs_key_gen");
        return Result<SecretKey>::ok(s);
    }
    // SecretKey::sk_to_pk (src/key_gen.rs:83-90)
    PublicKey sk_to_pk() const {
        auto ctx = detail::context(curve, 0, true, Bytes(sk.begin(), sk.end()), false);
        PublicKey p;
        p.curve = curve;
        p.pk.resize(4 * bbs_fp_bytes((int)curve));
        int inf = 0;
        detail::check(bbs_ctx_get_public_key(ctx.get(), p.pk.data(), &inf), "bbs_ctx_get_public_key");
        p.identity = inf != 0;
        return p;
    }
    // SecretKey::sign (src/sign.rs:32-60)
    Result<Signature> sign(const std::vector<Bytes>& msgs, const Bytes& header) const {
        auto ctx = detail::context(curve, msgs.size(), true, Bytes(sk.begin(), sk.end()), false);
        const Bytes sc = detail::msg_to_scalars(ctx.get(), curve, msgs);
        const uint64_t moff[2] = {0, (uint64_t)msgs.size()}, hoff[2] = {0, (uint64_t)header.size()};
        Signature sig;
        sig.record.resize(2 * bbs_fp_bytes((int)curve) + 32);
        int8_t st = 0;
        detail::check(bbs_core_sign_batch(ctx.get(), 1, detail::ptr(sc), moff, detail::ptr(header), hoff, sig.record.data(), &st), "bbs_core_sign_batch");
        detail::raise_if_not_variant(st);
        return st == 1 ? Result<Signature>::ok(std::move(sig)) : Result<Signature>::err(st);
    }
};

// calculate_random_scalars (src/utils/core_utilities.rs:70-81): 48 random bytes mod r each
inline Bytes calculate_random_scalars(Curve c, size_t count) {
    std::random_device rd;
    Bytes out(32 * count);
    for (size_t k = 0; k < count; k++) {
        uint8_t okm[48];
        for (int i = 0; i < 48; i += 4) { const uint32_t w = rd(); std::memcpy(okm + i, &w, 4); }
        detail::check(bbs_scalar_from_okm((int)c, okm, out.data() + 32 * k), "bbs_scalar_from_okm");
    }
    return out;
}

// proof_gen (src/proof_gen.rs:78-113)
inline Result<Proof> proof_gen(const PublicKey& pk, const Signature& sig, const Bytes& header, const Bytes& ph,
                               const std::vector<Bytes>& msgs, const std::vector<size_t>& disclosed_indexes) {
    const Curve c = pk.curve;
    const size_t L = msgs.size(), R = disclosed_indexes.size(), fpb = bbs_fp_bytes((int)c);
    auto ctx = detail::context(c, L, false, pk.pk, pk.identity);
    const Bytes sc = detail::msg_to_scalars(ctx.get(), c, msgs);
    const size_t n_rnd = 5 + L > R ? 5 + L - R : 0;                     // proof_gen.rs:145-149
    const Bytes rnd = calculate_random_scalars(c, n_rnd);
    std::vector<uint64_t> idx(disclosed_indexes.begin(), disclosed_indexes.end());
    idx.push_back(0);
    const uint64_t moff[2] = {0, (uint64_t)L}, ioff[2] = {0, (uint64_t)R}, roff[2] = {0, (uint64_t)n_rnd},
                   hoff[2] = {0, (uint64_t)header.size()}, poff[2] = {0, (uint64_t)ph.size()};
    Proof p;
    p.fixed.resize(6 * fpb + 128);
    Bytes cm(32 * (L + 1));
    uint64_t coff[2] = {0, 0};
    int8_t st = 0;
    detail::check(bbs_core_proof_gen_batch(ctx.get(), 1, sig.record.data(), detail::ptr(sc), moff, idx.data(), ioff, detail::ptr(rnd), roff,
                                           detail::ptr(header), hoff, detail::ptr(ph), poff, p.fixed.data(), cm.data(), coff, &st),
                  "bbs_core_proof_gen_batch");
    detail::raise_if_not_variant(st);
    if (st != 1) return Result<Proof>::err(st);
    p.commitments.assign(cm.begin(), cm.begin() + 32 * (coff[1] - coff[0]));
    return Result<Proof>::ok(std::move(p));
}

// proof_verify (src/proof_verify.rs:19-61): L is inferred as commitments + disclosed indexes
inline Result<bool> proof_verify(const PublicKey& pk, const Proof& proof, const Bytes& header, const Bytes& ph,
                                 const std::vector<Bytes>& disclosed_msgs, const std::vector<size_t>& disclosed_indexes) {
    const Curve c = pk.curve;
    const size_t U = proof.n_commitments(), R = disclosed_indexes.size();
    auto ctx = detail::context(c, U + R, false, pk.pk, pk.identity);
    const Bytes dm = detail::msg_to_scalars(ctx.get(), c, disclosed_msgs);
    std::vector<uint64_t> idx(disclosed_indexes.begin(), disclosed_indexes.end());
    idx.push_back(0);
    const uint64_t coff[2] = {0, (uint64_t)U}, moff[2] = {0, (uint64_t)disclosed_msgs.size()}, ioff[2] = {0, (uint64_t)R},
                   hoff[2] = {0, (uint64_t)header.size()}, poff[2] = {0, (uint64_t)ph.size()};
    int8_t st = 0;
    detail::check(bbs_core_proof_verify_batch(ctx.get(), 1, proof.fixed.data(), detail::ptr(proof.commitments), coff, detail::ptr(dm), moff,
                                              idx.data(), ioff, detail::ptr(header), hoff, detail::ptr(ph), poff, &st),
                  "bbs_core_proof_verify_batch");
    detail::raise_if_not_variant(st);
    return st >= 0 ? Result<bool>::ok(st == 1) : Result<bool>::err(st);
}

// proof_verify for n proofs of one issuer and one message count at once (what a verifier service calls: one engine
// batch instead of n calls); element i of the result is what proof_verify(...) returns for proof i
inline std::vector<Result<bool>> proof_verify_batch(const PublicKey& pk, const std::vector<Proof>& proofs,
                                                    const std::vector<Bytes>& headers, const std::vector<Bytes>& phs,
                                                    const std::vector<std::vector<Bytes>>& disclosed_msgs,
                                                    const std::vector<std::vector<size_t>>& disclosed_indexes, size_t message_count) {
    const Curve c = pk.curve;
    const size_t n = proofs.size();
    auto ctx = detail::context(c, message_count, false, pk.pk, pk.identity);
    std::vector<Bytes> all_msgs;
    for (const auto& item : disclosed_msgs) all_msgs.insert(all_msgs.end(), item.begin(), item.end());
    const Bytes dm = detail::msg_to_scalars(ctx.get(), c, all_msgs);
    Bytes fixed, cm, hb, pb;
    std::vector<uint64_t> coff{0}, moff{0}, idx, ioff{0}, hoff{0}, poff{0};
    for (size_t i = 0; i < n; i++) {
        fixed.insert(fixed.end(), proofs[i].fixed.begin(), proofs[i].fixed.end());
        cm.insert(cm.end(), proofs[i].commitments.begin(), proofs[i].commitments.end());
        coff.push_back(cm.size() / 32);
        moff.push_back(moff.back() + disclosed_msgs[i].size());
        idx.insert(idx.end(), disclosed_indexes[i].begin(), disclosed_indexes[i].end());
        ioff.push_back(idx.size());
        hb.insert(hb.end(), headers[i].begin(), headers[i].end());
        hoff.push_back(hb.size());
        pb.insert(pb.end(), phs[i].begin(), phs[i].end());
        poff.push_back(pb.size());
    }
    idx.push_back(0);
    std::vector<int8_t> st(n ? n : 1, 0);
    detail::check(bbs_core_proof_verify_batch(ctx.get(), n, detail::ptr(fixed), detail::ptr(cm), coff.data(), detail::ptr(dm), moff.data(),
                                              idx.data(), ioff.data(), detail::ptr(hb), hoff.data(), detail::ptr(pb), poff.data(), st.data()),
                  "bbs_core_proof_verify_batch");
    std::vector<Result<bool>> out;
    for (size_t i = 0; i < n; i++) {
        detail::raise_if_not_variant(st[i]);
        out.push_back(st[i] >= 0 ? Result<bool>::ok(st[i] == 1) : Result<bool>::err(st[i]));
    }
    return out;
}

}  // namespace bbs_plus
