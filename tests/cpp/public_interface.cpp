// The reference's own public-interface tests, restated against include/bbs_sign_amd.hpp (the C++ host mirror):
//   README.md:43-128 (both ciphersuites), src/tests/bbs_over_bls_tests.rs:41-84 (round trips) and :86-187
//   (test_invalid_proof, cases 1-4), src/key_gen.rs:127-216 (key_gen errors), src/tests/test_vector.rs:139-192
//   (key pair and signature known answers).
// Built by tests/test_cpp_interface.py with g++ -std=c++17 against the product library (GPU) or the CPU-side test
// build of the same stage code.
#include <cstdio>
#include <cstdlib>

#include "bbs_sign_amd.hpp"

using namespace bbs_plus;

#define CHECK(cond) do { if (!(cond)) { std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); std::exit(1); } } while (0)

static Bytes B(const char* s) { return Bytes(s, s + std::strlen(s)); }
static std::string hex(const uint8_t* p, size_t n) {
    static const char* d = "0123456789abcdef";
    std::string s;
    for (size_t i = 0; i < n; i++) { s += d[p[i] >> 4]; s += d[p[i] & 15]; }
    return s;
}
static Bytes unhex(const char* h) {
    Bytes b;
    for (size_t i = 0; h[i] && h[i + 1]; i += 2) b.push_back((uint8_t)std::strtol(std::string(h + i, 2).c_str(), nullptr, 16));
    return b;
}

static void readme_flow(Curve c) {                     // README.md:64-128
    const Bytes key_material(32, 5), key_dst = B("BBS-SIG-KEYGEN-SALT-");
    auto skr = SecretKey::key_gen(c, key_material, {}, key_dst);
    CHECK(skr.is_ok());
    const SecretKey sk = skr.unwrap();
    const PublicKey pk = sk.sk_to_pk();
    const std::vector<Bytes> msgs = {B("message1"), B("message2"), B("msg3"), B("msg4")};
    auto sig = sk.sign(msgs, {});
    CHECK(sig.is_ok());
    auto res = pk.verify(sig.unwrap(), {}, msgs);
    CHECK(res.is_ok() && res.unwrap());
    const std::vector<size_t> disclosed = {0, 2};
    const std::vector<Bytes> disclosed_msgs = {msgs[0], msgs[2]};
    auto proof = proof_gen(pk, sig.unwrap(), {}, {}, msgs, disclosed);
    CHECK(proof.is_ok());
    CHECK(proof_verify(pk, proof.unwrap(), {}, {}, disclosed_msgs, disclosed).unwrap());
    // and the negatives of sign_verify_tests.rs: wrong message, wrong header
    std::vector<Bytes> wrong = msgs;
    wrong[3] = B("msg5");
    CHECK(!pk.verify(sig.unwrap(), {}, wrong).unwrap());
    CHECK(!pk.verify(sig.unwrap(), B("h"), msgs).unwrap());
    CHECK(!proof_verify(pk, proof.unwrap(), {}, {}, {msgs[0], msgs[1]}, disclosed).unwrap());
}

static void known_answers() {                           // test_vector.rs:139-192
    const Bytes ikm = unhex("746869732d49532d6a7573742d616e2d546573742d494b4d2d746f2d67656e65726174652d246528724074232d6b6579");
    const Bytes key_info = unhex("746869732d49532d736f6d652d6b65792d6d657461646174612d746f2d62652d757365642d696e2d746573742d6b65792d67656e");
    const Bytes key_dst = B("BBS_BLS12381G1_XMD:SHA-256_SSWU_RO_H2G_HM2S_KEYGEN_DST_");
    auto sk = SecretKey::key_gen(Curve::Bls12_381, ikm, key_info, key_dst).unwrap();
    Bytes be(sk.sk.rbegin(), sk.sk.rend());
    CHECK(hex(be.data(), 32) == "60e55110f76883a13d030b2f6bd11883422d5abde717569fc0731f51237169fc");
    const PublicKey pk = sk.sk_to_pk();
    Bytes oct(96);
    CHECK(bbs_public_key_to_octets(0, pk.pk.data(), 0, oct.data()) == 0);
    CHECK(hex(oct.data(), 96) == "a820f230f6ae38503b86c70dc50b61c58a77e45c39ab25c0652bbaa8fa136f2851bd4781c9dcde39fc9d1d52c9e60268"
                                  "061e7d7632171d91aa8d460acee0e96f1e7c4cfb12d3ff9ab5d5dc91c277db75c845d649ef3c4f63aebc364cd55ded0c");
    const Bytes m1 = unhex("9872ad089e452c7b6e283dfac2a80d58e8d0ff71cc4d5e310a1debdda4a45f02");
    const Bytes header = unhex("11223344556677889900aabbccddeeff");
    auto sig = sk.sign({m1}, header).unwrap();
    Bytes so(80);
    CHECK(bbs_signature_to_octets(0, sig.record.data(), so.data()) == 0);
    CHECK(hex(so.data(), 80) == "84773160b824e194073a57493dac1a20b667af70cd2352d8af241c77658da5253aa8458317cca0eae615690d55b1f271"
                                 "64657dcafee1d5c1973947aa70e2cfbb4c892340be5969920d0916067b4565a0");
    CHECK(pk.verify(sig, header, {m1}).unwrap());
    // key_gen errors (key_gen.rs:55-61)
    CHECK(SecretKey::key_gen(Curve::Bls12_381, Bytes(31, 1), {}, key_dst).error == BBS_ST_INVALID_KEY_MATERIAL_LENGTH);
    CHECK(SecretKey::key_gen(Curve::Bls12_381, ikm, Bytes(65536, 0), key_dst).error == BBS_ST_INVALID_KEY_INFO_LENGTH);
}

static void round_trips_and_invalid_proofs() {          // bbs_over_bls_tests.rs:41-187
    const Curve c = Curve::Bls12_381;
    auto sk = SecretKey::key_gen(c, Bytes(32, 1), {}, B("BBS-SIG-KEYGEN-SALT-")).unwrap();
    const PublicKey pk = sk.sk_to_pk();
    std::mt19937 rng(7);
    auto random_msgs = [&](size_t n) { std::vector<Bytes> v(n, Bytes(5)); for (auto& m : v) for (auto& b : m) b = (uint8_t)rng(); return v; };
    struct Case { size_t count; std::vector<size_t> disclosed; const char* header; };
    const Case cases[] = {{0, {}, ""}, {0, {}, "abc"}, {1, {0}, "abc"}, {1, {}, "abc"}, {10, {0, 1, 2}, ""}, {5, {0, 4}, "defghjsdjdbcjbejd"}};
    for (const auto& cs : cases) {
        const auto msgs = random_msgs(cs.count);
        const Bytes header = B(cs.header);
        auto sig = sk.sign(msgs, header).unwrap();
        CHECK(pk.verify(sig, header, msgs).unwrap());
        auto proof = proof_gen(pk, sig, header, {}, msgs, cs.disclosed).unwrap();
        std::vector<Bytes> dm;
        for (size_t i : cs.disclosed) dm.push_back(msgs[i]);
        CHECK(proof_verify(pk, proof, header, {}, dm, cs.disclosed).unwrap());
    }
    // test_invalid_proof
    const auto msgs = random_msgs(10);
    auto sig = sk.sign(msgs, {}).unwrap();
    const std::vector<size_t> d = {0, 1, 5};
    const std::vector<Bytes> dm = {msgs[0], msgs[1], msgs[5]};
    auto proof = proof_gen(pk, sig, {}, {}, msgs, d).unwrap();
    CHECK(proof_verify(pk, proof, {}, {}, dm, d).unwrap());
    const size_t fpb = bbs_fp_bytes((int)c);
    Proof forged = proof;                               // case 1: a_bar = identity
    std::fill(forged.fixed.begin(), forged.fixed.begin() + 2 * fpb, 0);
    CHECK(!proof_verify(pk, forged, {}, {}, dm, d).unwrap());
    Proof dflt;                                         // case 2: Proof::default() -> Err (no commitments)
    dflt.fixed.assign(6 * fpb + 128, 0);
    CHECK(proof_verify(pk, dflt, {}, {}, dm, d).is_err());
    PublicKey forged_pk;                                // case 3: PublicKey::default()
    forged_pk.curve = c;
    forged_pk.identity = true;
    forged_pk.pk.assign(4 * fpb, 0);
    CHECK(!proof_verify(forged_pk, proof, {}, {}, dm, d).unwrap());
    Proof zeros = dflt;                                 // case 4: seven zero commitments
    zeros.commitments.assign(7 * 32, 0);
    CHECK(!proof_verify(pk, zeros, {}, {}, dm, d).unwrap());
    // proof_gen errors (proof_gen.rs:135-143): disclosed index out of range, more indexes than messages
    CHECK(proof_gen(pk, sig, {}, {}, msgs, {0, 10}).is_err());
    // the same checks through one engine batch: valid, forged a_bar, Err (index out of range), wrong message
    std::vector<Proof> ps = {proof, forged, proof, proof};
    std::vector<std::vector<Bytes>> dms = {dm, dm, dm, {msgs[0], msgs[2], msgs[5]}};
    std::vector<std::vector<size_t>> ds = {d, d, {0, 1, 10}, d};
    auto rs = proof_verify_batch(pk, ps, std::vector<Bytes>(4), std::vector<Bytes>(4), dms, ds, 10);
    CHECK(rs.size() == 4 && rs[0].unwrap() && !rs[1].unwrap() && rs[2].is_err() && !rs[3].unwrap());
}

int main() {
    known_answers();
    readme_flow(Curve::Bls12_381);
    readme_flow(Curve::Bn254);
    round_trips_and_invalid_proofs();
    std::puts("public interface (C++): all checks passed");
    return 0;
}
