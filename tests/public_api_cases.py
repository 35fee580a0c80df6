"""The reference's public-interface tests (src/tests/bbs_over_bls_tests.rs, src/tests/test_vector.rs)
re-stated over bbs_sign_amd.api -- messages are bytes, generators come from the library's own
create_generators.  Run on the GPU (product library) and through the host twin."""
import random

from bbs_sign_amd import BbsError, Proof, api
from oracle import bbs
from oracle.bbs import BLS_SUITE as S
from oracle.curves import BLS12_381 as C

H = bytes.fromhex
IKM = H("746869732d49532d6a7573742d616e2d546573742d494b4d2d746f2d67656e65726174652d246528724074232d6b6579")
KEY_INFO = H("746869732d49532d736f6d652d6b65792d6d657461646174612d746f2d62652d757365642d696e2d746573742d6b65792d67656e")
KEY_DST = H("4242535f424c53313233383147315f584d443a5348412d3235365f535357555f524f5f4832475f484d32535f4b455947454e5f4453545f")
M1 = H("9872ad089e452c7b6e283dfac2a80d58e8d0ff71cc4d5e310a1debdda4a45f02")
HEADER = H("11223344556677889900aabbccddeeff")
PH = H("bed231d880675ed101ead304512e043ade9958dd0241ea70b4b3957fba941501")
SIG_HEX = "84773160b824e194073a57493dac1a20b667af70cd2352d8af241c77658da5253aa8458317cca0eae615690d55b1f27164657dcafee1d5c1973947aa70e2cfbb4c892340be5969920d0916067b4565a0"
PROOF_HEX = "94916292a7a6bade28456c601d3af33fcf39278d6594b467e128a3f83686a104ef2b2fcf72df0215eeaf69262ffe8194a19fab31a82ddbe06908985abc4c9825788b8a1610942d12b7f5debbea8985296361206dbace7af0cc834c80f33e0aadaeea5597befbb651827b5eed5a66f1a959bb46cfd5ca1a817a14475960f69b32c54db7587b5ee3ab665fbd37b506830a49f21d592f5e634f47cee05a025a2f8f94e73a6c15f02301d1178a92873b6e8634bafe4983c3e15a663d64080678dbf29417519b78af042be2b3e1c4d08b8d520ffab008cbaaca5671a15b22c239b38e940cfeaa5e72104576a9ec4a6fad78c532381aeaa6fb56409cef56ee5c140d455feeb04426193c57086c9b6d397d9418"


def check_create_generators_kat(lib_path=None):      # test_vector.rs:123-136 (host code: no GPU needed)
    g = api.create_generators("bls12_381", 11, lib_path)
    want = {
        0: "a9ec65b70a7fbe40c874c9eb041c2cb0a7af36ccec1bea48fa2ba4c2eb67ef7f9ecb17ed27d38d27cdeddff44c8137be",
        1: "98cd5313283aaf5db1b3ba8611fe6070d19e605de4078c38df36019fbaad0bd28dd090fd24ed27f7f4d22d5ff5dea7d4",
        2: "a31fbe20c5c135bcaa8d9fc4e4ac665cc6db0226f35e737507e803044093f37697a9d452490a970eea6f9ad6c3dcaa3a",
        10: "a1f229540474f4d6f1134761b92b788128c7ac8dc9b0c52d59493132679673032ac7db3fb3d79b46b13c1c41ee495bca",
    }
    for i, h in want.items():
        assert bbs.g1_compress(C, g[i]).hex() == h
    assert g == bbs.create_generators(S, 11, S.api_id)
    # BN254: the reference's one known answer -- P1 (constants.rs:39-51) is the first generator under the seed
    # "...BP_MESSAGE_GENERATOR_SEED" -- through the library's host code, and the generators against the oracle
    from oracle.hashing import expand_message, i2osp
    bn = bbs.BN_SUITE
    v = expand_message(bn.api_id + b"BP_MESSAGE_GENERATOR_SEED", bn.api_id + b"SIG_GENERATOR_SEED_", 48)
    v = expand_message(v + i2osp(1, 8), bn.api_id + b"SIG_GENERATOR_SEED_", 48)
    assert api.hash_to_g1("bn254", v, bn.api_id + b"SIG_GENERATOR_DST_", lib_path) == bn.p1
    assert api.create_generators("bn254", 6, lib_path) == bbs.create_generators(bn, 6, bn.api_id)
    import json, os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bbs_golden.json")) as f:
        gold = json.load(f)["bn254_create_generators"]
    assert api.create_generators("bn254", 11, lib_path) == [(int(x, 16), int(y, 16)) for x, y in gold["generators"]]
    msgs = [b"", b"a", bytes(range(200))]
    for m in msgs:
        assert api.hash_to_g1("bn254", m, b"QUUX-V01-CS02-with-BN254G1_XMD:SHA-256_SVDW_RO_", lib_path) == \
            __import__("oracle.hashing", fromlist=["x"]).hash_to_g1_bn(m, b"QUUX-V01-CS02-with-BN254G1_XMD:SHA-256_SVDW_RO_")


def check_key_gen_kat(lib_path=None):                 # test_vector.rs:139-160, key_gen.rs:127-216
    sk = api.SecretKey.key_gen("bls12_381", IKM, KEY_INFO, KEY_DST, lib_path)
    assert bbs.scalar_be(C, sk.sk).hex() == "60e55110f76883a13d030b2f6bd11883422d5abde717569fc0731f51237169fc"
    for bad, variant in ((bytes(31), "InvalidKeyMaterialLength"),):
        try:
            api.SecretKey.key_gen("bls12_381", bad, b"", KEY_DST, lib_path)
            assert False
        except BbsError as e:
            assert e.variant == variant
    try:
        api.SecretKey.key_gen("bls12_381", IKM, bytes(65536), KEY_DST, lib_path)
        assert False
    except BbsError as e:
        assert e.variant == "InvalidKeyInfoLength"
    # bn254 key_gen agrees with the oracle too
    sk2 = api.SecretKey.key_gen("bn254", bytes([1] * 32), b"", b"BBS-SIG-KEYGEN-SALT-", lib_path)
    assert sk2.sk == bbs.key_gen(bbs.BN_SUITE, bytes([1] * 32), b"", b"BBS-SIG-KEYGEN-SALT-")


def check_vectors_public_api(lib_path=None):          # test_vector.rs:163-192, :199-260
    sk = api.SecretKey.key_gen("bls12_381", IKM, KEY_INFO, KEY_DST, lib_path)
    pk = sk.sk_to_pk()
    assert bbs.g2_compress(C, pk.pk).hex() == (
        "a820f230f6ae38503b86c70dc50b61c58a77e45c39ab25c0652bbaa8fa136f2851bd4781c9dcde39fc9d1d52c9e60268"
        "061e7d7632171d91aa8d460acee0e96f1e7c4cfb12d3ff9ab5d5dc91c277db75c845d649ef3c4f63aebc364cd55ded0c")
    sig = sk.sign([M1], HEADER)
    assert bbs.g1_compress(C, sig.a).hex() + bbs.scalar_be(C, sig.e).hex() == SIG_HEX
    assert pk.verify(sig, HEADER, [M1]) is True
    proof = api.proof_gen(pk, sig, HEADER, PH, [M1], [0], bbs.mocked_calculate_random_scalars(S, 5))
    got = (bbs.g1_compress(C, proof.a_bar) + bbs.g1_compress(C, proof.b_bar) + bbs.g1_compress(C, proof.d)).hex()
    got += "".join(bbs.scalar_be(C, x).hex() for x in (proof.e_cap, proof.r1_cap, proof.r3_cap, proof.challenge))
    assert got == PROOF_HEX
    assert api.proof_verify(pk, proof, HEADER, PH, [M1], [0]) is True


CASES = [(0, [], b""), (0, [], b"abc"), (1, [0], b"abc"), (1, [], b"abc"), (10, [0, 1, 2], b""),
         (10, [0, 4, 7, 9], b"def"), (5, [0, 4], b"defghjsdjdbcjbejd"), (5, [0, 1, 2, 3, 4], b"def")]


def check_readme_example_bn254(lib_path=None):          # README.md:64-128, BASELINE.json configs[0]
    """The reference's README flow on BN254: key_gen from IKM [5; 32], 4 messages, sign -> verify -> proof_gen ->
    proof_verify, every output against the oracle's public interface."""
    suite = bbs.BN_SUITE
    msgs = [b"message1", b"message2", b"msg3", b"msg4"]
    sk = api.SecretKey.key_gen("bn254", bytes([5] * 32), b"", b"BBS-SIG-KEYGEN-SALT-", lib_path)
    assert sk.sk == bbs.key_gen(suite, bytes([5] * 32), b"", b"BBS-SIG-KEYGEN-SALT-")
    pk = sk.sk_to_pk()
    assert pk.pk == bbs.sk_to_pk(suite, sk.sk)
    sig = sk.sign(msgs, b"")
    want = bbs.sign(suite, sk.sk, msgs, b"")
    assert (sig.a, sig.e) == (want.a, want.e)
    assert pk.verify(sig, b"", msgs) is True
    assert pk.verify(sig, b"", msgs[:3] + [b"msg5"]) is False
    rnd = bbs.seeded_random_scalars(suite, b"readme", suite.api_id + b"MOCK_RANDOM_SCALARS_DST_", 5 + 2)
    proof = api.proof_gen(pk, sig, b"", b"ph", msgs, [0, 2], rnd)
    wantp = bbs.proof_gen(suite, pk.pk, want, b"", b"ph", msgs, [0, 2], rnd)
    assert (proof.a_bar, proof.b_bar, proof.d, proof.e_cap, proof.r1_cap, proof.r3_cap, list(proof.commitments), proof.challenge) == \
           (wantp.a_bar, wantp.b_bar, wantp.d, wantp.e_cap, wantp.r1_cap, wantp.r3_cap, list(wantp.commitments), wantp.challenge)
    assert api.proof_verify(pk, proof, b"", b"ph", [msgs[0], msgs[2]], [0, 2]) is True
    assert api.proof_verify(pk, proof, b"", b"ph", [msgs[0], msgs[1]], [0, 2]) is False
    assert bbs.proof_verify(suite, pk.pk, wantp, b"", b"ph", [msgs[0], msgs[2]], [0, 2]) is True


def check_round_trips(lib_path=None, cases=CASES, curve="bls12_381"):    # bbs_over_bls_tests.rs:41-84, bbs_over_bn tests
    rng = random.Random(9)
    sk = api.SecretKey.key_gen(curve, bytes([1] * 32), b"", b"BBS-SIG-KEYGEN-SALT-", lib_path)
    pk = sk.sk_to_pk()
    for count, disclosed, header in cases:
        msgs = [bytes(rng.randrange(256) for _ in range(5)) for _ in range(count)]
        sig = sk.sign(msgs, header)
        assert pk.verify(sig, header, msgs) is True
        proof = api.proof_gen(pk, sig, header, b"", msgs, disclosed)          # random scalars drawn inside
        assert api.proof_verify(pk, proof, header, b"", [msgs[i] for i in disclosed], disclosed) is True
        if count:
            assert pk.verify(sig, header, msgs[:-1] + [msgs[-1] + b"!"]) is False


def check_invalid_proofs(lib_path=None):               # bbs_over_bls_tests.rs:86-187
    rng = random.Random(10)
    sk = api.SecretKey.key_gen("bls12_381", bytes([1] * 32), b"", b"BBS-SIG-KEYGEN-SALT-", lib_path)
    pk = sk.sk_to_pk()
    msgs = [bytes(rng.randrange(256) for _ in range(5)) for _ in range(10)]
    sig = sk.sign(msgs, b"")
    proof = api.proof_gen(pk, sig, b"", b"", msgs, [0, 1, 5])
    dm = [msgs[0], msgs[1], msgs[5]]
    assert api.proof_verify(pk, proof, b"", b"", dm, [0, 1, 5]) is True
    forged = Proof(None, proof.b_bar, proof.d, proof.e_cap, proof.r1_cap, proof.r3_cap, list(proof.commitments), proof.challenge)
    assert api.proof_verify(pk, forged, b"", b"", dm, [0, 1, 5]) is False                     # case 1
    try:                                                                                        # case 2
        api.proof_verify(pk, Proof(), b"", b"", dm, [0, 1, 5])
        assert False, "expected Err"
    except BbsError as e:
        assert e.variant == "InvalidDisclosedIndex"
    assert api.proof_verify(api.PublicKey("bls12_381", None, lib_path), proof, b"", b"", dm, [0, 1, 5]) is False   # case 3
    assert api.proof_verify(pk, Proof(commitments=[0] * 7), b"", b"", dm, [0, 1, 5]) is False   # case 4
