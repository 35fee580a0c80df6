"""Pins the oracle against every known-answer vector the reference's own tests hold for the
path (all BLS12-381): /root/reference/src/tests/test_vector.rs:56-260.  The hex strings below
are test DATA transcribed from those assertions (inputs and expected outputs)."""

from oracle import bbs
from oracle.bbs import BLS_SUITE as S
from oracle.curves import BLS12_381 as C
from oracle.hashing import hash_to_scalar

H = bytes.fromhex
IKM = H("746869732d49532d6a7573742d616e2d546573742d494b4d2d746f2d67656e65726174652d246528724074232d6b6579")
KEY_INFO = H("746869732d49532d736f6d652d6b65792d6d657461646174612d746f2d62652d757365642d696e2d746573742d6b65792d67656e")
KEY_DST = H("4242535f424c53313233383147315f584d443a5348412d3235365f535357555f524f5f4832475f484d32535f4b455947454e5f4453545f")
M1 = H("9872ad089e452c7b6e283dfac2a80d58e8d0ff71cc4d5e310a1debdda4a45f02")
HEADER = H("11223344556677889900aabbccddeeff")
PH = H("bed231d880675ed101ead304512e043ade9958dd0241ea70b4b3957fba941501")


def sc(x):
    return bbs.scalar_be(C, x).hex()


def test_constants_bls():  # test_vector.rs:56-69
    assert bbs.g1_compress(C, C.g1).hex() == "97f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb"
    assert bbs.g2_compress(C, C.g2).hex() == "93e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8"
    assert C.g1_is_on_curve(S.p1)
    assert bbs.g1_compress(C, S.p1).hex() == "a8ce256102840821a3e94ea9025e4662b205762f9776b3a766c872b948f1fd225e7c59698588e70d11406d161b4e28c9"


def test_hash_to_scalar():  # test_vector.rs:72-83
    dst = H("4242535f424c53313233383147315f584d443a5348412d3235365f535357555f524f5f4832475f484d32535f4832535f")
    assert sc(hash_to_scalar(C, M1, dst)) == "0f90cbee27beb214e6545becb8404640d3612da5d6758dffeccd77ed7169807c"


def test_mocked_random_scalars():  # test_vector.rs:86-97
    s = bbs.mocked_calculate_random_scalars(S, 10)
    assert sc(s[0]) == "04f8e2518993c4383957ad14eb13a023c4ad0c67d01ec86eeb902e732ed6df3f"
    assert sc(s[9]) == "485e2adab17b76f5334c95bf36c03ccf91cef77dcfcdc6b8a69e2090b3156663"


def test_msg_to_scalars():  # test_vector.rs:100-120
    dst = H("4242535f424c53313233383147315f584d443a5348412d3235365f535357555f524f5f4832475f484d32535f4d41505f4d53475f544f5f5343414c41525f41535f484153485f")
    assert dst == S.api_id + b"MAP_MSG_TO_SCALAR_AS_HASH_"
    assert sc(hash_to_scalar(C, M1, dst)) == "1cb5bb86114b34dc438a911617655a1db595abafac92f47c5001799cf624b430"
    assert sc(hash_to_scalar(C, b"", dst)) == "08e3afeb2b4f2b5f907924ef42856616e6f2d5f1fb373736db1cca32707a7d16"
    assert [sc(x) for x in bbs.msg_to_scalars(S, [M1, b""], S.api_id)] == [
        "1cb5bb86114b34dc438a911617655a1db595abafac92f47c5001799cf624b430",
        "08e3afeb2b4f2b5f907924ef42856616e6f2d5f1fb373736db1cca32707a7d16"]


def test_create_generators():  # test_vector.rs:123-136
    g = bbs.create_generators(S, 11, b"BBS_BLS12381G1_XMD:SHA-256_SSWU_RO_H2G_HM2S_")
    want = {
        0: "a9ec65b70a7fbe40c874c9eb041c2cb0a7af36ccec1bea48fa2ba4c2eb67ef7f9ecb17ed27d38d27cdeddff44c8137be",
        1: "98cd5313283aaf5db1b3ba8611fe6070d19e605de4078c38df36019fbaad0bd28dd090fd24ed27f7f4d22d5ff5dea7d4",
        2: "a31fbe20c5c135bcaa8d9fc4e4ac665cc6db0226f35e737507e803044093f37697a9d452490a970eea6f9ad6c3dcaa3a",
        10: "a1f229540474f4d6f1134761b92b788128c7ac8dc9b0c52d59493132679673032ac7db3fb3d79b46b13c1c41ee495bca",
    }
    for i, h in want.items():
        assert bbs.g1_compress(C, g[i]).hex() == h
        assert C.g1_mul(g[i], C.r) is None


def test_p1_is_first_bp_generator():  # test_vector.rs:15-19 (how P1 was produced)
    from oracle.hashing import expand_message, hash_to_g1_bls, i2osp
    api = S.api_id
    v = expand_message(api + b"BP_MESSAGE_GENERATOR_SEED", api + b"SIG_GENERATOR_SEED_", 48)
    v = expand_message(v + i2osp(1, 8), api + b"SIG_GENERATOR_SEED_", 48)
    assert hash_to_g1_bls(v, api + b"SIG_GENERATOR_DST_") == S.p1


def test_keygen():  # test_vector.rs:139-160
    sk = bbs.key_gen(S, IKM, KEY_INFO, KEY_DST)
    assert sc(sk) == "60e55110f76883a13d030b2f6bd11883422d5abde717569fc0731f51237169fc"
    pk = bbs.sk_to_pk(S, sk)
    assert bbs.g2_compress(C, pk).hex() == "a820f230f6ae38503b86c70dc50b61c58a77e45c39ab25c0652bbaa8fa136f2851bd4781c9dcde39fc9d1d52c9e60268061e7d7632171d91aa8d460acee0e96f1e7c4cfb12d3ff9ab5d5dc91c277db75c845d649ef3c4f63aebc364cd55ded0c"


SIG_HEX = "84773160b824e194073a57493dac1a20b667af70cd2352d8af241c77658da5253aa8458317cca0eae615690d55b1f27164657dcafee1d5c1973947aa70e2cfbb4c892340be5969920d0916067b4565a0"
PROOF_HEX = "94916292a7a6bade28456c601d3af33fcf39278d6594b467e128a3f83686a104ef2b2fcf72df0215eeaf69262ffe8194a19fab31a82ddbe06908985abc4c9825788b8a1610942d12b7f5debbea8985296361206dbace7af0cc834c80f33e0aadaeea5597befbb651827b5eed5a66f1a959bb46cfd5ca1a817a14475960f69b32c54db7587b5ee3ab665fbd37b506830a49f21d592f5e634f47cee05a025a2f8f94e73a6c15f02301d1178a92873b6e8634bafe4983c3e15a663d64080678dbf29417519b78af042be2b3e1c4d08b8d520ffab008cbaaca5671a15b22c239b38e940cfeaa5e72104576a9ec4a6fad78c532381aeaa6fb56409cef56ee5c140d455feeb04426193c57086c9b6d397d9418"


def test_sign_and_proof_vectors():  # test_vector.rs:163-192 and :199-260
    sk = bbs.key_gen(S, IKM, KEY_INFO, KEY_DST)
    pk = bbs.sk_to_pk(S, sk)
    sig = bbs.sign(S, sk, [M1], HEADER)
    assert bbs.g1_compress(C, sig.a).hex() + sc(sig.e) == SIG_HEX
    assert bbs.verify(S, pk, sig, HEADER, [M1]) is True

    rnd = bbs.mocked_calculate_random_scalars(S, 5 + 1 - 1)
    proof = bbs.proof_gen(S, pk, sig, HEADER, PH, [M1], [0], rnd)
    got = (bbs.g1_compress(C, proof.a_bar) + bbs.g1_compress(C, proof.b_bar) + bbs.g1_compress(C, proof.d)).hex()
    got += sc(proof.e_cap) + sc(proof.r1_cap) + sc(proof.r3_cap) + sc(proof.challenge)
    assert proof.commitments == []
    assert got == PROOF_HEX
    assert bbs.proof_verify(S, pk, proof, HEADER, PH, [M1], [0]) is True
    assert bbs.proof_verify(S, pk, proof, HEADER, PH + b"x", [M1], [0]) is False


def test_bn254_p1_is_first_bp_generator():
    """The reference's one BN254 known answer: P1 (constants.rs:39-51) is the first generator under the seed
    '...BP_MESSAGE_GENERATOR_SEED' (test_vector.rs:21-25, constants.rs comment).  Pins the restated SvdW
    hash-to-G1 (crate bn254_hash2curve 0.1.2 is not vendored)."""
    from oracle.bbs import BN_SUITE
    from oracle.hashing import expand_message, hash_to_g1_bn, i2osp
    api = BN_SUITE.api_id
    v = expand_message(api + b"BP_MESSAGE_GENERATOR_SEED", api + b"SIG_GENERATOR_SEED_", 48)
    v = expand_message(v + i2osp(1, 8), api + b"SIG_GENERATOR_SEED_", 48)
    assert hash_to_g1_bn(v, api + b"SIG_GENERATOR_DST_") == BN_SUITE.p1
    gens = bbs.create_generators(BN_SUITE, 3, api)
    assert all(BN_SUITE.curve.g1_is_on_curve(g) for g in gens) and len(set(gens)) == 3
