"""Pins the plain-C restatement (oracle/c) against the KAT-pinned Python oracle and the reference's own
full signature / proof vectors (src/tests/test_vector.rs:163-260)."""
import random

from oracle import bbs, c_port
from oracle.bbs import BLS_SUITE as S
from oracle.curves import BLS12_381 as C

H = bytes.fromhex
SIG_HEX = "84773160b824e194073a57493dac1a20b667af70cd2352d8af241c77658da5253aa8458317cca0eae615690d55b1f27164657dcafee1d5c1973947aa70e2cfbb4c892340be5969920d0916067b4565a0"


def test_reference_vectors_through_c():
    ikm = H("746869732d49532d6a7573742d616e2d546573742d494b4d2d746f2d67656e65726174652d246528724074232d6b6579")
    key_info = H("746869732d49532d736f6d652d6b65792d6d657461646174612d746f2d62652d757365642d696e2d746573742d6b65792d67656e")
    key_dst = H("4242535f424c53313233383147315f584d443a5348412d3235365f535357555f524f5f4832475f484d32535f4b455947454e5f4453545f")
    m1 = H("9872ad089e452c7b6e283dfac2a80d58e8d0ff71cc4d5e310a1debdda4a45f02")
    header = H("11223344556677889900aabbccddeeff")
    ph = H("bed231d880675ed101ead304512e043ade9958dd0241ea70b4b3957fba941501")
    sk = bbs.key_gen(S, ikm, key_info, key_dst)
    pk = c_port.sk_to_pk(sk)
    assert pk == bbs.sk_to_pk(S, sk)
    gens = bbs.create_generators(S, 2, S.api_id)
    msgs = bbs.msg_to_scalars(S, [m1], S.api_id)
    sig = c_port.core_sign(sk, gens, header, msgs, S.api_id)
    assert bbs.g1_compress(C, sig.a).hex() + bbs.scalar_be(C, sig.e).hex() == SIG_HEX
    assert c_port.core_verify(pk, sig, gens, header, msgs, S.api_id) is True
    assert c_port.core_verify(pk, sig, gens, header + b"x", msgs, S.api_id) is False
    rnd = bbs.mocked_calculate_random_scalars(S, 5)
    proof = c_port.core_proof_gen(pk, sig, header, gens, ph, msgs, [0], S.api_id, rnd)
    want = bbs.core_proof_gen(S, pk, sig, header, gens, ph, msgs, [0], S.api_id, rnd)
    assert proof == want
    assert c_port.core_proof_verify(pk, proof, gens, header, ph, msgs, [0], S.api_id) is True
    assert c_port.core_proof_verify(pk, proof, gens, header, ph + b"x", msgs, [0], S.api_id) is False


def test_c_matches_python_oracle_random():
    rng = random.Random(4)
    L = 6
    gens = bbs.create_generators(S, L + 1, S.api_id)
    sk = rng.randrange(1, C.r)
    pk = bbs.sk_to_pk(S, sk)
    for it in range(3):
        msgs = [rng.randrange(C.r) for _ in range(L)]
        hdr = bytes(rng.randrange(256) for _ in range(rng.choice([0, 7, 70])))
        ph = bytes(rng.randrange(256) for _ in range(rng.choice([0, 33])))
        disclosed = sorted(rng.sample(range(L), rng.randrange(0, L + 1)))
        rnd = [rng.randrange(1, C.r) for _ in range(5 + L - len(disclosed))]
        sig = c_port.core_sign(sk, gens, hdr, msgs, S.api_id)
        psig = bbs.core_sign(S, sk, gens, hdr, msgs, S.api_id)
        assert (sig.a, sig.e) == (psig.a, psig.e)
        proof = c_port.core_proof_gen(pk, sig, hdr, gens, ph, msgs, disclosed, S.api_id, rnd)
        assert proof == bbs.core_proof_gen(S, pk, psig, hdr, gens, ph, msgs, disclosed, S.api_id, rnd)
        dm = [msgs[i] for i in disclosed]
        assert c_port.core_proof_verify(pk, proof, gens, hdr, ph, dm, disclosed, S.api_id) is True
        bad = bbs.Proof(proof.a_bar, proof.b_bar, proof.d, (proof.e_cap + 1) % C.r, proof.r1_cap, proof.r3_cap, proof.commitments, proof.challenge)
        assert c_port.core_proof_verify(pk, bad, gens, hdr, ph, dm, disclosed, S.api_id) is False
        forged = bbs.Proof(None, proof.b_bar, proof.d, proof.e_cap, proof.r1_cap, proof.r3_cap, proof.commitments, proof.challenge)
        want = bbs.core_proof_verify(S, pk, forged, gens, hdr, ph, dm, disclosed, S.api_id)
        assert c_port.core_proof_verify(pk, forged, gens, hdr, ph, dm, disclosed, S.api_id) == want


def test_bn254_c_matches_python_oracle():
    """The BN254 build of the C restatement (-DORC_BN254: D-type twist, 6x+2 loop with the two Frobenius lines,
    Devegili-Scott-Dahab hard part, little-endian encodings) against the Python oracle: key, signature, proof bytes,
    booleans incl. a forged A = identity and a tampered scalar; the reference's one BN254 constant P1
    (src/constants.rs:39-51) is embedded in the C build and exercised through core_sign."""
    from oracle.bbs import BN_SUITE as SB
    from oracle.curves import BN254 as CB
    P = c_port.port("bn254")
    rng = random.Random(9)
    L = 5
    gens = bbs.synthetic_generators(SB, L + 1)
    sk = rng.randrange(1, CB.r)
    pk = P.sk_to_pk(sk)
    assert pk == bbs.sk_to_pk(SB, sk)
    for it in range(3):
        msgs = [rng.randrange(CB.r) for _ in range(L)]
        hdr = bytes(rng.randrange(256) for _ in range(rng.choice([0, 7, 70])))
        ph = bytes(rng.randrange(256) for _ in range(rng.choice([0, 33])))
        disclosed = sorted(rng.sample(range(L), rng.randrange(0, L + 1)))
        rnd = [rng.randrange(1, CB.r) for _ in range(5 + L - len(disclosed))]
        sig = P.core_sign(sk, gens, hdr, msgs, SB.api_id)
        psig = bbs.core_sign(SB, sk, gens, hdr, msgs, SB.api_id)
        assert (sig.a, sig.e) == (psig.a, psig.e)
        assert P.core_verify(pk, sig, gens, hdr, msgs, SB.api_id) is True
        assert P.core_verify(pk, sig, gens, hdr + b"x", msgs, SB.api_id) is False
        assert P.core_verify(pk, bbs.Signature(None, sig.e), gens, hdr, msgs, SB.api_id) is False
        proof = P.core_proof_gen(pk, sig, hdr, gens, ph, msgs, disclosed, SB.api_id, rnd)
        assert proof == bbs.core_proof_gen(SB, pk, psig, hdr, gens, ph, msgs, disclosed, SB.api_id, rnd)
        dm = [msgs[i] for i in disclosed]
        assert P.core_proof_verify(pk, proof, gens, hdr, ph, dm, disclosed, SB.api_id) is True
        bad = bbs.Proof(proof.a_bar, proof.b_bar, proof.d, (proof.e_cap + 1) % CB.r, proof.r1_cap, proof.r3_cap, proof.commitments, proof.challenge)
        assert P.core_proof_verify(pk, bad, gens, hdr, ph, dm, disclosed, SB.api_id) is False
        forged = bbs.Proof(None, proof.b_bar, proof.d, proof.e_cap, proof.r1_cap, proof.r3_cap, proof.commitments, proof.challenge)
        want = bbs.core_proof_verify(SB, pk, forged, gens, hdr, ph, dm, disclosed, SB.api_id)
        assert P.core_proof_verify(pk, forged, gens, hdr, ph, dm, disclosed, SB.api_id) == want
    # a proof of a forged signature: the challenge matches, only the pairing product decides (-> false)
    fsig = bbs.Signature(CB.g1_add(sig.a, CB.g1), sig.e)
    fp = P.core_proof_gen(pk, fsig, hdr, gens, ph, msgs, disclosed, SB.api_id, rnd)
    assert P.core_proof_verify(pk, fp, gens, hdr, ph, dm, disclosed, SB.api_id) is False


def test_config1_readme_example_bn254():
    """BASELINE configs[0] (README.md:64-81): BN254, 4 messages, IKM [5u8;32]: sign + verify through the C restatement."""
    from oracle import c_baseline
    r = c_baseline.config1(c_port.port("bn254"))
    assert r["verified"] and r["matches_python_oracle"]


def test_plain_msm_matches_python_oracle():
    """orc_g1_msm_plain (the checker of the device's multi-tile bucket MSM) against the Python oracle's sum, both curves,
    with the bucket method's edge cases: identity, equal points, P and -P, scalars 0 / 1 / r - 1."""
    from oracle.curves import BN254 as CB
    for curve, c in (("bls12_381", C), ("bn254", CB)):
        P = c_port.port(curve)
        rng = random.Random(17)
        pts = [c.g1_mul(c.g1, rng.randrange(1, c.r)) for _ in range(12)]
        sc = [rng.randrange(c.r) for _ in range(12)]
        sc[0], sc[1], sc[2] = 0, 1, c.r - 1
        pts[3] = None
        pts[5], sc[5] = pts[4], sc[4]
        pts[7], sc[7] = c.g1_neg(pts[6]), sc[6]
        want = None
        for p_, k in zip(pts, sc):
            want = c.g1_add(want, c.g1_mul(p_, k))
        assert P.g1_msm_plain(pts, sc) == want
        assert P.g1_msm_plain([pts[6], pts[7]], [5, 5]) is None
        assert P.g1_msm_plain([], []) is None


def test_bn254_svdw_sign_of_c3_three_implementations():
    """BN254 hash-to-G1 (crate bn254_hash2curve 0.1.2, not vendored; restated as RFC 9380 + Shallue-van de Woestijne).
    The map's constant c3 enters only through x1 = c2 - tv4, x2 = c2 + tv4: flipping its sign swaps the two candidates, which
    changes the result exactly when BOTH are on the curve (the RFC then takes x1).  The reference's one BN254 known answer
    (P1, constants.rs:39-51) has the masks (x2 only, x1 only) -- it cannot see the sign.  Here: inputs for which both
    candidates are squares in one or both of the two maps, through three separately written implementations -- the Python
    oracle, the C oracle (constants derived from Z in C) and the product's host code (host_h2c.hpp, through the test
    build) -- which must agree, and which all follow the RFC's sgn0(c3) = 0."""
    import ctypes
    from bbs_sign_amd import _lib, build
    from oracle.bbs import BN_SUITE
    from oracle.curves import BN254 as CB
    from oracle.hashing import SVDW_C3, expand_message, hash_to_g1_bn, i2osp
    assert SVDW_C3 % 2 == 0                                         # sgn0(c3) = 0 (RFC 9380 6.6.1)
    P = c_port.port("bn254")
    lib = _lib.load_library(build.build(twin=True, verbose=False))
    api = BN_SUITE.api_id
    dst = api + b"SIG_GENERATOR_DST_"
    # the known answer and what it exercises
    v = expand_message(api + b"BP_MESSAGE_GENERATOR_SEED", api + b"SIG_GENERATOR_SEED_", 48)
    v = expand_message(v + i2osp(1, 8), api + b"SIG_GENERATOR_SEED_", 48)
    pt, masks = P.bn_hash_to_g1(v, dst)
    assert pt == BN_SUITE.p1 and masks == (2, 1), masks              # only x2 / only x1 on the curve: c3's sign is not seen
    seen = {"first": 0, "second": 0, "both": 0}
    for i in range(64):
        msg = b"bn254-svdw-both-candidates-%d" % i
        pt, (m0, m1) = P.bn_hash_to_g1(msg, dst)
        b0, b1 = (m0 & 3) == 3, (m1 & 3) == 3
        if not (b0 or b1):
            continue
        seen["both" if (b0 and b1) else ("first" if b0 else "second")] += 1
        assert m0 in (1, 2, 4, 7) and m1 in (1, 2, 4, 7)              # g(x1) g(x2) g(x3) is a square: one or all three
        assert pt == hash_to_g1_bn(msg, dst), i
        out = (ctypes.c_uint8 * 64)()
        mb = (ctypes.c_uint8 * len(msg)).from_buffer_copy(msg)
        db = (ctypes.c_uint8 * len(dst)).from_buffer_copy(dst)
        assert lib.bbs_hash_to_g1(1, mb, len(msg), db, len(dst), out) == 0
        o = bytes(out)
        assert (int.from_bytes(o[:32], "little"), int.from_bytes(o[32:], "little")) == pt, i
        assert CB.g1_is_on_curve(pt)
    assert seen["first"] and seen["second"] and seen["both"], seen
