"""Wire codec (host side of the product library: no GPU needed): the octet strings of the reference's
vectors (src/tests/test_vector.rs:139-260) decode to the oracle's structures and re-encode to the same
bytes; malformed encodings are rejected."""
import os
import sys

import pytest

from bbs_sign_amd import BbsError, api
from oracle import bbs
from oracle.bbs import BLS_SUITE as S, BN_SUITE
from oracle.curves import BLS12_381 as C, BN254

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SIG_HEX = "84773160b824e194073a57493dac1a20b667af70cd2352d8af241c77658da5253aa8458317cca0eae615690d55b1f27164657dcafee1d5c1973947aa70e2cfbb4c892340be5969920d0916067b4565a0"
PROOF_HEX = "94916292a7a6bade28456c601d3af33fcf39278d6594b467e128a3f83686a104ef2b2fcf72df0215eeaf69262ffe8194a19fab31a82ddbe06908985abc4c9825788b8a1610942d12b7f5debbea8985296361206dbace7af0cc834c80f33e0aadaeea5597befbb651827b5eed5a66f1a959bb46cfd5ca1a817a14475960f69b32c54db7587b5ee3ab665fbd37b506830a49f21d592f5e634f47cee05a025a2f8f94e73a6c15f02301d1178a92873b6e8634bafe4983c3e15a663d64080678dbf29417519b78af042be2b3e1c4d08b8d520ffab008cbaaca5671a15b22c239b38e940cfeaa5e72104576a9ec4a6fad78c532381aeaa6fb56409cef56ee5c140d455feeb04426193c57086c9b6d397d9418"
PK_HEX = "a820f230f6ae38503b86c70dc50b61c58a77e45c39ab25c0652bbaa8fa136f2851bd4781c9dcde39fc9d1d52c9e60268061e7d7632171d91aa8d460acee0e96f1e7c4cfb12d3ff9ab5d5dc91c277db75c845d649ef3c4f63aebc364cd55ded0c"


@pytest.fixture(scope="session")
def lib():
    sys.path.insert(0, ROOT)
    from bbs_sign_amd import build as b
    return b.build(twin=False, verbose=False)


def test_vectors_round_trip(lib):
    sig = api.octets_to_signature("bls12_381", bytes.fromhex(SIG_HEX), lib)
    assert bbs.g1_compress(C, sig.a).hex() + bbs.scalar_be(C, sig.e).hex() == SIG_HEX
    assert C.g1_is_on_curve(sig.a)
    assert api.signature_to_octets("bls12_381", sig, lib).hex() == SIG_HEX
    proof = api.octets_to_proof("bls12_381", bytes.fromhex(PROOF_HEX), lib)
    assert proof.commitments == [] and C.g1_is_on_curve(proof.a_bar) and C.g1_is_on_curve(proof.d)
    assert api.proof_to_octets("bls12_381", proof, lib).hex() == PROOF_HEX
    pk = api.octets_to_public_key("bls12_381", bytes.fromhex(PK_HEX), lib)
    assert bbs.g2_compress(C, pk.pk).hex() == PK_HEX
    assert api.public_key_to_octets(pk).hex() == PK_HEX


def test_proof_with_commitments_and_bn254(lib):
    import random
    rng = random.Random(3)
    for suite, name in ((S, "bls12_381"), (BN_SUITE, "bn254")):
        c = suite.curve
        pts = [c.g1_mul(c.g1, rng.randrange(1, c.r)) for _ in range(3)]
        sc = [rng.randrange(c.r) for _ in range(4)]
        cms = [rng.randrange(c.r) for _ in range(5)]
        p = api.Proof(pts[0], pts[1], pts[2], sc[0], sc[1], sc[2], cms, sc[3])
        oct_ = api.proof_to_octets(name, p, lib)
        want = b"".join(bbs.g1_compress(c, q) for q in pts) + b"".join(bbs.scalar_be(c, x) for x in sc[:3] + cms + sc[3:])
        assert oct_ == want
        back = api.octets_to_proof(name, oct_, lib)
        assert (back.a_bar, back.b_bar, back.d, back.e_cap, back.r1_cap, back.r3_cap, back.commitments, back.challenge) == \
               (p.a_bar, p.b_bar, p.d, p.e_cap, p.r1_cap, p.r3_cap, p.commitments, p.challenge)
        pk = suite.curve.g2_mul(c.g2, rng.randrange(1, c.r))
        o = api.public_key_to_octets(api.PublicKey(name, pk, lib))
        assert o == bbs.g2_compress(c, pk)
        assert api.octets_to_public_key(name, o, lib).pk == pk


def test_malformed_octets_are_rejected(lib):
    good = bytearray.fromhex(SIG_HEX)
    cases = []
    b = bytearray(good); b[0] &= 0x7F; cases.append(bytes(b))                       # compression flag cleared
    b = bytearray(good); b[0] ^= 0x20; b[47] ^= 1; cases.append(bytes(b))          # x changed: (almost surely) not on the curve / subgroup
    cases.append(bytes([0xC0]) + bytes(47) + bytes(good[48:]))                      # A = identity
    cases.append(bytes(good[:48]) + bytes(32))                                      # e = 0
    cases.append(bytes(good[:48]) + C.r.to_bytes(32, "big"))                        # e = r: not canonical
    cases.append(bytes([0x9F]) + bytes([0xFF] * 47) + bytes(good[48:]))             # x >= p
    for o in cases:
        with pytest.raises(BbsError):
            api.octets_to_signature("bls12_381", o, lib)
    with pytest.raises(BbsError):
        api.octets_to_proof("bls12_381", bytes.fromhex(PROOF_HEX)[:-1], lib)
    # a point on the curve but outside the prime-order subgroup is rejected
    x = 1
    while True:
        y2 = (x ** 3 + 4) % C.p
        y = pow(y2, (C.p + 1) // 4, C.p)
        if y * y % C.p == y2 and C.g1_mul((x, y), C.r) is not None:
            break
        x += 1
    o = bytes(bbs.g1_compress(C, (x, y))) + bytes(good[48:])
    with pytest.raises(BbsError):
        api.octets_to_signature("bls12_381", o, lib)


# ---- batch decode (points decompressed and checked on the device) ------------------------------------------------
def _proof_octets(suite, pts, sc, cms):
    c = suite.curve
    return b"".join(bbs.g1_compress(c, q) for q in pts) + b"".join(bbs.scalar_be(c, x) for x in sc[:3] + cms + sc[3:])


def check_batch_decode(lib_path, curve_name):
    """bbs_proofs_from_octets_batch against bbs_proof_from_octets item by item: records, commitments and status codes,
    on valid proofs and on every kind of malformed one (bad flag, x >= p, not on the curve, on the curve but outside
    the prime-order subgroup incl. order 3, identity points, scalars >= r, bad lengths)."""
    import random
    from bbs_sign_amd import Engine
    suite = bbs.SUITES[curve_name]
    c = suite.curve
    rng = random.Random(17)
    fpb = c.fp_bytes
    items = []
    for k in range(10):
        pts = [c.g1_mul(c.g1, rng.randrange(1, c.r)) for _ in range(3)]
        sc = [rng.randrange(c.r) for _ in range(4)]
        cms = [rng.randrange(c.r) for _ in range(k % 4)]
        items.append(bytearray(_proof_octets(suite, pts, sc, cms)))
    bad = []
    b = bytearray(items[0]); b[fpb - 1 if curve_name == "bn254" else 0] ^= 0x20 if curve_name == "bls12_381" else 0x80; bad.append(b)  # other root: still valid
    if curve_name == "bls12_381":
        b = bytearray(items[1]); b[0] &= 0x7F; bad.append(b)                                   # not the compressed form
        b = bytearray(items[2]); b[fpb:2 * fpb] = bytes([0x9F]) + bytes([0xFF] * (fpb - 1)); bad.append(b)   # x >= p
        b = bytearray(items[3]); b[2 * fpb:3 * fpb] = bytes([0xC0]) + bytes(fpb - 1); bad.append(b)          # identity D
        b = bytearray(items[3]); b[0:fpb] = bytes([0xE0]) + bytes(fpb - 1); bad.append(b)                    # identity with sign bit
        b = bytearray(items[4]); b[0:fpb] = bbs.g1_compress(c, (0, 2)); bad.append(b)                         # order 3
        x = 7
        while True:
            y2 = (x ** 3 + 4) % c.p
            y = pow(y2, (c.p + 1) // 4, c.p)
            if y * y % c.p == y2 and c.g1_mul((x, y), c.r) is not None:
                break
            x += 1
        b = bytearray(items[5]); b[fpb:2 * fpb] = bbs.g1_compress(c, (x, y)); bad.append(b)                   # large order, not in G1
    else:
        b = bytearray(items[2]); b[fpb:2 * fpb] = bytes([0xFF] * (fpb - 1)) + bytes([0x3F]); bad.append(b)   # x >= p
        b = bytearray(items[3]); b[2 * fpb:3 * fpb] = bytes(fpb - 1) + bytes([0x40]); bad.append(b)          # identity D
    xx = 3
    while pow((xx ** 3 + (4 if curve_name == "bls12_381" else 3)) % c.p, (c.p - 1) // 2, c.p) == 1:
        xx += 1                                                                                 # x with no point on the curve
    enc = bytearray(bbs.g1_compress(c, c.g1))
    if curve_name == "bls12_381":
        enc = bytearray((xx).to_bytes(fpb, "big")); enc[0] |= 0x80
    else:
        enc = bytearray((xx).to_bytes(fpb, "little"))
    b = bytearray(items[6]); b[0:fpb] = enc; bad.append(b)                                                    # not on the curve
    b = bytearray(items[7]); b[3 * fpb:3 * fpb + 32] = c.r.to_bytes(32, "big"); bad.append(b)                # e^ = r
    b = bytearray(items[8]); b[-32:] = bytes([0xFF] * 32); bad.append(b)                                      # challenge >= r
    bad.append(items[9][:-1]); bad.append(items[9][:3 * fpb + 100]); bad.append(bytearray())                  # bad lengths
    allo = [bytes(x) for x in items + bad]
    eng = Engine(curve_name, lib_path=lib_path, window_bits=4)
    proofs, st = eng.proofs_from_octets_batch(allo)
    n_ok = 0
    for i, o in enumerate(allo):
        try:
            want = api.octets_to_proof(curve_name, o, lib_path)
            code = 1
        except BbsError as e:
            want, code = None, e.status
        assert int(st[i]) == code, (curve_name, i, int(st[i]), code)
        if want is not None:
            got = proofs[i]
            assert (got.a_bar, got.b_bar, got.d, got.e_cap, got.r1_cap, got.r3_cap, got.commitments, got.challenge) == \
                   (want.a_bar, want.b_bar, want.d, want.e_cap, want.r1_cap, want.r3_cap, want.commitments, want.challenge), i
            n_ok += 1
        else:
            assert proofs[i] is None
    assert n_ok >= 11
    ps, st = eng.proofs_from_octets_batch([])
    assert ps == [] and len(st) == 0
    # the decode stage as a primitive: random subgroup points (both roots), the identity, points outside the subgroup
    pts = [c.g1_mul(c.g1, rng.randrange(1, c.r)) for _ in range(40)] + [None]
    enc = [bytes(bbs.g1_compress(c, q)) for q in pts]
    dec, code = eng.g1_decompress_batch(enc)
    assert dec == pts and list(code) == [0] * 40 + [1]
    if curve_name == "bls12_381":
        h = 0x396c8c005555e1568c00aaab0000aaab
        outside = [(0, 2), (x, y), c.g1_mul((x, y), c.r), c.g1_mul((x, y), c.r * (h // 11)), c.g1_mul((x, y), 3), c.g1_add((x, y), c.g1)]
        outside = [q for q in outside if q is not None]
        dec, code = eng.g1_decompress_batch([bytes(bbs.g1_compress(c, q)) for q in outside])
        assert list(code) == [-41] * len(outside) and dec == [None] * len(outside) and len(outside) >= 4
    # signatures: the batch ingest against the per-item one, incl. identity A, e = 0, e >= r, a point outside G1
    sig_oct = [bytes(bbs.g1_compress(c, q)) + bbs.scalar_be(c, rng.randrange(1, c.r)) for q in pts[:6]]
    sig_oct.append(bytes(bbs.g1_compress(c, None)) + bbs.scalar_be(c, 5))
    sig_oct.append(bytes(bbs.g1_compress(c, pts[0])) + bytes(32))
    sig_oct.append(bytes(bbs.g1_compress(c, pts[1])) + c.r.to_bytes(32, "big"))
    if curve_name == "bls12_381":
        sig_oct.append(bytes(bbs.g1_compress(c, (0, 2))) + bbs.scalar_be(c, 9))
    sigs, st2 = eng.signatures_from_octets_batch(sig_oct)
    for i, o in enumerate(sig_oct):
        try:
            want = api.octets_to_signature(curve_name, o, lib_path)
            assert int(st2[i]) == 1 and (sigs[i].a, sigs[i].e) == (want.a, want.e), i
        except BbsError as e:
            assert int(st2[i]) == e.status and sigs[i] is None, (i, int(st2[i]), e.status)
    # and back: the batch encoder gives the per-item encoder's bytes
    good = [pr for pr in proofs if pr is not None]
    assert eng.proofs_to_octets_batch(good) == [api.proof_to_octets(curve_name, pr, lib_path) for pr in good]
    assert eng.proofs_to_octets_batch([]) == []
    eng.close()


@pytest.fixture(scope="session")
def twin():
    sys.path.insert(0, ROOT)
    from bbs_sign_amd import build as b
    return b.build(twin=True, verbose=False)


@pytest.mark.parametrize("curve_name", ["bls12_381", "bn254"])
def test_batch_decode_twin(twin, curve_name):
    check_batch_decode(twin, curve_name)


@pytest.mark.gpu
@pytest.mark.parametrize("curve_name", ["bls12_381", "bn254"])
def test_batch_decode_gpu(curve_name):
    check_batch_decode(None, curve_name)
