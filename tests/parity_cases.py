"""Parity checks of the C-ABI engine against the oracle and the committed golden fixtures.

The same bodies run (a) on the GPU through the product library (tests/test_parity_gpu.py,
``-m gpu``) and (b) on the CPU through the TEST-ONLY host twin of the same stage code
(tests/test_parity_hosttwin.py) -- (b) checks the host logic (validation order, packing, index
bookkeeping) in a container without a GPU; only (a) is a parity claim about the product.

Structure follows the reference's tests: known-answer vectors (src/tests/test_vector.rs),
round trips (src/tests/bbs_over_bls_tests.rs:41-84, core_sign_tests.rs:38-65), negative cases
(bbs_over_bls_tests.rs:86-187, core_sign_tests.rs:67-155, sign_verify_tests.rs:60-105).
"""
import json
import os
import random

import numpy as np

from bbs_sign_amd import BbsError, Engine, Proof, Signature
from bbs_sign_amd import _lib
from bbs_sign_amd.engine import _ragged_bytes as _engine_ragged
from oracle import bbs
from oracle.hashing import expand_message, i2osp

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden", "bbs_golden.json")
_golden = None


def golden():
    global _golden
    if _golden is None:
        with open(GOLDEN) as f:
            _golden = json.load(f)
    return _golden


def _i(h):
    return int(h, 16)


def _pt(p):
    return None if p is None else (_i(p[0]), _i(p[1]))


def _pt2(q):
    return None if q is None else ((_i(q[0][0]), _i(q[0][1])), (_i(q[1][0]), _i(q[1][1])))


# Form of the jobs the cases create (bbs_ctx_set_latency_mode): None = the library's default (AUTO: a job that is alone on its
# context gets the latency form), False / True = throughput / latency form for every job.  The test modules pin it so that
# a case exercises one known form; bench.py leaves it at None.
LATENCY_MODE = None


def make_engine(curve, gens, api_id, lib_path=None, sk=None, pk="unset", window_bits=None, device=0):
    if window_bits is None:
        window_bits = 4 if lib_path else 8
    eng = Engine(curve, device=device, lib_path=lib_path, window_bits=window_bits)
    if LATENCY_MODE is not None:
        eng.set_latency_mode(LATENCY_MODE)
    eng.set_generators(gens, api_id)
    if sk is not None:
        eng.set_secret_key(sk)
    if pk != "unset":
        eng.set_public_key(pk)
    return eng


def gens_for(suite, count):
    if suite.curve.name == "bls12_381":
        return bbs.create_generators(suite, count, suite.api_id)
    return bbs.synthetic_generators(suite, count)


def proof_eq(a, b):
    return (a.a_bar == b.a_bar and a.b_bar == b.b_bar and a.d == b.d and a.e_cap == b.e_cap
            and a.r1_cap == b.r1_cap and a.r3_cap == b.r3_cap and list(a.commitments) == list(b.commitments)
            and a.challenge == b.challenge)


def to_engine_proof(p):
    return Proof(p.a_bar, p.b_bar, p.d, p.e_cap, p.r1_cap, p.r3_cap, list(p.commitments), p.challenge)


# ------------------------------------------------------------------------------------------------
def check_kat_vectors(lib_path=None):
    """The reference's full signature and proof vectors (test_vector.rs:163-192, :199-260) through
    the engine's core_* entry points."""
    S = bbs.BLS_SUITE
    c = S.curve
    H = bytes.fromhex
    ikm = H("746869732d49532d6a7573742d616e2d546573742d494b4d2d746f2d67656e65726174652d246528724074232d6b6579")
    key_info = H("746869732d49532d736f6d652d6b65792d6d657461646174612d746f2d62652d757365642d696e2d746573742d6b65792d67656e")
    key_dst = H("4242535f424c53313233383147315f584d443a5348412d3235365f535357555f524f5f4832475f484d32535f4b455947454e5f4453545f")
    m1 = H("9872ad089e452c7b6e283dfac2a80d58e8d0ff71cc4d5e310a1debdda4a45f02")
    header = H("11223344556677889900aabbccddeeff")
    ph = H("bed231d880675ed101ead304512e043ade9958dd0241ea70b4b3957fba941501")
    sk = bbs.key_gen(S, ikm, key_info, key_dst)
    gens = bbs.create_generators(S, 2, S.api_id)
    eng = make_engine("bls12_381", gens, S.api_id, lib_path, sk=sk)
    assert eng.public_key_compressed().hex() == (
        "a820f230f6ae38503b86c70dc50b61c58a77e45c39ab25c0652bbaa8fa136f2851bd4781c9dcde39fc9d1d52c9e60268"
        "061e7d7632171d91aa8d460acee0e96f1e7c4cfb12d3ff9ab5d5dc91c277db75c845d649ef3c4f63aebc364cd55ded0c")
    # msg_to_scalars on the device (test_vector.rs:100-120)
    msg = eng.hash_to_scalar_batch([m1, b""], S.api_id + b"MAP_MSG_TO_SCALAR_AS_HASH_")
    assert bbs.scalar_be(c, msg[0]).hex() == "1cb5bb86114b34dc438a911617655a1db595abafac92f47c5001799cf624b430"
    assert bbs.scalar_be(c, msg[1]).hex() == "08e3afeb2b4f2b5f907924ef42856616e6f2d5f1fb373736db1cca32707a7d16"
    sig = eng.core_sign(header, [msg[0]])
    got = bbs.g1_compress(c, sig.a).hex() + bbs.scalar_be(c, sig.e).hex()
    assert got == ("84773160b824e194073a57493dac1a20b667af70cd2352d8af241c77658da5253aa8458317cca0eae615690d55b1f271"
                   "64657dcafee1d5c1973947aa70e2cfbb4c892340be5969920d0916067b4565a0")
    assert eng.core_verify(sig, header, [msg[0]]) is True
    rnd = bbs.mocked_calculate_random_scalars(S, 5)
    proof = eng.core_proof_gen(sig, header, ph, [msg[0]], [0], rnd)
    got = (bbs.g1_compress(c, proof.a_bar) + bbs.g1_compress(c, proof.b_bar) + bbs.g1_compress(c, proof.d)).hex()
    got += "".join(bbs.scalar_be(c, x).hex() for x in (proof.e_cap, proof.r1_cap, proof.r3_cap, proof.challenge))
    assert proof.commitments == []
    assert got == (
        "94916292a7a6bade28456c601d3af33fcf39278d6594b467e128a3f83686a104ef2b2fcf72df0215eeaf69262ffe8194a19fab31a82ddbe06908985abc4c9825788b8a1610942d12b7f5debbea8985296361206dbace7af0cc834c80f33e0aadaeea5597befbb651827b5eed5a66f1a959bb46cfd5ca1a817a14475960f69b32c54db7587b5ee3ab665fbd37b506830a49f21d592f5e634f47cee05a025a2f8f94e73a6c15f02301d1178a92873b6e8634bafe4983c3e15a663d64080678dbf29417519b78af042be2b3e1c4d08b8d520ffab008cbaaca5671a15b22c239b38e940cfeaa5e72104576a9ec4a6fad78c532381aeaa6fb56409cef56ee5c140d455feeb04426193c57086c9b6d397d9418")
    assert eng.core_proof_verify(proof, header, ph, [msg[0]], [0]) is True
    assert eng.core_proof_verify(proof, header, ph + b"x", [msg[0]], [0]) is False
    # the same two vectors as octet strings straight off the device (bbs_sign_octets_*, bbs_proof_gen_octets_*)
    so, st = eng.sign_octets_batch([[msg[0]]], [header])
    assert list(st) == [1] and so[0].hex() == ("84773160b824e194073a57493dac1a20b667af70cd2352d8af241c77658da5253aa8458317cca0eae615690d55b1f271"
                                               "64657dcafee1d5c1973947aa70e2cfbb4c892340be5969920d0916067b4565a0")
    po, st = eng.proof_gen_octets_batch([sig], [[msg[0]]], [[0]], [rnd], [header], [ph])
    assert list(st) == [1] and po[0].hex() == got


# ------------------------------------------------------------------------------------------------
def check_golden(curve, lib_path=None, max_L=None):
    """Golden fixtures (tests/golden): sign / verify / proof_gen / proof_verify, byte for byte."""
    g = golden()["suites"][curve]
    suite = bbs.SUITES[curve]
    api_id = bytes.fromhex(g["api_id"])
    sk = _i(g["sk"])
    n = 0
    for case in g["cases"]:
        L = case["L"]
        if max_L is not None and L > max_L:
            continue
        gens = [_pt(p) for p in case["generators"]]
        eng = make_engine(curve, gens, api_id, lib_path, sk=sk)
        assert eng.public_key() == _pt2(g["pk"])
        assert eng.public_key_compressed().hex() == g["pk_compressed"]
        msgs = [_i(m) for m in case["messages"]]
        header, ph = bytes.fromhex(case["header"]), bytes.fromhex(case["ph"])
        disclosed = case["disclosed"]
        rnd = [_i(x) for x in case["random_scalars"]]
        sig = eng.core_sign(header, msgs)
        assert sig.a == _pt(case["signature"]["a"]) and sig.e == _i(case["signature"]["e"]), (curve, L, "sign")
        assert eng.core_verify(sig, header, msgs) is True
        proof = eng.core_proof_gen(sig, header, ph, msgs, disclosed, rnd)
        gp = case["proof"]
        want = Proof(_pt(gp["a_bar"]), _pt(gp["b_bar"]), _pt(gp["d"]), _i(gp["e_cap"]), _i(gp["r1_cap"]), _i(gp["r3_cap"]),
                     [_i(x) for x in gp["commitments"]], _i(gp["challenge"]))
        assert proof_eq(proof, want), (curve, L, "proof_gen")
        assert eng.core_proof_verify(proof, header, ph, [msgs[i] for i in disclosed], disclosed) is True
        # one flipped bit anywhere must turn the boolean
        assert eng.core_proof_verify(proof, header + b"!", ph, [msgs[i] for i in disclosed], disclosed) is False
        eng.close()
        n += 1
    assert n > 0


def check_oracle_reproduces_golden(curve, max_L=10):
    g = golden()["suites"][curve]
    suite = bbs.SUITES[curve]
    api_id = bytes.fromhex(g["api_id"])
    sk = _i(g["sk"])
    pk = bbs.sk_to_pk(suite, sk)
    assert bbs.g2_compress(suite.curve, pk).hex() == g["pk_compressed"]
    for case in g["cases"]:
        if case["L"] > max_L:
            continue
        gens = [_pt(p) for p in case["generators"]]
        if curve == "bls12_381":
            assert gens == bbs.create_generators(suite, case["L"] + 1, api_id)
        msgs = [_i(m) for m in case["messages"]]
        header, ph = bytes.fromhex(case["header"]), bytes.fromhex(case["ph"])
        sig = bbs.core_sign(suite, sk, gens, header, msgs, api_id)
        assert sig.a == _pt(case["signature"]["a"]) and sig.e == _i(case["signature"]["e"])
        assert bbs.g1_compress(suite.curve, sig.a).hex() == case["signature"]["a_compressed"]
        rnd = [_i(x) for x in case["random_scalars"]]
        proof = bbs.core_proof_gen(suite, pk, sig, header, gens, ph, msgs, case["disclosed"], api_id, rnd)
        assert proof.challenge == _i(case["proof"]["challenge"])
        assert proof.commitments == [_i(x) for x in case["proof"]["commitments"]]


# ------------------------------------------------------------------------------------------------
def check_random_batch(curve, lib_path=None, n=6, L=5, seed=1):
    """Seeded random batch with ragged headers / disclosed sets, every output against the oracle."""
    rng = random.Random(seed)
    suite = bbs.SUITES[curve]
    c = suite.curve
    api_id = b"api-%d-" % seed if seed % 2 else suite.api_id     # core_* takes an arbitrary api_id
    gens = gens_for(suite, L + 1)
    sk = rng.randrange(1, c.r)
    pk = bbs.sk_to_pk(suite, sk)
    eng = make_engine(curve, gens, api_id, lib_path, sk=sk)
    assert eng.public_key() == pk
    msgs = [[rng.randrange(c.r) for _ in range(L)] for _ in range(n)]
    headers = [bytes(rng.randrange(256) for _ in range(rng.choice([0, 0, 3, 17, 64, 70]))) for _ in range(n)]
    phs = [bytes(rng.randrange(256) for _ in range(rng.choice([0, 5, 32, 100]))) for _ in range(n)]
    disclosed = [sorted(rng.sample(range(L), rng.randrange(0, L + 1))) for _ in range(n)]
    rnds = [[rng.randrange(1, c.r) for _ in range(5 + L - len(d))] for d in disclosed]
    sigs, st = eng.core_sign_batch(msgs, headers)
    assert list(st) == [1] * n
    want_sigs = [bbs.core_sign(suite, sk, gens, headers[i], msgs[i], api_id) for i in range(n)]
    for i in range(n):
        assert sigs[i].a == want_sigs[i].a and sigs[i].e == want_sigs[i].e, (curve, i, "sign")
    # verify: half the items forged in different ways (core_sign_tests.rs:67-155)
    vs = [Signature(s.a, s.e) for s in sigs]
    vm = [list(m) for m in msgs]
    vh = list(headers)
    expect = [1] * n
    for i in range(n):
        k = i % 4
        if k == 1:
            vs[i] = Signature(None, sigs[i].e); expect[i] = 0            # forged A = identity
        elif k == 2 and L:
            vm[i][0] = (vm[i][0] + 1) % c.r; expect[i] = 0               # forged message
        elif k == 3:
            vh[i] = vh[i] + b"x"; expect[i] = 0                          # forged header
    st = eng.core_verify_batch(vs, vm, vh)
    assert list(st) == expect, (curve, list(st), expect)
    for i in range(min(n, 3)):
        assert bbs.core_verify(suite, pk, bbs.Signature(vs[i].a, vs[i].e), gens, vh[i], vm[i], api_id) == bool(expect[i])
    proofs, st = eng.core_proof_gen_batch(sigs, msgs, disclosed, rnds, headers, phs)
    assert list(st) == [1] * n
    for i in range(n):
        want = bbs.core_proof_gen(suite, pk, want_sigs[i], headers[i], gens, phs[i], msgs[i], disclosed[i], api_id, rnds[i])
        assert proof_eq(proofs[i], want), (curve, i, "proof_gen")
    dm = [[msgs[i][j] for j in disclosed[i]] for i in range(n)]
    st = eng.core_proof_verify_batch(proofs, dm, disclosed, headers, phs)
    assert list(st) == [1] * n, (curve, list(st))
    # corrupt: one field per item
    bad = [to_engine_proof(p) for p in proofs]
    expect = [1] * n
    for i in range(n):
        k = i % 6
        p = bad[i]
        if k == 0:
            p.e_cap = (p.e_cap + 1) % c.r; expect[i] = 0
        elif k == 1:
            p.a_bar = None; expect[i] = 0                                  # bbs_over_bls_tests.rs:119-133
        elif k == 2:
            p.challenge = (p.challenge + 1) % c.r; expect[i] = 0
        elif k == 3 and p.commitments:
            p.commitments[-1] = (p.commitments[-1] + 5) % c.r; expect[i] = 0
        elif k == 4:
            p.d = c.g1_add(p.d, c.g1); expect[i] = 0
        elif k == 5 and curve == "bls12_381":
            p.d = (0, 2); expect[i] = 0                                    # on the curve, order 3: generic multiplication path
    st = eng.core_proof_verify_batch(bad, dm, disclosed, headers, phs)
    assert list(st) == expect, (curve, list(st), expect)
    for i in sorted(set(range(min(n, 2))) | ({5} if n > 5 else set())):
        op = bbs.Proof(bad[i].a_bar, bad[i].b_bar, bad[i].d, bad[i].e_cap, bad[i].r1_cap, bad[i].r3_cap,
                       bad[i].commitments, bad[i].challenge)
        assert bbs.core_proof_verify(suite, pk, op, gens, headers[i], phs[i], dm[i], disclosed[i], api_id) == bool(expect[i])
    eng.close()


# ------------------------------------------------------------------------------------------------
def check_error_semantics(curve, lib_path=None):
    """Err / panic / Ok(false) behaviour of the reference, per item, inside one batch."""
    rng = random.Random(7)
    suite = bbs.SUITES[curve]
    c = suite.curve
    api_id = suite.api_id
    L = 10
    gens = gens_for(suite, L + 1)
    sk = bbs.key_gen(suite, bytes([1] * 32), b"", b"BBS-SIG-KEYGEN-SALT-")
    pk = bbs.sk_to_pk(suite, sk)
    eng = make_engine(curve, gens, api_id, lib_path, sk=sk)
    msgs = [rng.randrange(c.r) for _ in range(L)]
    sig = eng.core_sign(b"", msgs)
    disclosed = [0, 1, 5]
    rnd = [rng.randrange(1, c.r) for _ in range(5 + L - 3)]
    proof = eng.core_proof_gen(sig, b"", b"", msgs, disclosed, rnd)
    dm = [msgs[i] for i in disclosed]
    default = Proof()
    zero7 = Proof(commitments=[0] * 7)
    items = [
        (proof, dm, disclosed),                                  # valid                       -> 1
        (default, dm, disclosed),                                # Proof::default(): l = 3, index 5 >= l -> Err (bbs_over_bls_tests.rs:137-152)
        (zero7, dm, disclosed),                                  # 7 zero commitments          -> Ok(false) (:172-186)
        (proof, dm, [1, 0, 5]),                                  # caller order matters        -> Ok(false) (proof_verify.rs:18)
        (proof, dm[:2], disclosed),                              # messages != indexes         -> Err -6
        (proof, dm, [0, 1, 1]),                                  # duplicate -> commitments[i] out of bounds -> panic
        (Proof(proof.a_bar, proof.b_bar, proof.d, proof.e_cap, proof.r1_cap, proof.r3_cap, proof.commitments[:-1],
               proof.challenge), dm, disclosed),                 # l = 9 != generators - 1     -> Err -1
    ]
    st = eng.core_proof_verify_batch([x[0] for x in items], [x[1] for x in items], [x[2] for x in items])
    assert list(st) == [1, -3, 0, 0, -6, -22, -1], list(st)
    for (p, m, d), s in zip(items, st):
        op = bbs.Proof(p.a_bar, p.b_bar, p.d, p.e_cap, p.r1_cap, p.r3_cap, list(p.commitments), p.challenge)
        try:
            want = int(bbs.core_proof_verify(suite, pk, op, gens, b"", b"", m, d, api_id))
        except bbs.BbsError as e:
            want = {"InvalidDisclosedIndex": -3, "InvalidIndicesAndMessagesLength": -6,
                    "InvalidMessageAndGeneratorsLength": -1}[e.variant]
        except bbs.BbsPanic:
            want = -22
        assert want == int(s), (want, int(s))
    # forged public key = default (identity) -> Ok(false), never an error (:156-169)
    eng2 = make_engine(curve, gens, api_id, lib_path, pk=None)
    assert eng2.core_proof_verify(proof, b"", b"", dm, disclosed) is False
    assert eng2.core_verify(sig, b"", msgs) is False
    # sign / verify length errors (sign.rs:77-79, verify.rs:69-71)
    sigs, st = eng.core_sign_batch([msgs, msgs[:-1]], [b"", b""])
    assert list(st) == [1, -1] and sigs[1] is None
    st = eng.core_verify_batch([sig, sig], [msgs, msgs + [1]], [b"", b""])
    assert list(st) == [1, -1]
    # proof_gen errors (proof_gen.rs:133-143, 233-235)
    def pg(ms, d, nr):
        try:
            eng.core_proof_gen(sig, b"", b"", ms, d, [1] * nr)
            return 1
        except BbsError as e:
            return e.status
    assert pg(msgs, list(range(L)) + [0], 4) == -2          # r > l
    assert pg(msgs, [0, L], 5 + L - 2) == -3                # index >= l
    assert pg(msgs, [2, 2], 5 + L - 2) == -4                # duplicates: scalars sized from the un-deduped length
    assert pg(msgs[:-1], [0], 5 + L - 2) == -1              # generators length
    # unsorted disclosed indexes are sorted by proof_gen (proof_gen.rs:151-161)
    p2 = eng.core_proof_gen(sig, b"", b"", msgs, [5, 0, 1], rnd)
    assert proof_eq(p2, proof)
    eng.close()
    eng2.close()


def check_empty_batches(curve, lib_path=None):
    """n = 0 through every batched entry point (the reference's functions are per item; an empty batch is the
    degenerate case of the batch boundary), in the per-item and the batch-verification mode."""
    suite = bbs.SUITES[curve]
    eng = make_engine(curve, gens_for(suite, 4), suite.api_id, lib_path, sk=12345)
    for mode in (False, True):
        eng.set_batch_verification(mode)
        sigs, st = eng.core_sign_batch([])
        assert sigs == [] and len(st) == 0
        assert len(eng.core_verify_batch([], [])) == 0
        proofs, st = eng.core_proof_gen_batch([], [], [], [])
        assert proofs == [] and len(st) == 0
        assert len(eng.core_proof_verify_batch([], [], [])) == 0
        job = eng.core_proof_verify_upload([], [], [])
        job.run(); job.wait()
        assert len(job.status()) == 0
        job.free()
        for job in (eng.core_proof_verify_submit([], [], []), eng.core_verify_submit([], []), eng.verify_octets_submit([], []),
                    eng.core_sign_submit([]), eng.core_proof_gen_submit([], [], [], [])):
            job.wait()
            assert len(job.result) == 0
            if job._decode is not None:
                out, st = job.output()
                assert out == [] and len(st) == 0
            job.free()
        assert len(eng.verify_octets_batch([], [])) == 0
        assert len(eng.proof_verify_octets_batch([], [], [])) == 0
    assert eng.hash_to_scalar_batch([], b"dst") == []
    out, st = eng.g1_msm_batch([], [], [])
    assert out == [] and len(st) == 0
    assert len(eng.pairing_product2_is_one_batch([], [])) == 0
    eng.close()


# ------------------------------------------------------------------------------------------------
def check_primitives(curve, lib_path=None):
    rng = random.Random(11)
    suite = bbs.SUITES[curve]
    c = suite.curve
    L = 3
    gens = gens_for(suite, L + 1)
    sk = rng.randrange(1, c.r)
    pk = bbs.sk_to_pk(suite, sk)
    eng = make_engine(curve, gens, b"x", lib_path, sk=sk)
    # hash_to_scalar: empty / short / block-boundary / long inputs
    msgs = [b"", b"a", bytes(55), bytes(56), bytes(range(64)), bytes(rng.randrange(256) for _ in range(300))]
    for dst in (b"", b"QUUX-V01-CS02", bytes(255)):
        got = eng.hash_to_scalar_batch(msgs, dst)
        from oracle.hashing import hash_to_scalar
        assert got == [hash_to_scalar(c, m, dst) for m in msgs]
    # MSM: fixed bases [P1, Q1, H1..], variable bases, edge scalars
    bases = [suite.p1] + gens
    n = 5
    fs = [[rng.randrange(c.r) for _ in range(L + 2)] for _ in range(n)]
    fs[0] = [0] * (L + 2)
    fs[1] = [c.r - 1] * (L + 2)
    vpts = [[c.g1_mul(c.g1, rng.randrange(1, c.r)), c.g1_mul(c.g1, rng.randrange(1, c.r))] for _ in range(n)]
    vsc = [[rng.randrange(c.r), rng.randrange(c.r)] for _ in range(n)]
    vpts[2][0] = None
    vpts[3][1] = vpts[3][0]; vsc[3][1] = (c.r - vsc[3][0]) % c.r          # cancels to the identity
    vpts[4] = [bases[0], bases[1]]                                         # equal to fixed bases (doubling path)
    out, st = eng.g1_msm_batch(fs, vpts, vsc)
    assert list(st) == [1] * n
    for i in range(n):
        want = None
        for k in range(L + 2):
            want = c.g1_add(want, c.g1_mul(bases[k], fs[i][k]))
        for k in range(2):
            want = c.g1_add(want, c.g1_mul(vpts[i][k], vsc[i][k]))
        assert out[i] == want, (curve, i)
    if curve == "bls12_381":
        # on-curve points OUTSIDE the prime-order subgroup, small orders included: the windowed scalar multiplication
        # must give the plain group-law result (its table set-up falls back to the generic path for orders < 16)
        h = 0x396c8c005555e1568c00aaab0000aaab
        x = 5
        while True:
            y2 = (x ** 3 + 4) % c.p
            y = pow(y2, (c.p + 1) // 4, c.p)
            if y * y % c.p == y2 and c.g1_mul((x, y), c.r) is not None:
                break
            x += 1
        Q = (x, y)
        odd = [(0, 2), c.g1_mul(Q, c.r * (h // 11)), c.g1_mul(Q, c.r * (h // 33)), c.g1_mul(Q, c.r * (h // (3 * 11 * 11))), Q,
               c.g1_mul(Q, c.r)]
        assert c.g1_mul(odd[0], 3) is None and c.g1_mul(odd[1], 11) is None and c.g1_mul(odd[2], 33) is None
        vp = [[pt, c.g1_mul(c.g1, 7)] for pt in odd]
        vk = [[rng.randrange(c.r), rng.randrange(c.r)] for _ in odd]
        vk[1][0] = 22; vk[2][0] = c.r - 1
        out, st = eng.g1_msm_batch([[0] * (L + 2)] * len(odd), vp, vk)
        assert list(st) == [1] * len(odd)
        for i in range(len(odd)):
            assert out[i] == c.g1_add(c.g1_mul(vp[i][0], vk[i][0]), c.g1_mul(vp[i][1], vk[i][1])), ("odd-order point", i)
    # pairing product e(Pa, pk) * e(Pb, BP2) == 1
    a = rng.randrange(1, c.r)
    Pa = [c.g1_mul(c.g1, a), c.g1_mul(c.g1, a), None, None, c.g1]
    Pb = [c.g1_neg(c.g1_mul(c.g1, a * sk % c.r)), c.g1_mul(c.g1, a * sk % c.r), None, c.g1, None]
    st = eng.pairing_product2_is_one_batch(Pa, Pb)
    want = [int(c.pairing_product_is_one([(Pa[i], pk), (Pb[i], c.g2)])) for i in range(5)]
    assert list(st) == want == [1, 0, 1, 0, 0]
    eng.close()


# ------------------------------------------------------------------------------------------------
def check_pippenger(curve, lib_path=None, n=40):
    """One large variable-base sum by the device's bucket method against the oracle's plain sum, with the
    cases that stress bucket accumulation: equal points in one bucket (doubling), P and -P in one bucket
    (cancellation), identity points, scalars 0 / 1 / r-1, an off-curve point and a non-canonical scalar."""
    rng = random.Random(21)
    suite = bbs.SUITES[curve]
    c = suite.curve
    eng = make_engine(curve, gens_for(suite, 2), b"x", lib_path)
    pts = [c.g1_mul(c.g1, rng.randrange(1, 1 << 40)) for _ in range(n)]
    sc = [rng.randrange(c.r) for _ in range(n)]
    sc[0], sc[1], sc[2] = 0, 1, c.r - 1
    pts[3] = None
    pts[5] = pts[4]; sc[5] = sc[4]                       # same point, same digits: doubling inside every bucket
    pts[7] = c.g1_neg(pts[6]); sc[7] = sc[6]             # cancels inside every bucket
    sc[8] = sc[9] = 0x0101010101010101010101010101010101010101010101010101010101010101 % c.r
    got, st = eng.g1_msm_pippenger(pts, sc)
    assert list(st) == [1] * n
    want = None
    for p_, k in zip(pts, sc):
        want = c.g1_add(want, c.g1_mul(p_, k))
    assert got == want, curve
    # everything cancels -> identity; empty input -> identity
    got, st = eng.g1_msm_pippenger([pts[6], pts[7]], [5, 5])
    assert got is None and list(st) == [1, 1]
    got, st = eng.g1_msm_pippenger([], [])
    assert got is None
    # flagged items contribute nothing
    bad_pt = (pts[10][0], (pts[10][1] + 1) % c.p)
    got, st = eng.g1_msm_pippenger([pts[10], bad_pt, pts[11], pts[12]], [3, 4, c.r, 7])
    assert list(st) == [1, -41, -40, 1]
    assert got == c.g1_add(c.g1_mul(pts[10], 3), c.g1_mul(pts[12], 7))
    eng.close()


def check_pippenger_tiles(curve, lib_path=None, n=9000, seed=27):
    """The MULTI-TILE path of the bucket kernel (pippenger.hpp k_pip_window: n > 4096 items -> n_tiles > 1, the i0 offsets
    of a tile's digits and points, the tile-sum array and PipTileSums) against (a) the plain-C oracle's sum of independent
    double-and-add multiplications and (b) the Python oracle through the points' known discrete logarithms
    (sum k_i [a_i] G = [sum k_i a_i] G).  The bucket method's edge cases sit ON the tile boundaries: the same point with
    the same scalar in items 4095 | 4096 (last of tile 0, first of tile 1: the tile sums then meet in PipTileSums), P and -P
    inside one tile and across a boundary, the identity as first / last item of a tile, scalars 0 / 1 / r - 1 in items
    4095, 4096, 8191 of further sub-cases, and an off-curve point and a non-canonical scalar in the last tile."""
    from oracle import c_port
    rng = random.Random(seed)
    suite = bbs.SUITES[curve]
    c = suite.curve
    P = c_port.port(curve)
    eng = make_engine(curve, gens_for(suite, 2), b"x", lib_path)
    T = 4096
    logs = [rng.randrange(1, 1 << 40) for _ in range(n)]
    pts = [c.g1_mul(c.g1, a) for a in logs]
    sc = [rng.randrange(c.r) for _ in range(n)]

    def put(i, log, k):
        if i < n:
            logs[i], sc[i] = log % c.r, k
            pts[i] = c.g1_mul(c.g1, log % c.r) if log % c.r else None

    last = n - 1
    tiles = (n + T - 1) // T
    # equal points with equal digits across the boundary 4095 | 4096; P, -P inside tile 0 and across 8191 | 8192
    put(T - 1, logs[7], sc[7]); put(T, logs[7], sc[7])
    put(100, logs[101], sc[101]); put(101, -logs[101], sc[101])
    put(2 * T - 1, logs[9], sc[9]); put(2 * T, -logs[9], sc[9])
    # identity as the first item of tile 1's neighbour and as the very last item
    put(T + 1, 0, sc[T + 1] if n > T + 1 else 0)
    if last > T + 5:
        put(last, 0, 5)
    # one byte pattern in every window (all 32 digits equal: the same bucket of every window in two tiles)
    put(5, logs[5], 0x0101010101010101010101010101010101010101010101010101010101010101 % c.r)
    put(T + 5, logs[T + 5] if n > T + 5 else 1, 0x0101010101010101010101010101010101010101010101010101010101010101 % c.r)
    got, st = eng.g1_msm_pippenger(pts, sc)
    assert list(st) == [1] * n
    want_c = P.g1_msm_plain(pts, sc)
    want_py = c.g1_mul(c.g1, sum(a * k for a, k in zip(logs, sc)) % c.r)
    assert want_c == want_py, "the two oracles disagree"
    assert got == want_c, (curve, n)
    # scalars 0 / 1 / r - 1 on the boundary items, an off-curve point and a non-canonical scalar in the last tile
    sc2 = list(sc)
    for i, k in ((T - 1, 0), (T, 1), (2 * T - 1, c.r - 1), (T - 2, c.r - 1), (T + 2, 0)):
        if i < n:
            sc2[i] = k
    pts2 = list(pts)
    bad = last - 1
    pts2[bad] = (pts[bad][0], (pts[bad][1] + 1) % c.p)
    sc2[last - 2] = c.r
    got, st = eng.g1_msm_pippenger(pts2, sc2)
    want_st = [1] * n
    want_st[bad], want_st[last - 2] = -41, -40
    assert list(st) == want_st
    keep = [i for i in range(n) if want_st[i] == 1]
    assert got == P.g1_msm_plain([pts2[i] for i in keep], [sc2[i] for i in keep]), (curve, n, "edge scalars")
    # everything cancels across tiles: item i of tile 0 against item i of the LAST tile (P, -P with the same scalar)
    m = min(T, n - (tiles - 1) * T)
    pts3 = [None] * n
    sc3 = [0] * n
    for i in range(m):
        pts3[i], sc3[i] = pts[i], sc[i]
        j = (tiles - 1) * T + i
        pts3[j], sc3[j] = (c.g1_neg(pts[i]) if pts[i] is not None else None), sc[i]
    if tiles > 1:
        got, st = eng.g1_msm_pippenger(pts3, sc3)
        assert got is None and list(st) == [1] * n
    eng.close()


def check_batch_verification(curve, lib_path=None, n=9, L=4, seed=5, window_bits=None):
    """Opt-in batch verification (one combined pairing check, per-item fallback) returns the same statuses as the
    default per-item mode and the oracle: all-valid batch; items that fail before the pairing (they are left out of
    the combination); a self-consistent proof of a forged signature (challenge matches, pairing fails: forces the
    fallback); host-validation errors."""
    rng = random.Random(seed)
    suite = bbs.SUITES[curve]
    c = suite.curve
    api_id = suite.api_id
    gens = gens_for(suite, L + 1)
    sk = rng.randrange(1, c.r)
    pk = bbs.sk_to_pk(suite, sk)
    exact = make_engine(curve, gens, api_id, lib_path, sk=sk, window_bits=window_bits)
    batch = make_engine(curve, gens, api_id, lib_path, sk=sk, window_bits=window_bits)
    batch.set_batch_verification(True, bytes(rng.randrange(256) for _ in range(32)))
    msgs = [[rng.randrange(c.r) for _ in range(L)] for _ in range(n)]
    headers = [bytes(rng.randrange(256) for _ in range(rng.choice([0, 3, 40]))) for _ in range(n)]
    phs = [bytes(rng.randrange(256) for _ in range(rng.choice([0, 9]))) for _ in range(n)]
    disclosed = [sorted(rng.sample(range(L), rng.randrange(0, L + 1))) for _ in range(n)]
    rnds = [[rng.randrange(1, c.r) for _ in range(5 + L - len(d))] for d in disclosed]
    sigs, st = exact.core_sign_batch(msgs, headers)
    assert list(st) == [1] * n
    proofs, st = exact.core_proof_gen_batch(sigs, msgs, disclosed, rnds, headers, phs)
    assert list(st) == [1] * n
    dm = [[msgs[i][j] for j in disclosed[i]] for i in range(n)]
    # 1. all valid
    assert list(batch.core_proof_verify_batch(proofs, dm, disclosed, headers, phs)) == [1] * n
    # 2. failures before the pairing only: the combined check passes for the rest
    bad = [to_engine_proof(p_) for p_ in proofs]
    bad[1].e_cap = (bad[1].e_cap + 1) % c.r
    bad[4].challenge = (bad[4].challenge + 1) % c.r
    idx2 = [list(d) for d in disclosed]
    idx2[6] = idx2[6] + [L]                                         # InvalidDisclosedIndex on the host
    dm2 = [list(d) for d in dm]
    dm2[6] = dm2[6] + [1]
    want = list(exact.core_proof_verify_batch(bad, dm2, idx2, headers, phs))
    assert want == [1, 0, 1, 1, 0, 1, -1, 1, 1][:n], want
    assert list(batch.core_proof_verify_batch(bad, dm2, idx2, headers, phs)) == want
    # 3. a proof generated from a forged signature: challenge matches, pairing product is not 1 -> fallback
    forged = [Signature(s.a, s.e) for s in sigs]
    forged[2] = Signature(c.g1_add(sigs[2].a, c.g1), sigs[2].e)
    forged[7] = Signature(c.g1_mul(sigs[7].a, 2), sigs[7].e)
    fp, st = exact.core_proof_gen_batch(forged, msgs, disclosed, rnds, headers, phs)
    assert list(st) == [1] * n
    fp[3].a_bar = None                                               # identity Abar (bbs_over_bls_tests.rs:119-133)
    want = list(exact.core_proof_verify_batch(fp, dm, disclosed, headers, phs))
    assert want == [1, 1, 0, 0, 1, 1, 1, 0, 1][:n], want
    assert list(batch.core_proof_verify_batch(fp, dm, disclosed, headers, phs)) == want
    op = bbs.Proof(fp[2].a_bar, fp[2].b_bar, fp[2].d, fp[2].e_cap, fp[2].r1_cap, fp[2].r3_cap, fp[2].commitments, fp[2].challenge)
    assert bbs.core_proof_verify(suite, pk, op, gens, headers[2], phs[2], dm[2], disclosed[2], api_id) is False
    # 3b. core_verify in the same mode: all valid, then forged A / identity A / wrong message / wrong header
    assert list(batch.core_verify_batch(sigs, msgs, headers)) == [1] * n
    vs = [Signature(s_.a, s_.e) for s_ in sigs]
    vm = [list(m) for m in msgs]
    vh = list(headers)
    vs[0] = Signature(c.g1_add(sigs[0].a, c.g1), sigs[0].e)
    vs[3] = Signature(None, sigs[3].e)
    vm[5][0] = (vm[5][0] + 1) % c.r
    vh[8] = vh[8] + b"x"
    vs[6] = Signature(sigs[6].a, (sigs[6].e + 1) % c.r)
    want_v = list(exact.core_verify_batch(vs, vm, vh))
    assert want_v == [0, 1, 1, 0, 1, 0, 0, 1, 0][:n], want_v
    assert list(batch.core_verify_batch(vs, vm, vh)) == want_v
    # 4. a resident job can be run again and a context can go back to the per-item mode
    job = batch.core_proof_verify_upload(fp, dm, disclosed, headers, phs)
    for _ in range(2):
        job.run()
        assert list(job.status()) == want
    job.free()
    batch.set_batch_verification(False)
    assert list(batch.core_proof_verify_batch(fp, dm, disclosed, headers, phs)) == want
    exact.close()
    batch.close()


def check_points_in_subgroup(curve, lib_path=None, n=10, L=4, seed=8, window_bits=None):
    """bbs_ctx_set_points_in_subgroup (GLV split of the variable-base terms on BLS12-381; BN254 has cofactor 1 and uses
    its split always, so there both engines run the same code and the comparison is against the oracle only): same
    statuses and group elements as the default path and the oracle for inputs in G1 -- valid proofs / signatures,
    tampered scalars and points, edge scalars through the MSM primitive -- alone and together with batch verification."""
    rng = random.Random(seed)
    suite = bbs.SUITES[curve]
    c = suite.curve
    api_id = suite.api_id
    gens = gens_for(suite, L + 1)
    sk = rng.randrange(1, c.r)
    pk = bbs.sk_to_pk(suite, sk)
    exact = make_engine(curve, gens, api_id, lib_path, sk=sk, window_bits=window_bits)
    fast = make_engine(curve, gens, api_id, lib_path, sk=sk, window_bits=window_bits)
    fast.set_points_in_subgroup(True)
    msgs = [[rng.randrange(c.r) for _ in range(L)] for _ in range(n)]
    headers = [bytes(rng.randrange(256) for _ in range(rng.choice([0, 5]))) for _ in range(n)]
    phs = [bytes(rng.randrange(256) for _ in range(rng.choice([0, 9]))) for _ in range(n)]
    disclosed = [sorted(rng.sample(range(L), rng.randrange(0, L + 1))) for _ in range(n)]
    rnds = [[rng.randrange(1, c.r) for _ in range(5 + L - len(d))] for d in disclosed]
    sigs, st = exact.core_sign_batch(msgs, headers)
    assert list(st) == [1] * n
    proofs, st = exact.core_proof_gen_batch(sigs, msgs, disclosed, rnds, headers, phs)
    assert list(st) == [1] * n
    dm = [[msgs[i][j] for j in disclosed[i]] for i in range(n)]
    # proof_gen under vouching produces the same proofs (its variable-base terms of A and B use the split too)
    proofs_fast, st = fast.core_proof_gen_batch(sigs, msgs, disclosed, rnds, headers, phs)
    assert list(st) == [1] * n
    for pa, pb in zip(proofs, proofs_fast):
        assert (pa.a_bar, pa.b_bar, pa.d, pa.e_cap, pa.r1_cap, pa.r3_cap, list(pa.commitments), pa.challenge) == \
               (pb.a_bar, pb.b_bar, pb.d, pb.e_cap, pb.r1_cap, pb.r3_cap, list(pb.commitments), pb.challenge)
    # valid inputs
    assert list(fast.core_proof_verify_batch(proofs, dm, disclosed, headers, phs)) == [1] * n
    assert list(fast.core_verify_batch(sigs, msgs, headers)) == [1] * n
    op = proofs[0]
    assert bbs.core_proof_verify(suite, pk, bbs.Proof(op.a_bar, op.b_bar, op.d, op.e_cap, op.r1_cap, op.r3_cap, op.commitments,
                                                      op.challenge), gens, headers[0], phs[0], dm[0], disclosed[0], api_id) is True
    # tampered (all points still in G1): scalars, swapped points, a proof of a forged signature, the identity
    bad = [to_engine_proof(p_) for p_ in proofs]
    bad[1].e_cap = (bad[1].e_cap + 1) % c.r
    bad[2].r1_cap = 0
    bad[3].r3_cap = c.r - 1
    bad[4].d = c.g1_mul(bad[4].d, 2)
    bad[5].a_bar, bad[5].b_bar = bad[5].b_bar, bad[5].a_bar
    bad[6].a_bar = None
    forged = [Signature(s_.a, s_.e) for s_ in sigs]
    forged[7] = Signature(c.g1_add(sigs[7].a, c.g1), sigs[7].e)
    fp, st = exact.core_proof_gen_batch(forged, msgs, disclosed, rnds, headers, phs)
    bad[7] = to_engine_proof(fp[7])
    want = list(exact.core_proof_verify_batch(bad, dm, disclosed, headers, phs))
    assert want[0] == 1 and want[1:8] == [0] * 7 and want[8:] == [1] * (n - 8), want
    assert list(fast.core_proof_verify_batch(bad, dm, disclosed, headers, phs)) == want
    vs = [Signature(s_.a, s_.e) for s_ in sigs]
    vs[0] = forged[7]
    vs[1] = Signature(sigs[1].a, (sigs[1].e + 1) % c.r)
    vs[2] = Signature(None, sigs[2].e)
    want_v = list(exact.core_verify_batch(vs, msgs, headers))
    assert want_v[:3] != [1, 1, 1] and want_v[3:] == [1] * (n - 3), want_v
    assert list(fast.core_verify_batch(vs, msgs, headers)) == want_v
    # the group elements themselves: variable-base terms with edge scalars through the MSM primitive
    lam = (c.x_param * c.x_param - 1) if curve == "bls12_381" else (1 << 127)
    edge = [0, 1, 2, lam - 1, lam, lam + 1, 2 * lam, c.r - 1, c.r - 2, (1 << 128) - 1, 1 << 128, c.r - lam, (c.r - 1) // 2]
    ks = [k % c.r for k in edge] + [rng.randrange(c.r) for _ in range(12)]
    pts = [c.g1_mul(c.g1, rng.randrange(1, c.r)) for _ in ks]
    pts[-1] = None
    fs = [[rng.randrange(c.r), 1] for _ in ks]
    got, st = fast.g1_msm_batch(fs, [[p_] for p_ in pts], [[k] for k in ks])
    ref, st_ref = exact.g1_msm_batch(fs, [[p_] for p_ in pts], [[k] for k in ks])
    assert list(st) == list(st_ref) == [1] * len(ks)
    assert got == ref
    for i in (0, 3, 4, 7, len(ks) - 2):
        fixed = c.g1_add(c.g1_mul(suite.p1, fs[i][0]), gens[0])
        assert got[i] == c.g1_add(fixed, c.g1_mul(pts[i], ks[i])), i
    # together with batch verification
    fast.set_batch_verification(True, bytes(rng.randrange(256) for _ in range(32)))
    assert list(fast.core_proof_verify_batch(bad, dm, disclosed, headers, phs)) == want
    assert list(fast.core_proof_verify_batch(proofs, dm, disclosed, headers, phs)) == [1] * n
    assert list(fast.core_verify_batch(vs, msgs, headers)) == want_v
    # and back to the default
    fast.set_batch_verification(False)
    fast.set_points_in_subgroup(False)
    assert list(fast.core_proof_verify_batch(bad, dm, disclosed, headers, phs)) == want
    exact.close()
    fast.close()


# ------------------------------------------------------------------------------------------------
def bench_engine(curve, L=32, lib_path=None, window_bits=None, device=0):
    """one issuer key (IKM [1u8;32], dst "BBS-SIG-KEYGEN-SALT-"), the suite's generators."""
    suite = bbs.SUITES[curve]
    gens = gens_for(suite, L + 1)
    sk = bbs.key_gen(suite, bytes([1] * 32), b"", b"BBS-SIG-KEYGEN-SALT-")
    eng = make_engine(curve, gens, suite.api_id, lib_path, sk=sk, window_bits=window_bits, device=device)
    return suite, eng, gens, sk


def bench_items(suite, eng, n, L=32, R=8, first_item=0, ids=None):
    """SURVEY 8d synthetic items b = first_item .. first_item + n (or the given ids): L 32-byte messages derived from
    (b, j), disclosed 0..R, proof_gen scalars from the seeded expander (src/utils/core_utilities.rs:84-100)."""
    api_id = suite.api_id
    ids = list(range(first_item, first_item + n)) if ids is None else list(ids)
    n = len(ids)
    raw = [expand_message(b"bbs-bench-msg" + i2osp(b, 8) + i2osp(j, 8), b"BBS_BENCH_MSG_DST_", 32)
           for b in ids for j in range(L)]
    flat = eng.hash_to_scalar_batch(raw, api_id + b"MAP_MSG_TO_SCALAR_AS_HASH_")
    msgs = [flat[b * L:(b + 1) * L] for b in range(n)]
    disclosed = [list(range(R))] * n
    rnds = [bbs.seeded_random_scalars(suite, b"bbs-bench-rnd" + i2osp(b, 8), api_id + b"MOCK_RANDOM_SCALARS_DST_", 5 + L - R)
            for b in ids]
    return msgs, disclosed, rnds


def bench_workload(curve, n, L=32, R=8, lib_path=None, window_bits=None, seed=0, device=0, first_item=0):
    """SURVEY 8d synthetic workload (bench_engine + bench_items).  Signatures and proofs are produced by the engine
    itself (checked by the caller)."""
    suite, eng, gens, sk = bench_engine(curve, L, lib_path, window_bits, device)
    msgs, disclosed, rnds = bench_items(suite, eng, n, L, R, first_item)
    return suite, eng, gens, sk, msgs, disclosed, rnds


def check_big_batch(curve, lib_path=None, n=4096, L=32, R=8, spot=2, window_bits=None):
    """Full-size batch through size-independent properties: sign -> verify all true -> proof_gen ->
    proof_verify all true; every 16th item corrupted -> exactly those false; a few items spot-checked
    against the oracle."""
    suite, eng, gens, sk, msgs, disclosed, rnds = bench_workload(curve, n, L, R, lib_path, window_bits)
    c = suite.curve
    pk = bbs.sk_to_pk(suite, sk)
    sigs, st = eng.core_sign_batch(msgs)
    assert (st == 1).all()
    st = eng.core_verify_batch(sigs, msgs)
    assert (st == 1).all()
    # every 16th signature forged (src/verify.rs:88-92 -> Ok(false)): A + P1 (items 0 mod 32: fails in the pairing only),
    # e + 1 (16 mod 32), and every 64th item additionally with one message changed
    forged, fmsgs = list(sigs), [list(m) for m in msgs]
    for i in range(0, n, 16):
        forged[i] = Signature(c.g1_add(sigs[i].a, c.g1), sigs[i].e) if i % 32 == 0 else Signature(sigs[i].a, (sigs[i].e + 1) % c.r)
        if i % 64 == 0:
            fmsgs[i][L - 1] = (fmsgs[i][L - 1] + 1) % c.r
    want_vf = [0 if i % 16 == 0 else 1 for i in range(n)]
    st = eng.core_verify_batch(forged, fmsgs)
    assert [int(x) for x in st] == want_vf
    eng.set_batch_verification(True)                     # the opt-in mode decides the same (fallback over the whole batch)
    st = eng.core_verify_batch(forged, fmsgs)
    assert [int(x) for x in st] == want_vf
    eng.set_batch_verification(False)
    proofs, st = eng.core_proof_gen_batch(sigs, msgs, disclosed, rnds)
    assert (st == 1).all()
    dm = [m[:R] for m in msgs]
    st = eng.core_proof_verify_batch(proofs, dm, disclosed)
    assert (st == 1).all()
    for i in range(0, n, 16):
        proofs[i].commitments[0] = (proofs[i].commitments[0] + 1) % c.r
    st = eng.core_proof_verify_batch(proofs, dm, disclosed)
    assert [int(x) for x in st] == [0 if i % 16 == 0 else 1 for i in range(n)]
    # the same two batches in batch-verification mode, then with one pairing-only failure (forged signature,
    # self-consistent proof) that forces the per-item fallback over the whole batch
    eng.set_batch_verification(True)
    st = eng.core_proof_verify_batch(proofs, dm, disclosed)
    assert [int(x) for x in st] == [0 if i % 16 == 0 else 1 for i in range(n)]
    k = n // 2 + 1
    fp, st = eng.core_proof_gen_batch([Signature(c.g1_add(sigs[k].a, c.g1), sigs[k].e)], [msgs[k]], [disclosed[k]], [rnds[k]])
    assert st[0] == 1
    good_k = proofs[k]
    proofs[k] = fp[0]
    st = eng.core_proof_verify_batch(proofs, dm, disclosed)
    assert [int(x) for x in st] == [0 if (i % 16 == 0 or i == k) else 1 for i in range(n)]
    proofs[k] = good_k
    eng.set_batch_verification(False)
    for i in ([1, n - 1][:spot]):
        want_sig = bbs.core_sign(suite, sk, gens, b"", msgs[i], suite.api_id)
        assert (sigs[i].a, sigs[i].e) == (want_sig.a, want_sig.e)
        want = bbs.core_proof_gen(suite, pk, want_sig, b"", gens, b"", msgs[i], disclosed[i], suite.api_id, rnds[i])
        assert proof_eq(proofs[i], want)
    eng.close()


def check_bv_tiles(curve, lib_path=None, n=16384, L=8, R=2, window_bits=None, job_form=None):
    """Batch verification on ONE job of more than 4096 items: the combination's bucket kernel then runs n_tiles > 1
    workgroups per (set, window) and PipTileSums adds them (pippenger.hpp:208-349).  A tile read at a wrong offset would not
    show on an all-valid batch (both point sets share the digits, the combined product stays 1): it shows as an item whose
    pairing product is NOT 1 being waved through.  So: a self-consistent proof of a forged signature (challenge matches,
    only the pairing decides, src/proof_verify.rs:112-115) is planted in every tile in turn -- first and last item of a
    tile included -- together with failures before the pairing (:108-110) in tiles 0 and 3, and the statuses must equal the
    per-item mode's, which are checked against the expected pattern and, for the planted items, the oracle."""
    suite, eng, gens, sk, msgs, disclosed, rnds = bench_workload(curve, n, L, R, lib_path, window_bits)
    c = suite.curve
    pk = bbs.sk_to_pk(suite, sk)
    T = 4096
    tiles = (n + T - 1) // T
    sigs, st = eng.core_sign_batch(msgs)
    assert (st == 1).all()
    proofs, st = eng.core_proof_gen_batch(sigs, msgs, disclosed, rnds)
    assert (st == 1).all()
    dm = [m[:R] for m in msgs]
    # planted pairing-only failures: one position per run, covering every tile and the boundary items of the tiles
    plant_runs = [[min(2 * T + 1234, n - 1)], [T], [n - 1], [min(3 * T, n - 1), T - 1]]
    planted = sorted({k for run in plant_runs for k in run})
    fsigs = [Signature(c.g1_add(sigs[k].a, c.g1), sigs[k].e) for k in planted]
    fps, st = eng.core_proof_gen_batch(fsigs, [msgs[k] for k in planted], [disclosed[k] for k in planted], [rnds[k] for k in planted])
    assert (st == 1).all()
    forged = dict(zip(planted, fps))
    k0 = planted[0]
    op = bbs.Proof(forged[k0].a_bar, forged[k0].b_bar, forged[k0].d, forged[k0].e_cap, forged[k0].r1_cap, forged[k0].r3_cap,
                   forged[k0].commitments, forged[k0].challenge)
    assert bbs.core_proof_verify(suite, pk, op, gens, b"", b"", dm[k0], disclosed[k0], suite.api_id) is False
    # failures before the pairing in tiles 0 and 3 (they never enter the combination)
    pre = [5, T - 2] + ([3 * T + 77] if n > 3 * T + 77 else [])
    for i in pre:
        proofs[i].commitments[0] = (proofs[i].commitments[0] + 1) % c.r
    eng.set_batch_verification(True)
    st = eng.core_proof_verify_batch(proofs, dm, disclosed)                  # combined checks pass: nobody falls back
    assert [int(x) for x in st] == [0 if i in pre else 1 for i in range(n)]
    for run in plant_runs:
        cur = list(proofs)
        for k in run:
            cur[k] = forged[k]
        want = [0 if (i in pre or i in run) else 1 for i in range(n)]
        eng.set_batch_verification(True)
        st_b = [int(x) for x in eng.core_proof_verify_batch(cur, dm, disclosed)]
        eng.set_batch_verification(False)
        st_e = [int(x) for x in eng.core_proof_verify_batch(cur, dm, disclosed)]
        assert st_e == want, (curve, run, "per-item mode")
        assert st_b == st_e, (curve, run, [i for i in range(n) if st_b[i] != st_e[i]][:8])
    # core_verify in the same mode (points A and e A - B computed on the device, Montgomery form): forged A in tiles 1 and 3
    vs = list(sigs)
    vbad = [T + 3, n - 2]
    for i in vbad:
        vs[i] = Signature(c.g1_add(sigs[i].a, c.g1), sigs[i].e)
    eng.set_batch_verification(True)
    st = eng.core_verify_batch(vs, msgs)
    assert [int(x) for x in st] == [0 if i in vbad else 1 for i in range(n)]
    assert tiles > 1
    eng.close()


def check_pool(lib_path=None, devices=(0, 0), per_curve=40, L=4, R=2, window_bits=None, max_batch=16):
    """bbs_pool: a mixed BN254 + BLS12-381 list fanned out over the pool's members BEHIND the C ABI (SURVEY 8(b) / 8(e);
    here both members are device 0 -- two context sets on one GPU, or the host twin).  Item i of the list is BLS12-381 for odd i,
    BN254 for even i; every 16th item has a commitment changed (fails at the challenge, src/proof_verify.rs:108-110), two
    items are self-consistent proofs of forged signatures (fail in the pairing only, :112-115), one has a disclosed index out
    of range (Err before any arithmetic, :139-150).  The pool's statuses -- written by the library into ONE array in list
    order through the sections' global indexes -- must equal what a single context per curve says about the same items,
    whatever the job size (shares cut into several jobs, a ragged last job, a member with an empty share)."""
    from bbs_sign_amd.pool import Pool
    total = 2 * per_curve
    curve_of_item = ["bls12_381" if (i & 1) else "bn254" for i in range(total)]
    items, single = {}, {}
    pool = Pool(list(devices), lib_path)
    assert pool.device_count() == len(devices)
    for curve in ("bls12_381", "bn254"):
        suite, eng, gens, sk = bench_engine(curve, L, lib_path, window_bits)
        c = suite.curve
        ids = [i for i in range(total) if curve_of_item[i] == curve]
        msgs, disclosed, rnds = bench_items(suite, eng, len(ids), L, R, ids=ids)
        sigs, st = eng.core_sign_batch(msgs)
        assert (st == 1).all()
        forged_at = {ids[3], ids[len(ids) - 1]}
        fs = [Signature(c.g1_add(s.a, c.g1), s.e) if g in forged_at else s for g, s in zip(ids, sigs)]
        proofs, st = eng.core_proof_gen_batch(fs, msgs, disclosed, rnds)
        assert (st == 1).all()
        dm = [m[:R] for m in msgs]
        didx = [list(d) for d in disclosed]
        for k, g in enumerate(ids):
            if g % 16 == 0:
                proofs[k].commitments[0] = (proofs[k].commitments[0] + 1) % c.r
        didx[5] = didx[5][:-1] + [L]                                      # InvalidDisclosedIndex
        items[curve] = (ids, proofs, dm, didx)
        single[curve] = eng.core_proof_verify_batch(proofs, dm, didx)
        want_c = [(-3 if k == 5 else (0 if (g % 16 == 0 or g in forged_at) else 1)) for k, g in enumerate(ids)]
        assert [int(x) for x in single[curve]] == want_c, (curve, [int(x) for x in single[curve]])
        pool.set_window_bits(curve, window_bits if window_bits is not None else (4 if lib_path else 8))
        pool.set_generators(curve, gens, suite.api_id)
        pool.set_public_key(curve, eng.public_key())
        eng.close()
    want = np.zeros(total, dtype=np.int8)
    for curve, (ids, _, _, _) in items.items():
        want[ids] = single[curve]

    def fetch(curve, ids):
        all_ids, proofs, dm, didx = items[curve]
        pos = {g: k for k, g in enumerate(all_ids)}
        sel = [pos[g] for g in ids]
        return [proofs[k] for k in sel], [dm[k] for k in sel], [didx[k] for k in sel]

    for mb in (max_batch, 7, 0):                                          # several jobs per share; ragged jobs; one job per share
        got = pool.proof_verify_mixed(curve_of_item, fetch, max_batch=mb)
        assert [int(x) for x in got] == [int(x) for x in want], (mb, [i for i in range(total) if got[i] != want[i]][:8])
    # several lists in flight (bbs_pool_proof_verify_submit): the members do not drain between them; every list's own statuses
    packed = [pool.pack(c2, *fetch(c2, [i for i in range(total) if curve_of_item[i] == c2]), global_index=[i for i in range(total) if curve_of_item[i] == c2])
              for c2 in sorted(items)]
    flying = [pool.submit_packed(packed, n_total=total, max_batch=mb) for mb in (max_batch, 5, 0, max_batch)]
    try:
        pool.set_window_bits("bls12_381", 8)
        raise AssertionError("the pool was reconfigured while lists were in flight")
    except Exception as e:
        assert getattr(e, "rc", None) == -102, e                          # BBS_E_STATE
    for f in flying:
        got = f.wait()
        assert [int(x) for x in got] == [int(x) for x in want]
    # sections without a global index: per-section status arrays, same verdicts
    secs = [pool.pack(curve, *items[curve][1:]) for curve in sorted(items)]
    got = pool.proof_verify_packed(secs, max_batch=max_batch)
    assert [int(x) for x in got] == [int(x) for c2 in sorted(items) for x in single[c2]]
    # a curve that was never configured is refused before anything runs; an empty list is fine
    p2 = Pool(list(devices), lib_path)
    try:
        p2.proof_verify_packed([pool.pack("bls12_381", *items["bls12_381"][1:])])
        raise AssertionError("an unconfigured pool verified a list")
    except Exception as e:
        assert getattr(e, "rc", None) == -102, e                          # BBS_E_STATE
    assert len(p2.proof_verify_packed([])) == 0
    p2.close()
    pool.close()


def check_mixed_curves_in_flight(lib_path=None, n=1024, per_curve=3, rounds=4):
    """BASELINE config 5 on one device: BN254 and BLS12-381 proof_verify batches resident and in flight together
    (every job on its own streams, two contexts), every 16th item corrupted: the statuses of every job are exact."""
    from bbs_sign_amd import Job
    jobs = []
    for curve in ("bls12_381", "bn254"):
        suite, eng, gens, sk, msgs, disclosed, rnds = bench_workload(curve, n, 32, 8, lib_path)
        sigs, st = eng.core_sign_batch(msgs)
        assert (st == 1).all()
        proofs, st = eng.core_proof_gen_batch(sigs, msgs, disclosed, rnds)
        assert (st == 1).all()
        for i in range(0, n, 16):
            proofs[i].commitments[0] = (proofs[i].commitments[0] + 1) % suite.curve.r
        dm = [m[:8] for m in msgs]
        jobs.append([eng.core_proof_verify_upload(proofs, dm, disclosed) for _ in range(per_curve)])
    order = [j for pair in zip(*jobs) for j in pair]
    for _ in range(rounds):
        for j in order:
            j.run()
    for j in order:
        j.wait()
    want = [0 if i % 16 == 0 else 1 for i in range(n)]
    for j in order:
        assert [int(x) for x in j.status()] == want
        j.free()


def check_batch_vs_c_oracle(lib_path=None, n=512, L=32, R=8, window_bits=None, curve="bls12_381"):
    """EVERY item of a BASELINE-shaped batch against the plain-C oracle (oracle/c), bit for bit:
    signatures, proofs and proof_verify booleans incl. corrupted items.  Two kinds of corrupted proofs: every 7th fails
    BEFORE the pairing (r1^ changed: the challenge no longer matches, src/proof_verify.rs:108-110), and a handful are
    self-consistent proofs of FORGED signatures (A + P1): their challenge matches and only the PER-ITEM pairing product can
    reject them (:112-115).  Those sit where a lane mapping could go wrong: last / first six-lane group of neighbouring
    pairing wavefronts (items 9 | 10 -- ten items per wavefront), last / first lane of neighbouring MSM wavefronts (63 | 64),
    the last full pairing wavefront's last group and the ragged last wavefront's first and last group (n - 7, n - 6, n - 1:
    4089, 4090, 4095 at n = 4096)."""
    import concurrent.futures as cf
    import os
    from oracle import c_port
    suite, eng, gens, sk, msgs, disclosed, rnds = bench_workload(curve, n, L, R, lib_path, window_bits)
    c = suite.curve
    api_id = suite.api_id
    c_port = c_port.port(curve)
    pk = c_port.sk_to_pk(sk)
    sigs, st = eng.core_sign_batch(msgs)
    assert (st == 1).all()
    proofs, st = eng.core_proof_gen_batch(sigs, msgs, disclosed, rnds)
    assert (st == 1).all()
    dm = [m[:R] for m in msgs]
    planted = sorted({k for k in (9, 10, 63, 64, n // 2 - 1, n - 7, n - 6, n - 1) if 0 <= k < n})
    fsig = {k: Signature(c.g1_add(sigs[k].a, c.g1), sigs[k].e) for k in planted}
    fps, fst = eng.core_proof_gen_batch([fsig[k] for k in planted], [msgs[k] for k in planted], [disclosed[k] for k in planted],
                                        [rnds[k] for k in planted])
    assert (fst == 1).all()
    for k, fp in zip(planted, fps):
        proofs[k] = fp
    for i in range(0, n, 7):
        if i not in fsig:
            proofs[i].r1_cap = (proofs[i].r1_cap + 1) % c.r
    st = eng.core_proof_verify_batch(proofs, dm, disclosed)              # per-item mode: every item its own pairing product
    # core_verify (src/verify.rs:53-93) of every item, every 7th signature forged (A + P1 / e + 1 alternating)
    vsigs = list(sigs)
    for i in range(0, n, 7):
        vsigs[i] = Signature(c.g1_add(sigs[i].a, c.g1), sigs[i].e) if i % 2 else Signature(sigs[i].a, (sigs[i].e + 1) % c.r)
    vst = eng.core_verify_batch(vsigs, msgs)

    def one(i):
        s = c_port.core_sign(sk, gens, b"", msgs[i], api_id)
        good = bbs.Signature(fsig[i].a, fsig[i].e) if i in fsig else bbs.Signature(sigs[i].a, sigs[i].e)   # the signature the proof was made from
        vf = c_port.core_verify(pk, bbs.Signature(vsigs[i].a, vsigs[i].e), gens, b"", msgs[i], api_id)
        p = c_port.core_proof_gen(pk, good, b"", gens, b"", msgs[i], disclosed[i], api_id, rnds[i])
        mine = bbs.Proof(proofs[i].a_bar, proofs[i].b_bar, proofs[i].d, proofs[i].e_cap, proofs[i].r1_cap, proofs[i].r3_cap,
                         proofs[i].commitments, proofs[i].challenge)
        v = c_port.core_proof_verify(pk, mine, gens, b"", b"", dm[i], disclosed[i], api_id)
        if i % 7 == 0 and i not in fsig:
            p.r1_cap = (p.r1_cap + 1) % c.r
        return (s.a, s.e) == (sigs[i].a, sigs[i].e), p == mine, int(v) == int(st[i]), int(vf) == int(vst[i])

    with cf.ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 1)) as ex:
        res = list(ex.map(one, range(n)))
    assert all(r[0] for r in res), "sign mismatch"
    assert all(r[1] for r in res), "proof_gen mismatch"
    assert all(r[2] for r in res), "proof_verify mismatch"
    assert all(r[3] for r in res), "verify mismatch"
    assert [int(x) for x in st] == [0 if (i % 7 == 0 or i in fsig) else 1 for i in range(n)]
    assert [int(x) for x in vst] == [0 if i % 7 == 0 else 1 for i in range(n)]
    # the opt-in batch-verification mode returns the same booleans on the same (partly corrupted) batch
    eng.set_batch_verification(True)
    assert [int(x) for x in eng.core_proof_verify_batch(proofs, dm, disclosed)] == [int(x) for x in st]
    eng.close()


# ------------------------------------------------------------------------------------------------
def check_window_widths(curve, lib_path=None, widths=(5, 7, 11, 13), L=2, seed=31):
    """Fixed-base tables at window widths whose digits straddle 32-bit words and whose last window is clamped at bit
    256 (stages.hpp fixed_msm_chunk_to): the group elements of the MSM primitive with edge scalars, and one
    sign -> verify -> proof_gen -> proof_verify round trip, against the oracle (src/proof_verify.rs:163-182)."""
    rng = random.Random(seed)
    suite = bbs.SUITES[curve]
    c = suite.curve
    api_id = suite.api_id
    gens = gens_for(suite, L + 1)
    bases = [suite.p1] + gens
    sk = rng.randrange(1, c.r)
    pk = bbs.sk_to_pk(suite, sk)
    for w in widths:
        eng = make_engine(curve, gens, api_id, lib_path, sk=sk, window_bits=w)
        top = (1 << 256) - 1
        edge = [0, 1, c.r - 1, (1 << w) - 1, 1 << w, (1 << (32 - 1)) | 1, ((1 << w) - 1) << (32 - w // 2),
                top % c.r, (top >> 1) % c.r, ((1 << 255) | (1 << 254)) % c.r, (c.r - 1) >> 1]
        fs = [[edge[(i + k) % len(edge)] for k in range(L + 2)] for i in range(len(edge))]
        fs += [[rng.randrange(c.r) for _ in range(L + 2)] for _ in range(4)]
        out, st = eng.g1_msm_batch(fs, [], [])
        assert list(st) == [1] * len(fs)
        for i, row in enumerate(fs):
            want = None
            for k in range(L + 2):
                want = c.g1_add(want, c.g1_mul(bases[k], row[k]))
            assert out[i] == want, (curve, w, i)
        msgs = [rng.randrange(c.r) for _ in range(L)]
        hdr, ph = b"hdr%d" % w, b"ph"
        sig = eng.core_sign(hdr, msgs)
        want = bbs.core_sign(suite, sk, gens, hdr, msgs, api_id)
        assert (sig.a, sig.e) == (want.a, want.e), (curve, w, "sign")
        assert eng.core_verify(sig, hdr, msgs) is True
        rnd = [rng.randrange(1, c.r) for _ in range(5 + L - 1)]
        proof = eng.core_proof_gen(sig, hdr, ph, msgs, [1], rnd)
        wantp = bbs.core_proof_gen(suite, pk, want, hdr, gens, ph, msgs, [1], api_id, rnd)
        assert proof_eq(proof, wantp), (curve, w, "proof_gen")
        assert eng.core_proof_verify(proof, hdr, ph, [msgs[1]], [1]) is True
        proof.commitments[0] = (proof.commitments[0] + 1) % c.r
        assert eng.core_proof_verify(proof, hdr, ph, [msgs[1]], [1]) is False
        eng.close()


def check_fail_closed_submit(curve, lib_path=None):
    """EVERY *_submit entry point of the C ABI ends in the one epilogue of capi.hip (submit_with_results): run, copy the statuses
    (and records) to page-locked memory, arm, deliver at bbs_job_wait -- and delivery refuses with BBS_E_STATE if ANY item is
    still undecided.  Shown for each entry point through its *_batch form (= submit + wait) with the stage that decides the
    items left out on purpose (BBS_FAULT_SKIP_STAGE, runtime.hpp): the call must fail with BBS_E_STATE, never hand out a
    status or a record, and work normally again once the fault is gone.  (src/proof_verify.rs:108-115, src/verify.rs:88-92: a
    verdict the pipeline did not reach must not read as Ok(true).)"""
    import os
    from bbs_sign_amd.engine import BbsRuntimeError
    from bbs_sign_amd import Issuer
    rng = random.Random(23)
    suite = bbs.SUITES[curve]
    c = suite.curve
    L, n = 3, 3
    gens = gens_for(suite, L + 1)
    sk = rng.randrange(1, c.r)
    eng = make_engine(curve, gens, suite.api_id, lib_path, sk=sk)
    eng.set_latency_mode(False)
    raw = [[bytes([65 + i, 48 + j]) * (1 + j) for j in range(L)] for i in range(n)]
    msgs = [eng.hash_to_scalar_batch(r, suite.api_id + b"MAP_MSG_TO_SCALAR_AS_HASH_") for r in raw]
    disclosed = [[0], [1, 2], []]
    rnds = [[rng.randrange(1, c.r) for _ in range(5 + L - len(d))] for d in disclosed]
    sigs, st = eng.core_sign_batch(msgs)
    assert list(st) == [1] * n
    proofs, st = eng.core_proof_gen_batch(sigs, msgs, disclosed, rnds)
    assert list(st) == [1] * n
    dm = [[msgs[i][j] for j in disclosed[i]] for i in range(n)]
    draw = [[raw[i][j] for j in disclosed[i]] for i in range(n)]
    sig_oct, st = eng.sign_octets_batch(msgs)
    assert list(st) == [1] * n
    pf_oct, st = eng.proof_gen_octets_batch(sigs, msgs, disclosed, rnds)
    assert list(st) == [1] * n
    VERIFY_SKIP = "pairing_6lane,pair_final_exp,pair_final"      # throughput form on the GPU / the host twin's one-lane stages
    calls = [
        ("bbs_core_proof_verify_submit", "pv_finish", lambda: eng.core_proof_verify_batch(proofs, dm, disclosed)),
        ("bbs_proof_verify_octets_submit", "pv_finish", lambda: eng.proof_verify_octets_batch(pf_oct, dm, disclosed)),
        ("bbs_proof_verify_wire_submit", "pv_finish", lambda: eng.proof_verify_wire_batch(pf_oct, draw, disclosed)),
        ("bbs_core_verify_submit", VERIFY_SKIP, lambda: eng.core_verify_batch(sigs, msgs)),
        ("bbs_verify_octets_submit", VERIFY_SKIP, lambda: eng.verify_octets_batch(sig_oct, msgs)),
        ("bbs_verify_wire_submit", VERIFY_SKIP, lambda: eng.verify_wire_batch(sig_oct, raw)),
        ("bbs_core_sign_submit", "sg_combine", lambda: eng.core_sign_batch(msgs)),
        ("bbs_sign_octets_submit", "sg_combine", lambda: eng.sign_octets_batch(msgs)),
        ("bbs_sign_wire_submit", "sg_combine", lambda: eng.sign_wire_batch(raw)),
        ("bbs_core_proof_gen_submit", "pg_finalize", lambda: eng.core_proof_gen_batch(sigs, msgs, disclosed, rnds)),
        ("bbs_proof_gen_octets_submit", "pg_finalize", lambda: eng.proof_gen_octets_batch(sigs, msgs, disclosed, rnds)),
        ("bbs_proof_gen_wire_submit", "pg_finalize", lambda: eng.proof_gen_wire_batch(sig_oct, raw, disclosed, rnds)),
    ]
    try:
        for name, skip, call in calls:
            os.environ["BBS_FAULT_SKIP_STAGE"] = skip
            try:
                got = call()
            except BbsRuntimeError as e:
                assert e.rc == -102, (name, e.rc)
            else:
                raise AssertionError("%s delivered %r although the stage that decides the items never ran" % (name, got))
            os.environ["BBS_FAULT_SKIP_STAGE"] = ""
            got = call()                                        # the fault gone: the same call answers
            st = got[1] if isinstance(got, tuple) else got
            assert [int(x) for x in st] == [1] * n, (name, list(st))
        # the issuer's routed calls end in the same submit forms (its own generators: create_generators by hash-to-curve)
        iss = Issuer(curve, suite.api_id, lib_path=lib_path, window_bits=4 if lib_path else 8)
        iss.set_secret_key(sk)
        i_sig, st = iss.sign(raw)
        assert list(st) == [1] * n
        i_pf, st = iss.proof_gen(i_sig, raw, disclosed, rnds)
        assert list(st) == [1] * n
        for name, skip, call in (("bbs_issuer_proof_verify", "pv_finish", lambda: iss.proof_verify(i_pf, draw, disclosed)),
                                 ("bbs_issuer_verify", VERIFY_SKIP, lambda: iss.verify(i_sig, raw)),
                                 ("bbs_issuer_sign", "sg_combine", lambda: iss.sign(raw)),
                                 ("bbs_issuer_proof_gen", "pg_finalize", lambda: iss.proof_gen(i_sig, raw, disclosed, rnds))):
            os.environ["BBS_FAULT_SKIP_STAGE"] = skip
            try:
                got = call()
            except BbsRuntimeError as e:
                assert e.rc == -102, (name, e.rc)
            else:
                raise AssertionError("%s delivered %r with undecided items" % (name, got))
            os.environ["BBS_FAULT_SKIP_STAGE"] = ""
            got = call()
            st = got[1] if isinstance(got, tuple) else got
            assert [int(x) for x in st] == [1] * n, (name, list(st))
        iss.close()
    finally:
        os.environ.pop("BBS_FAULT_SKIP_STAGE", None)
    eng.close()


def check_fail_closed(curve, lib_path=None):
    """A job whose kernels never ran has decided nothing: its statuses are the internal pending value, which the C ABI
    refuses to return (BBS_E_STATE) -- it must never read as Ok(true).  After a run the same job reports normally."""
    from bbs_sign_amd.engine import BbsRuntimeError
    rng = random.Random(17)
    suite = bbs.SUITES[curve]
    c = suite.curve
    L = 3
    gens = gens_for(suite, L + 1)
    eng = make_engine(curve, gens, suite.api_id, lib_path, sk=rng.randrange(1, c.r))
    msgs = [[rng.randrange(c.r) for _ in range(L)] for _ in range(3)]
    sigs, st = eng.core_sign_batch(msgs)
    assert list(st) == [1, 1, 1]
    disclosed = [[0], [1, 2], []]
    rnds = [[rng.randrange(1, c.r) for _ in range(5 + L - len(d))] for d in disclosed]
    proofs, st = eng.core_proof_gen_batch(sigs, msgs, disclosed, rnds)
    assert list(st) == [1, 1, 1]
    dm = [[msgs[i][j] for j in disclosed[i]] for i in range(3)]
    jobs = [eng.core_proof_verify_upload(proofs, dm, disclosed), eng.core_verify_upload(sigs, msgs), eng.core_sign_upload(msgs),
            eng.core_proof_gen_upload(sigs, msgs, disclosed, rnds)]
    for job in jobs:
        try:
            st = job.status()
        except BbsRuntimeError as e:
            assert e.rc == -102, e.rc
        else:
            raise AssertionError("a job that never ran returned statuses %r" % list(st))
    # ... and no records either: the output buffers of sign / proof_gen are only written by a run
    for job, fetch in ((jobs[2], "bbs_job_fetch_signatures"), (jobs[3], "bbs_job_fetch_proofs")):
        out = np.zeros(3 * (6 * eng.fpb + 128), dtype=np.uint8)
        cm = np.zeros(3 * L * 32, dtype=np.uint8)
        cmo = np.zeros(4, dtype=np.uint64)
        u8 = lambda x: x.ctypes.data_as(_lib.c_u8p)
        if fetch == "bbs_job_fetch_signatures":
            rc = eng.lib.bbs_job_fetch_signatures(job.h, u8(out))
        else:
            rc = eng.lib.bbs_job_fetch_proofs(job.h, u8(out), u8(cm), cmo.ctypes.data_as(_lib.c_u64p))
        assert rc == -102 and not out.any() and not cm.any(), (fetch, rc)
    for job in jobs:
        job.run(); job.wait()
        assert list(job.status()) == [1, 1, 1]
    got_s, st = jobs[2].signatures()
    assert [(s_.a, s_.e) for s_ in got_s] == [(s_.a, s_.e) for s_ in sigs]
    got_p, st = jobs[3].proofs()
    assert all(proof_eq(p_, q_) for p_, q_ in zip(got_p, proofs))
    for job in jobs:
        job.free()
    eng.close()


def check_submit(curve, lib_path=None, n=7, L=4, seed=23):
    """bbs_core_proof_verify_submit: several batches submitted back to back by one thread, statuses delivered at wait,
    equal to the one-shot call and to the oracle -- including items the device-side validation rejects."""
    rng = random.Random(seed)
    suite = bbs.SUITES[curve]
    c = suite.curve
    api_id = suite.api_id
    gens = gens_for(suite, L + 1)
    sk = rng.randrange(1, c.r)
    pk = bbs.sk_to_pk(suite, sk)
    eng = make_engine(curve, gens, api_id, lib_path, sk=sk)
    batches = []
    for b in range(3):
        msgs = [[rng.randrange(c.r) for _ in range(L)] for _ in range(n)]
        headers = [bytes(rng.randrange(256) for _ in range(rng.choice([0, 3, 33]))) for _ in range(n)]
        phs = [bytes(rng.randrange(256) for _ in range(rng.choice([0, 7]))) for _ in range(n)]
        disclosed = [sorted(rng.sample(range(L), rng.randrange(0, L + 1))) for _ in range(n)]
        rnds = [[rng.randrange(1, c.r) for _ in range(5 + L - len(d))] for d in disclosed]
        sigs, st = eng.core_sign_batch(msgs, headers)
        proofs, st = eng.core_proof_gen_batch(sigs, msgs, disclosed, rnds, headers, phs)
        assert list(st) == [1] * n
        dm = [[msgs[i][j] for j in disclosed[i]] for i in range(n)]
        idx = [list(d) for d in disclosed]
        k = b % n
        proofs[k].r3_cap = (proofs[k].r3_cap + 1) % c.r                  # Ok(false)
        proofs[(k + 1) % n].e_cap = c.r                                   # not canonical: -40
        idx[(k + 2) % n] = idx[(k + 2) % n] + [L + 3]                     # InvalidDisclosedIndex: -3
        dm[(k + 2) % n] = dm[(k + 2) % n] + [5]
        if len(idx[(k + 3) % n]) >= 1:
            idx[(k + 3) % n] = idx[(k + 3) % n] + [idx[(k + 3) % n][0]]   # duplicate: the reference panics, -22 (or -1)
            dm[(k + 3) % n] = dm[(k + 3) % n] + [7]
        batches.append((proofs, dm, idx, headers, phs))
    # core_verify through its submit form: two batches in flight, a forged message, a non-canonical e, a wrong length
    vjobs, vwant = [], []
    for b in range(2):
        msgs = [[rng.randrange(c.r) for _ in range(L)] for _ in range(n)]
        headers = [bytes(rng.randrange(256) for _ in range(rng.choice([0, 9]))) for _ in range(n)]
        sigs, st = eng.core_sign_batch(msgs, headers)
        vm = [list(m) for m in msgs]
        vm[b][0] = (vm[b][0] + 1) % c.r
        vs = [Signature(s_.a, s_.e) for s_ in sigs]
        vs[b + 2] = Signature(sigs[b + 2].a, c.r + 5)
        vm[b + 4] = vm[b + 4][:-1]
        vjobs.append(eng.core_verify_submit(vs, vm, headers))
        vwant.append([int(x) for x in eng.core_verify_batch(vs, vm, headers)])
    for job, want in zip(vjobs, vwant):
        job.wait()
        assert [int(x) for x in job.result] == want and 0 in want and -40 in want and -1 in want, want
        job.free()
    # core_sign / core_proof_gen through their submit forms: two batches of each in flight, records delivered at wait;
    # a wrong message count (Err), a scalar >= r (malformed), a signature with e >= r; against the oracle and the one-shot call
    sjobs, pjobs, sin, pin = [], [], [], []
    for b in range(2):
        msgs = [[rng.randrange(c.r) for _ in range(L)] for _ in range(n)]
        headers = [bytes(rng.randrange(256) for _ in range(rng.choice([0, 11, 65]))) for _ in range(n)]
        phs = [bytes(rng.randrange(256) for _ in range(rng.choice([0, 4]))) for _ in range(n)]
        disclosed = [sorted(rng.sample(range(L), rng.randrange(0, L + 1))) for _ in range(n)]
        rnds = [[rng.randrange(1, c.r) for _ in range(5 + L - len(d))] for d in disclosed]
        good, st = eng.core_sign_batch(msgs, headers)
        assert list(st) == [1] * n
        sm = [list(m) for m in msgs]
        sm[b] = sm[b][:-1]
        sm[b + 2][L - 1] = c.r + 1
        sin.append((sm, headers))
        sjobs.append(eng.core_sign_submit(sm, headers))
        ps = [Signature(s_.a, s_.e) for s_ in good]
        ps[b + 1] = Signature(good[b + 1].a, c.r)
        pin.append((ps, msgs, disclosed, rnds, headers, phs))
        pjobs.append(eng.core_proof_gen_submit(ps, msgs, disclosed, rnds, headers, phs))
    for b, job in enumerate(sjobs):
        sm, headers = sin[b]
        job.wait()
        sigs, st = job.output()
        one, st1 = eng.core_sign_batch(sm, headers)
        assert list(st) == list(st1) == [int(x) for x in job.result] and list(st).count(1) == n - 2 and st[b] < 0 and st[b + 2] == -40, list(st)
        for i in range(n):
            if st[i] != 1:
                assert sigs[i] is None and one[i] is None
                continue
            w = bbs.core_sign(suite, sk, gens, headers[i], sm[i], api_id)
            assert sigs[i].a == w.a and sigs[i].e == w.e and one[i].a == w.a and one[i].e == w.e, (curve, b, i, "sign submit")
        job.free()
    for b, job in enumerate(pjobs):
        ps, msgs, disclosed, rnds, headers, phs = pin[b]
        job.wait()
        proofs, st = job.output()
        one, st1 = eng.core_proof_gen_batch(ps, msgs, disclosed, rnds, headers, phs)
        assert list(st) == list(st1) == [int(x) for x in job.result] and list(st).count(1) == n - 1 and st[b + 1] == -40, list(st)
        for i in range(n):
            if st[i] != 1:
                assert proofs[i] is None and one[i] is None
                continue
            w = bbs.core_proof_gen(suite, pk, bbs.Signature(ps[i].a, ps[i].e), headers[i], gens, phs[i], msgs[i], disclosed[i], api_id, rnds[i])
            assert proof_eq(proofs[i], w) and proof_eq(one[i], w), (curve, b, i, "proof_gen submit")
        job.free()
    jobs = [eng.core_proof_verify_submit(*b) for b in batches]            # all in flight
    for job, b in zip(jobs, batches):
        job.wait()
        got = [int(x) for x in job.result]
        assert got == [int(x) for x in eng.core_proof_verify_batch(*b)], got
        assert -40 in got and -3 in got and 0 in got and 1 in got, got
        proofs, dm, idx, headers, phs = b
        for i in range(n):
            if got[i] in (0, 1):
                p = proofs[i]
                op = bbs.Proof(p.a_bar, p.b_bar, p.d, p.e_cap, p.r1_cap, p.r3_cap, list(p.commitments), p.challenge)
                assert int(bbs.core_proof_verify(suite, pk, op, gens, headers[i], phs[i], dm[i], idx[i], api_id)) == got[i]
        job.free()
    # completion-order retire (bbs_jobs_wait_any): jobs of three operations in flight, taken in whatever order they finish;
    # every job is delivered exactly once, to ITS OWN buffers, with the statuses of the one-shot call; a job that is freed
    # leaves the set (None entries are skipped); a resident job that was never run is ignored; the set without a run job
    # is BBS_E_STATE; bbs_job_poll agrees with what was delivered
    from bbs_sign_amd import Job
    from bbs_sign_amd.engine import BbsRuntimeError
    want = [[int(x) for x in eng.core_proof_verify_batch(*b)] for b in batches]
    sm, sh = sin[0]
    sig_want, sig_st = eng.core_sign_batch(sm, sh)
    never_run = eng.core_proof_verify_upload(*batches[0])
    try:
        Job.wait_any([never_run, None])
    except BbsRuntimeError as e:
        assert e.rc == -102, e.rc
    else:
        raise AssertionError("wait_any over jobs that were never run returned")
    for rep in range(3):
        live = [eng.core_proof_verify_submit(*batches[k % 3]) for k in range(rep, rep + 5)] + [eng.core_sign_submit(sm, sh), never_run, None]
        kinds = [(k % 3) for k in range(rep, rep + 5)] + ["sign", "idle", None]
        seen = 0
        while any(j is not None and j is not never_run for j in live):
            k = Job.wait_any(live)
            job = live[k]
            assert job is not None and job is not never_run and job.done()
            if kinds[k] == "sign":
                sigs2, st2 = job.output()
                assert list(st2) == list(sig_st) and [(s_.a, s_.e) if s_ else None for s_ in sigs2] == [(s_.a, s_.e) if s_ else None for s_ in sig_want]
            else:
                assert [int(x) for x in job.result] == want[kinds[k]], (curve, rep, k)
            job.free()
            live[k] = None
            seen += 1
        assert seen == 6
    assert not never_run.done()
    never_run.run()
    assert Job.wait_any([never_run]) == 0 and [int(x) for x in never_run.status()] == want[0]
    never_run.free()
    eng.close()


def check_latency_mode(curve, lib_path=None, n=12, L=4, seed=41):
    """bbs_ctx_set_latency_mode (T1's three terms on three lanes, summed afterwards): the same statuses as the default
    joint chain and as the oracle -- valid proofs, every tampered field that enters T1 or T2, identity and small-order
    points -- alone and together with batch verification and subgroup vouching."""
    rng = random.Random(seed)
    suite = bbs.SUITES[curve]
    c = suite.curve
    api_id = suite.api_id
    gens = gens_for(suite, L + 1)
    sk = rng.randrange(1, c.r)
    pk = bbs.sk_to_pk(suite, sk)
    exact = make_engine(curve, gens, api_id, lib_path, sk=sk)
    exact.set_latency_mode(False)
    fast = make_engine(curve, gens, api_id, lib_path, sk=sk)
    fast.set_latency_mode(True)
    msgs = [[rng.randrange(c.r) for _ in range(L)] for _ in range(n)]
    headers = [bytes(rng.randrange(256) for _ in range(rng.choice([0, 5]))) for _ in range(n)]
    phs = [bytes(rng.randrange(256) for _ in range(rng.choice([0, 9]))) for _ in range(n)]
    disclosed = [sorted(rng.sample(range(L), rng.randrange(0, L + 1))) for _ in range(n)]
    rnds = [[rng.randrange(1, c.r) for _ in range(5 + L - len(d))] for d in disclosed]
    sigs, st = exact.core_sign_batch(msgs, headers)
    proofs, st = exact.core_proof_gen_batch(sigs, msgs, disclosed, rnds, headers, phs)
    assert list(st) == [1] * n
    dm = [[msgs[i][j] for j in disclosed[i]] for i in range(n)]
    assert list(fast.core_proof_verify_batch(proofs, dm, disclosed, headers, phs)) == [1] * n
    # core_verify in both forms (the latency form splits its two Miller loops too): forged A, forged e, identity
    vs = [Signature(s_.a, s_.e) for s_ in sigs]
    vs[1] = Signature(c.g1_add(sigs[1].a, c.g1), sigs[1].e)
    vs[2] = Signature(sigs[2].a, (sigs[2].e + 1) % c.r)
    vs[3] = Signature(None, sigs[3].e)
    want_v = [0 if i in (1, 2, 3) else 1 for i in range(n)]
    assert list(exact.core_verify_batch(vs, msgs, headers)) == want_v
    assert list(fast.core_verify_batch(vs, msgs, headers)) == want_v
    bad = [to_engine_proof(p_) for p_ in proofs]
    bad[1].e_cap = (bad[1].e_cap + 1) % c.r
    bad[2].r1_cap = 0
    bad[3].r3_cap = c.r - 1
    bad[4].d = c.g1_mul(bad[4].d, 2)
    bad[5].a_bar, bad[5].b_bar = bad[5].b_bar, bad[5].a_bar
    bad[6].a_bar = None
    bad[7].challenge = (bad[7].challenge + 1) % c.r
    bad[8].b_bar = None
    if curve == "bls12_381":
        bad[9].d = (0, 2)                                            # on the curve, order 3
    bad[10].d = None
    want = list(exact.core_proof_verify_batch(bad, dm, disclosed, headers, phs))
    assert want[0] == 1 and want[11] == 1 and 0 in want, want
    assert list(fast.core_proof_verify_batch(bad, dm, disclosed, headers, phs)) == want
    for i in (1, 4, 6):
        p = bad[i]
        op = bbs.Proof(p.a_bar, p.b_bar, p.d, p.e_cap, p.r1_cap, p.r3_cap, list(p.commitments), p.challenge)
        assert int(bbs.core_proof_verify(suite, pk, op, gens, headers[i], phs[i], dm[i], disclosed[i], api_id)) == want[i]
    fast.set_batch_verification(True, bytes(rng.randrange(256) for _ in range(32)))
    assert list(fast.core_proof_verify_batch(bad, dm, disclosed, headers, phs)) == want
    fast.set_batch_verification(False)
    fast.set_points_in_subgroup(True)
    ok = [i for i in range(n) if i != 9]                            # vouching excludes the small-order point
    pick = lambda xs: [xs[i] for i in ok]
    assert list(fast.core_proof_verify_batch(pick(bad), pick(dm), pick(disclosed), pick(headers), pick(phs))) == pick(want)
    exact.close()
    fast.close()


def check_large_shapes(curve, lib_path=None, L=100, n=3, seed=51):
    """Shapes well beyond the BASELINE one: many messages (the reference's benches go to 128, benches/sign.rs:40), long
    ragged headers / presentation headers (several SHA-256 blocks more than the cached prefix), everything or nothing
    disclosed.  Signature, proof and booleans against the oracle."""
    rng = random.Random(seed)
    suite = bbs.SUITES[curve]
    c = suite.curve
    api_id = suite.api_id
    gens = bbs.synthetic_generators(suite, L + 1)           # hash-to-curve of 100+ generators is not what is tested here
    sk = rng.randrange(1, c.r)
    pk = bbs.sk_to_pk(suite, sk)
    eng = make_engine(curve, gens, api_id, lib_path, sk=sk)
    msgs = [[rng.randrange(c.r) for _ in range(L)] for _ in range(n)]
    headers = [bytes(rng.randrange(256) for _ in range(k)) for k in (1000, 0, 63, 129)[:n]]
    phs = [bytes(rng.randrange(256) for _ in range(k)) for k in (0, 300, 64, 1)[:n]]
    disclosed = [list(range(L)), [], sorted(rng.sample(range(L), L // 3)), [L - 1]][:n]
    rnds = [[rng.randrange(1, c.r) for _ in range(5 + L - len(d))] for d in disclosed]
    sigs, st = eng.core_sign_batch(msgs, headers)
    assert list(st) == [1] * n
    want = bbs.core_sign(suite, sk, gens, headers[0], msgs[0], api_id)
    assert (sigs[0].a, sigs[0].e) == (want.a, want.e)
    assert list(eng.core_verify_batch(sigs, msgs, headers)) == [1] * n
    proofs, st = eng.core_proof_gen_batch(sigs, msgs, disclosed, rnds, headers, phs)
    assert list(st) == [1] * n
    for i in range(min(n, 2)):
        ws = bbs.core_sign(suite, sk, gens, headers[i], msgs[i], api_id)
        wp = bbs.core_proof_gen(suite, pk, ws, headers[i], gens, phs[i], msgs[i], disclosed[i], api_id, rnds[i])
        assert proof_eq(proofs[i], wp), (curve, i)
    dm = [[msgs[i][j] for j in disclosed[i]] for i in range(n)]
    assert list(eng.core_proof_verify_batch(proofs, dm, disclosed, headers, phs)) == [1] * n
    phs2 = list(phs)
    phs2[0] = phs2[0] + b"x"
    hd2 = list(headers)
    hd2[-1] = hd2[-1][:-1] if hd2[-1] else b"y"
    assert list(eng.core_proof_verify_batch(proofs, dm, disclosed, hd2, phs2))[0] == 0
    assert list(eng.core_proof_verify_batch(proofs, dm, disclosed, hd2, phs))[-1] == 0
    eng.close()


def check_proof_verify_octets(curve, lib_path=None, n=14, L=5, seed=61, disclose_all_3=False):
    """bbs_proof_verify_octets_*: proof OCTET strings in, statuses out, decoding on the device.  Against (a) the
    composition it replaces, bbs_proofs_from_octets_batch -> bbs_core_proof_verify_batch, item by item, and (b) the oracle:
    valid proofs, tampered scalars, every kind of malformed encoding (truncated string, missing compression flag,
    x with no square root, a point outside the subgroup, the identity, a scalar >= r) and the reference's own errors."""
    from bbs_sign_amd import api
    rng = random.Random(seed)
    suite = bbs.SUITES[curve]
    c = suite.curve
    api_id = suite.api_id
    gens = gens_for(suite, L + 1)
    sk = rng.randrange(1, c.r)
    pk = bbs.sk_to_pk(suite, sk)
    eng = make_engine(curve, gens, api_id, lib_path, sk=sk)
    msgs = [[rng.randrange(c.r) for _ in range(L)] for _ in range(n)]
    headers = [bytes(rng.randrange(256) for _ in range(rng.choice([0, 5, 70]))) for _ in range(n)]
    phs = [bytes(rng.randrange(256) for _ in range(rng.choice([0, 9]))) for _ in range(n)]
    disclosed = [sorted(rng.sample(range(L), rng.randrange(0, L + 1))) for _ in range(n)]
    if disclose_all_3:                                   # item 3 without commitments: see the "one scalar short" case below
        disclosed[3] = list(range(L))
    rnds = [[rng.randrange(1, c.r) for _ in range(5 + L - len(d))] for d in disclosed]
    sigs, st = eng.core_sign_batch(msgs, headers)
    proofs, st = eng.core_proof_gen_batch(sigs, msgs, disclosed, rnds, headers, phs)
    assert list(st) == [1] * n
    dm = [[msgs[i][j] for j in disclosed[i]] for i in range(n)]
    octs = [bytearray(api.proof_to_octets(curve, p_, lib_path)) for p_ in proofs]
    fpb = c.fp_bytes
    idx = [list(d) for d in disclosed]
    dmm = [list(d) for d in dm]
    # 1: e^ tampered (valid encoding, Ok(false)); 2: truncated by one byte; 3: truncated by one scalar (l != L: Err -1)
    octs[1][3 * fpb + 31] ^= 1
    octs[2] = octs[2][:-1]
    octs[3] = octs[3][:-32]
    # 4: compression flag missing (BLS) / infinity flag with a non-zero value (BN254) -> malformed
    if curve == "bls12_381":
        octs[4][0] &= 0x7F
    else:
        octs[4][fpb - 1] |= 0x40
    # 5: x of Bbar replaced by a value with no square root on the curve
    x = 7
    while pow((x ** 3 + c.b) % c.p, (c.p - 1) // 2, c.p) == 1:
        x += 1
    enc = bytearray(bbs.g1_compress(c, (x, 0)))          # flags + x; y is irrelevant: there is none
    octs[5][fpb:2 * fpb] = enc
    # 6: D = an on-curve point of order 3, outside the prime-order subgroup (BLS12-381 only)
    if curve == "bls12_381":
        octs[6][2 * fpb:3 * fpb] = bbs.g1_compress(c, (0, 2))
    # 7: Abar = the identity (rejected by octets_to_proof); 8: r1^ >= r; 9: challenge >= r
    octs[7][0:fpb] = bbs.g1_compress(c, None)
    octs[8][3 * fpb + 32:3 * fpb + 64] = (c.r + 3).to_bytes(32, "big")
    octs[9][-32:] = ((1 << 256) - 1).to_bytes(32, "big")
    # 10: disclosed index >= l; 11: duplicate disclosed index (only when something is disclosed)
    idx[10] = idx[10] + [L + 2]; dmm[10] = dmm[10] + [1]
    if idx[11]:
        idx[11] = idx[11] + [idx[11][0]]; dmm[11] = dmm[11] + [2]
    octs = [bytes(o) for o in octs]
    got = [int(x_) for x_ in eng.proof_verify_octets_batch(octs, dmm, idx, headers, phs)]
    # (a) the composition
    dec, dst = eng.proofs_from_octets_batch(octs)
    want = []
    for i in range(n):
        if dst[i] != 1:
            want.append(int(dst[i]))
        else:
            want.append(int(eng.core_proof_verify_batch([dec[i]], [dmm[i]], [idx[i]], [headers[i]], [phs[i]])[0]))
    assert got == want, (curve, got, want)
    # 3: one scalar short.  With undisclosed messages that is a proof for l - 1 messages (Err: l != L); with every message
    # disclosed there is no commitment to drop and the string falls below the floor of 3 points + 4 scalars: malformed
    want3 = (-42,) if len(disclosed[3]) == L else (-1, -3)
    assert got[0] == 1 and got[1] == 0 and got[2] == -42 and got[3] in want3 and got[4] == -40 and got[5] == -41, got
    assert got[7] == -42 and got[8] == -40 and got[9] == -40 and got[10] == -3 and got[12] == 1 and got[13] == 1, got
    if curve == "bls12_381":
        assert got[6] == -41, got
    # (b) the oracle on the well-formed ones
    for i in (0, 1, 12):
        p = dec[i]
        op = bbs.Proof(p.a_bar, p.b_bar, p.d, p.e_cap, p.r1_cap, p.r3_cap, list(p.commitments), p.challenge)
        assert int(bbs.core_proof_verify(suite, pk, op, gens, headers[i], phs[i], dmm[i], idx[i], api_id)) == got[i]
    # submit form, two batches in flight, and the reference's own proof vector through the wire path (BLS12-381)
    nn, keep, args = eng._oct_inputs(octs, dmm, idx, headers, phs)
    jobs = [eng.proof_verify_octets_submit_packed(nn, args) for _ in range(2)]
    for j in jobs:
        j.wait()
        assert [int(x_) for x_ in j.result] == want
        j.free()
    eng.close()
    if curve == "bls12_381":
        S = bbs.BLS_SUITE
        H = bytes.fromhex
        kat = H("94916292a7a6bade28456c601d3af33fcf39278d6594b467e128a3f83686a104ef2b2fcf72df0215eeaf69262ffe8194a19fab31a82ddbe06908985abc4c9825788b8a1610942d12b7f5debbea8985296361206dbace7af0cc834c80f33e0aadaeea5597befbb651827b5eed5a66f1a959bb46cfd5ca1a817a14475960f69b32c54db7587b5ee3ab665fbd37b506830a49f21d592f5e634f47cee05a025a2f8f94e73a6c15f02301d1178a92873b6e8634bafe4983c3e15a663d64080678dbf29417519b78af042be2b3e1c4d08b8d520ffab008cbaaca5671a15b22c239b38e940cfeaa5e72104576a9ec4a6fad78c532381aeaa6fb56409cef56ee5c140d455feeb04426193c57086c9b6d397d9418")
        ikm = H("746869732d49532d6a7573742d616e2d546573742d494b4d2d746f2d67656e65726174652d246528724074232d6b6579")
        key_info = H("746869732d49532d736f6d652d6b65792d6d657461646174612d746f2d62652d757365642d696e2d746573742d6b65792d67656e")
        key_dst = H("4242535f424c53313233383147315f584d443a5348412d3235365f535357555f524f5f4832475f484d32535f4b455947454e5f4453545f")
        sk2 = bbs.key_gen(S, ikm, key_info, key_dst)
        e2 = make_engine("bls12_381", bbs.create_generators(S, 2, S.api_id), S.api_id, lib_path, sk=sk2)
        m1 = bbs.msg_to_scalars(S, [H("9872ad089e452c7b6e283dfac2a80d58e8d0ff71cc4d5e310a1debdda4a45f02")], S.api_id)
        hdr, ph = H("11223344556677889900aabbccddeeff"), H("bed231d880675ed101ead304512e043ade9958dd0241ea70b4b3957fba941501")
        assert list(e2.proof_verify_octets_batch([kat], [m1], [[0]], [hdr], [ph])) == [1]      # test_vector.rs:199-260
        assert list(e2.proof_verify_octets_batch([kat], [m1], [[0]], [hdr], [ph + b"x"])) == [0]
        e2.close()


def check_verify_octets(curve, lib_path=None, n=16, L=4, seed=71):
    """bbs_verify_octets_*: signature OCTET strings in (compress(A) || e), statuses out, decoding on the device.  Against
    (a) the composition it replaces, bbs_signatures_from_octets_batch -> bbs_core_verify_batch, item by item, (b) the
    oracle on the decodable ones, (c) the reference's signature vector (src/tests/test_vector.rs:163-193)."""
    from bbs_sign_amd import api
    rng = random.Random(seed)
    suite = bbs.SUITES[curve]
    c = suite.curve
    api_id = suite.api_id
    gens = gens_for(suite, L + 1)
    sk = rng.randrange(1, c.r)
    pk = bbs.sk_to_pk(suite, sk)
    eng = make_engine(curve, gens, api_id, lib_path, sk=sk)
    fpb = c.fp_bytes
    msgs = [[rng.randrange(c.r) for _ in range(L)] for _ in range(n)]
    headers = [bytes(rng.randrange(256) for _ in range(rng.choice([0, 5, 70]))) for _ in range(n)]
    sigs, st = eng.core_sign_batch(msgs, headers)
    assert list(st) == [1] * n
    octs = [bytearray(api.signature_to_octets(curve, s_, lib_path)) for s_ in sigs]
    vm = [list(m) for m in msgs]
    vh = list(headers)
    vm[1][L - 1] = (vm[1][L - 1] + 1) % c.r                 # 1: forged message -> Ok(false)
    vh[2] = vh[2] + b"x"                                    # 2: forged header -> Ok(false)
    octs[3][fpb + 31] ^= 1                                  # 3: e altered (still < r) -> Ok(false)
    if curve == "bls12_381":                                # 4: compression flag missing / infinity flag with a value
        octs[4][0] &= 0x7F
    else:
        octs[4][fpb - 1] |= 0x40
    x = 7                                                   # 5: x with no square root on the curve
    while pow((x ** 3 + c.b) % c.p, (c.p - 1) // 2, c.p) == 1:
        x += 1
    octs[5][0:fpb] = bbs.g1_compress(c, (x, 0))
    if curve == "bls12_381":                                # 6: on the curve, order 3: outside the subgroup
        octs[6][0:fpb] = bbs.g1_compress(c, (0, 2))
    octs[7][0:fpb] = bbs.g1_compress(c, None)              # 7: A = identity (rejected by octets_to_signature)
    octs[8][fpb:] = (c.r + 2).to_bytes(32, "big")           # 8: e >= r
    octs[9][fpb:] = bytes(32)                               # 9: e = 0
    vm[10] = vm[10][:-1]                                    # 10: wrong message count (Err)
    vm[11][0] = c.r                                         # 11: message >= r
    octs[12][fpb:] = (c.r + 2).to_bytes(32, "big")          # 12: undecodable AND a wrong message count: the decoder's verdict
    vm[12] = vm[12] + [1]
    if curve == "bls12_381":                                # 13: another point of the curve in the subgroup: Ok(false)
        octs[13][0:fpb] = bbs.g1_compress(c, c.g1)
    octs = [bytes(o) for o in octs]
    got = [int(x_) for x_ in eng.verify_octets_batch(octs, vm, vh)]
    dec, dst = eng.signatures_from_octets_batch(octs)
    want = []
    for i in range(n):
        if dst[i] != 1:
            want.append(int(dst[i]))
        else:
            want.append(int(eng.core_verify_batch([dec[i]], [vm[i]], [vh[i]])[0]))
    assert got == want, (curve, got, want)
    assert got[0] == 1 and got[1] == 0 and got[2] == 0 and got[3] == 0 and got[4] == -40 and got[5] == -41 and got[7] == -42, got
    assert got[8] == -40 and got[9] == -42 and got[10] == -1 and got[11] == -40 and got[12] == -40 and got[14] == 1 and got[15] == 1, got
    if curve == "bls12_381":
        assert got[6] == -41 and got[13] == 0, got
    for i in (0, 1, 2, 3, 13, 14):
        if dst[i] == 1:
            assert int(bbs.core_verify(suite, pk, bbs.Signature(dec[i].a, dec[i].e), gens, vh[i], vm[i], api_id)) == got[i], i
    # the same under batch verification and with subgroup vouching (the decoder has checked membership either way)
    eng.set_batch_verification(True)
    assert [int(x_) for x_ in eng.verify_octets_batch(octs, vm, vh)] == want
    eng.set_batch_verification(False)
    eng.set_points_in_subgroup(True)
    assert [int(x_) for x_ in eng.verify_octets_batch(octs, vm, vh)] == want
    eng.set_points_in_subgroup(False)
    # submit form, two batches in flight; the empty batch
    jobs = [eng.verify_octets_submit(octs, vm, vh) for _ in range(2)]
    for j in jobs:
        j.wait()
        assert [int(x_) for x_ in j.result] == want
        j.free()
    assert list(eng.verify_octets_batch([], [], [])) == []
    # a string of the wrong length is malformed (the Python mirror pads it for the fixed-stride ABI and reports -42)
    short = [octs[0][:-1], octs[14], octs[15] + b"\x00"]
    assert list(eng.verify_octets_batch(short, [vm[0], vm[14], vm[15]], [vh[0], vh[14], vh[15]])) == [-42, 1, -42]
    j = eng.verify_octets_submit(short, [vm[0], vm[14], vm[15]], [vh[0], vh[14], vh[15]])
    j.wait()
    assert list(j.result) == [-42, 1, -42]
    j.free()
    eng.close()
    if curve == "bls12_381":
        S = bbs.BLS_SUITE
        H = bytes.fromhex
        kat = H("84773160b824e194073a57493dac1a20b667af70cd2352d8af241c77658da5253aa8458317cca0eae615690d55b1f27164657dcafee1d5c1973947aa70e2cfbb4c892340be5969920d0916067b4565a0")
        ikm = H("746869732d49532d6a7573742d616e2d546573742d494b4d2d746f2d67656e65726174652d246528724074232d6b6579")
        key_info = H("746869732d49532d736f6d652d6b65792d6d657461646174612d746f2d62652d757365642d696e2d746573742d6b65792d67656e")
        key_dst = H("4242535f424c53313233383147315f584d443a5348412d3235365f535357555f524f5f4832475f484d32535f4b455947454e5f4453545f")
        sk2 = bbs.key_gen(S, ikm, key_info, key_dst)
        e2 = make_engine("bls12_381", bbs.create_generators(S, 2, S.api_id), S.api_id, lib_path, sk=sk2)
        m1 = bbs.msg_to_scalars(S, [H("9872ad089e452c7b6e283dfac2a80d58e8d0ff71cc4d5e310a1debdda4a45f02")], S.api_id)
        hdr = H("11223344556677889900aabbccddeeff")
        assert list(e2.verify_octets_batch([kat], [m1], [hdr])) == [1]
        assert list(e2.verify_octets_batch([kat], [m1], [hdr + b"x"])) == [0]
        e2.close()


def check_threads(lib_path=None, threads=4, rounds=3, n=9, L=3, seed=81):
    """Several host threads on ONE configured context per curve at the same time (every call builds its own
    job; pools and the context's counters are locked): each thread signs, generates proofs, verifies -- submit forms and
    one-shot calls mixed -- on its own data, with a tampered item per batch; results against the oracle."""
    import threading
    engines, ctxs = {}, {}
    for curve in ("bls12_381", "bn254"):
        suite = bbs.SUITES[curve]
        rng = random.Random(seed)
        sk = rng.randrange(1, suite.curve.r)
        gens = gens_for(suite, L + 1)
        engines[curve] = make_engine(curve, gens, suite.api_id, lib_path, sk=sk)
        ctxs[curve] = (suite, sk, bbs.sk_to_pk(suite, sk), gens)
    errors = []

    def worker(t):
        try:
            rng = random.Random(seed * 100 + t)
            for rd in range(rounds):
                curve = ("bls12_381", "bn254")[(t + rd) % 2]
                eng = engines[curve]
                suite, sk, pk, gens = ctxs[curve]
                c = suite.curve
                msgs = [[rng.randrange(c.r) for _ in range(L)] for _ in range(n)]
                headers = [bytes(rng.randrange(256) for _ in range(rng.choice([0, 6]))) for _ in range(n)]
                disclosed = [sorted(rng.sample(range(L), rng.randrange(0, L + 1))) for _ in range(n)]
                rnds = [[rng.randrange(1, c.r) for _ in range(5 + L - len(d))] for d in disclosed]
                sj = eng.core_sign_submit(msgs, headers)
                sj.wait()
                sigs, st = sj.output()
                sj.free()
                assert list(st) == [1] * n
                w = bbs.core_sign(suite, sk, gens, headers[t % n], msgs[t % n], suite.api_id)
                assert sigs[t % n].a == w.a and sigs[t % n].e == w.e
                vm = [list(m) for m in msgs]
                vm[rd][0] = (vm[rd][0] + 1) % c.r
                vj = eng.core_verify_submit(sigs, vm, headers)
                proofs, st = eng.core_proof_gen_batch(sigs, msgs, disclosed, rnds, headers)
                assert list(st) == [1] * n
                wp = bbs.core_proof_gen(suite, pk, bbs.Signature(sigs[0].a, sigs[0].e), headers[0], gens, b"", msgs[0], disclosed[0],
                                        suite.api_id, rnds[0])
                assert proof_eq(proofs[0], wp)
                dm = [[msgs[i][j] for j in disclosed[i]] for i in range(n)]
                proofs[rd + 1].e_cap = (proofs[rd + 1].e_cap + 1) % c.r
                pj = eng.core_proof_verify_submit(proofs, dm, disclosed, headers)
                vj.wait(); pj.wait()
                assert [int(x) for x in vj.result] == [0 if i == rd else 1 for i in range(n)]
                assert [int(x) for x in pj.result] == [0 if i == rd + 1 else 1 for i in range(n)]
                vj.free(); pj.free()
        except BaseException as e:                      # noqa: BLE001 -- reported by the main thread
            import traceback
            errors.append((t, traceback.format_exc()))

    ths = [threading.Thread(target=worker, args=(t,)) for t in range(threads)]
    for th in ths:
        th.start()
    for th in ths:
        th.join()
    for e in engines.values():
        e.close()
    assert not errors, errors[0][1]


def check_fixed_base_tree(curve, lib_path=None, L=6, seed=91, window_bits=None, n_pv=10):
    """bbs_ctx_set_fixed_base_tree: the fixed-base sum as one tree of affine additions per item.  (a) Group elements of the
    MSM primitive against the oracle's plain sum, with generators chosen to hit every exceptional case of affine addition
    -- repeated generators (equal table entries: doubling), opposite generators (cancellation to the identity, at the
    leaves and higher up), zero digits and all-zero scalars (identity operands), odd and even numbers of terms; (b) the
    same proof_verify / verify / sign / proof_gen results as the chunked sums and the oracle."""
    rng = random.Random(seed)
    suite = bbs.SUITES[curve]
    c = suite.curve
    api_id = suite.api_id
    gens = gens_for(suite, L + 1)
    neg = lambda pt: (pt[0], (c.p - pt[1]) % c.p)
    # Q1, H1..HL with H2 = H1, H3 = -H1, H5 = -H4 (caller-supplied generators are arbitrary points)
    g2 = list(gens)
    g2[2] = g2[1]
    g2[3] = neg(g2[1])
    g2[5] = neg(g2[4])
    for gset in (gens, g2):
        bases = [suite.p1] + gset
        eng = make_engine(curve, gset, api_id, lib_path, sk=rng.randrange(1, c.r), window_bits=window_bits)
        eng.set_fixed_base_tree(True)
        top = (1 << 256) - 1
        rows = [[0] * (L + 2), [1] * (L + 2), [c.r - 1] * (L + 2), [top % c.r] * (L + 2)]
        rows.append([0, 0, 5, 5, 5, 0, 0, 0][:L + 2])                       # H1 + H2 (= 2 H1) + H3 (= -H1): doubling and cancellation
        rows.append([0, 0, 7, 0, 7, 0, 0, 0][:L + 2])                       # H1 + (-H1) = identity at the leaves, everything else zero
        rows.append([0, 0, 0, 0, 0, 9, 9, 0][:L + 2])                       # H4 + (-H4)
        rows.append([3, 0, 11, 11, 0, 0, 0, 2][:L + 2])
        rows += [[rng.randrange(c.r) for _ in range(L + 2)] for _ in range(6)]
        rows += [[rng.randrange(1 << 20) for _ in range(L + 2)] for _ in range(3)]       # mostly zero digits
        for nf in (L + 2, 1, 3):                                             # odd / even numbers of table points
            fs = [r_[:nf] for r_ in rows]
            out, st = eng.g1_msm_batch(fs, [], [])
            assert list(st) == [1] * len(fs)
            for i, row in enumerate(fs):
                want = None
                for k in range(nf):
                    want = c.g1_add(want, c.g1_mul(bases[k], row[k]))
                assert out[i] == want, (curve, gset is g2, nf, i, row)
        eng.close()
    # (b) the four operations: tree on vs off vs oracle
    sk = rng.randrange(1, c.r)
    pk = bbs.sk_to_pk(suite, sk)
    off = make_engine(curve, gens, api_id, lib_path, sk=sk, window_bits=window_bits)
    on = make_engine(curve, gens, api_id, lib_path, sk=sk, window_bits=window_bits)
    on.set_fixed_base_tree(True)
    n = n_pv
    msgs = [[rng.randrange(c.r) for _ in range(L)] for _ in range(n)]
    msgs[0] = [0] * L
    headers = [bytes(rng.randrange(256) for _ in range(rng.choice([0, 7, 66]))) for _ in range(n)]
    phs = [bytes(rng.randrange(256) for _ in range(rng.choice([0, 3]))) for _ in range(n)]
    disclosed = [sorted(rng.sample(range(L), rng.randrange(0, L + 1))) for _ in range(n)]
    rnds = [[rng.randrange(1, c.r) for _ in range(5 + L - len(d))] for d in disclosed]
    sigs, st = on.core_sign_batch(msgs, headers)
    sigs0, st0 = off.core_sign_batch(msgs, headers)
    assert list(st) == list(st0) == [1] * n
    for i in range(n):
        w = bbs.core_sign(suite, sk, gens, headers[i], msgs[i], api_id) if i < 3 else sigs0[i]
        assert (sigs[i].a, sigs[i].e) == (w.a, w.e) == (sigs0[i].a, sigs0[i].e), (curve, i, "sign")
    vm = [list(m) for m in msgs]
    vm[1][0] = (vm[1][0] + 1) % c.r
    assert list(on.core_verify_batch(sigs, vm, headers)) == list(off.core_verify_batch(sigs, vm, headers)) == [1] + [0] + [1] * (n - 2)
    proofs, st = on.core_proof_gen_batch(sigs, msgs, disclosed, rnds, headers, phs)
    proofs0, st0 = off.core_proof_gen_batch(sigs, msgs, disclosed, rnds, headers, phs)
    assert list(st) == list(st0) == [1] * n and all(proof_eq(a_, b_) for a_, b_ in zip(proofs, proofs0))
    assert proof_eq(proofs[2], bbs.core_proof_gen(suite, pk, bbs.Signature(sigs[2].a, sigs[2].e), headers[2], gens, phs[2], msgs[2],
                                                 disclosed[2], api_id, rnds[2]))
    dm = [[msgs[i][j] for j in disclosed[i]] for i in range(n)]
    bad = [to_engine_proof(p_) for p_ in proofs]
    bad[3].commitments = list(bad[3].commitments)
    if bad[3].commitments:
        bad[3].commitments[0] = (bad[3].commitments[0] + 1) % c.r
    else:
        bad[3].e_cap = (bad[3].e_cap + 1) % c.r
    bad[4].challenge = (bad[4].challenge + 1) % c.r
    got = list(on.core_proof_verify_batch(bad, dm, disclosed, headers, phs))
    assert got == list(off.core_proof_verify_batch(bad, dm, disclosed, headers, phs)) == [1, 1, 1, 0, 0] + [1] * (n - 5), got
    for i in (0, 3):
        op = bbs.Proof(bad[i].a_bar, bad[i].b_bar, bad[i].d, bad[i].e_cap, bad[i].r1_cap, bad[i].r3_cap, bad[i].commitments, bad[i].challenge)
        assert int(bbs.core_proof_verify(suite, pk, op, gens, headers[i], phs[i], dm[i], disclosed[i], api_id)) == got[i]
    j = on.core_proof_verify_submit(bad, dm, disclosed, headers, phs)
    j.wait()
    assert [int(x) for x in j.result] == got
    j.free()
    on.close(); off.close()


def check_octets_out(curve, lib_path=None, n=12, L=5, seed=95):
    """bbs_sign_octets_* / bbs_proof_gen_octets_*: the results as octet strings compressed on the device, against the
    host encoder applied to the records of the core_* calls (bbs_signature_to_octets / bbs_proof_to_octets, themselves
    pinned by the reference's vectors), failed items included, and back through the wire-form verifiers."""
    from bbs_sign_amd import api
    rng = random.Random(seed)
    suite = bbs.SUITES[curve]
    c = suite.curve
    api_id = suite.api_id
    gens = gens_for(suite, L + 1)
    sk = rng.randrange(1, c.r)
    eng = make_engine(curve, gens, api_id, lib_path, sk=sk)
    msgs = [[rng.randrange(c.r) for _ in range(L)] for _ in range(n)]
    headers = [bytes(rng.randrange(256) for _ in range(rng.choice([0, 9, 64]))) for _ in range(n)]
    phs = [bytes(rng.randrange(256) for _ in range(rng.choice([0, 5]))) for _ in range(n)]
    sm = [list(m) for m in msgs]
    sm[2] = sm[2][:-1]                                     # Err: wrong message count
    sm[5][1] = c.r + 9                                     # malformed scalar
    sigs, st = eng.core_sign_batch(sm, headers)
    octs, st2 = eng.sign_octets_batch(sm, headers)
    assert list(st) == list(st2) and st[2] < 0 and st[5] == -40 and list(st).count(1) == n - 2
    for i in range(n):
        want = api.signature_to_octets(curve, sigs[i], lib_path) if st[i] == 1 else b""
        assert octs[i] == want, (curve, i, "sign octets")
    good, st = eng.core_sign_batch(msgs, headers)
    disclosed = [sorted(rng.sample(range(L), rng.randrange(0, L + 1))) for _ in range(n)]
    disclosed[0] = list(range(L))                          # no commitments at all
    disclosed[1] = []                                      # every message undisclosed
    rnds = [[rng.randrange(1, c.r) for _ in range(5 + L - len(d))] for d in disclosed]
    ps = [Signature(s_.a, s_.e) for s_ in good]
    ps[3] = Signature(good[3].a, c.r + 1)                  # malformed e
    di = [list(d) for d in disclosed]
    di[4] = di[4] + [L + 1]; rn4 = rnds[4]                 # InvalidDisclosedIndex (the scalar count is not looked at then)
    proofs, st = eng.core_proof_gen_batch(ps, msgs, di, rnds, headers, phs)
    pocts, st2 = eng.proof_gen_octets_batch(ps, msgs, di, rnds, headers, phs)
    # item 4: one index too many and out of range -- r > l is the reference's FIRST check (proof_gen.rs:135-137, -2) when the
    # item already disclosed everything, else the index check (-3)
    want4 = -2 if len(di[4]) > L else -3
    assert list(st) == list(st2) and st[3] == -40 and st[4] == want4 and list(st).count(1) == n - 2, list(st)
    for i in range(n):
        want = api.proof_to_octets(curve, proofs[i], lib_path) if st[i] == 1 else b""
        assert pocts[i] == want, (curve, i, "proof octets", len(pocts[i]), len(want))
    # round trip through the wire-form verifiers
    ok = [i for i in range(n) if st[i] == 1]
    dm = [[msgs[i][j] for j in disclosed[i]] for i in ok]
    assert list(eng.proof_verify_octets_batch([pocts[i] for i in ok], dm, [disclosed[i] for i in ok], [headers[i] for i in ok],
                                              [phs[i] for i in ok])) == [1] * len(ok)
    so, st = eng.sign_octets_batch(msgs, headers)
    assert list(eng.verify_octets_batch(so, msgs, headers)) == [1] * n
    # two submits in flight, the empty batch
    jobs = [eng.proof_gen_octets_submit(ps, msgs, di, rnds, headers, phs), eng.sign_octets_submit(sm, headers)]
    for j in jobs:
        j.wait()
    assert jobs[0].output()[0] == pocts and jobs[1].output()[0] == octs
    for j in jobs:
        j.free()
    assert eng.sign_octets_batch([])[0] == [] and eng.proof_gen_octets_batch([], [], [], [])[0] == []
    eng.close()


def check_proof_verify_wire(curve, lib_path=None, n=12, L=5, seed=97):
    """bbs_proof_verify_wire_*: proof octet strings and the disclosed messages as RAW BYTES in, statuses out -- the
    reference's public proof_verify (src/proof_verify.rs:19-61) in one call.  Against (a) the composition it replaces,
    bbs_hash_to_scalar_batch -> bbs_proof_verify_octets_batch, item by item, (b) the oracle's public proof_verify where
    the context's generators are the suite's (BLS12-381), (c) the reference's proof vector with its raw message."""
    from bbs_sign_amd import api
    rng = random.Random(seed)
    suite = bbs.SUITES[curve]
    c = suite.curve
    api_id = suite.api_id
    gens = gens_for(suite, L + 1)
    sk = rng.randrange(1, c.r)
    pk = bbs.sk_to_pk(suite, sk)
    eng = make_engine(curve, gens, api_id, lib_path, sk=sk)
    raw = [[bytes(rng.randrange(256) for _ in range(rng.choice([0, 1, 32, 55, 56, 64, 200]))) for _ in range(L)] for _ in range(n)]
    flat = eng.hash_to_scalar_batch([m for item in raw for m in item], api_id + b"MAP_MSG_TO_SCALAR_AS_HASH_")
    msgs = [flat[i * L:(i + 1) * L] for i in range(n)]
    assert msgs[0] == bbs.msg_to_scalars(suite, raw[0], api_id)
    headers = [bytes(rng.randrange(256) for _ in range(rng.choice([0, 5, 70]))) for _ in range(n)]
    phs = [bytes(rng.randrange(256) for _ in range(rng.choice([0, 9]))) for _ in range(n)]
    disclosed = [sorted(rng.sample(range(L), rng.randrange(0, L + 1))) for _ in range(n)]
    disclosed[0] = []                                      # nothing disclosed: no message to hash
    disclosed[1] = list(range(L))
    rnds = [[rng.randrange(1, c.r) for _ in range(5 + L - len(d))] for d in disclosed]
    sigs, st = eng.core_sign_batch(msgs, headers)
    proofs, st = eng.core_proof_gen_batch(sigs, msgs, disclosed, rnds, headers, phs)
    assert list(st) == [1] * n
    octs = [bytearray(api.proof_to_octets(curve, p_, lib_path)) for p_ in proofs]
    idx = [list(d) for d in disclosed]
    draw = [[raw[i][j] for j in disclosed[i]] for i in range(n)]
    if draw[2]:
        draw[2][0] = draw[2][0] + b"!"                     # 2: a disclosed message altered -> Ok(false)
    else:
        octs[2][-1] ^= 1
    octs[3] = octs[3][:-1]                                 # 3: malformed string
    idx[4] = idx[4] + [L + 2]; draw[4] = draw[4] + [b"extra"]          # 4: index out of range
    draw[5] = draw[5] + [b"one too many"]                  # 5: more messages than indexes
    octs[6][3 * c.fp_bytes + 31] ^= 1                      # 6: e^ altered
    octs = [bytes(o) for o in octs]
    got = [int(x) for x in eng.proof_verify_wire_batch(octs, draw, idx, headers, phs)]
    # (a) the composition
    dsc = eng.hash_to_scalar_batch([m for item in draw for m in item], api_id + b"MAP_MSG_TO_SCALAR_AS_HASH_")
    dm, k = [], 0
    for item in draw:
        dm.append(dsc[k:k + len(item)]); k += len(item)
    want = [int(x) for x in eng.proof_verify_octets_batch(octs, dm, idx, headers, phs)]
    assert got == want, (curve, got, want)
    assert got[0] == 1 and got[1] == 1 and got[2] == 0 and got[3] == -42 and got[5] == -6 and got[6] == 0 and got[7:] == [1] * (n - 7), got
    assert got[4] in (-3, -1), got
    # (b) the oracle's public function (BLS12-381: the context's generators are create_generators(L + 1, api_id))
    if curve == "bls12_381":
        for i in (0, 1, 2, 7):
            p = proofs[i]
            op = bbs.Proof(p.a_bar, p.b_bar, p.d, p.e_cap, p.r1_cap, p.r3_cap, list(p.commitments), p.challenge)
            if i == 2 and not disclosed[2]:
                continue
            assert int(bbs.proof_verify(suite, pk, op, headers[i], phs[i], draw[i], idx[i])) == got[i], i
    # submit form, two in flight; the empty batch
    jobs = [eng.proof_verify_wire_submit(octs, draw, idx, headers, phs) for _ in range(2)]
    for j in jobs:
        j.wait()
        assert [int(x) for x in j.result] == want
        j.free()
    assert list(eng.proof_verify_wire_batch([], [], [])) == []
    eng.close()
    # a DST longer than 255 bytes: the reference's expand_message panics inside msg_to_scalars for items with messages
    long_id = b"x" * 240
    e3 = make_engine(curve, gens, long_id, lib_path, sk=sk)
    got3 = [int(x) for x in e3.proof_verify_wire_batch(octs[:2], draw[:2], idx[:2], headers[:2], phs[:2])]
    assert got3[1] == -23 and got3[0] != -23, got3          # item 0 discloses nothing: nothing is hashed for it
    e3.close()
    if curve == "bls12_381":
        S = bbs.BLS_SUITE
        H = bytes.fromhex
        kat = H("94916292a7a6bade28456c601d3af33fcf39278d6594b467e128a3f83686a104ef2b2fcf72df0215eeaf69262ffe8194a19fab31a82ddbe06908985abc4c9825788b8a1610942d12b7f5debbea8985296361206dbace7af0cc834c80f33e0aadaeea5597befbb651827b5eed5a66f1a959bb46cfd5ca1a817a14475960f69b32c54db7587b5ee3ab665fbd37b506830a49f21d592f5e634f47cee05a025a2f8f94e73a6c15f02301d1178a92873b6e8634bafe4983c3e15a663d64080678dbf29417519b78af042be2b3e1c4d08b8d520ffab008cbaaca5671a15b22c239b38e940cfeaa5e72104576a9ec4a6fad78c532381aeaa6fb56409cef56ee5c140d455feeb04426193c57086c9b6d397d9418")
        ikm = H("746869732d49532d6a7573742d616e2d546573742d494b4d2d746f2d67656e65726174652d246528724074232d6b6579")
        key_info = H("746869732d49532d736f6d652d6b65792d6d657461646174612d746f2d62652d757365642d696e2d746573742d6b65792d67656e")
        key_dst = H("4242535f424c53313233383147315f584d443a5348412d3235365f535357555f524f5f4832475f484d32535f4b455947454e5f4453545f")
        sk2 = bbs.key_gen(S, ikm, key_info, key_dst)
        e2 = make_engine("bls12_381", bbs.create_generators(S, 2, S.api_id), S.api_id, lib_path, sk=sk2)
        m1 = H("9872ad089e452c7b6e283dfac2a80d58e8d0ff71cc4d5e310a1debdda4a45f02")
        hdr, ph = H("11223344556677889900aabbccddeeff"), H("bed231d880675ed101ead304512e043ade9958dd0241ea70b4b3957fba941501")
        assert list(e2.proof_verify_wire_batch([kat], [[m1]], [[0]], [hdr], [ph])) == [1]      # test_vector.rs:199-260, raw message
        assert list(e2.proof_verify_wire_batch([kat], [[m1 + b"x"]], [[0]], [hdr], [ph])) == [0]
        e2.close()


def check_sign_verify_wire(curve, lib_path=None, n=10, L=4, seed=99):
    """bbs_sign_wire_* / bbs_verify_wire_*: the reference's public sign (src/sign.rs:32-60) and verify (src/verify.rs:18-50)
    in one call each -- raw messages in, signatures as octet strings out / in.  Against the oracle's public functions
    (BLS12-381: the context's generators are the suite's), the composition hash -> core call -> host encoder, and the
    reference's signature vector with its raw message."""
    from bbs_sign_amd import api
    rng = random.Random(seed)
    suite = bbs.SUITES[curve]
    c = suite.curve
    api_id = suite.api_id
    gens = gens_for(suite, L + 1)
    sk = rng.randrange(1, c.r)
    pk = bbs.sk_to_pk(suite, sk)
    eng = make_engine(curve, gens, api_id, lib_path, sk=sk)
    raw = [[bytes(rng.randrange(256) for _ in range(rng.choice([0, 3, 32, 64, 119, 120, 300]))) for _ in range(L)] for _ in range(n)]
    headers = [bytes(rng.randrange(256) for _ in range(rng.choice([0, 8, 65]))) for _ in range(n)]
    sraw = [list(m) for m in raw]
    sraw[2] = sraw[2][:-1]                                 # Err: one message short
    sraw[3] = sraw[3] + [b"extra"]                         # Err: one too many
    octs, st = eng.sign_wire_batch(sraw, headers)
    assert [int(x) for x in st] == [1, 1, -1, -1] + [1] * (n - 4), list(st)
    flat = eng.hash_to_scalar_batch([m for item in raw for m in item], api_id + b"MAP_MSG_TO_SCALAR_AS_HASH_")
    msgs = [flat[i * L:(i + 1) * L] for i in range(n)]
    sigs, st0 = eng.core_sign_batch(msgs, headers)
    for i in range(n):
        if st[i] == 1:
            assert octs[i] == api.signature_to_octets(curve, sigs[i], lib_path), (curve, i)
        else:
            assert octs[i] == b""
    if curve == "bls12_381":
        w = bbs.sign(suite, sk, raw[0], headers[0])
        assert octs[0] == bbs.g1_compress(c, w.a) + bbs.scalar_be(c, w.e)
    good = [api.signature_to_octets(curve, s_, lib_path) for s_ in sigs]
    vraw = [list(m) for m in raw]
    vraw[1][L - 1] = vraw[1][L - 1] + b"."                 # forged message -> Ok(false)
    vh = list(headers)
    vh[4] = vh[4] + b"x"                                   # forged header
    vo = list(good)
    vo[5] = vo[5][:-1] + bytes([vo[5][-1] ^ 1])            # e altered
    vo[6] = vo[6][:-2]                                     # malformed length
    vraw[7] = vraw[7][:-1]                                 # wrong count
    got = [int(x) for x in eng.verify_wire_batch(vo, vraw, vh)]
    assert got == [1, 0, 1, 1, 0, 0, -42, -1] + [1] * (n - 8), got
    if curve == "bls12_381":
        for i in (0, 1, 4):
            assert int(bbs.verify(suite, pk, bbs.Signature(sigs[i].a, sigs[i].e), vh[i], vraw[i])) == got[i], i
    assert eng.sign_wire_batch([])[0] == [] and list(eng.verify_wire_batch([], [])) == []
    # item offsets that do not start at zero (a window into a larger list): items 2 .. 4 of the same arrays
    mb_, mbo_, mio_ = eng._raw_msgs(raw)
    ob_, _ = eng._sig_octets(good)
    hb_, ho_ = _engine_ragged(headers)
    st_w = np.full(3, -128, dtype=np.int8)
    rec_o = c.fp_bytes + 32
    rc = eng.lib.bbs_verify_wire_batch(eng.h, 3, ob_[2 * rec_o:].ctypes.data_as(_lib.c_u8p), mb_.ctypes.data_as(_lib.c_u8p),
                                       mbo_.ctypes.data_as(_lib.c_u64p), mio_[2:].ctypes.data_as(_lib.c_u64p),
                                       hb_.ctypes.data_as(_lib.c_u8p), ho_[2:].ctypes.data_as(_lib.c_u64p), st_w.ctypes.data_as(_lib.c_i8p))
    assert rc == 0 and list(st_w) == [1, 1, 1], (rc, list(st_w))
    # public proof_gen on the wire: signature octets + raw messages in, proof octets out; then the public proof_verify
    disclosed = [sorted(rng.sample(range(L), rng.randrange(0, L + 1))) for _ in range(n)]
    rnds = [[rng.randrange(1, c.r) for _ in range(5 + L - len(d))] for d in disclosed]
    phs = [bytes(rng.randrange(256) for _ in range(rng.choice([0, 6]))) for _ in range(n)]
    po = list(good)
    po[3] = po[3][:c.fp_bytes] + (c.r + 4).to_bytes(32, "big")           # e >= r: malformed signature
    po[4] = bbs.g1_compress(c, None) + po[4][c.fp_bytes:]               # A = identity
    pocts, pst = eng.proof_gen_wire_batch(po, raw, disclosed, rnds, headers, phs)
    assert [int(x) for x in pst] == [1, 1, 1, -40, -42] + [1] * (n - 5), list(pst)
    cproofs, cst = eng.core_proof_gen_batch(sigs, msgs, disclosed, rnds, headers, phs)
    for i in range(n):
        assert pocts[i] == (api.proof_to_octets(curve, cproofs[i], lib_path) if pst[i] == 1 else b""), (curve, i)
    if curve == "bls12_381":
        w = bbs.proof_gen(suite, pk, bbs.Signature(sigs[0].a, sigs[0].e), headers[0], phs[0], raw[0], disclosed[0], rnds[0])
        assert pocts[0] == api.proof_to_octets(curve, Proof(w.a_bar, w.b_bar, w.d, w.e_cap, w.r1_cap, w.r3_cap, list(w.commitments), w.challenge), lib_path)
    okk = [i for i in range(n) if pst[i] == 1]
    assert list(eng.proof_verify_wire_batch([pocts[i] for i in okk], [[raw[i][j] for j in disclosed[i]] for i in okk],
                                            [disclosed[i] for i in okk], [headers[i] for i in okk], [phs[i] for i in okk])) == [1] * len(okk)
    eng.close()
    e3 = make_engine(curve, gens, b"y" * 235, lib_path, sk=sk)       # DST of msg_to_scalars longer than 255 bytes
    o3, st3 = e3.sign_wire_batch(raw[:2], headers[:2])
    assert [int(x) for x in st3] == [-23, -23] and o3 == [b"", b""]
    e3.close()
    if curve == "bls12_381":
        S = bbs.BLS_SUITE
        H = bytes.fromhex
        ikm = H("746869732d49532d6a7573742d616e2d546573742d494b4d2d746f2d67656e65726174652d246528724074232d6b6579")
        key_info = H("746869732d49532d736f6d652d6b65792d6d657461646174612d746f2d62652d757365642d696e2d746573742d6b65792d67656e")
        key_dst = H("4242535f424c53313233383147315f584d443a5348412d3235365f535357555f524f5f4832475f484d32535f4b455947454e5f4453545f")
        sk2 = bbs.key_gen(S, ikm, key_info, key_dst)
        e2 = make_engine("bls12_381", bbs.create_generators(S, 2, S.api_id), S.api_id, lib_path, sk=sk2)
        m1 = H("9872ad089e452c7b6e283dfac2a80d58e8d0ff71cc4d5e310a1debdda4a45f02")
        hdr = H("11223344556677889900aabbccddeeff")
        o, st = e2.sign_wire_batch([[m1]], [hdr])                    # src/tests/test_vector.rs:163-193, raw message in, octets out
        assert list(st) == [1] and o[0].hex() == ("84773160b824e194073a57493dac1a20b667af70cd2352d8af241c77658da5253aa8458317cca0eae615690d55b1f271"
                                                  "64657dcafee1d5c1973947aa70e2cfbb4c892340be5969920d0916067b4565a0")
        assert list(e2.verify_wire_batch(o, [[m1]], [hdr])) == [1]
        assert list(e2.verify_wire_batch(o, [[m1 + b"x"]], [hdr])) == [0]
        e2.close()



def check_issuer_mixed_lengths(curve, lib_path=None, seed=131, lengths=(3, 0, 1, 5, 3, 2, 5, 1, 7, 3), oracle_items=(0, 1, 3, 5), window_bits="fixed"):
    """bbs_issuer_*: ONE call over items whose numbers of messages differ -- every item gets the generators of its own
    length, as the reference's public functions do (create_generators(messages.len() + 1): src/sign.rs:44-49,
    src/verify.rs:30-35, src/proof_gen.rs:91-96; commitments + disclosed indexes + 1: src/proof_verify.rs:40-43).
    Signatures and proofs byte for byte against the oracle's PUBLIC sign / proof_gen, booleans against its public verify /
    proof_verify (a subset: pairings in pure Python are slow), incl. forged items, a malformed proof string and an item
    longer than the issuer's limit."""
    from bbs_sign_amd import Issuer, api
    rng = random.Random(seed)
    suite = bbs.SUITES[curve]
    c = suite.curve
    n = len(lengths)
    sk = rng.randrange(1, c.r)
    pk = bbs.sk_to_pk(suite, sk)
    # window_bits = 0: the library picks the width of every context's tables from the free device memory
    iss = Issuer(curve, suite.api_id, lib_path=lib_path, max_messages=6, window_bits=(4 if lib_path else 8) if window_bits == "fixed" else window_bits)
    iss.set_secret_key(sk)
    raw = [[bytes(rng.randrange(256) for _ in range(rng.choice([0, 5, 32, 70]))) for _ in range(L)] for L in lengths]
    headers = [bytes(rng.randrange(256) for _ in range(rng.choice([0, 9, 66]))) for _ in range(n)]
    too_long = [i for i, L in enumerate(lengths) if L > 6]
    assert too_long, "the case needs an item above the issuer's limit"
    # ---- sign
    octs, st = iss.sign(raw, headers)
    assert [int(x) for x in st] == [-1 if i in too_long else 1 for i in range(n)], list(st)
    sigs = []
    for i in range(n):
        if i in too_long:
            assert octs[i] == b""
            sigs.append(None)
            continue
        w = bbs.sign(suite, sk, raw[i], headers[i])
        sigs.append(w)
        assert octs[i] == api.signature_to_octets(curve, Signature(w.a, w.e), lib_path), (curve, "sign", i)
    assert iss.context_count() == len({L for L in lengths if L <= 6})
    # ---- verify: forged message / header / e, one malformed string
    vo = [o if o else bytes(c.fp_bytes + 32) for o in octs]
    vraw = [list(m) for m in raw]
    vh = list(headers)
    i_fm = next(i for i in range(n) if lengths[i] >= 1 and i not in (4, 5, 6) and i not in too_long)   # an item with a message to forge
    vraw[i_fm][0] = vraw[i_fm][0] + b"!"
    vh[4] = vh[4] + b"x"
    vo[5] = vo[5][:-1] + bytes([vo[5][-1] ^ 1])
    vo[6] = vo[6][:-3]
    got = [int(x) for x in iss.verify(vo, vraw, vh)]
    want = [1] * n
    want[i_fm] = want[4] = want[5] = 0
    want[6] = -42
    for i in too_long:
        want[i] = -1
    assert got == want, (got, want)
    for i in (0, 1, 3, 4):                                                 # (items whose signature string is untouched)
        if i not in too_long:
            assert int(bbs.verify(suite, pk, sigs[i], vh[i], vraw[i])) == got[i], (curve, "verify", i)
    # ---- proof_gen
    ok = [i for i in range(n) if i not in too_long]
    disclosed = [sorted(rng.sample(range(L), rng.randrange(0, L + 1))) if L else [] for L in lengths]
    rnds = [[rng.randrange(1, c.r) for _ in range(5 + L - len(d))] for L, d in zip(lengths, disclosed)]
    phs = [bytes(rng.randrange(256) for _ in range(rng.choice([0, 7]))) for _ in range(n)]
    so = [o if o else bytes(c.fp_bytes + 32) for o in octs]
    poct, st = iss.proof_gen(so, raw, disclosed, rnds, headers, phs)
    assert [int(x) for x in st] == [-1 if i in too_long else 1 for i in range(n)], list(st)
    proofs = {}
    for i in ok:
        w = bbs.proof_gen(suite, pk, sigs[i], headers[i], phs[i], raw[i], disclosed[i], rnds[i])
        proofs[i] = w
        assert poct[i] == api.proof_to_octets(curve, Proof(w.a_bar, w.b_bar, w.d, w.e_cap, w.r1_cap, w.r3_cap, list(w.commitments), w.challenge), lib_path), (curve, "proof_gen", i)
    # ---- proof_verify: the message count of an item comes from its own octet string and index list
    dm = [[raw[i][j] for j in disclosed[i]] for i in range(n)]
    vp = [p if p else bytes(3 * c.fp_bytes + 128) for p in poct]
    vdm = [list(m) for m in dm]
    vph = list(phs)
    vph[0] = vph[0] + b"?"                                                 # forged presentation header -> Ok(false)
    i_msg = next((i for i in ok if disclosed[i] and i not in (0,)), None)
    if i_msg is not None:
        vdm[i_msg][0] = vdm[i_msg][0] + b"~"                               # forged disclosed message -> Ok(false)
    i_cut = next(i for i in ok if i not in (0, i_msg))
    vp[i_cut] = vp[i_cut][:-5]                                             # malformed: no message count can be read
    long_i = too_long[0]
    vp[long_i] = bytes(3 * c.fp_bytes + 32 * (4 + 7))                      # 7 commitments: above the limit
    vdisc = list(disclosed)
    vdisc[long_i] = []
    vdm[long_i] = []
    got = [int(x) for x in iss.proof_verify(vp, vdm, vdisc, headers, vph)]
    want = [1] * n
    want[0] = 0
    if i_msg is not None:
        want[i_msg] = 0
    want[i_cut] = -42
    want[long_i] = -1
    assert got == want, (got, want)
    for i in oracle_items:
        if i in (i_cut, long_i):
            continue
        assert int(bbs.proof_verify(suite, pk, proofs[i], headers[i], vph[i], vdm[i], disclosed[i])) == got[i], (curve, "proof_verify", i)
    # the asynchronous form: three lists in flight (the second one is the tampered list), the same statuses
    n_a, keep_a, args_a = iss.pack_proof_verify(vp, vdm, vdisc, headers, vph)
    gp = [p if p else bytes(3 * c.fp_bytes + 128) for p in poct]
    n_b, keep_b, args_b = iss.pack_proof_verify([gp[i] for i in ok], [dm[i] for i in ok], [disclosed[i] for i in ok],
                                                [headers[i] for i in ok], [phs[i] for i in ok])
    jobs = [iss.proof_verify_submit_packed(n_b, args_b), iss.proof_verify_submit_packed(n_a, args_a), iss.proof_verify_submit_packed(n_b, args_b)]
    for j in jobs:
        j.wait()
    assert [int(x) for x in jobs[0].result] == [1] * len(ok) and [int(x) for x in jobs[2].result] == [1] * len(ok)
    assert [int(x) for x in jobs[1].result] == want
    for j in jobs:
        j.free()
    # empty calls
    assert list(iss.proof_verify([], [], [])) == [] and list(iss.verify([], [])) == []
    iss.close()


def check_issuer_threads(curve, lib_path=None, threads=3, rounds=2, seed=151):
    """One issuer used from several host threads at once, every thread with its own mix of lengths (so that contexts are
    created concurrently on first use): the signatures verify, a forged one does not, the statuses are per item."""
    import threading
    from bbs_sign_amd import Issuer
    suite = bbs.SUITES[curve]
    iss = Issuer(curve, suite.api_id, lib_path=lib_path, max_messages=8, window_bits=4 if lib_path else 8)
    iss.set_secret_key(77 + seed)
    errors = []

    def worker(t):
        try:
            rng = random.Random(seed + t)
            for r in range(rounds):
                lens = [(t + r + k) % 5 for k in range(6)]
                raw = [[bytes(rng.randrange(256) for _ in range(rng.choice([1, 20, 65]))) for _ in range(L)] for L in lens]
                octs, st = iss.sign(raw)
                assert [int(x) for x in st] == [1] * len(lens), list(st)
                bad = list(octs)
                bad[2] = bad[2][:-1] + bytes([bad[2][-1] ^ 1])
                got = [int(x) for x in iss.verify(bad, raw)]
                assert got == [0 if i == 2 else 1 for i in range(len(lens))], got
        except Exception as e:                                   # noqa: BLE001
            errors.append((t, repr(e)))

    ts = [threading.Thread(target=worker, args=(t,)) for t in range(threads)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    assert iss.context_count() == 5
    iss.close()


def check_issuer_budget(curve, lib_path=None, seed=171):
    """bbs_issuer's resident contexts are BOUNDED (bbs_issuer_set_budget): message counts keep arriving -- in proof_verify they
    are read from untrusted octet strings -- and idle contexts leave least-recently-used first; a length that comes back is
    rebuilt and gives the same signatures (byte for byte against the oracle's public sign, src/sign.rs:32-60).  A call whose
    groups need more contexts than the limit allows fails only the group that found no slot (BBS_ST_NO_RESOURCES, -43:
    not computed) -- the others are served.  Configuration calls are refused (BBS_E_STATE) while a routed list is in flight
    and take effect for every context, resident or rebuilt, afterwards."""
    from bbs_sign_amd import Issuer, api
    from bbs_sign_amd.engine import BbsRuntimeError
    rng = random.Random(seed)
    suite = bbs.SUITES[curve]
    c = suite.curve
    sk = rng.randrange(1, c.r)
    iss = Issuer(curve, suite.api_id, lib_path=lib_path, max_messages=8, window_bits=4 if lib_path else 8)
    iss.set_secret_key(sk)
    iss.set_budget(2, 0)

    def msgs_of(L):
        return [bytes(rng.randrange(256) for _ in range(rng.choice([1, 9, 40]))) for _ in range(L)]

    def want_sig(raw):
        w = bbs.sign(suite, sk, raw, b"")
        return api.signature_to_octets(curve, Signature(w.a, w.e), lib_path)

    # lengths 1, 2, 3, 1, 4 one call each: never more than two contexts resident, every signature right
    first = {}
    for L in (1, 2, 3, 1, 4, 2):
        raw = [msgs_of(L), msgs_of(L)]
        octs, st = iss.sign(raw)
        assert [int(x) for x in st] == [1, 1], (L, list(st))
        assert octs[0] == want_sig(raw[0]) and octs[1] == want_sig(raw[1]), (curve, L)
        assert iss.context_count() <= 2, iss.context_count()
        first.setdefault(L, (raw[0], octs[0]))
    assert iss.table_bytes() > 0
    # verify with rebuilt contexts: the signatures made before the evictions still verify
    for L, (raw0, o0) in first.items():
        assert [int(x) for x in iss.verify([o0], [raw0])] == [1], L
        assert iss.context_count() <= 2
    # one call with THREE lengths at a limit of two: the groups are served in order of message count; the last finds every
    # resident context busy (pinned by this very list) -> its items are -43, the others are computed
    raw = [msgs_of(3), msgs_of(1), msgs_of(2), msgs_of(1), msgs_of(3)]
    octs, st = iss.sign(raw)
    assert [int(x) for x in st] == [-43, 1, 1, 1, -43], list(st)
    assert octs[0] == b"" and octs[4] == b"" and octs[1] == want_sig(raw[1]) and octs[2] == want_sig(raw[2])
    # ... and the same list passes once the limit allows three
    iss.set_budget(3, 0)
    octs, st = iss.sign(raw)
    assert [int(x) for x in st] == [1] * 5 and octs[0] == want_sig(raw[0])
    assert iss.context_count() == 3
    # shrinking the budget evicts idle contexts at once
    iss.set_budget(1, 0)
    assert iss.context_count() == 1
    iss.set_budget(4, 0)
    # configuration while a routed list is in flight: refused; afterwards: applied to resident and future contexts
    sig1 = first[1][1]
    n_a, keep_a, args_a = iss.pack_proof_verify([], [], [])
    octs2, st2 = iss.sign([first[2][0]])
    vjob = None
    try:
        import ctypes
        ob, bad = iss._sig_octets([sig1])
        # (a routed verify in flight through the submit form)
        from bbs_sign_amd.engine import _u8, _u64, _ragged_bytes, Engine
        mb, mbo, mio = Engine._raw_msgs([first[1][0]])
        hb, ho = _ragged_bytes([b""])
        stv = np.full(1, -128, dtype=np.int8)
        j = ctypes.c_void_p()
        rc = iss.lib.bbs_issuer_verify_submit(iss.h, 1, _u8(ob), _u8(mb), _u64(mbo), _u64(mio), _u8(hb), _u64(ho), stv.ctypes.data_as(_lib.c_i8p), ctypes.byref(j))
        assert rc == 0
        vjob = j
        try:
            iss.set_secret_key(sk + 1)
        except BbsRuntimeError as e:
            assert e.rc == -102, e.rc
        else:
            raise AssertionError("set_secret_key was accepted while a routed list was in flight")
        assert iss.lib.bbs_issuer_job_wait(vjob) == 0 and int(stv[0]) == 1
    finally:
        if vjob is not None:
            iss.lib.bbs_issuer_job_free(vjob)
    iss.set_secret_key(sk + 1)                                # now idle: accepted; contexts pick the key up at their next use
    assert [int(x) for x in iss.verify([sig1], [first[1][0]])] == [0]          # the old signature is not valid under the new key
    o_new, st_new = iss.sign([first[1][0]])
    w = bbs.sign(suite, sk + 1, first[1][0], b"")
    assert list(st_new) == [1] and o_new[0] == api.signature_to_octets(curve, Signature(w.a, w.e), lib_path)
    # a context handed out raw (bbs_issuer_context, e.g. to warm it up) is never evicted, and does not block configuration
    iss.warm(5)
    iss.set_budget(1, 0)
    assert iss.context_count() == 1                           # the warmed one; everything else was idle and left
    iss.set_secret_key(sk + 2)
    m5, m2 = msgs_of(5), msgs_of(2)
    o5, st5 = iss.sign([m5, m2])
    assert [int(x) for x in st5] == [1, -43], list(st5)       # the one slot is taken by the context that cannot leave
    w5 = bbs.sign(suite, sk + 2, m5, b"")
    assert o5[0] == api.signature_to_octets(curve, Signature(w5.a, w5.e), lib_path) and o5[1] == b""
    iss.close()


def check_proof_gen_unusual_points(curve, lib_path=None, seed=181, L=5):
    """core_proof_gen (src/proof_gen.rs:116-208) with signature points A that the fast paths of the variable-base parts must
    hand to their fallbacks -- the identity, an on-curve point of order 3 (outside the prime-order subgroup; BLS12-381) --
    beside ordinary items in the SAME batch (a wavefront then runs the comb over 64-bit pieces, the joint chains and the
    separate multiplications side by side), and scalars whose 64-bit pieces are 0, 1 and all ones.
    Every proof of an item whose A is in the prime-order subgroup (or the identity) equals the oracle's, which multiplies by
    plain double-and-add in the reference's operation order.  For A OUTSIDE the subgroup the engine's documented
    restructuring -- Abar e and Abar e~ as (r1 r2 e mod r) A and (r1 r2 e~ mod r) A -- is not the reference's
    integer product (the point's order does not divide r); the reference's typed interface cannot hold such a point
    (ark-serialize checks subgroup membership), and the case pins what the engine computes instead: the same formulas
    evaluated with the oracle's group law, identical in all three layouts of the job."""
    rng = random.Random(seed)
    suite = bbs.SUITES[curve]
    c = suite.curve
    api_id = suite.api_id
    gens = gens_for(suite, L + 1)
    sk = rng.randrange(1, c.r)
    pk = bbs.sk_to_pk(suite, sk)
    eng = make_engine(curve, gens, api_id, lib_path, sk=sk)
    n = 7
    msgs = [[rng.randrange(c.r) for _ in range(L)] for _ in range(n)]
    headers = [bytes(rng.randrange(256) for _ in range(rng.choice([0, 6]))) for _ in range(n)]
    phs = [bytes(rng.randrange(256) for _ in range(rng.choice([0, 3]))) for _ in range(n)]
    disclosed = [sorted(rng.sample(range(L), rng.randrange(0, L + 1))) for _ in range(n)]
    rnds = [[rng.randrange(1, c.r) for _ in range(5 + L - len(d))] for d in disclosed]
    sigs, st = eng.core_sign_batch(msgs, headers)
    assert list(st) == [1] * n
    sigs = list(sigs)
    sigs[1] = Signature(None, sigs[1].e)                               # A = identity
    outside = []
    if curve == "bls12_381":
        assert c.g1_is_on_curve((0, 2)) and c.g1_mul((0, 2), 3) is None
        sigs[4] = Signature((0, 2), sigs[4].e)                         # A of order 3
        outside = [4]
    sigs[5] = Signature(c.g1_neg(sigs[5].a), 1)                        # e = 1, another ordinary point
    # scalars with zero and all-ones 64-bit pieces (the comb's recoding of every piece)
    rnds[2][0] = (1 << 64)                                             # r1 = 2^64: pieces (0, 1, 0, 0)
    rnds[2][1] = (1 << 192) + 1
    rnds[3][2] = ((1 << 64) - 1) << 64                                 # e~ with an all-ones piece
    rnds[6][3] = c.r - 1
    proofs, st = eng.core_proof_gen_batch(sigs, msgs, disclosed, rnds, headers, phs)
    assert list(st) == [1] * n, list(st)
    for i in range(n):
        if i in outside:
            continue
        want = bbs.core_proof_gen(suite, pk, bbs.Signature(sigs[i].a, sigs[i].e), headers[i], gens, phs[i], msgs[i], disclosed[i], api_id, rnds[i])
        assert proof_eq(proofs[i], want), (curve, i)
    for i in outside:
        # the three points that do not depend on the challenge, by the engine's formulas with the oracle's group law
        r1, r2 = rnds[i][0], rnds[i][1]
        domain = bbs.calculate_domain(suite, pk, gens[0], gens[1:], headers[i], api_id)
        B = c.g1_add(suite.p1, c.g1_mul(gens[0], domain))
        for g, m in zip(gens[1:], msgs[i]):
            B = c.g1_add(B, c.g1_mul(g, m))
        A, e = sigs[i].a, sigs[i].e
        k = r1 * r2 % c.r
        assert proofs[i].d == c.g1_mul(B, r2) and proofs[i].a_bar == c.g1_mul(A, k)
        assert proofs[i].b_bar == c.g1_add(c.g1_mul(B, k), c.g1_neg(c.g1_mul(A, k * e % c.r)))
    eng.close()
