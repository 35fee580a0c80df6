#!/usr/bin/env python3
"""Generates tests/golden/bbs_golden.json with the oracle (oracle/), AFTER the oracle has passed
the reference's known-answer vectors (tests/test_oracle_kat.py).

The reference is Rust and cannot run in this image, so these are not reference outputs: they are
outputs of the KAT-pinned oracle, committed so that (a) the oracle itself cannot drift silently,
(b) the GPU parity tests have fixed expected bytes at the BASELINE shape (L = 32, R = 8) and at the
reference's own round-trip shapes (src/tests/bbs_over_bls_tests.rs:41-48), for both curves.

    python tests/golden/make_golden.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import bbs  # noqa: E402
from oracle.hashing import expand_message, i2osp  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "bbs_golden.json")

# (L, disclosed, header) -- bbs_over_bls_tests.rs:41-48 plus the BASELINE shape
CASES = [
    (0, [], b""), (0, [], b"abc"), (1, [0], b"abc"), (1, [], b"abc"),
    (10, [0, 1, 2], b""), (10, [0, 4, 7, 9], b"def"), (5, [0, 4], b"defghjsdjdbcjbejd"),
    (5, [0, 1, 2, 3, 4], b"def"), (32, list(range(8)), b""),
]


def hx(v, n):
    return int(v).to_bytes(n, "big").hex()


def pt(c, p):
    return None if p is None else [hx(p[0], c.fp_bytes), hx(p[1], c.fp_bytes)]


def pt2(c, q):
    return None if q is None else [[hx(q[0][0], c.fp_bytes), hx(q[0][1], c.fp_bytes)],
                                   [hx(q[1][0], c.fp_bytes), hx(q[1][1], c.fp_bytes)]]


def main():
    doc = {"note": "oracle-generated (see make_golden.py); integers are big-endian hex", "suites": {}}
    for name, suite in bbs.SUITES.items():
        c = suite.curve
        api_id = suite.api_id
        # benches/proof_verify.rs:118-121 key: IKM [1u8;32], empty key_info
        sk = bbs.key_gen(suite, bytes([1] * 32), b"", b"BBS-SIG-KEYGEN-SALT-")
        pk = bbs.sk_to_pk(suite, sk)
        entry = {"api_id": api_id.hex(), "sk": hx(sk, 32), "pk": pt2(c, pk),
                 "pk_compressed": bbs.g2_compress(c, pk).hex(), "cases": []}
        for ci, (L, disclosed, header) in enumerate(CASES):
            if name == "bls12_381":
                gens = bbs.create_generators(suite, L + 1, api_id)
            else:
                gens = bbs.synthetic_generators(suite, L + 1)
            msgs_bytes = [expand_message(b"bbs-golden-msg" + i2osp(ci, 8) + i2osp(j, 8), b"BBS_GOLDEN_MSG_DST_", 32)
                          for j in range(L)]
            msgs = bbs.msg_to_scalars(suite, msgs_bytes, api_id)
            ph = b"" if ci % 2 == 0 else b"presentation-header-%d" % ci
            rnd = bbs.seeded_random_scalars(suite, b"bbs-golden-rnd" + i2osp(ci, 8), api_id + b"MOCK_RANDOM_SCALARS_DST_",
                                            5 + L - len(disclosed))
            sig = bbs.core_sign(suite, sk, gens, header, msgs, api_id)
            assert bbs.core_verify(suite, pk, sig, gens, header, msgs, api_id)
            proof = bbs.core_proof_gen(suite, pk, sig, header, gens, ph, msgs, disclosed, api_id, rnd)
            assert bbs.core_proof_verify(suite, pk, proof, gens, header, ph, [msgs[i] for i in disclosed], disclosed, api_id)
            dom = bbs.calculate_domain(suite, pk, gens[0], gens[1:], header, api_id)
            entry["cases"].append({
                "L": L, "disclosed": disclosed, "header": header.hex(), "ph": ph.hex(),
                "generators": [pt(c, g) for g in gens],
                "messages": [hx(m, 32) for m in msgs],
                "random_scalars": [hx(s, 32) for s in rnd],
                "domain": hx(dom, 32),
                "signature": {"a": pt(c, sig.a), "e": hx(sig.e, 32),
                              "a_compressed": bbs.g1_compress(c, sig.a).hex()},
                "proof": {"a_bar": pt(c, proof.a_bar), "b_bar": pt(c, proof.b_bar), "d": pt(c, proof.d),
                          "e_cap": hx(proof.e_cap, 32), "r1_cap": hx(proof.r1_cap, 32), "r3_cap": hx(proof.r3_cap, 32),
                          "commitments": [hx(x, 32) for x in proof.commitments], "challenge": hx(proof.challenge, 32)},
            })
            print(name, "case", ci, "L", L, "ok", flush=True)
        doc["suites"][name] = entry
    # BN254 create_generators (Shallue-van de Woestijne hash-to-G1, pinned by the reference's P1 constant):
    # the first 11 generators of the ciphersuite, and P1 re-derived from its seed
    bn = bbs.BN_SUITE
    g = bbs.create_generators(bn, 11, bn.api_id)
    v = expand_message(bn.api_id + b"BP_MESSAGE_GENERATOR_SEED", bn.api_id + b"SIG_GENERATOR_SEED_", 48)
    v = expand_message(v + i2osp(1, 8), bn.api_id + b"SIG_GENERATOR_SEED_", 48)
    from oracle.hashing import hash_to_g1_bn
    assert hash_to_g1_bn(v, bn.api_id + b"SIG_GENERATOR_DST_") == bn.p1
    doc["bn254_create_generators"] = {"api_id": bn.api_id.hex(), "generators": [pt(bn.curve, x) for x in g],
                                      "p1": pt(bn.curve, bn.p1)}
    with open(OUT, "w") as f:
        json.dump(doc, f, indent=0)
    print("wrote", OUT)


if __name__ == "__main__":
    main()
