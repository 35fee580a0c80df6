"""The oracle reproduces the committed golden fixtures (guards against oracle drift)."""
import pytest

import parity_cases as pc


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_oracle_reproduces_golden(curve):
    pc.check_oracle_reproduces_golden(curve, max_L=10)


def test_bn254_generators_fixture():
    """The committed BN254 generators (tests/golden/make_golden.py) are what the oracle's SvdW hash-to-G1 gives."""
    import json
    import os
    from oracle import bbs
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "bbs_golden.json")) as f:
        gold = json.load(f)["bn254_create_generators"]
    bn = bbs.BN_SUITE
    assert bytes.fromhex(gold["api_id"]) == bn.api_id
    assert bbs.create_generators(bn, 11, bn.api_id) == [(int(x, 16), int(y, 16)) for x, y in gold["generators"]]
    assert (int(gold["p1"][0], 16), int(gold["p1"][1], 16)) == bn.p1
