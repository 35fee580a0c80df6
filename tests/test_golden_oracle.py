"""The oracle reproduces the committed golden fixtures (guards against oracle drift)."""
import pytest

import parity_cases as pc


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_oracle_reproduces_golden(curve):
    pc.check_oracle_reproduces_golden(curve, max_L=10)
