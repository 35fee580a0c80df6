"""Public interface (bytes messages, library-made generators) on the GPU."""
import pytest

import public_api_cases as pa

pytestmark = pytest.mark.gpu


def test_vectors_through_public_api():
    pa.check_create_generators_kat(None)
    pa.check_key_gen_kat(None)
    pa.check_vectors_public_api(None)


def test_round_trips():
    pa.check_round_trips(None)


def test_invalid_proofs():
    pa.check_invalid_proofs(None)


def test_bn254_public_interface():
    pa.check_readme_example_bn254(None)
    pa.check_round_trips(None, pa.CASES, curve="bn254")
