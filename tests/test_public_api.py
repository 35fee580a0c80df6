"""Public-interface mirror: host-side pieces (create_generators, key_gen: no GPU involved) against the
reference's vectors, and the full interface through the host twin."""
import os
import sys

import pytest

import public_api_cases as pa

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="session")
def product():
    sys.path.insert(0, ROOT)
    from bbs_sign_amd import build as b
    return b.build(twin=False, verbose=False)


@pytest.fixture(scope="session")
def twin():
    sys.path.insert(0, ROOT)
    from bbs_sign_amd import build as b
    return b.build(twin=True, verbose=False)


def test_create_generators_kat_product_host_code(product):
    pa.check_create_generators_kat(product)


def test_key_gen_kat_product_host_code(product):
    pa.check_key_gen_kat(product)


def test_vectors_through_public_api_twin(twin):
    pa.check_vectors_public_api(twin)


def test_round_trips_twin(twin):
    pa.check_round_trips(twin, pa.CASES[:4] + pa.CASES[6:7])


def test_invalid_proofs_twin(twin):
    pa.check_invalid_proofs(twin)


def test_readme_example_bn254_twin(twin):
    pa.check_readme_example_bn254(twin)


def test_round_trips_bn254_twin(twin):
    pa.check_round_trips(twin, pa.CASES[:3], curve="bn254")
