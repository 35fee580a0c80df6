"""GPU parity tests proper: the product library (hand-written HIP for gfx950) through the C ABI,
against the oracle on the same seeded inputs, the committed golden fixtures, the reference's
known-answer vectors, and size-independent properties at BASELINE.json's full batch size."""
import pytest

import parity_cases as pc

pytestmark = pytest.mark.gpu


def test_kat_vectors():
    pc.check_kat_vectors(None)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_golden(curve):
    pc.check_golden(curve, None)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_random_batch(curve):
    pc.check_random_batch(curve, None, n=24, L=5, seed=1)
    pc.check_random_batch(curve, None, n=70, L=3, seed=2)     # crosses a wavefront boundary


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_error_semantics(curve):
    pc.check_error_semantics(curve, None)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_primitives(curve):
    pc.check_primitives(curve, None)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_pippenger(curve):
    pc.check_pippenger(curve, None, n=200)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_batch_verification(curve):
    pc.check_batch_verification(curve, None)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_points_in_subgroup(curve):
    pc.check_points_in_subgroup(curve, None)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_full_batch_4096(curve):
    pc.check_big_batch(curve, None, n=4096, L=32, R=8)


def test_every_item_against_c_oracle():
    pc.check_batch_vs_c_oracle(None, n=1024)


def test_mixed_curves_in_flight():
    pc.check_mixed_curves_in_flight(None)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_empty_batches(curve):
    pc.check_empty_batches(curve, None)
