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


# Fixed-base window widths: 8 = the library default, 16 / 20 = what bench.py runs (BN254 / BLS12-381 headline).  Width 20
# is the one whose digits straddle 32-bit words, whose last window is clamped and whose tables take 52 GB at L = 32.
WIDTHS = [("bls12_381", 8), ("bls12_381", 16), ("bls12_381", 20), ("bn254", 8), ("bn254", 16), ("bn254", 20)]


@pytest.mark.parametrize("curve,window_bits", WIDTHS)
def test_batch_verification(curve, window_bits):
    pc.check_batch_verification(curve, None, window_bits=window_bits)


@pytest.mark.parametrize("curve,window_bits", WIDTHS)
def test_points_in_subgroup(curve, window_bits):
    pc.check_points_in_subgroup(curve, None, window_bits=window_bits)


@pytest.mark.parametrize("curve,window_bits", WIDTHS)
def test_full_batch_4096(curve, window_bits):
    pc.check_big_batch(curve, None, n=4096, L=32, R=8, window_bits=window_bits)


@pytest.mark.parametrize("curve,window_bits", [("bls12_381", 8), ("bls12_381", 16), ("bls12_381", 20), ("bn254", 16)])
def test_every_item_against_c_oracle(curve, window_bits):
    pc.check_batch_vs_c_oracle(None, n=1024, window_bits=window_bits, curve=curve)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_window_widths(curve):
    pc.check_window_widths(curve, None, widths=(5, 7, 11, 13, 17, 19, 20, 22), L=2)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_fail_closed(curve):
    pc.check_fail_closed(curve, None)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_submit(curve):
    pc.check_submit(curve, None)
    pc.check_submit(curve, None, n=130, L=6, seed=24)         # crosses wavefront boundaries


def test_mixed_curves_in_flight():
    pc.check_mixed_curves_in_flight(None)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_empty_batches(curve):
    pc.check_empty_batches(curve, None)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_latency_mode(curve):
    pc.check_latency_mode(curve, None)
    pc.check_latency_mode(curve, None, n=80, L=6, seed=42)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_large_shapes(curve):
    pc.check_large_shapes(curve, None, L=100, n=4)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_proof_verify_octets(curve):
    pc.check_proof_verify_octets(curve, None)
    pc.check_proof_verify_octets(curve, None, n=150, L=7, seed=62)
    pc.check_proof_verify_octets(curve, None, seed=63, disclose_all_3=True)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_verify_octets(curve):
    pc.check_verify_octets(curve, None)
    pc.check_verify_octets(curve, None, n=200, L=6, seed=72)


def test_threads():
    pc.check_threads(None, threads=6, rounds=4, n=64, L=5)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
@pytest.mark.parametrize("window_bits", [8, 13, 20])
def test_fixed_base_tree(curve, window_bits):
    if curve == "bn254" and window_bits == 20:
        window_bits = 16
    pc.check_fixed_base_tree(curve, None, window_bits=window_bits, n_pv=70)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_octets_out(curve):
    pc.check_octets_out(curve, None)
    pc.check_octets_out(curve, None, n=300, L=9, seed=96)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_proof_verify_wire(curve):
    pc.check_proof_verify_wire(curve, None)
    pc.check_proof_verify_wire(curve, None, n=200, L=8, seed=98)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_sign_verify_wire(curve):
    pc.check_sign_verify_wire(curve, None)
    pc.check_sign_verify_wire(curve, None, n=150, L=9, seed=100)

