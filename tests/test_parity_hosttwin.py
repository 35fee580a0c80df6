"""Host-logic checks through the TEST-ONLY host twin (same stage code compiled for x86): validation
order, packing, index bookkeeping, byte formats.  NOT a parity claim about the GPU product -- that is
tests/test_parity_gpu.py (-m gpu)."""
import os
import subprocess
import sys

import pytest

import parity_cases as pc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="session")
def twin():
    sys.path.insert(0, ROOT)
    from bbs_sign_amd import build as b
    return b.build(twin=True, verbose=False)


def test_kat_vectors(twin):
    pc.check_kat_vectors(twin)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_golden_small(twin, curve):
    pc.check_golden(curve, twin, max_L=10)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_random_batch(twin, curve):
    pc.check_random_batch(curve, twin, n=6, L=5, seed=1)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_error_semantics(twin, curve):
    pc.check_error_semantics(curve, twin)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_primitives(twin, curve):
    pc.check_primitives(curve, twin)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_pippenger(twin, curve):
    pc.check_pippenger(curve, twin, n=24)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_batch_verification(twin, curve):
    pc.check_batch_verification(curve, twin)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_points_in_subgroup(twin, curve):
    pc.check_points_in_subgroup(curve, twin)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_empty_batches(twin, curve):
    pc.check_empty_batches(curve, twin)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_window_widths(twin, curve):
    # odd widths: digits straddle 32-bit words, the last window is clamped at bit 256
    pc.check_window_widths(curve, twin, widths=(5, 7, 11, 13))


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_fail_closed(twin, curve):
    pc.check_fail_closed(curve, twin)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_submit(twin, curve):
    pc.check_submit(curve, twin)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_latency_mode(twin, curve):
    pc.check_latency_mode(curve, twin)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_large_shapes(twin, curve):
    pc.check_large_shapes(curve, twin, L=40, n=2)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_proof_verify_octets(twin, curve):
    pc.check_proof_verify_octets(curve, twin)
    pc.check_proof_verify_octets(curve, twin, seed=63, disclose_all_3=True)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_verify_octets(twin, curve):
    pc.check_verify_octets(curve, twin)


def test_threads(twin):
    pc.check_threads(twin)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_fixed_base_tree(twin, curve):
    pc.check_fixed_base_tree(curve, twin)
    pc.check_fixed_base_tree(curve, twin, L=6, seed=92, window_bits=7, n_pv=6)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_octets_out(twin, curve):
    pc.check_octets_out(curve, twin)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_proof_verify_wire(twin, curve):
    pc.check_proof_verify_wire(curve, twin)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_sign_verify_wire(twin, curve):
    pc.check_sign_verify_wire(curve, twin)



@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_issuer_mixed_lengths(twin, curve):
    pc.check_issuer_mixed_lengths(curve, twin)


@pytest.mark.job_form(True)
@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_latency_form_of_every_job(twin, curve):
    """The latency form of a job (on the host twin: T1 on three lanes; the pairing stages are the one-lane ones in both forms)
    through the cases that create jobs of every kind, incl. empty batches and batch verification."""
    pc.check_random_batch(curve, twin, n=6, L=3, seed=3)
    pc.check_empty_batches(curve, twin)
    pc.check_batch_verification(curve, twin)
    pc.check_submit(curve, twin)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_issuer_threads(twin, curve):
    pc.check_issuer_threads(curve, twin)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_issuer_budget(twin, curve):
    pc.check_issuer_budget(curve, twin)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_proof_gen_unusual_points(twin, curve):
    pc.check_proof_gen_unusual_points(curve, twin)


@pytest.mark.job_form(True)
@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_proof_gen_unusual_points_latency_form(twin, curve):
    pc.check_proof_gen_unusual_points(curve, twin)


def test_pool_two_members(twin):
    """bbs_pool (SURVEY 8(b) / 8(e) behind the C ABI): a mixed-curve list over two members, merged statuses = one context per curve"""
    pc.check_pool(twin, devices=(0, 0), per_curve=24)


def test_pool_three_members_uneven_shares(twin):
    pc.check_pool(twin, devices=(0, 0, 0), per_curve=13, max_batch=3)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_fail_closed_every_submit_entry_point(twin, curve):
    pc.check_fail_closed_submit(curve, twin)

