"""The synthetic bench workload of SURVEY.md 8(d), produced with the PRODUCT's own host functions and hashlib only.

bench.py, bench_extras.py and bench_mixed.py take their inputs from here, so that nothing the bench measures or feeds on
comes from the test infrastructure (oracle/, tests/): the issuer key by bbs_key_gen (IKM [1u8; 32], dst
"BBS-SIG-KEYGEN-SALT-", benches/proof_verify.rs:118-121), the generators by bbs_create_generators, message b of item j =
expand_message("bbs-bench-msg" || I2OSP(b, 8) || I2OSP(j, 8), "BBS_BENCH_MSG_DST_", 32) hashed to a scalar on the device
(msg_to_scalars), disclosed indexes 0 .. R, proof_gen's random scalars = expand_message(seed, dst, 48 * count) cut into
48-byte strings reduced mod r (src/utils/core_utilities.rs:84-100).  tests/test_workload.py checks that the items are the
ones tests/parity_cases.py derives with the oracle.

The interface is the subset of tests/parity_cases.py the bench files use (bench_engine, bench_items, expand_message,
i2osp, to_engine_proof, seeded_random_scalars), so bench_mixed.run_mixed accepts either module.
"""
from __future__ import annotations

import hashlib
from collections import namedtuple
from typing import Optional

from . import api
from .engine import Engine, Proof

Curve = namedtuple("Curve", "name r")
Suite = namedtuple("Suite", "curve api_id")
SUITES = {c: Suite(Curve(c, api.SCALAR_ORDER[c]), api.api_id(c)) for c in ("bls12_381", "bn254")}


def i2osp(v: int, n: int) -> bytes:
    return int(v).to_bytes(n, "big")


def expand_message(msg: bytes, dst: bytes, len_in_bytes: int) -> bytes:
    """expand_message_xmd with SHA-256 (RFC 9380 5.3.1; src/utils/utilities_helper.rs:42-97)."""
    ell = (len_in_bytes + 31) // 32
    if ell > 255 or len(dst) > 255:
        raise ValueError("expand_message: length out of range")
    dst_prime = dst + bytes([len(dst)])
    b0 = hashlib.sha256(bytes(64) + msg + bytes([(len_in_bytes >> 8) & 0xFF, len_in_bytes & 0xFF, 0]) + dst_prime).digest()
    bi = hashlib.sha256(b0 + b"\x01" + dst_prime).digest()
    out = bi
    for i in range(2, ell + 1):
        bi = hashlib.sha256(bytes(x ^ y for x, y in zip(b0, bi)) + bytes([i]) + dst_prime).digest()
        out += bi
    return out[:len_in_bytes]


def seeded_random_scalars(suite: Suite, seed: bytes, dst: bytes, count: int):
    """src/utils/core_utilities.rs:84-100: 48 uniform bytes per scalar, big-endian, mod r."""
    v = expand_message(seed, dst, 48 * count)
    return [int.from_bytes(v[48 * i:48 * (i + 1)], "big") % suite.curve.r for i in range(count)]


def to_engine_proof(p):
    return Proof(p.a_bar, p.b_bar, p.d, p.e_cap, p.r1_cap, p.r3_cap, list(p.commitments), p.challenge)


def bench_engine(curve: str, L: int = 32, lib_path: Optional[str] = None, window_bits: Optional[int] = None, device: int = 0):
    """One issuer key (IKM [1u8; 32], empty key_info, dst "BBS-SIG-KEYGEN-SALT-"), the suite's generators.
    -> (suite, engine, generators, secret key)"""
    suite = SUITES[curve]
    gens = api.create_generators(curve, L + 1, lib_path)
    sk = api.SecretKey.key_gen(curve, bytes([1] * 32), b"", b"BBS-SIG-KEYGEN-SALT-", lib_path, device).sk
    if window_bits is None and lib_path is not None:
        window_bits = 4
    eng = Engine(curve, device=device, lib_path=lib_path, window_bits=window_bits)
    eng.set_generators(gens, suite.api_id)
    eng.set_secret_key(sk)
    return suite, eng, gens, sk


def bench_items(suite: Suite, eng: Engine, n: int, L: int = 32, R: int = 8, first_item: int = 0, ids=None):
    """Items b = first_item .. first_item + n (or the given ids): (message scalars, disclosed indexes, random scalars)."""
    ids = list(range(first_item, first_item + n)) if ids is None else [int(b) for b in ids]
    n = len(ids)
    raw = [expand_message(b"bbs-bench-msg" + i2osp(b, 8) + i2osp(j, 8), b"BBS_BENCH_MSG_DST_", 32) for b in ids for j in range(L)]
    flat = eng.hash_to_scalar_batch(raw, suite.api_id + b"MAP_MSG_TO_SCALAR_AS_HASH_")
    msgs = [flat[b * L:(b + 1) * L] for b in range(n)]
    disclosed = [list(range(R))] * n
    rnds = [seeded_random_scalars(suite, b"bbs-bench-rnd" + i2osp(b, 8), suite.api_id + b"MOCK_RANDOM_SCALARS_DST_", 5 + L - R) for b in ids]
    return msgs, disclosed, rnds
