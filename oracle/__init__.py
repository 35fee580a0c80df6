"""CPU oracle for the BBS+ hot path (TEST INFRASTRUCTURE ONLY).

This package is a big-integer restatement of the reference algorithm
(hashcloak/bbs_sign, Rust on arkworks 0.4) for the path named by
BASELINE.json:north_star.  It is the *checker*: only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it.  The product (``bbs_sign_amd``) never imports, links or executes anything
in here.

Parity pinning: the reference is Rust and cannot be compiled in this image
(no cargo/rustc; the arithmetic lives in un-vendored crates: ark-ff/ark-ec/
ark-serialize 0.4.2, ark-bls12-381 0.4.0, ark-bn254 0.4.0, zkcrypto bls12_381
@9ea427c, bn254_hash2curve 0.1.2, sha2 0.10.6).  The oracle is therefore pinned
by every known-answer vector the reference's own tests hold for this path
(src/tests/test_vector.rs:56-260, all BLS12-381) -- see tests/test_oracle_kat.py.
BN254 has one known answer in the reference, P1 (src/constants.rs:39-51): it pins
the restated BN254 hash-to-G1 (Shallue-van de Woestijne, oracle/hashing.py) and
hence create_generators.  The reference holds no BN254 *byte* vector, so the
ark-serialize formats hashed into domain / challenge on BN254 are "parity
unpinned"; BN254 results are otherwise pinned as group elements / booleans through
the reference's round-trip and negative test structure.
"""
