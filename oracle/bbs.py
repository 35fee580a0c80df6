"""BBS+ sign / verify / proof_gen / proof_verify -- restatement of the reference.

ORACLE (test infrastructure, see oracle/__init__.py).  Every function cites the
reference lines it follows; operation order is the reference's (per-call
domain, independent scalar multiplications, two full pairings).  Scalars are
Python ints in [0, r); points are ``None`` or affine (x, y).

Errors: the reference returns ``Err(variant)``; here ``BbsError(variant)`` is
raised with the same variant name.  Reference panics are ``BbsPanic``.
"""

from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional, Sequence

from .curves import BLS12_381, BN254, Curve
from .hashing import expand_message, from_okm, hash_to_g1_bls, hash_to_g1_bn, hash_to_scalar, i2osp


class BbsError(Exception):
    def __init__(self, variant: str):
        super().__init__(variant)
        self.variant = variant


class BbsPanic(Exception):
    pass


# ------------------------------------------------------------------ constants
@dataclass(frozen=True)
class Suite:
    """src/constants.rs:13-89."""
    curve: Curve
    ciphersuite_id: bytes
    p1: tuple

    @property
    def api_id(self) -> bytes:          # sign.rs:44, verify.rs:31, proof_gen.rs:94, proof_verify.rs:35
        return self.ciphersuite_id + b"H2G_HM2S_"


BLS_SUITE = Suite(
    BLS12_381,
    b"BBS_BLS12381G1_XMD:SHA-256_SSWU_RO_",
    (
        1355253221325668152696183518801331769866100080859571110928822005264442742039790254588065001486134245057142899747017,
        2563071790429735027383427649950865259619709115697058137448106859255609577834149037543606665262210555960464099235249,
    ),
)
BN_SUITE = Suite(
    BN254,
    b"BBS_QUUX-V01-CS02-with-BN254G1_XMD:SHA-256_SVDW_RO_",
    (
        7738860219269362160002109478394842060990190871738832255540382874922375322334,
        8255268479661695615178834896135584953541182794935974658059743263102507888551,
    ),
)
SUITES = {"bls12_381": BLS_SUITE, "bn254": BN_SUITE}


# -------------------------------------------------------------- serialisation
def scalar_be(curve: Curve, s: int) -> bytes:
    """LE serialize_compressed then .reverse() (sign.rs:92-96) == 32-byte big-endian."""
    return int(s % curve.r).to_bytes(32, "big")


def _fp_largest(curve, y):
    return y > (curve.p - 1) // 2


def g1_compress(curve: Curve, P) -> bytes:
    """ark-serialize compressed G1.  BLS12-381: 48 B big-endian x, flags in the first byte
    (0x80 compressed, 0x40 infinity, 0x20 y lexicographically largest) -- pinned by
    test_vector.rs:56-69.  BN254: ark-ec default, 32 B little-endian x, flags in the last
    byte (0x80 y > -y, 0x40 infinity) [crate knowledge, unpinned]."""
    n = curve.fp_bytes
    if curve.name == "bls12_381":
        if P is None:
            return bytes([0xC0]) + bytes(n - 1)
        b = bytearray(P[0].to_bytes(n, "big"))
        b[0] |= 0x80
        if _fp_largest(curve, P[1]):
            b[0] |= 0x20
        return bytes(b)
    if P is None:
        return bytes(n - 1) + bytes([0x40])
    b = bytearray(P[0].to_bytes(n, "little"))
    if _fp_largest(curve, P[1]):
        b[n - 1] |= 0x80
    return bytes(b)


def _fp2_largest(curve, y):
    # compare c1 first, then c0 (both ark QuadExtField::cmp and the zkcrypto rule)
    if y[1] != 0:
        return _fp_largest(curve, y[1])
    return _fp_largest(curve, y[0])


def g2_compress(curve: Curve, Q) -> bytes:
    """Compressed G2: BLS12-381 = x.c1 || x.c0 big-endian, flags first byte (pinned by
    test_vector.rs:62-64,158-159); BN254 = x.c0 || x.c1 little-endian, flags last byte."""
    n = curve.fp_bytes
    if curve.name == "bls12_381":
        if Q is None:
            return bytes([0xC0]) + bytes(2 * n - 1)
        (x0, x1), y = Q
        b = bytearray(x1.to_bytes(n, "big") + x0.to_bytes(n, "big"))
        b[0] |= 0x80
        if _fp2_largest(curve, y):
            b[0] |= 0x20
        return bytes(b)
    if Q is None:
        return bytes(2 * n - 1) + bytes([0x40])
    (x0, x1), y = Q
    b = bytearray(x0.to_bytes(n, "little") + x1.to_bytes(n, "little"))
    if _fp2_largest(curve, y):
        b[2 * n - 1] |= 0x80
    return bytes(b)


def g1_decompress_bls(data: bytes):
    """Inverse of g1_compress for BLS12-381 (used to read the reference's hex vectors)."""
    c = BLS12_381
    if data[0] & 0x40:
        return None
    x = int.from_bytes(bytes([data[0] & 0x1F]) + data[1:], "big")
    y2 = (x * x * x + c.b) % c.p
    y = pow(y2, (c.p + 1) // 4, c.p)
    assert y * y % c.p == y2
    if _fp_largest(c, y) != bool(data[0] & 0x20):
        y = c.p - y
    return (x, y)


# ------------------------------------------------------------------ key_gen
def key_gen(suite: Suite, key_material: bytes, key_info: bytes, key_dst: bytes) -> int:
    """src/key_gen.rs:46-81."""
    if len(key_material) < 32:
        raise BbsError("InvalidKeyMaterialLength")
    if len(key_info) > 65535:
        raise BbsError("InvalidKeyInfoLength")
    derive_input = key_material + i2osp(len(key_info), 2) + key_info
    sk = hash_to_scalar(suite.curve, derive_input, key_dst)
    if sk == 0:
        raise BbsError("InvalidSecretKey")
    return sk


def sk_to_pk(suite: Suite, sk: int):
    """src/key_gen.rs:83-90."""
    return suite.curve.g2_mul(suite.curve.g2, sk)


# ------------------------------------------------------- interface utilities
_GEN_CACHE = {}


def create_generators(suite: Suite, count: int, api_id: bytes):
    """src/utils/interface_utilities.rs:47-73 with the suite's hash-to-G1 (:24-44)."""
    h2g = hash_to_g1_bls if suite.curve.name == "bls12_381" else hash_to_g1_bn
    key = (suite.curve.name, api_id)
    have = _GEN_CACHE.setdefault(key, {"v": None, "gens": []})
    seed_dst = api_id + b"SIG_GENERATOR_SEED_"
    generator_dst = api_id + b"SIG_GENERATOR_DST_"
    generator_seed = api_id + b"MESSAGE_GENERATOR_SEED"
    if have["v"] is None:
        have["v"] = expand_message(generator_seed, seed_dst, 48)
    while len(have["gens"]) < count:
        i = len(have["gens"])
        have["v"] = expand_message(have["v"] + i2osp(i + 1, 8), seed_dst, 48)
        have["gens"].append(h2g(have["v"], generator_dst))
    return list(have["gens"][:count])


def synthetic_generators(suite: Suite, count: int, tag: bytes = b"synthetic-generators"):
    """NOT in the reference: deterministic stand-in generators k_i * BP1 with k_i hashed from
    (tag, i).  Used for BN254, whose hash-to-curve backend is not restated; core_* takes
    generators as an input, exactly like core_sign_tests.rs:51-64."""
    c = suite.curve
    return [c.g1_mul(c.g1, hash_to_scalar(c, tag + i2osp(i, 8), b"ORACLE_SYNTH_GEN_") or 1)
            for i in range(count)]


def msg_to_scalars(suite: Suite, messages: Sequence[bytes], api_id: bytes) -> List[int]:
    """src/utils/interface_utilities.rs:76-88."""
    map_dst = api_id + b"MAP_MSG_TO_SCALAR_AS_HASH_"
    return [hash_to_scalar(suite.curve, m, map_dst) for m in messages]


def calculate_domain(suite: Suite, pk, q_1, h_points, header: bytes, api_id: bytes) -> int:
    """src/utils/core_utilities.rs:24-63."""
    c = suite.curve
    dom_octs = i2osp(len(h_points), 8) + g1_compress(c, q_1)
    for h in h_points:
        dom_octs += g1_compress(c, h)
    dom_octs += api_id
    dom_input = g2_compress(c, pk) + dom_octs + i2osp(len(header), 8) + header
    return hash_to_scalar(c, dom_input, api_id + b"H2S_")


def seeded_random_scalars(suite: Suite, seed: bytes, dst: bytes, count: int) -> List[int]:
    """src/utils/core_utilities.rs:84-100."""
    v = expand_message(seed, dst, 48 * count)
    return [from_okm(suite.curve, v[48 * i:48 * (i + 1)]) for i in range(count)]


def mocked_calculate_random_scalars(suite: Suite, count: int) -> List[int]:
    """src/utils/core_utilities.rs:103-113 (dst is the BLS string for every curve)."""
    dst = b"BBS_BLS12381G1_XMD:SHA-256_SSWU_RO_H2G_HM2S_MOCK_RANDOM_SCALARS_DST_"
    seed = bytes.fromhex("332e313431353932363533353839373933323338343632363433333833323739")
    return seeded_random_scalars(suite, seed, dst, count)


# ----------------------------------------------------------------- core_sign
@dataclass
class Signature:
    a: Optional[tuple]
    e: int


def compute_b(suite: Suite, generators, domain: int, messages: Sequence[int]):
    """B = P1 + Q1*domain + sum H_i*m_i  (sign.rs:120-126, verify.rs:81-86, proof_gen.rs:249-253)."""
    c = suite.curve
    b = c.g1_add(suite.p1, c.g1_mul(generators[0], domain))
    for i in range(1, len(messages) + 1):
        b = c.g1_add(b, c.g1_mul(generators[i], messages[i - 1]))
    return b


def core_sign(suite: Suite, sk: int, generators, header: bytes, messages: Sequence[int],
              api_id: bytes) -> Signature:
    """src/sign.rs:63-133."""
    c = suite.curve
    if len(messages) + 1 != len(generators):
        raise BbsError("InvalidMessageAndGeneratorsLength")
    pk = sk_to_pk(suite, sk)
    domain = calculate_domain(suite, pk, generators[0], generators[1:len(messages) + 1], header, api_id)
    ser = scalar_be(c, sk) + b"".join(scalar_be(c, m) for m in messages) + scalar_be(c, domain)
    e = hash_to_scalar(c, ser, api_id + b"H2S_")
    b = compute_b(suite, generators, domain, messages)
    sk_plus_e = (sk + e) % c.r
    if sk_plus_e == 0:
        raise BbsPanic("(sk + e) has no inverse (sign.rs:129 unwrap)")
    a = c.g1_mul(b, pow(sk_plus_e, -1, c.r))
    return Signature(a, e)


def core_verify(suite: Suite, pk, signature: Signature, generators, header: bytes,
                messages: Sequence[int], api_id: bytes) -> bool:
    """src/verify.rs:53-93."""
    c = suite.curve
    if len(messages) + 1 != len(generators):
        raise BbsError("InvalidMessageAndGeneratorsLength")
    domain = calculate_domain(suite, pk, generators[0], generators[1:len(messages) + 1], header, api_id)
    b = compute_b(suite, generators, domain, messages)
    q = c.g2_add(pk, c.g2_mul(c.g2, signature.e))
    return c.pairing_product_is_one([(signature.a, q), (b, c.g2_neg(c.g2))])


# ------------------------------------------------------------------ proof_gen
@dataclass
class Proof:
    a_bar: Optional[tuple] = None
    b_bar: Optional[tuple] = None
    d: Optional[tuple] = None
    e_cap: int = 0
    r1_cap: int = 0
    r3_cap: int = 0
    commitments: List[int] = field(default_factory=list)
    challenge: int = 0


@dataclass
class InitProof:
    points: list        # [a_bar, b_bar, d, t1, t2]
    scalar: int         # domain


def proof_init(suite: Suite, pk, signature: Signature, generators, random_scalars, header: bytes,
               messages: Sequence[int], undisclosed_indexes: Sequence[int], api_id: bytes) -> InitProof:
    """src/proof_gen.rs:211-269."""
    c = suite.curve
    r = c.r
    if len(messages) + 1 != len(generators):
        raise BbsError("InvalidMessageAndGeneratorsLength")
    if len(random_scalars) != len(undisclosed_indexes) + 5:
        raise BbsError("InvalidRandomScalarsAndUndisclosedIndicesLength")
    if len(undisclosed_indexes) > len(messages):
        raise BbsError("InvalidUndisclosedIndicesLength")
    domain = calculate_domain(suite, pk, generators[0], generators[1:len(messages) + 1], header, api_id)
    b = compute_b(suite, generators, domain, messages)
    rs = random_scalars
    d = c.g1_mul(b, rs[1])
    a_bar = c.g1_mul(signature.a, rs[0] * rs[1] % r)
    b_bar = c.g1_add(c.g1_mul(d, rs[0]), c.g1_neg(c.g1_mul(a_bar, signature.e)))
    t1 = c.g1_add(c.g1_mul(a_bar, rs[2]), c.g1_mul(d, rs[3]))
    t2 = c.g1_mul(d, rs[4])
    msg_generators = generators[1:len(messages) + 1]
    for i in range(5, len(rs)):
        t2 = c.g1_add(t2, c.g1_mul(msg_generators[undisclosed_indexes[i - 5]], rs[i]))
    return InitProof([a_bar, b_bar, d, t1, t2], domain)


def proof_challenge_calculate(suite: Suite, init_res: InitProof, disclosed_messages: Sequence[int],
                              disclosed_indexes: Sequence[int], ph: bytes, api_id: bytes) -> int:
    """src/proof_gen.rs:272-328."""
    c = suite.curve
    if len(disclosed_messages) != len(disclosed_indexes):
        raise BbsError("InvalidIndicesAndMessagesLength")
    ser = i2osp(len(disclosed_indexes), 8)
    for idx, m in zip(disclosed_indexes, disclosed_messages):
        ser += i2osp(idx, 8) + scalar_be(c, m)
    for pt in init_res.points:
        ser += g1_compress(c, pt)
    ser += scalar_be(c, init_res.scalar)
    ser += i2osp(len(ph), 8) + ph
    return hash_to_scalar(c, ser, api_id + b"H2S_")


def proof_finalize(suite: Suite, init_res: InitProof, challenge: int, e_value: int, random_scalars,
                   undisclosed_messages: Sequence[int]) -> Proof:
    """src/proof_gen.rs:331-365."""
    r = suite.curve.r
    rs = random_scalars
    if len(rs) != len(undisclosed_messages) + 5:
        raise BbsError("InvalidRandomScalarsAndUndisclosedIndicesLength")
    if rs[1] % r == 0:
        raise BbsPanic("r2 has no inverse (proof_gen.rs:346 unwrap)")
    r3 = pow(rs[1], -1, r)
    e_cap = (rs[2] + e_value * challenge) % r
    r1_cap = (rs[3] - rs[0] * challenge) % r
    r3_cap = (rs[4] - r3 * challenge) % r
    commitments = [(rs[i + 5] + undisclosed_messages[i] * challenge) % r
                   for i in range(len(undisclosed_messages))]
    return Proof(init_res.points[0], init_res.points[1], init_res.points[2], e_cap, r1_cap, r3_cap,
                 commitments, challenge)


def core_proof_gen(suite: Suite, pk, signature: Signature, header: bytes, generators, ph: bytes,
                   messages: Sequence[int], disclosed_indexes: Sequence[int], api_id: bytes,
                   random_scalars: Sequence[int]) -> Proof:
    """src/proof_gen.rs:116-208.  ``random_scalars`` replaces the draw at :145-149 (it must have
    5 + L - R entries, R counted before dedup, exactly as the reference sizes it)."""
    l = len(messages)
    r_ = len(disclosed_indexes)
    if r_ > l:
        raise BbsError("InvalidDisclosedIndicesLength")
    for idx in disclosed_indexes:
        if idx >= l:
            raise BbsError("InvalidDisclosedIndex")
    if len(random_scalars) != 5 + l - r_:
        raise ValueError("caller must supply 5 + L - R random scalars")
    disclosed = sorted(set(disclosed_indexes))
    undisclosed = sorted(set(range(l)) - set(disclosed))
    init_res = proof_init(suite, pk, signature, generators, random_scalars, header, messages,
                          undisclosed, api_id)
    disclosed_messages = [messages[i] for i in disclosed]
    undisclosed_messages = [messages[i] for i in undisclosed]
    challenge = proof_challenge_calculate(suite, init_res, disclosed_messages, disclosed, ph, api_id)
    return proof_finalize(suite, init_res, challenge, signature.e, random_scalars, undisclosed_messages)


# --------------------------------------------------------------- proof_verify
def proof_verify_init(suite: Suite, pk, proof: Proof, generators, header: bytes,
                      disclosed_messages: Sequence[int], disclosed_indexes: Sequence[int],
                      api_id: bytes) -> InitProof:
    """src/proof_verify.rs:119-188."""
    c = suite.curve
    u = len(proof.commitments)
    r_ = len(disclosed_indexes)
    l = r_ + u
    for idx in disclosed_indexes:
        if idx >= l:
            raise BbsError("InvalidDisclosedIndex")
    if len(disclosed_messages) != r_:
        raise BbsError("InvalidIndicesAndMessagesLength")
    if len(generators) != l + 1:
        raise BbsError("InvalidMessageAndGeneratorsLength")
    undisclosed = sorted(set(range(l)) - set(disclosed_indexes))
    domain = calculate_domain(suite, pk, generators[0], generators[1:l + 1], header, api_id)
    t1 = c.g1_add(c.g1_add(c.g1_mul(proof.b_bar, proof.challenge), c.g1_mul(proof.a_bar, proof.e_cap)),
                  c.g1_mul(proof.d, proof.r1_cap))
    bv = c.g1_add(suite.p1, c.g1_mul(generators[0], domain))
    msg_generators = generators[1:l + 1]
    for i, idx in enumerate(disclosed_indexes):
        bv = c.g1_add(bv, c.g1_mul(msg_generators[idx], disclosed_messages[i]))
    t2 = c.g1_add(c.g1_mul(bv, proof.challenge), c.g1_mul(proof.d, proof.r3_cap))
    for i, idx in enumerate(undisclosed):
        # the reference indexes proof.commitments[i] for every undisclosed index; with duplicate
        # disclosed indexes the undisclosed set is larger than the commitments -> Rust panics
        if i >= len(proof.commitments):
            raise BbsPanic("index out of bounds: proof.commitments (proof_verify.rs:177-179)")
        t2 = c.g1_add(t2, c.g1_mul(msg_generators[idx], proof.commitments[i]))
    return InitProof([proof.a_bar, proof.b_bar, proof.d, t1, t2], domain)


def core_proof_verify(suite: Suite, pk, proof: Proof, generators, header: bytes, ph: bytes,
                      disclosed_messages: Sequence[int], disclosed_indexes: Sequence[int],
                      api_id: bytes) -> bool:
    """src/proof_verify.rs:64-116."""
    c = suite.curve
    init_res = proof_verify_init(suite, pk, proof, generators, header, disclosed_messages,
                                 disclosed_indexes, api_id)
    challenge = proof_challenge_calculate(suite, init_res, disclosed_messages, disclosed_indexes, ph, api_id)
    if challenge != proof.challenge % c.r:
        return False
    return c.pairing_product_is_one([(proof.a_bar, pk), (proof.b_bar, c.g2_neg(c.g2))])


# ------------------------------------------------------------ public wrappers
def sign(suite: Suite, sk: int, messages: Sequence[bytes], header: bytes) -> Signature:
    """src/sign.rs:32-60."""
    api_id = suite.api_id
    scalars = msg_to_scalars(suite, messages, api_id)
    generators = create_generators(suite, len(messages) + 1, api_id)
    return core_sign(suite, sk, generators, header, scalars, api_id)


def verify(suite: Suite, pk, signature: Signature, header: bytes, messages: Sequence[bytes]) -> bool:
    """src/verify.rs:18-50."""
    api_id = suite.api_id
    scalars = msg_to_scalars(suite, messages, api_id)
    generators = create_generators(suite, len(messages) + 1, api_id)
    return core_verify(suite, pk, signature, generators, header, scalars, api_id)


def proof_gen(suite: Suite, pk, signature: Signature, header: bytes, ph: bytes,
              messages: Sequence[bytes], disclosed_indexes: Sequence[int], random_scalars) -> Proof:
    """src/proof_gen.rs:78-113."""
    api_id = suite.api_id
    scalars = msg_to_scalars(suite, messages, api_id)
    generators = create_generators(suite, len(messages) + 1, api_id)
    return core_proof_gen(suite, pk, signature, header, generators, ph, scalars, disclosed_indexes,
                          api_id, random_scalars)


def proof_verify(suite: Suite, pk, proof: Proof, header: bytes, ph: bytes,
                 disclosed_messages: Sequence[bytes], disclosed_indexes: Sequence[int]) -> bool:
    """src/proof_verify.rs:19-61."""
    api_id = suite.api_id
    scalars = msg_to_scalars(suite, disclosed_messages, api_id)
    generators = create_generators(suite, len(proof.commitments) + len(disclosed_indexes) + 1, api_id)
    return core_proof_verify(suite, pk, proof, generators, header, ph, scalars, disclosed_indexes, api_id)
