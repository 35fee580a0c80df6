"""Mirror of the reference's PUBLIC interface (README.md:43-128) over the MI355X engine:

    SecretKey.key_gen / .sk_to_pk / .sign        src/key_gen.rs:46-90, src/sign.rs:32-60
    PublicKey.verify                              src/verify.rs:18-50
    proof_gen / proof_verify                      src/proof_gen.rs:78-113, src/proof_verify.rs:19-61

Messages are byte strings, exactly as in the reference.  What the reference recomputes on every
call -- create_generators (33 hash-to-curve operations at L = 32) and the api_id strings -- is
computed once per (ciphersuite, L) by the host side of the library (bbs_create_generators) and kept
in an engine context; msg_to_scalars runs on the device (bbs_hash_to_scalar_batch).

Both ciphersuites of the reference (src/constants.rs): BLS12-381 (simplified SWU hash-to-curve) and BN254
(Shallue-van de Woestijne, restated from RFC 9380 and pinned by the reference's P1 constant).
"""
from __future__ import annotations

import ctypes
import secrets
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from .engine import BbsError, BbsRuntimeError, Engine, Proof, Signature, _bytes_arr, _u8

CIPHERSUITE_ID = {"bls12_381": b"BBS_BLS12381G1_XMD:SHA-256_SSWU_RO_",
                  "bn254": b"BBS_QUUX-V01-CS02-with-BN254G1_XMD:SHA-256_SVDW_RO_"}
SCALAR_ORDER = {
    "bls12_381": 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001,
    "bn254": 21888242871839275222246405745257275088548364400416034343698204186575808495617,
}
KEYGEN_ERRORS = {-7: "InvalidKeyMaterialLength", -8: "InvalidKeyInfoLength", -9: "InvalidSecretKey"}


def api_id(curve: str) -> bytes:
    return CIPHERSUITE_ID[curve] + b"H2G_HM2S_"


_gen_cache: Dict[Tuple[str, int, Optional[str]], list] = {}
_eng_cache: Dict[tuple, Engine] = {}


def create_generators(curve: str, count: int, lib_path: Optional[str] = None) -> list:
    """create_generators::<E, H>(count, api_id) (src/utils/interface_utilities.rs:47-73)."""
    key = (curve, count, lib_path)
    if key not in _gen_cache:
        lib = _lib.load_library(lib_path)
        fpb = int(lib.bbs_fp_bytes(0 if curve == "bls12_381" else 1))
        aid = _bytes_arr(api_id(curve))
        out = np.zeros(max(count, 1) * 2 * fpb, dtype=np.uint8)
        rc = lib.bbs_create_generators(0 if curve == "bls12_381" else 1, count, _u8(aid), len(api_id(curve)), _u8(out))
        if rc:
            raise BbsRuntimeError(rc, "bbs_create_generators")
        b = out.tobytes()
        gens = []
        for k in range(count):
            x = int.from_bytes(b[k * 2 * fpb:k * 2 * fpb + fpb], "little")
            y = int.from_bytes(b[k * 2 * fpb + fpb:(k + 1) * 2 * fpb], "little")
            gens.append(None if x == 0 and y == 0 else (x, y))
        _gen_cache[key] = gens
    return list(_gen_cache[key])


def _engine(curve: str, L: int, *, sk: Optional[int] = None, pk="unset", device: int = 0,
            lib_path: Optional[str] = None, window_bits: Optional[int] = None) -> Engine:
    key = (curve, L, sk, None if pk is None else (pk if pk == "unset" else tuple(map(tuple, pk))), device, lib_path)
    eng = _eng_cache.get(key)
    if eng is None:
        if window_bits is None and lib_path is not None:
            window_bits = 4              # host-twin test library: small tables
        eng = Engine(curve, device=device, lib_path=lib_path, window_bits=window_bits)
        eng.set_generators(create_generators(curve, L + 1, lib_path), api_id(curve))
        if sk is not None:
            eng.set_secret_key(sk)
        elif pk != "unset":
            eng.set_public_key(pk)
        _eng_cache[key] = eng
    return eng


def clear_caches():
    for e in _eng_cache.values():
        e.close()
    _eng_cache.clear()
    _gen_cache.clear()


def msg_to_scalars(eng: Engine, curve: str, messages: Sequence[bytes]) -> List[int]:
    """msg_to_scalars (src/utils/interface_utilities.rs:76-88), on the device."""
    if not messages:
        return []
    return eng.hash_to_scalar_batch(list(messages), api_id(curve) + b"MAP_MSG_TO_SCALAR_AS_HASH_")


class PublicKey:
    """key_gen.rs:12-15; pk is None (identity, `PublicKey::default()`) or ((x0, x1), (y0, y1))."""

    def __init__(self, curve: str, pk, lib_path: Optional[str] = None, device: int = 0):
        self.curve, self.pk, self.lib_path, self.device = curve, pk, lib_path, device

    def verify(self, signature: Signature, header: bytes, messages: Sequence[bytes]) -> bool:
        """PublicKey::verify (src/verify.rs:18-50)."""
        eng = _engine(self.curve, len(messages), pk=self.pk, device=self.device, lib_path=self.lib_path)
        return eng.core_verify(signature, header, msg_to_scalars(eng, self.curve, messages))


class SecretKey:
    def __init__(self, curve: str, sk: int, lib_path: Optional[str] = None, device: int = 0):
        self.curve, self.sk, self.lib_path, self.device = curve, sk, lib_path, device

    @classmethod
    def key_gen(cls, curve: str, key_material: bytes, key_info: bytes, key_dst: bytes,
                lib_path: Optional[str] = None, device: int = 0) -> "SecretKey":
        """SecretKey::key_gen (src/key_gen.rs:46-81); errors as BbsError(KeyGenError variant)."""
        lib = _lib.load_library(lib_path)
        km, ki, kd = _bytes_arr(key_material), _bytes_arr(key_info), _bytes_arr(key_dst)
        out = np.zeros(32, dtype=np.uint8)
        rc = lib.bbs_key_gen(0 if curve == "bls12_381" else 1, _u8(km), len(key_material), _u8(ki), len(key_info),
                             _u8(kd), len(key_dst), _u8(out))
        if rc in KEYGEN_ERRORS:
            e = BbsError(rc)
            e.variant = KEYGEN_ERRORS[rc]
            raise e
        if rc:
            raise BbsRuntimeError(rc, "bbs_key_gen")
        return cls(curve, int.from_bytes(out.tobytes(), "little"), lib_path, device)

    def sk_to_pk(self) -> PublicKey:
        """SecretKey::sk_to_pk (src/key_gen.rs:83-90)."""
        eng = _engine(self.curve, 0, sk=self.sk, device=self.device, lib_path=self.lib_path)
        return PublicKey(self.curve, eng.public_key(), self.lib_path, self.device)

    def sign(self, messages: Sequence[bytes], header: bytes) -> Signature:
        """SecretKey::sign (src/sign.rs:32-60)."""
        eng = _engine(self.curve, len(messages), sk=self.sk, device=self.device, lib_path=self.lib_path)
        return eng.core_sign(header, msg_to_scalars(eng, self.curve, messages))


def hash_to_g1(curve: str, msg: bytes, dst: bytes, lib_path: Optional[str] = None):
    """The suite's hash-to-G1 (src/utils/interface_utilities.rs:24-44), host side of the library."""
    lib = _lib.load_library(lib_path)
    cid = 0 if curve == "bls12_381" else 1
    fpb = int(lib.bbs_fp_bytes(cid))
    out = np.zeros(2 * fpb, dtype=np.uint8)
    m, d = _bytes_arr(msg), _bytes_arr(dst)
    rc = lib.bbs_hash_to_g1(cid, _u8(m), len(msg), _u8(d), len(dst), _u8(out))
    if rc:
        raise BbsRuntimeError(rc, "bbs_hash_to_g1")
    b = out.tobytes()
    x, y = int.from_bytes(b[:fpb], "little"), int.from_bytes(b[fpb:], "little")
    return None if x == 0 and y == 0 else (x, y)


def calculate_random_scalars(curve: str, count: int) -> List[int]:
    """calculate_random_scalars (src/utils/core_utilities.rs:70-81): 48 random bytes mod r each."""
    r = SCALAR_ORDER[curve]
    return [int.from_bytes(secrets.token_bytes(48), "big") % r for _ in range(count)]


def proof_gen(pk: PublicKey, signature: Signature, header: bytes, ph: bytes, messages: Sequence[bytes],
              disclosed_indexes: Sequence[int], random_scalars: Optional[Sequence[int]] = None) -> Proof:
    """proof_gen (src/proof_gen.rs:78-113).  ``random_scalars`` (5 + L - R of them) replaces the draw at
    :145-149; by default they are drawn here exactly as the reference draws them."""
    L, R = len(messages), len(disclosed_indexes)
    eng = _engine(pk.curve, L, pk=pk.pk, device=pk.device, lib_path=pk.lib_path)
    if random_scalars is None:
        random_scalars = calculate_random_scalars(pk.curve, max(5 + L - R, 0))
    return eng.core_proof_gen(signature, header, ph, msg_to_scalars(eng, pk.curve, messages), disclosed_indexes,
                              random_scalars)


def proof_verify(pk: PublicKey, proof: Proof, header: bytes, ph: bytes, disclosed_messages: Sequence[bytes],
                 disclosed_indexes: Sequence[int]) -> bool:
    """proof_verify (src/proof_verify.rs:19-61): L is inferred as commitments + disclosed indexes."""
    L = len(proof.commitments) + len(disclosed_indexes)
    eng = _engine(pk.curve, L, pk=pk.pk, device=pk.device, lib_path=pk.lib_path)
    return eng.core_proof_verify(proof, header, ph, msg_to_scalars(eng, pk.curve, disclosed_messages), disclosed_indexes)


# ---------------------------------------------------------------------------------- wire codec
def _curve_id(curve: str) -> int:
    return 0 if curve == "bls12_381" else 1


def _codec_check(rc: int, where: str):
    if rc == 0:
        return
    if rc in (-40, -41, -42):
        raise BbsError(rc)
    raise BbsRuntimeError(rc, where)


def _g1_rec(fpb, p) -> bytes:
    return bytes(2 * fpb) if p is None else int(p[0]).to_bytes(fpb, "little") + int(p[1]).to_bytes(fpb, "little")


def _g1_unrec(fpb, b):
    x, y = int.from_bytes(b[:fpb], "little"), int.from_bytes(b[fpb:2 * fpb], "little")
    return None if x == 0 and y == 0 else (x, y)


def signature_to_octets(curve: str, sig: Signature, lib_path: Optional[str] = None) -> bytes:
    """signature = compress(A) || I2OSP(e, 32)  (the byte string of src/tests/test_vector.rs:188-191)."""
    lib = _lib.load_library(lib_path)
    fpb = int(lib.bbs_fp_bytes(_curve_id(curve)))
    rec = _bytes_arr(_g1_rec(fpb, sig.a) + int(sig.e).to_bytes(32, "little"))
    out = np.zeros(fpb + 32, dtype=np.uint8)
    _codec_check(lib.bbs_signature_to_octets(_curve_id(curve), _u8(rec), _u8(out)), "bbs_signature_to_octets")
    return out.tobytes()


def octets_to_signature(curve: str, octets: bytes, lib_path: Optional[str] = None) -> Signature:
    lib = _lib.load_library(lib_path)
    fpb = int(lib.bbs_fp_bytes(_curve_id(curve)))
    if len(octets) != fpb + 32:
        raise BbsError(-42)
    rec = np.zeros(2 * fpb + 32, dtype=np.uint8)
    _codec_check(lib.bbs_signature_from_octets(_curve_id(curve), _u8(_bytes_arr(octets)), _u8(rec)), "bbs_signature_from_octets")
    b = rec.tobytes()
    return Signature(_g1_unrec(fpb, b), int.from_bytes(b[2 * fpb:], "little"))


def proof_to_octets(curve: str, proof: Proof, lib_path: Optional[str] = None) -> bytes:
    lib = _lib.load_library(lib_path)
    fpb = int(lib.bbs_fp_bytes(_curve_id(curve)))
    pf = _bytes_arr(_g1_rec(fpb, proof.a_bar) + _g1_rec(fpb, proof.b_bar) + _g1_rec(fpb, proof.d)
                    + b"".join(int(x).to_bytes(32, "little") for x in (proof.e_cap, proof.r1_cap, proof.r3_cap, proof.challenge)))
    cm = _bytes_arr(b"".join(int(x).to_bytes(32, "little") for x in proof.commitments))
    out = np.zeros(3 * fpb + 32 * (4 + len(proof.commitments)), dtype=np.uint8)
    _codec_check(lib.bbs_proof_to_octets(_curve_id(curve), _u8(pf), _u8(cm), len(proof.commitments), _u8(out)), "bbs_proof_to_octets")
    return out.tobytes()


def octets_to_proof(curve: str, octets: bytes, lib_path: Optional[str] = None) -> Proof:
    lib = _lib.load_library(lib_path)
    fpb = int(lib.bbs_fp_bytes(_curve_id(curve)))
    cap = max(len(octets) // 32, 1)
    pf = np.zeros(6 * fpb + 128, dtype=np.uint8)
    cm = np.zeros(cap * 32, dtype=np.uint8)
    n = ctypes.c_size_t(0)
    _codec_check(lib.bbs_proof_from_octets(_curve_id(curve), _u8(_bytes_arr(octets)), len(octets), _u8(pf), _u8(cm), cap,
                                           ctypes.byref(n)), "bbs_proof_from_octets")
    b, c = pf.tobytes(), cm.tobytes()
    pts = [_g1_unrec(fpb, b[k * 2 * fpb:(k + 1) * 2 * fpb]) for k in range(3)]
    sc = [int.from_bytes(b[6 * fpb + 32 * k:6 * fpb + 32 * (k + 1)], "little") for k in range(4)]
    return Proof(pts[0], pts[1], pts[2], sc[0], sc[1], sc[2],
                 [int.from_bytes(c[32 * k:32 * k + 32], "little") for k in range(n.value)], sc[3])


def public_key_to_octets(pk: PublicKey) -> bytes:
    lib = _lib.load_library(pk.lib_path)
    fpb = int(lib.bbs_fp_bytes(_curve_id(pk.curve)))
    out = np.zeros(2 * fpb, dtype=np.uint8)
    if pk.pk is None:
        rec, inf = _bytes_arr(bytes(4 * fpb)), 1
    else:
        (x0, x1), (y0, y1) = pk.pk
        rec, inf = _bytes_arr(b"".join(int(v).to_bytes(fpb, "little") for v in (x0, x1, y0, y1))), 0
    _codec_check(lib.bbs_public_key_to_octets(_curve_id(pk.curve), _u8(rec), inf, _u8(out)), "bbs_public_key_to_octets")
    return out.tobytes()


def octets_to_public_key(curve: str, octets: bytes, lib_path: Optional[str] = None, device: int = 0) -> PublicKey:
    lib = _lib.load_library(lib_path)
    fpb = int(lib.bbs_fp_bytes(_curve_id(curve)))
    if len(octets) != 2 * fpb:
        raise BbsError(-42)
    rec = np.zeros(4 * fpb, dtype=np.uint8)
    inf = ctypes.c_int(0)
    _codec_check(lib.bbs_public_key_from_octets(_curve_id(curve), _u8(_bytes_arr(octets)), _u8(rec), ctypes.byref(inf)),
                 "bbs_public_key_from_octets")
    if inf.value:
        return PublicKey(curve, None, lib_path, device)
    b = rec.tobytes()
    f = [int.from_bytes(b[i * fpb:(i + 1) * fpb], "little") for i in range(4)]
    return PublicKey(curve, ((f[0], f[1]), (f[2], f[3])), lib_path, device)
