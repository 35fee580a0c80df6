"""Mirror of the reference's PUBLIC interface (README.md:43-128) over the MI355X engine:

    SecretKey.key_gen / .sk_to_pk / .sign        src/key_gen.rs:46-90, src/sign.rs:32-60
    PublicKey.verify                              src/verify.rs:18-50
    proof_gen / proof_verify                      src/proof_gen.rs:78-113, src/proof_verify.rs:19-61

Messages are byte strings, exactly as in the reference.  What the reference recomputes on every
call -- create_generators (33 hash-to-curve operations at L = 32) and the api_id strings -- is
computed once per (ciphersuite, L) by the host side of the library (bbs_create_generators) and kept
in an engine context; msg_to_scalars runs on the device (bbs_hash_to_scalar_batch).

BLS12-381 only at this level: the BN254 hash-to-curve backend of the reference (crate
bn254_hash2curve, SvdW) is not restated; BN254 is served at the core_* level (bbs_sign_amd.Engine)
with caller-supplied generators, as the reference's own core tests do.
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
        if rc == -106:
            raise NotImplementedError("create_generators for %s (SvdW hash-to-curve is not restated)" % curve)
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
