"""Host-side mirror of the reference's core_* interface over the C ABI (ctypes).

Values are exchanged as plain Python integers / tuples so that parity tests read like the
reference's own tests:
    scalar  : int in [0, r)
    G1 point: None (identity) or (x, y)
    G2 point: None (identity) or ((x0, x1), (y0, y1))
    Signature(a, e) / Proof(a_bar, b_bar, d, e_cap, r1_cap, r3_cap, commitments, challenge)
mirror src/sign.rs:18-22 and src/proof_gen.rs:29-39.

Errors: a negative per-item status becomes ``BbsError(variant)`` with the reference's variant
name (single-item helpers) or is returned as the raw status (batch helpers).
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

import numpy as np

from . import _lib

BLS12_381 = 0
BN254 = 1
CURVE_IDS = {"bls12_381": BLS12_381, "bn254": BN254}

STATUS_NAMES = {
    1: "Ok(true)", 0: "Ok(false)",
    -1: "InvalidMessageAndGeneratorsLength",
    -2: "InvalidDisclosedIndicesLength",
    -3: "InvalidDisclosedIndex",
    -4: "InvalidRandomScalarsAndUndisclosedIndicesLength",
    -5: "InvalidUndisclosedIndicesLength",
    -6: "InvalidIndicesAndMessagesLength",
    -20: "Panic(sk+e has no inverse)",
    -21: "Panic(r2 has no inverse)",
    -22: "Panic(index out of bounds: proof.commitments)",
    -23: "Panic(dst size is invalid)",
    -40: "NonCanonical",
    -41: "NotOnCurve",
    -42: "InvalidEncoding",
}
ERRORS = {-100: "BBS_E_ARG", -101: "BBS_E_HIP", -102: "BBS_E_STATE", -103: "BBS_E_PUBLIC_KEY",
          -104: "BBS_E_NO_DEVICE", -105: "BBS_E_NOMEM"}


class BbsError(Exception):
    """A reference ``Err(variant)`` / panic for one item."""

    def __init__(self, status: int):
        self.status = int(status)
        self.variant = STATUS_NAMES.get(int(status), "status %d" % status)
        super().__init__(self.variant)


class BbsRuntimeError(RuntimeError):
    """Batch-level failure (bad argument, HIP error, missing key/generators)."""

    def __init__(self, rc: int, where: str):
        self.rc = rc
        super().__init__("%s failed: %s" % (where, ERRORS.get(rc, rc)))


@dataclass
class Signature:
    a: Optional[tuple]
    e: int


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


def _u8(buf) -> "ctypes.POINTER(ctypes.c_uint8)":
    return buf.ctypes.data_as(_lib.c_u8p)


def _u64(buf):
    return buf.ctypes.data_as(_lib.c_u64p)


def _bytes_arr(b: bytes) -> np.ndarray:
    return np.frombuffer(bytes(b) + b"\0", dtype=np.uint8).copy()


def _ragged_bytes(items: Sequence[bytes]):
    off = np.zeros(len(items) + 1, dtype=np.uint64)
    for i, it in enumerate(items):
        off[i + 1] = off[i] + len(it)
    return _bytes_arr(b"".join(bytes(x) for x in items)), off


class Engine:
    """One context: (curve, GPU, generator set + api_id, issuer key)."""

    def __init__(self, curve, device: int = 0, lib_path: Optional[str] = None, window_bits: Optional[int] = None):
        self.lib = _lib.load_library(lib_path)
        self.curve = CURVE_IDS[curve] if isinstance(curve, str) else int(curve)
        self.fpb = int(self.lib.bbs_fp_bytes(self.curve))
        h = ctypes.c_void_p()
        rc = self.lib.bbs_ctx_create(self.curve, device, ctypes.byref(h))
        if rc:
            raise BbsRuntimeError(rc, "bbs_ctx_create")
        self.h = h
        self.L = None
        if window_bits is not None:
            self._chk(self.lib.bbs_ctx_set_window_bits(self.h, window_bits), "bbs_ctx_set_window_bits")

    def close(self):
        if getattr(self, "h", None):
            self.lib.bbs_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _chk(rc, where):
        if rc:
            raise BbsRuntimeError(rc, where)

    # ------------------------------------------------------------------ encoders
    def _fp(self, v: int) -> bytes:
        return int(v).to_bytes(self.fpb, "little")

    @staticmethod
    def _fr(v: int) -> bytes:
        return int(v).to_bytes(32, "little")

    def _g1(self, p) -> bytes:
        if p is None:
            return bytes(2 * self.fpb)
        return self._fp(p[0]) + self._fp(p[1])

    def _g1_dec(self, b: bytes):
        x = int.from_bytes(b[:self.fpb], "little")
        y = int.from_bytes(b[self.fpb:2 * self.fpb], "little")
        return None if (x == 0 and y == 0) else (x, y)

    def _scalars(self, rows: Sequence[Sequence[int]]):
        off = np.zeros(len(rows) + 1, dtype=np.uint64)
        chunks = []
        for i, row in enumerate(rows):
            off[i + 1] = off[i] + len(row)
            chunks.extend(self._fr(s) for s in row)
        return _bytes_arr(b"".join(chunks)), off

    @staticmethod
    def _indexes(rows: Sequence[Sequence[int]]):
        off = np.zeros(len(rows) + 1, dtype=np.uint64)
        flat = []
        for i, row in enumerate(rows):
            off[i + 1] = off[i] + len(row)
            flat.extend(int(x) for x in row)
        return np.array(flat + [0], dtype=np.uint64), off

    def _sigs(self, sigs: Sequence[Signature]) -> np.ndarray:
        return _bytes_arr(b"".join(self._g1(s.a) + self._fr(s.e) for s in sigs))

    # --------------------------------------------------------------------- setup
    def set_generators(self, generators: Sequence, api_id: bytes):
        """``generators`` = [Q1, H_1..H_L] -- the `generators: &[E::G1]` of every core_* fn."""
        buf = _bytes_arr(b"".join(self._g1(g) for g in generators))
        aid = _bytes_arr(api_id)
        self._chk(self.lib.bbs_ctx_set_generators(self.h, _u8(buf), len(generators), _u8(aid), len(api_id)),
                  "bbs_ctx_set_generators")
        self.L = len(generators) - 1

    def set_public_key(self, pk):
        if pk is None:
            self._chk(self.lib.bbs_ctx_set_public_key(self.h, None, 1), "bbs_ctx_set_public_key")
            return
        (x0, x1), (y0, y1) = pk
        buf = _bytes_arr(self._fp(x0) + self._fp(x1) + self._fp(y0) + self._fp(y1))
        self._chk(self.lib.bbs_ctx_set_public_key(self.h, _u8(buf), 0), "bbs_ctx_set_public_key")

    def set_secret_key(self, sk: int):
        buf = _bytes_arr(self._fr(sk))
        self._chk(self.lib.bbs_ctx_set_secret_key(self.h, _u8(buf)), "bbs_ctx_set_secret_key")

    def set_points_in_subgroup(self, vouched: bool):
        """bbs_ctx_set_points_in_subgroup: the caller vouches that every G1 input is in the prime-order subgroup."""
        self._chk(self.lib.bbs_ctx_set_points_in_subgroup(self.h, 1 if vouched else 0), "bbs_ctx_set_points_in_subgroup")

    def set_fixed_base_tree(self, enabled: bool):
        """bbs_ctx_set_fixed_base_tree: fixed-base sums as one tree of affine additions per item (jobs created afterwards)."""
        self._chk(self.lib.bbs_ctx_set_fixed_base_tree(self.h, 1 if enabled else 0), "bbs_ctx_set_fixed_base_tree")

    def set_latency_mode(self, enabled):
        """bbs_ctx_set_latency_mode: the latency form of a job (T1 as three multiplications on three lanes, the two Miller
        loops of a pairing product on separate lane groups).  False / 0 = never, True / 1 = always, "auto" / 2 = the
        library's default: a job gets it when at most one other job of the context is alive."""
        mode = 2 if enabled in ("auto", 2) else (1 if enabled else 0)
        self._chk(self.lib.bbs_ctx_set_latency_mode(self.h, mode), "bbs_ctx_set_latency_mode")

    def set_batch_verification(self, enabled: bool, seed: Optional[bytes] = None):
        """Opt-in random-linear-combination batch verification for core_proof_verify (include/bbs_sign_amd.h);
        seed=None draws the secret seed from the operating system."""
        if seed is not None and len(seed) != 32:
            raise ValueError("seed must be 32 bytes")
        buf = _bytes_arr(seed) if seed is not None else None
        self._chk(self.lib.bbs_ctx_set_batch_verification(self.h, 1 if enabled else 0, _u8(buf) if buf is not None else None),
                  "bbs_ctx_set_batch_verification")

    def set_stage_timing(self, enabled: bool):
        """bbs_ctx_set_stage_timing: jobs created afterwards record HIP events around every stage (Job.stage_times)."""
        self._chk(self.lib.bbs_ctx_set_stage_timing(self.h, 1 if enabled else 0), "bbs_ctx_set_stage_timing")

    def public_key(self):
        """sk_to_pk (src/key_gen.rs:83-90) of the secret key set on this context."""
        out = np.zeros(4 * self.fpb, dtype=np.uint8)
        inf = ctypes.c_int(0)
        self._chk(self.lib.bbs_ctx_get_public_key(self.h, _u8(out), ctypes.byref(inf)), "bbs_ctx_get_public_key")
        if inf.value:
            return None
        b = out.tobytes()
        f = [int.from_bytes(b[i * self.fpb:(i + 1) * self.fpb], "little") for i in range(4)]
        return ((f[0], f[1]), (f[2], f[3]))

    def public_key_compressed(self) -> bytes:
        out = np.zeros(2 * self.fpb, dtype=np.uint8)
        n = ctypes.c_size_t(0)
        self._chk(self.lib.bbs_ctx_get_public_key_compressed(self.h, _u8(out), out.size, ctypes.byref(n)),
                  "bbs_ctx_get_public_key_compressed")
        return out.tobytes()[:n.value]

    # ------------------------------------------------------------ batched core_*
    def _pv_inputs(self, proofs, disclosed_msgs, disclosed_idx, headers, phs):
        n = len(proofs)
        rec = b"".join(
            self._g1(p.a_bar) + self._g1(p.b_bar) + self._g1(p.d) + self._fr(p.e_cap) + self._fr(p.r1_cap)
            + self._fr(p.r3_cap) + self._fr(p.challenge) for p in proofs)
        pf = _bytes_arr(rec)
        cm, cmo = self._scalars([p.commitments for p in proofs])
        dm, dmo = self._scalars(disclosed_msgs)
        di, dio = self._indexes(disclosed_idx)
        hb, ho = _ragged_bytes(headers if headers is not None else [b""] * n)
        pb, po = _ragged_bytes(phs if phs is not None else [b""] * n)
        keep = (pf, cm, cmo, dm, dmo, di, dio, hb, ho, pb, po)
        args = (_u8(pf), _u8(cm), _u64(cmo), _u8(dm), _u64(dmo), _u64(di), _u64(dio), _u8(hb), _u64(ho), _u8(pb), _u64(po))
        return n, keep, args

    def core_proof_verify_batch(self, proofs, disclosed_msgs, disclosed_idx, headers=None, phs=None) -> np.ndarray:
        """core_proof_verify (src/proof_verify.rs:64-116) over a batch; returns int8 statuses."""
        n, keep, args = self._pv_inputs(proofs, disclosed_msgs, disclosed_idx, headers, phs)
        st = np.zeros(max(n, 1), dtype=np.int8)
        self._chk(self.lib.bbs_core_proof_verify_batch(self.h, n, *args, st.ctypes.data_as(_lib.c_i8p)),
                  "bbs_core_proof_verify_batch")
        return st[:n]

    def core_proof_verify_upload(self, proofs, disclosed_msgs, disclosed_idx, headers=None, phs=None) -> "Job":
        n, keep, args = self._pv_inputs(proofs, disclosed_msgs, disclosed_idx, headers, phs)
        j = ctypes.c_void_p()
        self._chk(self.lib.bbs_core_proof_verify_upload(self.h, n, *args, ctypes.byref(j)), "bbs_core_proof_verify_upload")
        return Job(self, j, n)

    def core_proof_verify_submit(self, proofs, disclosed_msgs, disclosed_idx, headers=None, phs=None) -> "Job":
        """bbs_core_proof_verify_submit: everything enqueued, nothing waited for; ``job.wait()`` then ``job.result``."""
        n, keep, args = self._pv_inputs(proofs, disclosed_msgs, disclosed_idx, headers, phs)
        return self.submit_packed(n, args)

    def submit_packed(self, n, args) -> "Job":
        """The same from already packed ctypes arguments (``_pv_inputs``): the per-batch host work of a serving loop."""
        st = np.full(max(n, 1), -128, dtype=np.int8)
        j = ctypes.c_void_p()
        self._chk(self.lib.bbs_core_proof_verify_submit(self.h, n, *args, st.ctypes.data_as(_lib.c_i8p), ctypes.byref(j)),
                  "bbs_core_proof_verify_submit")
        job = Job(self, j, n)
        job.result = st[:n]
        return job

    def core_verify_batch(self, signatures, messages, headers=None) -> np.ndarray:
        """core_verify (src/verify.rs:53-93) over a batch."""
        n = len(signatures)
        sg = self._sigs(signatures)
        ms, mo = self._scalars(messages)
        hb, ho = _ragged_bytes(headers if headers is not None else [b""] * n)
        st = np.zeros(max(n, 1), dtype=np.int8)
        self._chk(self.lib.bbs_core_verify_batch(self.h, n, _u8(sg), _u8(ms), _u64(mo), _u8(hb), _u64(ho),
                                                 st.ctypes.data_as(_lib.c_i8p)), "bbs_core_verify_batch")
        return st[:n]

    def _oct_inputs(self, octets, disclosed_msgs, disclosed_idx, headers, phs):
        n = len(octets)
        ob, oo = _ragged_bytes(octets)
        dm, dmo = self._scalars(disclosed_msgs)
        di, dio = self._indexes(disclosed_idx)
        hb, ho = _ragged_bytes(headers if headers is not None else [b""] * n)
        pb, po = _ragged_bytes(phs if phs is not None else [b""] * n)
        keep = (ob, oo, dm, dmo, di, dio, hb, ho, pb, po)
        args = (_u8(ob), _u64(oo), _u8(dm), _u64(dmo), _u64(di), _u64(dio), _u8(hb), _u64(ho), _u8(pb), _u64(po))
        return n, keep, args

    def proof_verify_octets_batch(self, octets, disclosed_msgs, disclosed_idx, headers=None, phs=None) -> np.ndarray:
        """bbs_proof_verify_octets_batch: proof octet strings in (decoded and subgroup-checked on the device), statuses out."""
        n, keep, args = self._oct_inputs(octets, disclosed_msgs, disclosed_idx, headers, phs)
        st = np.full(max(n, 1), -128, dtype=np.int8)
        self._chk(self.lib.bbs_proof_verify_octets_batch(self.h, n, *args, st.ctypes.data_as(_lib.c_i8p)), "bbs_proof_verify_octets_batch")
        return st[:n]

    @staticmethod
    def _raw_msgs(items):
        """items[i] = the messages of item i as byte strings -> (flat bytes, per-message byte offsets, per-item message offsets)."""
        mb, mbo = _ragged_bytes([m for item in items for m in item])
        mio = np.zeros(len(items) + 1, dtype=np.uint64)
        for i, item in enumerate(items):
            mio[i + 1] = mio[i] + len(item)
        return mb, mbo, mio

    def verify_wire_batch(self, sig_octets, messages_raw, headers=None) -> np.ndarray:
        """bbs_verify_wire_batch: the reference's public verify -- signature octets and raw messages in, statuses out."""
        n = len(sig_octets)
        ob, bad = self._sig_octets(sig_octets)
        mb, mbo, mio = self._raw_msgs(messages_raw)
        hb, ho = _ragged_bytes(headers if headers is not None else [b""] * n)
        st = np.full(max(n, 1), -128, dtype=np.int8)
        self._chk(self.lib.bbs_verify_wire_batch(self.h, n, _u8(ob), _u8(mb), _u64(mbo), _u64(mio), _u8(hb), _u64(ho),
                                                 st.ctypes.data_as(_lib.c_i8p)), "bbs_verify_wire_batch")
        st[bad] = -42
        return st[:n]

    def sign_wire_batch(self, messages_raw, headers=None):
        """bbs_sign_wire_batch: the reference's public sign -- raw messages in, (signature octet strings, statuses) out."""
        n = len(messages_raw)
        mb, mbo, mio = self._raw_msgs(messages_raw)
        hb, ho = _ragged_bytes(headers if headers is not None else [b""] * n)
        st = np.full(max(n, 1), -128, dtype=np.int8)
        rec = self.fpb + 32
        out = np.zeros(max(n, 1) * rec, dtype=np.uint8)
        self._chk(self.lib.bbs_sign_wire_batch(self.h, n, _u8(mb), _u64(mbo), _u64(mio), _u8(hb), _u64(ho), _u8(out),
                                               st.ctypes.data_as(_lib.c_i8p)), "bbs_sign_wire_batch")
        return [out[i * rec:(i + 1) * rec].tobytes() if st[i] == 1 else b"" for i in range(n)], st[:n]

    def proof_gen_wire_batch(self, sig_octets, messages_raw, disclosed_idx, random_scalars, headers=None, phs=None):
        """bbs_proof_gen_wire_batch: the reference's public proof_gen -- signature octets, raw messages, indexes and the
        random scalars in, (proof octet strings, statuses) out."""
        n = len(sig_octets)
        ob, bad = self._sig_octets(sig_octets)
        mb, mbo, mio = self._raw_msgs(messages_raw)
        di, dio = self._indexes(disclosed_idx)
        rs, ro = self._scalars(random_scalars)
        hb, ho = _ragged_bytes(headers if headers is not None else [b""] * n)
        pb, po = _ragged_bytes(phs if phs is not None else [b""] * n)
        cap = sum(3 * self.fpb + 32 * (4 + len(m)) for m in messages_raw)
        st = np.full(max(n, 1), -128, dtype=np.int8)
        oc = np.zeros(max(cap, 1), dtype=np.uint8)
        oo = np.zeros(n + 1, dtype=np.uint64)
        self._chk(self.lib.bbs_proof_gen_wire_batch(self.h, n, _u8(ob), _u8(mb), _u64(mbo), _u64(mio), _u64(di), _u64(dio), _u8(rs), _u64(ro),
                                                    _u8(hb), _u64(ho), _u8(pb), _u64(po), _u8(oc), _u64(oo),
                                                    st.ctypes.data_as(_lib.c_i8p)), "bbs_proof_gen_wire_batch")
        out = [oc[int(oo[i]):int(oo[i + 1])].tobytes() for i in range(n)]
        for i in bad:
            st[i] = -42
            out[i] = b""
        return out, st[:n]

    def _wire_inputs(self, octets, disclosed_raw, disclosed_idx, headers, phs):
        """disclosed_raw[i]: the disclosed messages of item i as byte strings (hashed on the device)."""
        n = len(octets)
        ob, oo = _ragged_bytes(octets)
        flat = [m for item in disclosed_raw for m in item]
        mb, mbo = _ragged_bytes(flat)
        mio = np.zeros(n + 1, dtype=np.uint64)
        for i, item in enumerate(disclosed_raw):
            mio[i + 1] = mio[i] + len(item)
        di, dio = self._indexes(disclosed_idx)
        hb, ho = _ragged_bytes(headers if headers is not None else [b""] * n)
        pb, po = _ragged_bytes(phs if phs is not None else [b""] * n)
        keep = (ob, oo, mb, mbo, mio, di, dio, hb, ho, pb, po)
        args = (_u8(ob), _u64(oo), _u8(mb), _u64(mbo), _u64(mio), _u64(di), _u64(dio), _u8(hb), _u64(ho), _u8(pb), _u64(po))
        return n, keep, args

    def proof_verify_wire_batch(self, octets, disclosed_raw, disclosed_idx, headers=None, phs=None) -> np.ndarray:
        """bbs_proof_verify_wire_batch: proof octets and raw disclosed messages in, statuses out (the reference's public
        proof_verify for this context's number of messages)."""
        n, keep, args = self._wire_inputs(octets, disclosed_raw, disclosed_idx, headers, phs)
        st = np.full(max(n, 1), -128, dtype=np.int8)
        self._chk(self.lib.bbs_proof_verify_wire_batch(self.h, n, *args, st.ctypes.data_as(_lib.c_i8p)), "bbs_proof_verify_wire_batch")
        return st[:n]

    def proof_verify_wire_submit(self, octets, disclosed_raw, disclosed_idx, headers=None, phs=None) -> "Job":
        n, keep, args = self._wire_inputs(octets, disclosed_raw, disclosed_idx, headers, phs)
        st = np.full(max(n, 1), -128, dtype=np.int8)
        j = ctypes.c_void_p()
        self._chk(self.lib.bbs_proof_verify_wire_submit(self.h, n, *args, st.ctypes.data_as(_lib.c_i8p), ctypes.byref(j)),
                  "bbs_proof_verify_wire_submit")
        job = Job(self, j, n)
        job.result = st[:n]
        return job

    def proof_verify_octets_submit_packed(self, n, args) -> "Job":
        st = np.full(max(n, 1), -128, dtype=np.int8)
        j = ctypes.c_void_p()
        self._chk(self.lib.bbs_proof_verify_octets_submit(self.h, n, *args, st.ctypes.data_as(_lib.c_i8p), ctypes.byref(j)),
                  "bbs_proof_verify_octets_submit")
        job = Job(self, j, n)
        job.result = st[:n]
        return job

    def core_verify_submit(self, signatures, messages, headers=None) -> "Job":
        """bbs_core_verify_submit: everything enqueued, nothing waited for; ``job.wait()`` then ``job.result``."""
        n = len(signatures)
        sg = self._sigs(signatures)
        ms, mo = self._scalars(messages)
        hb, ho = _ragged_bytes(headers if headers is not None else [b""] * n)
        st = np.full(max(n, 1), -128, dtype=np.int8)
        j = ctypes.c_void_p()
        self._chk(self.lib.bbs_core_verify_submit(self.h, n, _u8(sg), _u8(ms), _u64(mo), _u8(hb), _u64(ho),
                                                  st.ctypes.data_as(_lib.c_i8p), ctypes.byref(j)), "bbs_core_verify_submit")
        job = Job(self, j, n)
        job.result = st[:n]
        return job

    def _sig_octets(self, octets):
        """-> (flat buffer of n strings of fp_bytes + 32 octets, indexes of the items whose string has another length).
        The C ABI takes a fixed stride; a string of the wrong length is malformed (-42, as octets_to_signature gives) and
        travels as zeros, its status is overwritten afterwards."""
        want = self.fpb + 32
        bad = [i for i, o in enumerate(octets) if len(o) != want]
        flat = b"".join(o if len(o) == want else bytes(want) for o in octets)
        return (_bytes_arr(flat) if octets else np.zeros(1, dtype=np.uint8)), bad

    def verify_octets_batch(self, sig_octets, messages, headers=None) -> np.ndarray:
        """bbs_verify_octets_batch: signature octet strings in (decoded and subgroup-checked on the device), statuses out."""
        n = len(sig_octets)
        ob, bad = self._sig_octets(sig_octets)
        ms, mo = self._scalars(messages)
        hb, ho = _ragged_bytes(headers if headers is not None else [b""] * n)
        st = np.full(max(n, 1), -128, dtype=np.int8)
        self._chk(self.lib.bbs_verify_octets_batch(self.h, n, _u8(ob), _u8(ms), _u64(mo), _u8(hb), _u64(ho),
                                                   st.ctypes.data_as(_lib.c_i8p)), "bbs_verify_octets_batch")
        st[bad] = -42
        return st[:n]

    def verify_octets_submit(self, sig_octets, messages, headers=None) -> "Job":
        n = len(sig_octets)
        ob, bad = self._sig_octets(sig_octets)
        ms, mo = self._scalars(messages)
        hb, ho = _ragged_bytes(headers if headers is not None else [b""] * n)
        st = np.full(max(n, 1), -128, dtype=np.int8)
        j = ctypes.c_void_p()
        self._chk(self.lib.bbs_verify_octets_submit(self.h, n, _u8(ob), _u8(ms), _u64(mo), _u8(hb), _u64(ho),
                                                    st.ctypes.data_as(_lib.c_i8p), ctypes.byref(j)), "bbs_verify_octets_submit")
        job = Job(self, j, n)
        job.result = st[:n]
        job._fixup = (lambda: st.__setitem__(bad, -42)) if bad else None
        return job

    def core_verify_upload(self, signatures, messages, headers=None) -> "Job":
        n = len(signatures)
        sg = self._sigs(signatures)
        ms, mo = self._scalars(messages)
        hb, ho = _ragged_bytes(headers if headers is not None else [b""] * n)
        j = ctypes.c_void_p()
        self._chk(self.lib.bbs_core_verify_upload(self.h, n, _u8(sg), _u8(ms), _u64(mo), _u8(hb), _u64(ho),
                                                  ctypes.byref(j)), "bbs_core_verify_upload")
        return Job(self, j, n)

    def _dec_sigs(self, out: np.ndarray, st: np.ndarray, n: int):
        rec = 2 * self.fpb + 32
        b = out.tobytes()
        sigs = []
        for i in range(n):
            if st[i] != 1:
                sigs.append(None)
                continue
            r = b[i * rec:(i + 1) * rec]
            sigs.append(Signature(self._g1_dec(r), int.from_bytes(r[2 * self.fpb:], "little")))
        return sigs

    def core_sign_batch(self, messages, headers=None):
        """core_sign (src/sign.rs:63-133) over a batch -> (signatures | None, statuses)."""
        n = len(messages)
        ms, mo = self._scalars(messages)
        hb, ho = _ragged_bytes(headers if headers is not None else [b""] * n)
        st = np.zeros(max(n, 1), dtype=np.int8)
        out = np.zeros(max(n, 1) * (2 * self.fpb + 32), dtype=np.uint8)
        self._chk(self.lib.bbs_core_sign_batch(self.h, n, _u8(ms), _u64(mo), _u8(hb), _u64(ho), _u8(out),
                                               st.ctypes.data_as(_lib.c_i8p)), "bbs_core_sign_batch")
        return self._dec_sigs(out, st, n), st[:n]

    def core_sign_submit(self, messages, headers=None) -> "Job":
        """bbs_core_sign_submit: everything enqueued, nothing waited for; ``job.wait()`` then ``job.result`` (statuses)
        and ``job.output()`` -> (signatures | None, statuses)."""
        n = len(messages)
        ms, mo = self._scalars(messages)
        hb, ho = _ragged_bytes(headers if headers is not None else [b""] * n)
        st = np.full(max(n, 1), -128, dtype=np.int8)
        out = np.zeros(max(n, 1) * (2 * self.fpb + 32), dtype=np.uint8)
        j = ctypes.c_void_p()
        self._chk(self.lib.bbs_core_sign_submit(self.h, n, _u8(ms), _u64(mo), _u8(hb), _u64(ho), _u8(out),
                                                st.ctypes.data_as(_lib.c_i8p), ctypes.byref(j)), "bbs_core_sign_submit")
        job = Job(self, j, n)
        job.result = st[:n]
        job._decode = lambda: (self._dec_sigs(out, st, n), st[:n])
        return job

    def sign_octets_submit(self, messages, headers=None) -> "Job":
        """bbs_sign_octets_submit: signatures leave as octet strings; ``job.wait()`` then ``job.output()`` ->
        (list of bytes, b"" for failed items; statuses)."""
        n = len(messages)
        ms, mo = self._scalars(messages)
        hb, ho = _ragged_bytes(headers if headers is not None else [b""] * n)
        st = np.full(max(n, 1), -128, dtype=np.int8)
        rec = self.fpb + 32
        out = np.zeros(max(n, 1) * rec, dtype=np.uint8)
        j = ctypes.c_void_p()
        self._chk(self.lib.bbs_sign_octets_submit(self.h, n, _u8(ms), _u64(mo), _u8(hb), _u64(ho), _u8(out),
                                                  st.ctypes.data_as(_lib.c_i8p), ctypes.byref(j)), "bbs_sign_octets_submit")
        job = Job(self, j, n)
        job.result = st[:n]
        job._decode = lambda: ([out[i * rec:(i + 1) * rec].tobytes() if st[i] == 1 else b"" for i in range(n)], st[:n])
        return job

    def sign_octets_batch(self, messages, headers=None):
        job = self.sign_octets_submit(messages, headers)
        job.wait()
        r = job.output()
        job.free()
        return r

    def proof_gen_octets_submit(self, signatures, messages, disclosed_idx, random_scalars, headers=None, phs=None) -> "Job":
        """bbs_proof_gen_octets_submit: proofs leave as octet strings; ``job.output()`` -> (list of bytes, statuses)."""
        n, keep, args = self._pg_inputs(signatures, messages, disclosed_idx, random_scalars, headers, phs)
        cap = sum(3 * self.fpb + 32 * (4 + len(m)) for m in messages)
        st = np.full(max(n, 1), -128, dtype=np.int8)
        oc = np.zeros(max(cap, 1), dtype=np.uint8)
        oo = np.zeros(n + 1, dtype=np.uint64)
        j = ctypes.c_void_p()
        self._chk(self.lib.bbs_proof_gen_octets_submit(self.h, n, *args, _u8(oc), _u64(oo), st.ctypes.data_as(_lib.c_i8p),
                                                       ctypes.byref(j)), "bbs_proof_gen_octets_submit")
        job = Job(self, j, n)
        job.result = st[:n]
        job._decode = lambda: ([oc[int(oo[i]):int(oo[i + 1])].tobytes() for i in range(n)], st[:n])
        return job

    def proof_gen_octets_batch(self, signatures, messages, disclosed_idx, random_scalars, headers=None, phs=None):
        job = self.proof_gen_octets_submit(signatures, messages, disclosed_idx, random_scalars, headers, phs)
        job.wait()
        r = job.output()
        job.free()
        return r

    def core_sign_upload(self, messages, headers=None) -> "Job":
        n = len(messages)
        ms, mo = self._scalars(messages)
        hb, ho = _ragged_bytes(headers if headers is not None else [b""] * n)
        j = ctypes.c_void_p()
        self._chk(self.lib.bbs_core_sign_upload(self.h, n, _u8(ms), _u64(mo), _u8(hb), _u64(ho), ctypes.byref(j)),
                  "bbs_core_sign_upload")
        return Job(self, j, n)

    def _pg_inputs(self, signatures, messages, disclosed_idx, random_scalars, headers, phs):
        n = len(signatures)
        sg = self._sigs(signatures)
        ms, mo = self._scalars(messages)
        di, dio = self._indexes(disclosed_idx)
        rs, ro = self._scalars(random_scalars)
        hb, ho = _ragged_bytes(headers if headers is not None else [b""] * n)
        pb, po = _ragged_bytes(phs if phs is not None else [b""] * n)
        keep = (sg, ms, mo, di, dio, rs, ro, hb, ho, pb, po)
        args = (_u8(sg), _u8(ms), _u64(mo), _u64(di), _u64(dio), _u8(rs), _u64(ro), _u8(hb), _u64(ho), _u8(pb), _u64(po))
        return n, keep, args

    def _dec_proofs(self, pf: np.ndarray, cm: np.ndarray, cmo: np.ndarray, st: np.ndarray, n: int):
        rec = 6 * self.fpb + 128
        b = pf.tobytes()
        c = cm.tobytes()
        proofs = []
        for i in range(n):
            if st[i] != 1:
                proofs.append(None)
                continue
            r = b[i * rec:(i + 1) * rec]
            pts = [self._g1_dec(r[k * 2 * self.fpb:(k + 1) * 2 * self.fpb]) for k in range(3)]
            sc = [int.from_bytes(r[6 * self.fpb + 32 * k:6 * self.fpb + 32 * (k + 1)], "little") for k in range(4)]
            cms = [int.from_bytes(c[32 * k:32 * (k + 1)], "little") for k in range(int(cmo[i]), int(cmo[i + 1]))]
            proofs.append(Proof(pts[0], pts[1], pts[2], sc[0], sc[1], sc[2], cms, sc[3]))
        return proofs

    def core_proof_gen_batch(self, signatures, messages, disclosed_idx, random_scalars, headers=None, phs=None):
        """core_proof_gen (src/proof_gen.rs:116-208) over a batch -> (proofs | None, statuses).
        ``random_scalars[i]`` are the 5 + L - R scalars the reference draws at :145-149."""
        n, keep, args = self._pg_inputs(signatures, messages, disclosed_idx, random_scalars, headers, phs)
        total = sum(len(m) for m in messages)
        st = np.zeros(max(n, 1), dtype=np.int8)
        pf = np.zeros(max(n, 1) * (6 * self.fpb + 128), dtype=np.uint8)
        cm = np.zeros(max(total, 1) * 32, dtype=np.uint8)
        cmo = np.zeros(n + 1, dtype=np.uint64)
        self._chk(self.lib.bbs_core_proof_gen_batch(self.h, n, *args, _u8(pf), _u8(cm), _u64(cmo),
                                                    st.ctypes.data_as(_lib.c_i8p)), "bbs_core_proof_gen_batch")
        return self._dec_proofs(pf, cm, cmo, st, n), st[:n]

    def core_proof_gen_submit(self, signatures, messages, disclosed_idx, random_scalars, headers=None, phs=None) -> "Job":
        """bbs_core_proof_gen_submit; ``job.wait()`` then ``job.result`` and ``job.output()`` -> (proofs | None, statuses)."""
        n, keep, args = self._pg_inputs(signatures, messages, disclosed_idx, random_scalars, headers, phs)
        total = sum(len(m) for m in messages)
        st = np.full(max(n, 1), -128, dtype=np.int8)
        pf = np.zeros(max(n, 1) * (6 * self.fpb + 128), dtype=np.uint8)
        cm = np.zeros(max(total, 1) * 32, dtype=np.uint8)
        cmo = np.zeros(n + 1, dtype=np.uint64)
        j = ctypes.c_void_p()
        self._chk(self.lib.bbs_core_proof_gen_submit(self.h, n, *args, _u8(pf), _u8(cm), _u64(cmo),
                                                     st.ctypes.data_as(_lib.c_i8p), ctypes.byref(j)), "bbs_core_proof_gen_submit")
        job = Job(self, j, n)
        job.result = st[:n]
        job._decode = lambda: (self._dec_proofs(pf, cm, cmo, st, n), st[:n])
        return job

    def core_proof_gen_upload(self, signatures, messages, disclosed_idx, random_scalars, headers=None, phs=None) -> "Job":
        n, keep, args = self._pg_inputs(signatures, messages, disclosed_idx, random_scalars, headers, phs)
        j = ctypes.c_void_p()
        self._chk(self.lib.bbs_core_proof_gen_upload(self.h, n, *args, ctypes.byref(j)), "bbs_core_proof_gen_upload")
        job = Job(self, j, n)
        job.total_msgs = sum(len(m) for m in messages)
        return job

    # --------------------------------------------------------- single-item mirror
    @staticmethod
    def _one(st):
        s = int(st[0])
        if s < 0:
            raise BbsError(s)
        return s

    def core_sign(self, header: bytes, messages: Sequence[int]) -> Signature:
        sigs, st = self.core_sign_batch([list(messages)], [header])
        self._one(st)
        return sigs[0]

    def core_verify(self, signature: Signature, header: bytes, messages: Sequence[int]) -> bool:
        return bool(self._one(self.core_verify_batch([signature], [list(messages)], [header])))

    def core_proof_gen(self, signature: Signature, header: bytes, ph: bytes, messages: Sequence[int],
                       disclosed_indexes: Sequence[int], random_scalars: Sequence[int]) -> Proof:
        proofs, st = self.core_proof_gen_batch([signature], [list(messages)], [list(disclosed_indexes)],
                                               [list(random_scalars)], [header], [ph])
        self._one(st)
        return proofs[0]

    def core_proof_verify(self, proof: Proof, header: bytes, ph: bytes, disclosed_messages: Sequence[int],
                          disclosed_indexes: Sequence[int]) -> bool:
        st = self.core_proof_verify_batch([proof], [list(disclosed_messages)], [list(disclosed_indexes)], [header], [ph])
        return bool(self._one(st))

    # ------------------------------------------------------------------ primitives
    def hash_to_scalar_batch(self, msgs: Sequence[bytes], dst: bytes) -> List[int]:
        n = len(msgs)
        mb, mo = _ragged_bytes(msgs)
        d = _bytes_arr(dst)
        out = np.zeros(max(n, 1) * 32, dtype=np.uint8)
        self._chk(self.lib.bbs_hash_to_scalar_batch(self.h, n, _u8(mb), _u64(mo), _u8(d), len(dst), _u8(out)),
                  "bbs_hash_to_scalar_batch")
        b = out.tobytes()
        return [int.from_bytes(b[32 * i:32 * (i + 1)], "little") for i in range(n)]

    def g1_msm_batch(self, fixed_scalars, var_points, var_scalars):
        n = len(fixed_scalars)
        nf = len(fixed_scalars[0]) if n else 0
        nv = len(var_points[0]) if (n and var_points) else 0
        fs = _bytes_arr(b"".join(self._fr(s) for row in fixed_scalars for s in row))
        vpb = _bytes_arr(b"".join(self._g1(p) for row in (var_points or []) for p in row))
        vs = _bytes_arr(b"".join(self._fr(s) for row in (var_scalars or []) for s in row))
        out = np.zeros(max(n, 1) * 2 * self.fpb, dtype=np.uint8)
        st = np.zeros(max(n, 1), dtype=np.int8)
        self._chk(self.lib.bbs_g1_msm_batch(self.h, n, _u8(fs), nf, _u8(vpb), _u8(vs), nv, _u8(out),
                                            st.ctypes.data_as(_lib.c_i8p)), "bbs_g1_msm_batch")
        b = out.tobytes()
        return [self._g1_dec(b[i * 2 * self.fpb:(i + 1) * 2 * self.fpb]) if st[i] == 1 else None for i in range(n)], st[:n]

    def g1_msm_pippenger(self, points, scalars):
        """sum_i scalars[i] * points[i] by the device's bucket method; returns (affine point or None, status)."""
        n = len(points)
        pb = _bytes_arr(b"".join(self._g1(p) for p in points))
        sb = _bytes_arr(b"".join(self._fr(s) for s in scalars))
        out = np.zeros(2 * self.fpb, dtype=np.uint8)
        inf = ctypes.c_int(0)
        st = np.zeros(max(n, 1), dtype=np.int8)
        self._chk(self.lib.bbs_g1_msm_pippenger(self.h, n, _u8(pb), _u8(sb), _u8(out), ctypes.byref(inf),
                                                st.ctypes.data_as(_lib.c_i8p)), "bbs_g1_msm_pippenger")
        return (None if inf.value else self._g1_dec(out.tobytes())), st[:n]

    def g1_decompress_batch(self, compressed: Sequence[bytes]):
        """bbs_g1_decompress_batch -> (list of affine points / None for the identity or a rejected encoding, codes)."""
        n = len(compressed)
        buf = _bytes_arr(b"".join(compressed))
        out = np.zeros(max(n, 1) * 2 * self.fpb, dtype=np.uint8)
        code = np.zeros(max(n, 1), dtype=np.int8)
        self._chk(self.lib.bbs_g1_decompress_batch(self.h, n, _u8(buf), _u8(out), code.ctypes.data_as(_lib.c_i8p)), "bbs_g1_decompress_batch")
        b = out.tobytes()
        return [self._g1_dec(b[i * 2 * self.fpb:(i + 1) * 2 * self.fpb]) if code[i] == 0 else None for i in range(n)], code[:n]

    def signatures_from_octets_batch(self, octets: Sequence[bytes]):
        """bbs_signatures_from_octets_batch -> (list of Signature or None, statuses)."""
        n = len(octets)
        buf = _bytes_arr(b"".join(octets))
        rec = 2 * self.fpb + 32
        out = np.zeros(max(n, 1) * rec, dtype=np.uint8)
        st = np.zeros(max(n, 1), dtype=np.int8)
        self._chk(self.lib.bbs_signatures_from_octets_batch(self.h, n, _u8(buf), _u8(out), st.ctypes.data_as(_lib.c_i8p)),
                  "bbs_signatures_from_octets_batch")
        return self._dec_sigs(out, st, n), st[:n]

    def proofs_to_octets_batch(self, proofs: Sequence[Proof]) -> List[bytes]:
        """bbs_proofs_to_octets_batch: the octet strings of n proofs (host, one call)."""
        n = len(proofs)
        rec = b"".join(self._g1(p.a_bar) + self._g1(p.b_bar) + self._g1(p.d) + self._fr(p.e_cap) + self._fr(p.r1_cap)
                       + self._fr(p.r3_cap) + self._fr(p.challenge) for p in proofs)
        pf = _bytes_arr(rec)
        cm, cmo = self._scalars([p.commitments for p in proofs])
        total = sum(3 * self.fpb + 32 * (4 + len(p.commitments)) for p in proofs)
        out = np.zeros(max(total, 1), dtype=np.uint8)
        oo = np.zeros(n + 1, dtype=np.uint64)
        st = np.zeros(max(n, 1), dtype=np.int8)
        self._chk(self.lib.bbs_proofs_to_octets_batch(self.curve, n, _u8(pf), _u8(cm), _u64(cmo), _u8(out),
                                                      _u64(oo), st.ctypes.data_as(_lib.c_i8p)), "bbs_proofs_to_octets_batch")
        b = out.tobytes()
        return [b[int(oo[i]):int(oo[i + 1])] if st[i] == 1 else None for i in range(n)]

    def proofs_from_octets_batch(self, octets: Sequence[bytes]):
        """bbs_proofs_from_octets_batch: n proof octet strings -> (list of Proof or None, int8 statuses); the point
        decompression and subgroup checks run on the device."""
        n = len(octets)
        flat, off = _ragged_bytes(octets)
        rec = 6 * self.fpb + 128
        pf = np.zeros(max(n, 1) * rec, dtype=np.uint8)
        cap = max(sum(len(o) for o in octets) // 32, 1)
        cm = np.zeros(cap * 32, dtype=np.uint8)
        cmo = np.zeros(n + 1, dtype=np.uint64)
        st = np.zeros(max(n, 1), dtype=np.int8)
        self._chk(self.lib.bbs_proofs_from_octets_batch(self.h, n, _u8(flat), _u64(off), _u8(pf), _u8(cm), _u64(cmo),
                                                        st.ctypes.data_as(_lib.c_i8p)), "bbs_proofs_from_octets_batch")
        proofs = self._dec_proofs(pf, cm, cmo, st, n)
        return proofs, st[:n]

    def pairing_product2_is_one_batch(self, pa, pb) -> np.ndarray:
        n = len(pa)
        a = _bytes_arr(b"".join(self._g1(p) for p in pa))
        b = _bytes_arr(b"".join(self._g1(p) for p in pb))
        st = np.zeros(max(n, 1), dtype=np.int8)
        self._chk(self.lib.bbs_pairing_product2_is_one_batch(self.h, n, _u8(a), _u8(b), st.ctypes.data_as(_lib.c_i8p)),
                  "bbs_pairing_product2_is_one_batch")
        return st[:n]


class Job:
    """A device-resident batch (bbs_job): run() is asynchronous on the context's stream."""

    def __init__(self, eng: Engine, handle, n: int):
        self.eng, self.h, self.n = eng, handle, n
        self.total_msgs = 0
        self.result = None          # submit form: the statuses, valid after wait()
        self._decode = None         # submit form of sign / proof_gen: decodes the delivered records
        self._fixup = None          # applied to `result` once the statuses have been delivered
        self._waited = False

    def run(self):
        Engine._chk(self.eng.lib.bbs_job_run(self.h), "bbs_job_run")

    def wait(self):
        Engine._chk(self.eng.lib.bbs_job_wait(self.h), "bbs_job_wait")
        self._waited = True
        if self._fixup is not None:
            self._fixup()

    def done(self) -> bool:
        """bbs_job_poll: everything enqueued for the job has finished (wait() will not block)."""
        return self.eng.lib.bbs_job_poll(self.h) == 1

    @staticmethod
    def wait_any(jobs) -> int:
        """bbs_jobs_wait_any: sleeps until one of `jobs` (entries may be None) has finished, delivers it as wait() does and
        returns its position -- the job that finished FIRST, whatever the order of submission."""
        live = [(k, j) for k, j in enumerate(jobs) if j is not None and j.h]
        if not live:
            raise ValueError("wait_any: no job")
        arr = (ctypes.c_void_p * len(live))(*[j.h for _, j in live])
        idx = ctypes.c_size_t(0)
        rc = live[0][1].eng.lib.bbs_jobs_wait_any(arr, len(live), ctypes.byref(idx))
        Engine._chk(rc, "bbs_jobs_wait_any")
        k, j = live[idx.value]
        j._waited = True
        if j._fixup is not None:
            j._fixup()
        return k

    def output(self):
        """Submit form of sign / proof_gen, after wait(): (signatures | proofs with None for failed items, statuses)."""
        if self._decode is None or not self._waited:
            raise RuntimeError("output(): a sign / proof_gen submit job that has been waited for")
        return self._decode()

    def device_bytes(self) -> int:
        return int(self.eng.lib.bbs_job_device_bytes(self.h))

    def status(self) -> np.ndarray:
        st = np.zeros(max(self.n, 1), dtype=np.int8)
        Engine._chk(self.eng.lib.bbs_job_fetch_status(self.h, st.ctypes.data_as(_lib.c_i8p)), "bbs_job_fetch_status")
        return st[:self.n]

    def signatures(self):
        st = self.status()
        out = np.zeros(max(self.n, 1) * (2 * self.eng.fpb + 32), dtype=np.uint8)
        Engine._chk(self.eng.lib.bbs_job_fetch_signatures(self.h, _u8(out)), "bbs_job_fetch_signatures")
        return self.eng._dec_sigs(out, st, self.n), st

    def proofs(self):
        st = self.status()
        pf = np.zeros(max(self.n, 1) * (6 * self.eng.fpb + 128), dtype=np.uint8)
        cm = np.zeros(max(self.total_msgs, 1) * 32, dtype=np.uint8)
        cmo = np.zeros(self.n + 1, dtype=np.uint64)
        Engine._chk(self.eng.lib.bbs_job_fetch_proofs(self.h, _u8(pf), _u8(cm), _u64(cmo)), "bbs_job_fetch_proofs")
        return self.eng._dec_proofs(pf, cm, cmo, st, self.n), st

    def run_timed(self, reps: int = 1, per_stage: bool = True):
        """-> (total_ms, {stage: ms}) measured with HIP events on the context's own stream."""
        tot = ctypes.c_float(0)
        ks = (ctypes.c_float * 16)()
        ns = ctypes.c_int(0)
        Engine._chk(self.eng.lib.bbs_job_run_timed(self.h, reps, ctypes.byref(tot), ks if per_stage else None, 16,
                                                   ctypes.byref(ns)), "bbs_job_run_timed")
        names = [self.eng.lib.bbs_job_stage_name(self.h, k).decode() for k in range(ns.value)]
        return tot.value, ({names[k]: ks[k] for k in range(ns.value)} if per_stage else {})

    def stage_times(self):
        """-> (total_ms, {stage: ms}) of the last run of a job created under Engine.set_stage_timing(True); after wait()."""
        tot = ctypes.c_float(0)
        ks = (ctypes.c_float * 16)()
        ns = ctypes.c_int(0)
        Engine._chk(self.eng.lib.bbs_job_stage_times(self.h, ctypes.byref(tot), ks, 16, ctypes.byref(ns)), "bbs_job_stage_times")
        names = [self.eng.lib.bbs_job_stage_name(self.h, k).decode() for k in range(ns.value)]
        return tot.value, {names[k]: ks[k] for k in range(ns.value)}

    @staticmethod
    def run_many_timed(jobs, steps: int):
        """Step k runs on jobs[k % len(jobs)], all batches in flight -> (total_ms, {stage: summed ms})."""
        eng = jobs[0].eng
        arr = (ctypes.c_void_p * len(jobs))(*[j.h for j in jobs])
        tot = ctypes.c_float(0)
        ks = (ctypes.c_float * 16)()
        ns = ctypes.c_int(0)
        Engine._chk(eng.lib.bbs_jobs_run_timed(arr, len(jobs), steps, ctypes.byref(tot), ks, 16, ctypes.byref(ns)),
                    "bbs_jobs_run_timed")
        names = [eng.lib.bbs_job_stage_name(jobs[0].h, k).decode() for k in range(ns.value)]
        return tot.value, {names[k]: ks[k] for k in range(ns.value)}

    def free(self):
        if self.h:
            self.eng.lib.bbs_job_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Issuer:
    """bbs_issuer: the reference's four PUBLIC functions over batches whose items differ in their number of messages
    (generators chosen by the item's own length: src/sign.rs:44-49, verify.rs:30-35, proof_gen.rs:91-96,
    proof_verify.rs:40-43).  Octet strings and raw messages in and out; statuses as numpy int8."""

    def __init__(self, curve, api_id: bytes, device: int = 0, lib_path: Optional[str] = None, max_messages: Optional[int] = None,
                 window_bits: Optional[int] = None):
        self.lib = _lib.load_library(lib_path)
        self.curve = CURVE_IDS[curve] if isinstance(curve, str) else int(curve)
        self.fpb = int(self.lib.bbs_fp_bytes(self.curve))
        h = ctypes.c_void_p()
        aid = _bytes_arr(api_id)
        Engine._chk(self.lib.bbs_issuer_create(self.curve, device, _u8(aid), len(api_id), ctypes.byref(h)), "bbs_issuer_create")
        self.h = h
        if max_messages is not None or window_bits is not None:
            Engine._chk(self.lib.bbs_issuer_set_limits(self.h, 1024 if max_messages is None else max_messages, window_bits or 0),
                        "bbs_issuer_set_limits")

    def close(self):
        if getattr(self, "h", None):
            self.lib.bbs_issuer_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _fp(self, v: int) -> bytes:
        return int(v).to_bytes(self.fpb, "little")

    def set_public_key(self, pk):
        if pk is None:
            Engine._chk(self.lib.bbs_issuer_set_public_key(self.h, None, 1), "bbs_issuer_set_public_key")
            return
        (x0, x1), (y0, y1) = pk
        buf = _bytes_arr(self._fp(x0) + self._fp(x1) + self._fp(y0) + self._fp(y1))
        Engine._chk(self.lib.bbs_issuer_set_public_key(self.h, _u8(buf), 0), "bbs_issuer_set_public_key")

    def set_secret_key(self, sk: int):
        buf = _bytes_arr(int(sk).to_bytes(32, "little"))
        Engine._chk(self.lib.bbs_issuer_set_secret_key(self.h, _u8(buf)), "bbs_issuer_set_secret_key")

    def set_modes(self, latency_mode=2, batch_verification=False, points_in_subgroup=False):
        Engine._chk(self.lib.bbs_issuer_set_modes(self.h, 2 if latency_mode in ("auto", 2) else (1 if latency_mode else 0),
                                                  1 if batch_verification else 0, 1 if points_in_subgroup else 0), "bbs_issuer_set_modes")

    def context_count(self) -> int:
        return int(self.lib.bbs_issuer_context_count(self.h))

    def set_budget(self, max_contexts: int = 64, max_table_bytes: int = 0):
        """bbs_issuer_set_budget: resident contexts / table bytes (0 = half of the free device memory); idle ones beyond leave at once."""
        Engine._chk(self.lib.bbs_issuer_set_budget(self.h, max_contexts, max_table_bytes), "bbs_issuer_set_budget")

    def table_bytes(self) -> int:
        return int(self.lib.bbs_issuer_table_bytes(self.h))

    def warm(self, message_count: int):
        c = ctypes.c_void_p()
        Engine._chk(self.lib.bbs_issuer_context(self.h, message_count, ctypes.byref(c)), "bbs_issuer_context")

    def _sig_octets(self, octets):
        want = self.fpb + 32
        bad = [i for i, o in enumerate(octets) if len(o) != want]
        flat = b"".join(o if len(o) == want else bytes(want) for o in octets)
        return (_bytes_arr(flat) if octets else np.zeros(1, dtype=np.uint8)), bad

    def pack_proof_verify(self, proof_octets, disclosed_raw, disclosed_idx, headers=None, phs=None):
        """The flat host buffers of one bbs_issuer_proof_verify call: (n, arrays to keep alive, ctypes arguments)."""
        n = len(proof_octets)
        ob, oo = _ragged_bytes(proof_octets)
        mb, mbo, mio = Engine._raw_msgs(disclosed_raw)
        di, dio = Engine._indexes(disclosed_idx)
        hb, ho = _ragged_bytes(headers if headers is not None else [b""] * n)
        pb, po = _ragged_bytes(phs if phs is not None else [b""] * n)
        keep = (ob, oo, mb, mbo, mio, di, dio, hb, ho, pb, po)
        return n, keep, (_u8(ob), _u64(oo), _u8(mb), _u64(mbo), _u64(mio), _u64(di), _u64(dio), _u8(hb), _u64(ho), _u8(pb), _u64(po))

    def proof_verify_packed(self, n, args) -> np.ndarray:
        st = np.full(max(n, 1), -128, dtype=np.int8)
        Engine._chk(self.lib.bbs_issuer_proof_verify(self.h, n, *args, st.ctypes.data_as(_lib.c_i8p)), "bbs_issuer_proof_verify")
        return st[:n]

    def proof_verify(self, proof_octets, disclosed_raw, disclosed_idx, headers=None, phs=None) -> np.ndarray:
        n, keep, args = self.pack_proof_verify(proof_octets, disclosed_raw, disclosed_idx, headers, phs)
        return self.proof_verify_packed(n, args)

    def proof_verify_submit_packed(self, n, args) -> "IssuerJob":
        """bbs_issuer_proof_verify_submit: returns at once; ``job.wait()`` delivers ``job.result`` (the statuses)."""
        st = np.full(max(n, 1), -128, dtype=np.int8)
        j = ctypes.c_void_p()
        Engine._chk(self.lib.bbs_issuer_proof_verify_submit(self.h, n, *args, st.ctypes.data_as(_lib.c_i8p), ctypes.byref(j)),
                    "bbs_issuer_proof_verify_submit")
        return IssuerJob(self.lib, j, st[:n])

    def verify(self, sig_octets, messages_raw, headers=None) -> np.ndarray:
        n = len(sig_octets)
        ob, bad = self._sig_octets(sig_octets)
        mb, mbo, mio = Engine._raw_msgs(messages_raw)
        hb, ho = _ragged_bytes(headers if headers is not None else [b""] * n)
        st = np.full(max(n, 1), -128, dtype=np.int8)
        Engine._chk(self.lib.bbs_issuer_verify(self.h, n, _u8(ob), _u8(mb), _u64(mbo), _u64(mio), _u8(hb), _u64(ho),
                                               st.ctypes.data_as(_lib.c_i8p)), "bbs_issuer_verify")
        for i in bad:
            st[i] = -42
        return st[:n]

    def sign(self, messages_raw, headers=None):
        """-> (signature octet strings, b"" where status != 1; statuses)"""
        n = len(messages_raw)
        mb, mbo, mio = Engine._raw_msgs(messages_raw)
        hb, ho = _ragged_bytes(headers if headers is not None else [b""] * n)
        st = np.full(max(n, 1), -128, dtype=np.int8)
        so = self.fpb + 32
        out = np.zeros(max(n, 1) * so, dtype=np.uint8)
        Engine._chk(self.lib.bbs_issuer_sign(self.h, n, _u8(mb), _u64(mbo), _u64(mio), _u8(hb), _u64(ho), _u8(out),
                                             st.ctypes.data_as(_lib.c_i8p)), "bbs_issuer_sign")
        return [bytes(out[i * so:(i + 1) * so]) if st[i] == 1 else b"" for i in range(n)], st[:n]

    def proof_gen(self, sig_octets, messages_raw, disclosed_idx, random_scalars, headers=None, phs=None):
        """-> (proof octet strings, b"" where status != 1; statuses).  random_scalars[i]: 5 + U_i integers."""
        n = len(sig_octets)
        ob, bad = self._sig_octets(sig_octets)
        mb, mbo, mio = Engine._raw_msgs(messages_raw)
        di, dio = Engine._indexes(disclosed_idx)
        ro = np.zeros(n + 1, dtype=np.uint64)
        chunks = []
        for i, row in enumerate(random_scalars):
            ro[i + 1] = ro[i] + len(row)
            chunks.extend(int(s).to_bytes(32, "little") for s in row)
        rs = _bytes_arr(b"".join(chunks))
        hb, ho = _ragged_bytes(headers if headers is not None else [b""] * n)
        pb, po = _ragged_bytes(phs if phs is not None else [b""] * n)
        st = np.full(max(n, 1), -128, dtype=np.int8)
        cap = sum(3 * self.fpb + 32 * (4 + len(m)) for m in messages_raw) + 8
        out = np.zeros(cap, dtype=np.uint8)
        off = np.zeros(n + 1, dtype=np.uint64)
        Engine._chk(self.lib.bbs_issuer_proof_gen(self.h, n, _u8(ob), _u8(mb), _u64(mbo), _u64(mio), _u64(di), _u64(dio), _u8(rs), _u64(ro),
                                                  _u8(hb), _u64(ho), _u8(pb), _u64(po), _u8(out), _u64(off), st.ctypes.data_as(_lib.c_i8p)),
                    "bbs_issuer_proof_gen")
        for i in bad:
            st[i] = -42
        return [bytes(out[int(off[i]):int(off[i + 1])]) if st[i] == 1 else b"" for i in range(n)], st[:n]


class IssuerJob:
    """A routed issuer call in flight (bbs_issuer_job)."""

    def __init__(self, lib, handle, result):
        self.lib, self.h, self.result = lib, handle, result

    def wait(self):
        Engine._chk(self.lib.bbs_issuer_job_wait(self.h), "bbs_issuer_job_wait")

    def free(self):
        if self.h:
            self.lib.bbs_issuer_job_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass
