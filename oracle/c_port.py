"""ctypes access to the plain-C restatement (oracle/c/bbs_oracle.c).  ORACLE = test infrastructure.

build() compiles it with gcc into oracle/c/libbbs_oracle.so (git-ignored).  BLS12-381 only; used as
the CPU baseline of bench.py and to check whole GPU batches item by item."""
import ctypes
import os
import subprocess

from . import bbs as pyo

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "c", "bbs_oracle.c")
LIB = os.path.join(HERE, "c", "libbbs_oracle.so")
_lib = None
u8p = ctypes.POINTER(ctypes.c_uint8)
u64p = ctypes.POINTER(ctypes.c_uint64)


def build(force=False):
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        # no -march flags: the .so built here travels to the GPU box, whose host CPU may differ
        subprocess.run(["gcc", "-O3", "-shared", "-fPIC", "-o", LIB, SRC], check=True)
    return LIB


def port(curve):
    """The C port of one curve as an object with sk_to_pk / core_sign / core_verify / core_proof_gen / core_proof_verify."""
    import sys
    if curve == "bls12_381":
        return sys.modules[__name__]
    raise NotImplementedError("oracle/c restates BLS12-381 only")


def lib():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
    return _lib


def _b(data: bytes):
    return (ctypes.c_uint8 * max(len(data), 1)).from_buffer_copy(bytes(data) + (b"" if data else b"\0"))


def _fp(v):
    return int(v).to_bytes(48, "little")


def _fr(v):
    return int(v).to_bytes(32, "little")


def _g1(p):
    return bytes(96) if p is None else _fp(p[0]) + _fp(p[1])


def _g1_dec(b):
    x, y = int.from_bytes(b[:48], "little"), int.from_bytes(b[48:96], "little")
    return None if x == 0 and y == 0 else (x, y)


def _pk(pk):
    if pk is None:
        return bytes(192), 1
    (x0, x1), (y0, y1) = pk
    return _fp(x0) + _fp(x1) + _fp(y0) + _fp(y1), 0


def sk_to_pk(sk):
    out = (ctypes.c_uint8 * 192)()
    lib().orc_sk_to_pk(_b(_fr(sk)), out)
    f = [int.from_bytes(bytes(out)[48 * i:48 * (i + 1)], "little") for i in range(4)]
    return ((f[0], f[1]), (f[2], f[3]))


def core_sign(sk, generators, header, messages, api_id):
    L = len(messages)
    out = (ctypes.c_uint8 * 128)()
    rc = lib().orc_core_sign(_b(_fr(sk)), L, _b(b"".join(_g1(g) for g in generators)), _b(api_id), ctypes.c_size_t(len(api_id)),
                             _b(header), ctypes.c_size_t(len(header)), _b(b"".join(_fr(m) for m in messages)), out)
    if rc != 1:
        raise pyo.BbsPanic("sk + e == 0")
    o = bytes(out)
    return pyo.Signature(_g1_dec(o[:96]), int.from_bytes(o[96:], "little"))


def core_verify(pk, signature, generators, header, messages, api_id):
    pkb, inf = _pk(pk)
    sig = _g1(signature.a) + _fr(signature.e)
    return bool(lib().orc_core_verify(_b(pkb), inf, len(messages), _b(b"".join(_g1(g) for g in generators)), _b(api_id),
                                      ctypes.c_size_t(len(api_id)), _b(header), ctypes.c_size_t(len(header)),
                                      _b(b"".join(_fr(m) for m in messages)), _b(sig)))


def core_proof_gen(pk, signature, header, generators, ph, messages, disclosed_sorted, api_id, random_scalars):
    pkb, inf = _pk(pk)
    L, R = len(messages), len(disclosed_sorted)
    sig = _g1(signature.a) + _fr(signature.e)
    idx = (ctypes.c_uint64 * max(R, 1))(*disclosed_sorted)
    pf = (ctypes.c_uint8 * 416)()
    cm = (ctypes.c_uint8 * max(32 * (L - R), 1))()
    rc = lib().orc_core_proof_gen(_b(pkb), inf, L, _b(b"".join(_g1(g) for g in generators)), _b(api_id), ctypes.c_size_t(len(api_id)),
                                  _b(header), ctypes.c_size_t(len(header)), _b(ph), ctypes.c_size_t(len(ph)),
                                  _b(b"".join(_fr(m) for m in messages)), _b(sig), idx, ctypes.c_size_t(R),
                                  _b(b"".join(_fr(s) for s in random_scalars)), pf, cm)
    if rc != 1:
        raise pyo.BbsPanic("r2 == 0")
    o, c = bytes(pf), bytes(cm)
    sc = [int.from_bytes(o[288 + 32 * k:320 + 32 * k], "little") for k in range(4)]
    return pyo.Proof(_g1_dec(o[:96]), _g1_dec(o[96:192]), _g1_dec(o[192:288]), sc[0], sc[1], sc[2],
                     [int.from_bytes(c[32 * k:32 * k + 32], "little") for k in range(L - R)], sc[3])


def proof_fixed_bytes(proof):
    return (_g1(proof.a_bar) + _g1(proof.b_bar) + _g1(proof.d) + _fr(proof.e_cap) + _fr(proof.r1_cap) + _fr(proof.r3_cap)
            + _fr(proof.challenge))


def core_proof_verify(pk, proof, generators, header, ph, disclosed_messages, disclosed_indexes, api_id):
    """Valid inputs only (distinct in-range indexes, matching lengths): error paths are the Python oracle's."""
    pkb, inf = _pk(pk)
    R = len(disclosed_indexes)
    L = R + len(proof.commitments)
    idx = (ctypes.c_uint64 * max(R, 1))(*disclosed_indexes)
    return bool(lib().orc_core_proof_verify(_b(pkb), inf, L, _b(b"".join(_g1(g) for g in generators)), _b(api_id), ctypes.c_size_t(len(api_id)),
                                            _b(header), ctypes.c_size_t(len(header)), _b(ph), ctypes.c_size_t(len(ph)),
                                            _b(proof_fixed_bytes(proof)), _b(b"".join(_fr(c) for c in proof.commitments)),
                                            _b(b"".join(_fr(m) for m in disclosed_messages)), idx, ctypes.c_size_t(R)))
