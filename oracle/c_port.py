"""ctypes access to the plain-C restatement (oracle/c/bbs_oracle.c).  ORACLE = test infrastructure.

build() compiles it with gcc, once per curve, into oracle/c/libbbs_oracle_<curve>.so (git-ignored; no -march flags:
the .so built in the build container travels to the GPU box).  Used as the CPU baseline of bench.py and to check whole
GPU batches item by item.  `port(curve)` gives the C port of one curve; the module-level functions are the BLS12-381
port (kept for the tests that predate the BN254 build)."""
import ctypes
import os
import subprocess

from . import bbs as pyo

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "c", "bbs_oracle.c")
LIBS = {"bls12_381": os.path.join(HERE, "c", "libbbs_oracle_bls12_381.so"), "bn254": os.path.join(HERE, "c", "libbbs_oracle_bn254.so")}
FLAGS = {"bls12_381": [], "bn254": ["-DORC_BN254"]}
u8p = ctypes.POINTER(ctypes.c_uint8)
u64p = ctypes.POINTER(ctypes.c_uint64)


def build(force=False, curve=None):
    """Compile the port(s); returns the path of the BLS12-381 library (or of `curve`)."""
    for c in ([curve] if curve else list(LIBS)):
        lib = LIBS[c]
        if force or not os.path.exists(lib) or os.path.getmtime(lib) < os.path.getmtime(SRC):
            subprocess.run(["gcc", "-O3", "-shared", "-fPIC"] + FLAGS[c] + ["-o", lib, SRC], check=True)
    return LIBS[curve or "bls12_381"]


def _b(data: bytes):
    return (ctypes.c_uint8 * max(len(data), 1)).from_buffer_copy(bytes(data) + (b"" if data else b"\0"))


def _fr(v):
    return int(v).to_bytes(32, "little")


class CPort:
    """The C port of one curve: sk_to_pk / core_sign / core_verify / core_proof_gen / core_proof_verify with the
    signatures of the Python oracle (oracle/bbs.py), values as Python integers / tuples."""

    def __init__(self, curve):
        self.curve = curve
        self.lib = ctypes.CDLL(build(curve=curve))
        self.fpb = int(self.lib.orc_fp_bytes())
        assert self.fpb == (48 if curve == "bls12_381" else 32)

    def _fp(self, v):
        return int(v).to_bytes(self.fpb, "little")

    def _g1(self, p):
        return bytes(2 * self.fpb) if p is None else self._fp(p[0]) + self._fp(p[1])

    def _g1_dec(self, b):
        x, y = int.from_bytes(b[:self.fpb], "little"), int.from_bytes(b[self.fpb:2 * self.fpb], "little")
        return None if x == 0 and y == 0 else (x, y)

    def _pk(self, pk):
        if pk is None:
            return bytes(4 * self.fpb), 1
        (x0, x1), (y0, y1) = pk
        return self._fp(x0) + self._fp(x1) + self._fp(y0) + self._fp(y1), 0

    def _gens(self, generators):
        return _b(b"".join(self._g1(g) for g in generators))

    def sk_to_pk(self, sk):
        out = (ctypes.c_uint8 * (4 * self.fpb))()
        self.lib.orc_sk_to_pk(_b(_fr(sk)), out)
        f = [int.from_bytes(bytes(out)[self.fpb * i:self.fpb * (i + 1)], "little") for i in range(4)]
        return ((f[0], f[1]), (f[2], f[3]))

    def g1_msm_plain(self, points, scalars):
        """sum_i scalars[i] * points[i] by independent double-and-add multiplications (None = identity)."""
        assert len(points) == len(scalars)
        out = (ctypes.c_uint8 * (2 * self.fpb))()
        self.lib.orc_g1_msm_plain(ctypes.c_size_t(len(points)), _b(b"".join(self._g1(p) for p in points)),
                                  _b(b"".join(_fr(k) for k in scalars)), out)
        return self._g1_dec(bytes(out))

    def bn_hash_to_g1(self, msg: bytes, dst: bytes):
        """BN254 build only: RFC 9380 hash_to_curve with the SvdW map -> (point, (mask0, mask1)); mask bit k set = candidate
        x_(k+1) of that map was on the curve (bits 0 and 1 both set = the case that sees the sign of the constant c3)."""
        assert self.curve == "bn254"
        out = (ctypes.c_uint8 * (2 * self.fpb))()
        sq = (ctypes.c_int * 2)()
        self.lib.orc_bn_hash_to_g1(_b(msg), ctypes.c_size_t(len(msg)), _b(dst), ctypes.c_size_t(len(dst)), out, sq)
        return self._g1_dec(bytes(out)), (int(sq[0]), int(sq[1]))

    def core_sign(self, sk, generators, header, messages, api_id):
        L = len(messages)
        out = (ctypes.c_uint8 * (2 * self.fpb + 32))()
        rc = self.lib.orc_core_sign(_b(_fr(sk)), L, self._gens(generators), _b(api_id), ctypes.c_size_t(len(api_id)),
                                    _b(header), ctypes.c_size_t(len(header)), _b(b"".join(_fr(m) for m in messages)), out)
        if rc != 1:
            raise pyo.BbsPanic("sk + e == 0")
        o = bytes(out)
        return pyo.Signature(self._g1_dec(o), int.from_bytes(o[2 * self.fpb:], "little"))

    def core_verify(self, pk, signature, generators, header, messages, api_id):
        pkb, inf = self._pk(pk)
        sig = self._g1(signature.a) + _fr(signature.e)
        return bool(self.lib.orc_core_verify(_b(pkb), inf, len(messages), self._gens(generators), _b(api_id),
                                             ctypes.c_size_t(len(api_id)), _b(header), ctypes.c_size_t(len(header)),
                                             _b(b"".join(_fr(m) for m in messages)), _b(sig)))

    def core_proof_gen(self, pk, signature, header, generators, ph, messages, disclosed_sorted, api_id, random_scalars):
        pkb, inf = self._pk(pk)
        L, R = len(messages), len(disclosed_sorted)
        f = self.fpb
        sig = self._g1(signature.a) + _fr(signature.e)
        idx = (ctypes.c_uint64 * max(R, 1))(*disclosed_sorted)
        pf = (ctypes.c_uint8 * (6 * f + 128))()
        cm = (ctypes.c_uint8 * max(32 * (L - R), 1))()
        rc = self.lib.orc_core_proof_gen(_b(pkb), inf, L, self._gens(generators), _b(api_id), ctypes.c_size_t(len(api_id)),
                                         _b(header), ctypes.c_size_t(len(header)), _b(ph), ctypes.c_size_t(len(ph)),
                                         _b(b"".join(_fr(m) for m in messages)), _b(sig), idx, ctypes.c_size_t(R),
                                         _b(b"".join(_fr(s) for s in random_scalars)), pf, cm)
        if rc != 1:
            raise pyo.BbsPanic("r2 == 0")
        o, c = bytes(pf), bytes(cm)
        sc = [int.from_bytes(o[6 * f + 32 * k:6 * f + 32 * k + 32], "little") for k in range(4)]
        return pyo.Proof(self._g1_dec(o[:2 * f]), self._g1_dec(o[2 * f:4 * f]), self._g1_dec(o[4 * f:6 * f]), sc[0], sc[1], sc[2],
                         [int.from_bytes(c[32 * k:32 * k + 32], "little") for k in range(L - R)], sc[3])

    def proof_fixed_bytes(self, proof):
        return (self._g1(proof.a_bar) + self._g1(proof.b_bar) + self._g1(proof.d) + _fr(proof.e_cap) + _fr(proof.r1_cap)
                + _fr(proof.r3_cap) + _fr(proof.challenge))

    def core_proof_verify(self, pk, proof, generators, header, ph, disclosed_messages, disclosed_indexes, api_id):
        """Valid inputs only (distinct in-range indexes, matching lengths): error paths are the Python oracle's."""
        pkb, inf = self._pk(pk)
        R = len(disclosed_indexes)
        L = R + len(proof.commitments)
        idx = (ctypes.c_uint64 * max(R, 1))(*disclosed_indexes)
        return bool(self.lib.orc_core_proof_verify(_b(pkb), inf, L, self._gens(generators), _b(api_id), ctypes.c_size_t(len(api_id)),
                                                   _b(header), ctypes.c_size_t(len(header)), _b(ph), ctypes.c_size_t(len(ph)),
                                                   _b(self.proof_fixed_bytes(proof)), _b(b"".join(_fr(c) for c in proof.commitments)),
                                                   _b(b"".join(_fr(m) for m in disclosed_messages)), idx, ctypes.c_size_t(R)))


_ports = {}


def port(curve) -> CPort:
    if curve not in LIBS:
        raise NotImplementedError("oracle/c restates bls12_381 and bn254 only")
    if curve not in _ports:
        _ports[curve] = CPort(curve)
    return _ports[curve]


# ---- the BLS12-381 port as module-level functions -------------------------------------------------------------------
def sk_to_pk(sk):
    return port("bls12_381").sk_to_pk(sk)


def core_sign(sk, generators, header, messages, api_id):
    return port("bls12_381").core_sign(sk, generators, header, messages, api_id)


def core_verify(pk, signature, generators, header, messages, api_id):
    return port("bls12_381").core_verify(pk, signature, generators, header, messages, api_id)


def core_proof_gen(pk, signature, header, generators, ph, messages, disclosed_sorted, api_id, random_scalars):
    return port("bls12_381").core_proof_gen(pk, signature, header, generators, ph, messages, disclosed_sorted, api_id, random_scalars)


def proof_fixed_bytes(proof):
    return port("bls12_381").proof_fixed_bytes(proof)


def core_proof_verify(pk, proof, generators, header, ph, disclosed_messages, disclosed_indexes, api_id):
    return port("bls12_381").core_proof_verify(pk, proof, generators, header, ph, disclosed_messages, disclosed_indexes, api_id)
