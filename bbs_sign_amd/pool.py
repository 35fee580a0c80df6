"""bbs_pool: core_proof_verify over a list of proofs fanned out over several GPUs BEHIND the C ABI (include/bbs_sign_amd.h,
csrc/pool.hpp).  One process, one submitting thread per device inside the library, statuses
written straight into the caller's array -- the path a Rust / C host gets by linking the library.  The other multi-GPU
path of this package, `mixed.py` (one process per GPU, torch.distributed, one all_gather of statuses), partitions by the
same rule (`sharding.shard_plan`: by curve, then contiguous ceil(n / devices) items per device).

The reference verifies one proof per call (src/proof_verify.rs:19-61); a caller with a list loops over it.
"""
import ctypes
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import _lib
from .engine import CURVE_IDS, BbsRuntimeError, Engine, _bytes_arr, _u8, _u64


class _Packer(Engine):
    """Engine's encoders for a curve, without a context of its own."""

    def __init__(self, lib, curve):       # noqa: D401 (no super().__init__: nothing is created on the device)
        self.lib = lib
        self.curve = CURVE_IDS[curve] if isinstance(curve, str) else int(curve)
        self.fpb = int(lib.bbs_fp_bytes(self.curve))
        self.h = None
        self.L = None


class PackedSection:
    """The items of one curve of a list, packed for struct bbs_pv_list (host buffers, kept alive with the object)."""

    def __init__(self, packer: _Packer, proofs, disclosed_msgs, disclosed_idx, headers=None, phs=None,
                 global_index: Optional[Sequence[int]] = None):
        self.curve = packer.curve
        self.n, self.keep, self.args = packer._pv_inputs(proofs, disclosed_msgs, disclosed_idx, headers, phs)
        self.global_index = None if global_index is None else np.asarray(list(global_index) + [0], dtype=np.uint64)

    def fill(self, rec: "_lib.PvList", status: np.ndarray):
        pf, cm, cmo, dm, dmo, di, dio, hb, ho, pb, po = self.args
        rec.curve, rec.n = self.curve, self.n
        rec.proofs_fixed, rec.commitments, rec.commit_off = pf, cm, cmo
        rec.disclosed_msgs, rec.dmsg_off, rec.disclosed_idx, rec.didx_off = dm, dmo, di, dio
        rec.headers, rec.hdr_off, rec.ph, rec.ph_off = hb, ho, pb, po
        rec.global_index = _u64(self.global_index) if self.global_index is not None else None
        rec.status = status.ctypes.data_as(_lib.c_i8p)


class PoolJob:
    """A list in flight on the pool (bbs_pool_job)."""

    def __init__(self, pool, h, recs, sections, outs, whole, n_total):
        self.pool, self.h, self.recs, self.sections, self.outs, self.whole, self.n_total = pool, h, recs, sections, outs, whole, n_total

    def wait(self) -> np.ndarray:
        rc = self.pool.lib.bbs_pool_job_wait(self.h)
        self.pool.lib.bbs_pool_job_free(self.h)
        self.h = None
        if rc:
            raise BbsRuntimeError(rc, "bbs_pool_job_wait")
        if self.whole is not None:
            return self.whole[:self.n_total]
        return np.concatenate([o[:s.n] for o, s in zip(self.outs, self.sections)]) if self.sections else np.zeros(0, dtype=np.int8)


class Pool:
    """One context per (curve, member device); `devices` may repeat an id (two context sets on one GPU)."""

    def __init__(self, devices: Sequence[int], lib_path: Optional[str] = None):
        self.lib = _lib.load_library(lib_path)
        ids = (ctypes.c_int * len(devices))(*[int(d) for d in devices])
        h = ctypes.c_void_p()
        rc = self.lib.bbs_pool_create(ids, len(devices), ctypes.byref(h))
        if rc:
            raise BbsRuntimeError(rc, "bbs_pool_create")
        self.h = h
        self.packers: Dict[int, _Packer] = {}

    def close(self):
        if getattr(self, "h", None):
            self.lib.bbs_pool_destroy(self.h)
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

    def packer(self, curve) -> _Packer:
        cid = CURVE_IDS[curve] if isinstance(curve, str) else int(curve)
        if cid not in self.packers:
            self.packers[cid] = _Packer(self.lib, cid)
        return self.packers[cid]

    def device_count(self) -> int:
        return int(self.lib.bbs_pool_device_count(self.h))

    def set_window_bits(self, curve, bits: int):
        self._chk(self.lib.bbs_pool_set_window_bits(self.h, self.packer(curve).curve, bits), "bbs_pool_set_window_bits")

    def set_generators(self, curve, generators: Sequence, api_id: bytes):
        pk = self.packer(curve)
        buf = _bytes_arr(b"".join(pk._g1(g) for g in generators))
        aid = _bytes_arr(api_id)
        self._chk(self.lib.bbs_pool_set_generators(self.h, pk.curve, _u8(buf), len(generators), _u8(aid), len(api_id)),
                  "bbs_pool_set_generators")

    def set_public_key(self, curve, pk_point):
        pk = self.packer(curve)
        if pk_point is None:
            self._chk(self.lib.bbs_pool_set_public_key(self.h, pk.curve, None, 1), "bbs_pool_set_public_key")
            return
        (x0, x1), (y0, y1) = pk_point
        buf = _bytes_arr(pk._fp(x0) + pk._fp(x1) + pk._fp(y0) + pk._fp(y1))
        self._chk(self.lib.bbs_pool_set_public_key(self.h, pk.curve, _u8(buf), 0), "bbs_pool_set_public_key")

    def set_inflight(self, jobs_per_member: int):
        self._chk(self.lib.bbs_pool_set_inflight(self.h, jobs_per_member), "bbs_pool_set_inflight")

    def context(self, curve, member: int) -> ctypes.c_void_p:
        h = ctypes.c_void_p()
        self._chk(self.lib.bbs_pool_context(self.h, self.packer(curve).curve, member, ctypes.byref(h)), "bbs_pool_context")
        return h

    def pack(self, curve, proofs, disclosed_msgs, disclosed_idx, headers=None, phs=None, global_index=None) -> PackedSection:
        return PackedSection(self.packer(curve), proofs, disclosed_msgs, disclosed_idx, headers, phs, global_index)

    def submit_packed(self, sections: List[PackedSection], n_total: Optional[int] = None, max_batch: int = 0) -> "PoolJob":
        """bbs_pool_proof_verify_submit on packed sections; ``job.wait()`` returns the statuses.  n_total given: every section
        carries a global index and ONE status array of the whole list comes back; else the sections' status arrays,
        concatenated in section order.  The sections must stay alive until wait() (the job keeps a reference)."""
        recs = (_lib.PvList * max(1, len(sections)))()
        if n_total is not None:
            whole = np.full(max(n_total, 1), -128, dtype=np.int8)
            outs = [whole] * len(sections)
        else:
            whole = None
            outs = [np.full(max(s.n, 1), -128, dtype=np.int8) for s in sections]
        for r, s, o in zip(recs, sections, outs):
            s.fill(r, o)
        h = ctypes.c_void_p()
        self._chk(self.lib.bbs_pool_proof_verify_submit(self.h, recs, len(sections), max_batch, ctypes.byref(h)), "bbs_pool_proof_verify_submit")
        return PoolJob(self, h, recs, sections, outs, whole, n_total)

    def proof_verify_packed(self, sections: List[PackedSection], n_total: Optional[int] = None, max_batch: int = 0) -> np.ndarray:
        """submit + wait"""
        return self.submit_packed(sections, n_total, max_batch).wait()

    def proof_verify_mixed(self, curve_of_item: Sequence[str], fetch_items, max_batch: int = 0) -> np.ndarray:
        """A list whose item i is of curve curve_of_item[i]; fetch_items(curve, ids) -> (proofs, disclosed_msgs,
        disclosed_idx[, headers, phs]) for those global ids.  Returns the statuses of the whole list, in list order."""
        by_curve: Dict[str, List[int]] = {}
        for i, c in enumerate(curve_of_item):
            by_curve.setdefault(c, []).append(i)
        sections = [self.pack(c, *fetch_items(c, ids), global_index=ids) for c, ids in sorted(by_curve.items())]
        return self.proof_verify_packed(sections, n_total=len(curve_of_item), max_batch=max_batch)
