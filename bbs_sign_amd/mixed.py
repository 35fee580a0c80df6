"""core_proof_verify over a LIST of proofs of mixed curves, sharded over the GPUs of one node.

The reference verifies one proof per call (src/proof_verify.rs:19-61); a caller with a list simply loops.  Here the
list (BASELINE configs[4]: 65 536 proofs, half BN254, half BLS12-381) is partitioned by `sharding.shard_plan` --
split by curve first, contiguous ranges per rank inside a curve -- every rank owns one context per curve, cuts its
share into batches, keeps them in flight through bbs_core_proof_verify_submit, and the statuses (one int8 per item)
are exchanged with ONE all_gather (RCCL over xGMI for backend nccl).  No other collective: items are independent.
"""
from typing import Callable, Dict, List, Sequence

import numpy as np

from .sharding import merge_status, shard_plan


class PackedBatch:
    """One batch of one curve, packed for bbs_core_proof_verify_submit (host buffers)."""

    def __init__(self, eng, curve, ids, proofs, disclosed_msgs, disclosed_idx, headers=None, phs=None):
        self.eng, self.curve, self.ids = eng, curve, list(ids)
        self.n, self.keep, self.args = eng._pv_inputs(proofs, disclosed_msgs, disclosed_idx, headers, phs)


MIN_BATCH = 4096     # measured (tools/mixed_share_sweep.sh, profiles/r03_b_mixed_share_sweep.log): bigger jobs win


def batch_size_for(share: int, batch_max: int = 4096, inflight: int = 8, min_batch: int = MIN_BATCH) -> int:
    """Batch size for one curve's share of a rank: ceil(share / inflight), clamped to [min_batch, batch_max] and rounded
    up to whole wavefronts.  Under strong scaling the share shrinks with the number of ranks (65 536 items / 2 curves /
    8 ranks = 4096 = ONE job per curve).  Cutting such a share into many small jobs so that "enough" are in flight is
    the wrong cure -- measured on one MI355X with a rank's share of the list at 8 GPUs (8192 items): 2 jobs of 4096 ->
    8.2 ms per step, 4 x 2048 -> 8.9 ms, 8 x 1024 -> 12.4 ms, 16 x 512 -> 17.4 ms: a job's critical path (~6 ms) does not
    shrink with its size, every job costs the submitting thread ~0.4 ms, and 2 streams x 16 jobs exceed the hardware
    queues.  What shortens a step with few jobs is the LATENCY FORM of a job (bbs_ctx_set_latency_mode, automatic when
    at most two jobs of a context are alive).  min_batch therefore defaults to batch_max's 4096; the knob stays for
    A/B runs (bench.py --min-batch)."""
    if share <= 0:
        return max(1, min(batch_max, min_batch))
    size = -(-share // max(1, inflight))                 # ceil(share / inflight)
    size = max(min_batch, min(batch_max, size))
    return min(batch_max, -(-size // 64) * 64)


def prepare_rank(engines: Dict[str, object], plan_for_rank: Dict[str, List[int]],
                 fetch_items: Callable[[str, Sequence[int]], tuple], batch: int = 4096, inflight: int = 8,
                 min_batch: int = MIN_BATCH) -> List[PackedBatch]:
    """Cut this rank's share into batches per curve (batch_size_for: at most `batch` items, by default as few jobs as
    that allows).  fetch_items(curve, ids) returns (proofs, disclosed_msgs, disclosed_idx[, headers, phs]) for those
    global item ids."""
    out = []
    for curve in sorted(plan_for_rank):
        ids = plan_for_rank[curve]
        size = batch_size_for(len(ids), batch, inflight, min(min_batch, batch))
        for lo in range(0, len(ids), size):
            part = ids[lo:lo + size]
            out.append(PackedBatch(engines[curve], curve, part, *fetch_items(curve, part)))
    return out


class ListPipeline:
    """One rank's jobs of CONSECUTIVE lists kept in flight: the jobs of list k + 1 are submitted before the statuses of
    list k are collected, gathered and merged, so the device never drains between lists and the exchange (one all_gather
    of int8 per list) and the merge run beside the next list's kernels.  At 8 GPUs a rank owns two jobs per list; run one
    list at a time the GPU idles through every list's ramp, drain, gather and merge (round 3: 7.2 - 7.7 ms per 8192-item
    share against 4.35 ms of critical path).  Jobs are retired in completion order (bbs_jobs_wait_any), at most `inflight`
    outstanding; submission alternates curves so that BN254 and BLS12-381 jobs overlap on the device."""

    def __init__(self, batches: List[PackedBatch], inflight: int = 8):
        self.by_curve: Dict[str, List[PackedBatch]] = {}
        for b in batches:
            self.by_curve.setdefault(b.curve, []).append(b)
        self.order: List[PackedBatch] = []
        queues = [list(v) for _, v in sorted(self.by_curve.items())]
        while any(queues):
            for q in queues:
                if q:
                    self.order.append(q.pop(0))
        self.inflight = max(1, inflight)
        self.pending = []            # (list id, batch, job) in submission order
        self.results: Dict[int, Dict[int, np.ndarray]] = {}
        self.next_id = 0

    def _retire_one(self):
        from .engine import Job
        k = Job.wait_any([j for _, _, j in self.pending])      # raises if an item was left undecided (BBS_E_STATE)
        lid, b, job = self.pending.pop(k)
        self.results[lid][id(b)] = job.result.copy()
        job.free()

    def submit_list(self) -> int:
        """Enqueue one pass over this rank's share; returns the list's handle for collect()."""
        lid = self.next_id
        self.next_id += 1
        self.results[lid] = {}
        for b in self.order:
            while len(self.pending) >= self.inflight:
                self._retire_one()
            self.pending.append((lid, b, b.eng.submit_packed(b.n, b.args)))
        return lid

    def collect(self, lid: int) -> Dict[str, np.ndarray]:
        """Wait until every job of list `lid` has delivered -> {curve: statuses in plan order}.  Jobs of later lists that
        finish meanwhile are retired too (their statuses wait for their own collect)."""
        while len(self.results[lid]) < len(self.order):
            self._retire_one()
        res = self.results.pop(lid)
        return {curve: (np.concatenate([res[id(b)] for b in bs]) if bs else np.zeros(0, dtype=np.int8))
                for curve, bs in self.by_curve.items()}

    def lists_in_flight_for(self, want_jobs: int) -> int:
        """How many lists to keep submitted ahead so that about `want_jobs` jobs are alive (at least two lists)."""
        return max(2, -(-max(1, want_jobs) // max(1, len(self.order))))


def run_rank(batches: List[PackedBatch], inflight: int = 8) -> Dict[str, np.ndarray]:
    """One list, start to finish: submit every batch (at most `inflight` outstanding, one submitting thread, completion-order
    retire), return {curve: statuses in plan order}."""
    pipe = ListPipeline(batches, inflight)
    return pipe.collect(pipe.submit_list())


class StatusExchange:
    """The ONE collective of the path -- an all_gather of int8 statuses per list (each rank's share in plan order, padded to
    the largest share) followed by sharding.merge_status -- taken OFF the submitting thread.

    Round 4 ran it synchronously between two lists: pageable copy to the device, blocking all_gather, `.cpu()`.  With the
    chip full of one-wavefront-per-SIMD kernels that run for milliseconds, the collective's own kernel waits for a free slot
    and the thread that should be retiring and submitting jobs sleeps on it -- 8 % of a rank's step with ONE rank
    (5.29 vs 4.90 ms per 8192-item list, profiles/r04_u_mixed_rccl_world1.json), and with eight ranks it would also sleep
    until the slowest rank arrives.  Now: start() only ENQUEUES -- page-locked staging buffer, copy, all_gather
    (async_op=True) and the copy back, all on a side stream of torch's -- and returns a handle; finish() is called one list
    LATER, when the exchange has long completed, and merges.  Nothing on the submitting thread waits for a collective in
    steady state.  Every rank starts its exchanges in list order, so the collectives match up.

    dist=None: single process (start / finish only reorder).  CPU process groups (gloo, the world-2 / world-4 tests): the
    same protocol with async work handles."""

    def __init__(self, plan, rank: int, n_items: int, dist=None, device="cpu", slots: int = 4):
        self.plan, self.rank, self.n_items, self.dist, self.device = plan, rank, n_items, dist, device
        self.world = len(plan)
        self.curves = sorted({c for p in plan for c in p})
        self.sizes = [sum(len(p.get(c, [])) for c in self.curves) for p in plan]
        self.pad = max(self.sizes) if self.sizes else 0
        self.on_gpu = dist is not None and str(device).startswith("cuda")
        self.free_slots, self.side = [], None
        if dist is not None:
            import torch
            self.torch = torch
            if self.on_gpu:
                self.side = torch.cuda.Stream(device=device)
            for _ in range(max(1, slots)):
                self.free_slots.append(self._new_slot())

    def _new_slot(self):
        torch = self.torch
        pin = self.on_gpu
        slot = {"in_host": torch.empty(max(1, self.pad), dtype=torch.int8, pin_memory=pin),
                "out_host": torch.empty(max(1, self.pad) * self.world, dtype=torch.int8, pin_memory=pin)}
        if self.on_gpu:
            slot["in_dev"] = torch.empty(max(1, self.pad), dtype=torch.int8, device=self.device)
            slot["out_dev"] = torch.empty(max(1, self.pad) * self.world, dtype=torch.int8, device=self.device)
            slot["done"] = torch.cuda.Event()
        return slot

    def _flat(self, mine):
        flat = np.concatenate([np.asarray(mine.get(c, np.zeros(0, dtype=np.int8)), dtype=np.int8) for c in self.curves]) \
            if self.curves else np.zeros(0, dtype=np.int8)
        assert len(flat) == self.sizes[self.rank], (len(flat), self.sizes[self.rank])
        return flat

    def start(self, mine: Dict[str, np.ndarray]):
        """Enqueue the exchange of this rank's statuses of one list; returns a handle for finish()."""
        flat = self._flat(mine)
        if self.dist is None:
            return {"local": flat}
        torch, dist = self.torch, self.dist
        slot = self.free_slots.pop() if self.free_slots else self._new_slot()
        buf = slot["in_host"].numpy()
        buf[:] = -128                                       # padding = the internal "undecided" value, never a result
        buf[:len(flat)] = flat
        if self.on_gpu:
            with torch.cuda.stream(self.side):
                slot["in_dev"].copy_(slot["in_host"], non_blocking=True)
                work = dist.all_gather_into_tensor(slot["out_dev"], slot["in_dev"], async_op=True)
                work.wait()                                 # RCCL: orders the side stream behind the collective, does not block the host
                slot["out_host"].copy_(slot["out_dev"], non_blocking=True)
                slot["done"].record(self.side)
            slot["work"] = None
        else:
            outs = list(slot["out_host"].view(self.world, -1).unbind(0))
            slot["work"] = dist.all_gather(outs, slot["in_host"], async_op=True)
        return slot

    def finish(self, handle) -> np.ndarray:
        """Wait for the exchange started as `handle` (normally long complete) and merge -> statuses of the whole list."""
        if "local" in handle:
            gathered = [handle["local"]]
        else:
            if self.on_gpu:
                handle["done"].synchronize()
            else:
                handle["work"].wait()
            rows = handle["out_host"].numpy().reshape(self.world, -1)
            gathered = [rows[r, :self.sizes[r]].copy() for r in range(self.world)]
            handle["work"] = None
            self.free_slots.append(handle)
        per_rank = []
        for r, p in enumerate(self.plan):
            d, at = {}, 0
            for c in self.curves:
                k = len(p.get(c, []))
                d[c] = gathered[r][at:at + k]
                at += k
            per_rank.append(d)
        return merge_status(self.plan, per_rank, self.n_items)


def gather_statuses(plan, rank: int, mine: Dict[str, np.ndarray], n_items: int, dist=None, device="cpu") -> np.ndarray:
    """One all_gather of int8 statuses, then sharding.merge_status, start to finish (a caller with ONE list; a loop over
    lists keeps a StatusExchange and finishes every exchange one list late).  dist=None: single process."""
    x = StatusExchange(plan, rank, n_items, dist, device, slots=1)
    return x.finish(x.start(mine))


def proof_verify_mixed(engines, curve_of_item: Sequence[str], fetch_items, world: int = 1, rank: int = 0, dist=None,
                       device="cpu", batch: int = 4096, inflight: int = 8) -> np.ndarray:
    """The whole path for one call: plan, pack, run, gather.  Every rank returns the merged statuses."""
    plan = shard_plan(curve_of_item, world)
    batches = prepare_rank(engines, plan[rank], fetch_items, batch, inflight)
    mine = run_rank(batches, max(inflight, 1))
    return gather_statuses(plan, rank, mine, len(curve_of_item), dist, device)
