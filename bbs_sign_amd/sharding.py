"""Static sharding of a global batch of independent items over the GPUs of one node.

Items of the BBS+ hot path share no mutable state, so multi-GPU is a partition, not
a collective: every rank verifies a contiguous range and only a per-rank pass count (or the status
bytes) is exchanged.  Mixed batches are split by curve first (a BLS12-381 item costs ~2.5x a BN254
item), then evenly inside each curve, so every rank gets the same cost."""
from typing import Dict, List, Sequence, Tuple


def shard_range(n_items: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) of `n_items` for `rank` of `world`; sizes differ by at most one."""
    if world < 1 or not (0 <= rank < world) or n_items < 0:
        raise ValueError("bad shard request")
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_plan(curve_of_item: Sequence[str], world: int) -> List[Dict[str, List[int]]]:
    """For a mixed batch (curve name per item) return, per rank, {curve: [global item ids]}."""
    by_curve: Dict[str, List[int]] = {}
    for i, c in enumerate(curve_of_item):
        by_curve.setdefault(c, []).append(i)
    plan: List[Dict[str, List[int]]] = [dict() for _ in range(world)]
    for c, ids in sorted(by_curve.items()):
        for r in range(world):
            lo, hi = shard_range(len(ids), world, r)
            plan[r][c] = ids[lo:hi]
    return plan


_INDEX_CACHE = {}


def merge_status(plan: List[Dict[str, List[int]]], per_rank_status: List[Dict[str, Sequence[int]]], n_items: int):
    """Inverse of shard_plan for the gathered per-rank status lists -> numpy int8 array of n_items statuses.  Vectorised
    (one fancy-index store per rank and curve): every rank merges the WHOLE list after every step, and a Python loop over
    65 536 items costs more than the GPU spends verifying a rank's share of them at 8 GPUs."""
    import numpy as np
    key = id(plan)
    cached = _INDEX_CACHE.get(key)
    if cached is None or cached[0] is not plan:            # the id lists as arrays, once per plan
        if len(_INDEX_CACHE) > 8:
            _INDEX_CACHE.clear()
        cached = (plan, [{c: np.asarray(ids, dtype=np.int64) for c, ids in shard.items()} for shard in plan])
        _INDEX_CACHE[key] = cached
    index = cached[1]
    out = np.full(n_items, -128, dtype=np.int8)           # -128 = the library's "undecided": never a result
    seen = np.zeros(n_items, dtype=bool)
    for r, shard in enumerate(plan):
        for c, ids in shard.items():
            st = np.asarray(per_rank_status[r][c], dtype=np.int8)
            if len(st) != len(ids):
                raise ValueError("rank %d returned %d statuses for %d items" % (r, len(st), len(ids)))
            if len(ids):
                out[index[r][c]] = st
                seen[index[r][c]] = True
    if not seen.all():
        raise ValueError("items without a status")
    return out
