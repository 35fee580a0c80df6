"""The N > 1 path on CPU, world sizes 2 and 4 over gloo.
 * test_two_rank_gloo_shard_and_gather: partition + gather plumbing with a stand-in per-item result.
 * test_two_rank_engine_backed_mixed_list: bench.py --config mixed65536's own driver (bench_mixed.run_mixed) with each
   rank running the HOST TWIN engine (the same stage code compiled for x86, tests/hosttwin) on its shard of a small
   mixed BN254 + BLS12-381 list: two contexts per rank, distinct data per rank, submit path, one all_gather of int8
   statuses, merge, comparison with the expected pattern on every rank."""
import os
import socket
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from bbs_sign_amd.sharding import merge_status, shard_plan, shard_range  # noqa: E402


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 4096, 65536, 65537):
        for w in (1, 2, 3, 8):
            parts = [shard_range(n, w, r) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(parts[i][1] == parts[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in parts]
            assert max(sizes) - min(sizes) <= 1


def test_plan_balances_each_curve():
    curves = ["bls12_381" if i % 3 else "bn254" for i in range(100)]
    plan = shard_plan(curves, 8)
    for c in ("bls12_381", "bn254"):
        sizes = [len(p[c]) for p in plan]
        assert sum(sizes) == curves.count(c) and max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, n_items, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    curves = ["bls12_381" if i % 2 else "bn254" for i in range(n_items)]
    plan = shard_plan(curves, world)
    # stand-in per-item result: a deterministic function of the global item id
    truth = [1 if (i * 7 + 3) % 5 else 0 for i in range(n_items)]
    mine = {c: [truth[i] for i in ids] for c, ids in plan[rank].items()}
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    passed = torch.tensor([sum(sum(v) for v in mine.values())], dtype=torch.int64)
    dist.all_reduce(passed, op=dist.ReduceOp.SUM)
    if rank == 0:
        q.put((merge_status(plan, gathered, n_items).tolist() == truth, int(passed.item()) == sum(truth)))
    dist.destroy_process_group()


def test_two_rank_gloo_shard_and_gather():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 37, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok = q.get(timeout=120)
    for p in procs:
        p.join(60)
    assert ok == (True, True)


def _engine_worker(rank, world, port, twin, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import argparse
    import json
    import parity_cases as pc
    import bench_mixed
    dist.init_process_group("gloo", rank=rank, world_size=world)
    args = argparse.Namespace(batch=5, inflight=3, window_bits=4, warmup=1, steps=4, backend="gloo")
    lines = []
    bench_mixed.run_mixed(args, pc, torch, dist, rank, 0, world, "cpu", dist.barrier, total=18 * world, lib_path=twin,
                          emit=lines.append, L=4, R=2)
    if rank == 0:
        q.put(json.loads(lines[0]))
    dist.destroy_process_group()


def test_strong_scaling_plan():
    """BASELINE configs[4] at 1/2/4/8 ranks: every rank gets the same number of items of each curve, cut into the fewest
    jobs of at most 4096 items (measured: smaller jobs are slower, bbs_sign_amd/mixed.py batch_size_for); the knob that
    cuts a share into `inflight` jobs still works."""
    from bbs_sign_amd.mixed import batch_size_for
    curves = ["bls12_381" if (i & 1) else "bn254" for i in range(65536)]
    for world in (1, 2, 4, 8):
        plan = shard_plan(curves, world)
        for shard in plan:
            for c, ids in shard.items():
                assert len(ids) == 32768 // world
                size = batch_size_for(len(ids), 4096, 8)
                assert size == 4096
                assert -(-len(ids) // size) == max(1, 8 // world)
    assert batch_size_for(4096, 4096, 8, 512) == 512 and batch_size_for(16384, 4096, 8, 512) == 2048
    assert batch_size_for(100, 4096, 8, 512) == 512 and batch_size_for(0, 4096, 8) == 4096
    assert batch_size_for(9, 5, 3, 5) == 5               # test-sized lists: the cap wins


import pytest  # noqa: E402


@pytest.mark.parametrize("world", [2, 4])
def test_two_rank_engine_backed_mixed_list(world):
    from bbs_sign_amd import build as b
    twin = b.build(twin=True, verbose=False)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_engine_worker, args=(r, world, port, twin, q)) for r in range(world)]
    for p in procs:
        p.start()
    import queue
    line = None
    for _ in range(120):
        try:
            line = q.get(timeout=5)
            break
        except queue.Empty:
            if any(p.exitcode not in (None, 0) for p in procs):
                break
    for p in procs:
        p.join(60 if line is not None else 1)
        if p.is_alive():
            p.terminate()
    assert line is not None, "a rank died: exit codes %r" % [p.exitcode for p in procs]
    assert all(p.exitcode == 0 for p in procs)
    assert line["n_gpus"] == world and line["scaling"] == "strong" and line["checks"]["merged_statuses_exact_every_step"] is True
    assert line["config"]["batches_per_rank"] == 4          # 9 items per curve per rank in batches of 5
    assert line["config"]["items_per_rank"] == 18
