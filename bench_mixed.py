"""bench.py --config mixed65536: BASELINE.json configs[4] -- 65 536 proof_verify, half BN254 and half BLS12-381
(split assumed 50/50), sharded over the GPUs of one node.

A step = the whole list once: `sharding.shard_plan` splits it by curve and then into contiguous ranges per rank, every
rank owns two contexts (BLS12-381 with --window-bits, BN254 with 16-bit windows), cuts its share into 4096-item
batches in HOST buffers, runs them through bbs_core_proof_verify_submit (one submitting thread, --inflight outstanding,
curves alternating), and the statuses are exchanged with ONE all_gather of int8 (RCCL for --backend nccl) and merged
with `sharding.merge_status`.  Consecutive steps are PIPELINED (mixed.ListPipeline): the jobs of the next lists are
submitted before the statuses of the oldest one are collected, so gather and merge run beside kernels and a rank that owns
two jobs per list (8 GPUs) never drains; every list's merged statuses are still checked.  Strong scaling: the list is fixed, `value` = 65 536 x steps / wall.  Every 16th item of
the global list is corrupted (one commitment incremented) and the merged statuses are compared with that pattern on
every rank after every step.

Items are generated per rank for that rank's ids only (the global id decides the item: distinct data on every rank).
"""
import json
import time


def run_mixed(args, pc, torch, dist, rank, local_rank, world, red_dev, barrier, total=65536, lib_path=None, emit=None, L=32, R=8):
    from bbs_sign_amd import mixed
    from bbs_sign_amd.sharding import shard_plan
    batch = args.batch
    if emit is None:
        def emit(line):
            print(line, flush=True)
    curve_of_item = ["bls12_381" if (i & 1) else "bn254" for i in range(total)]
    plan = shard_plan(curve_of_item, world)
    engines, suites = {}, {}
    for curve, w in (("bls12_381", args.window_bits), ("bn254", min(args.window_bits, 16))):
        suites[curve], engines[curve], _, _ = pc.bench_engine(curve, L, lib_path, w, device=local_rank)
        if getattr(args, "latency_mode", None) is not None:
            engines[curve].set_latency_mode(bool(args.latency_mode))
    import numpy as np
    expect = np.array([0 if i % 16 == 0 else 1 for i in range(total)], dtype=np.int8)

    def fetch_items(curve, ids):
        """The rank's items by global id (the SURVEY 8d workload with item number = global id)."""
        suite, eng = suites[curve], engines[curve]
        out_p, out_dm, out_di = [], [], []
        for lo in range(0, len(ids), 2048):
            part = ids[lo:lo + 2048]
            msgs, disclosed, rnds = pc.bench_items(suite, eng, len(part), L, R, ids=part)
            sigs, st = eng.core_sign_batch(msgs)
            assert (st == 1).all()
            proofs, st = eng.core_proof_gen_batch(sigs, msgs, disclosed, rnds)
            assert (st == 1).all()
            for g, p in zip(part, proofs):
                if g % 16 == 0:
                    p.commitments[0] = (p.commitments[0] + 1) % suite.curve.r
            out_p += proofs; out_dm += [m[:R] for m in msgs]; out_di += disclosed
        return out_p, out_dm, out_di

    t_prep = time.perf_counter()
    min_batch = getattr(args, "min_batch", None) or mixed.MIN_BATCH
    batches = mixed.prepare_rank(engines, plan[rank], fetch_items, batch, args.inflight, min_batch)
    t_prep = time.perf_counter() - t_prep
    depth = max(1, args.inflight)
    pipe = mixed.ListPipeline(batches, depth)
    ahead = getattr(args, "lists_in_flight", None) or pipe.lists_in_flight_for(max(depth, 8))     # measured: 8192-item share 7.3 / 5.2 / 5.0 / 4.9 ms per list at 1 / 2 / 3 / 4 lists ahead (profiles/r04_b_*)

    exch = mixed.StatusExchange(plan, rank, total, dist, red_dev)

    def run_lists(count):
        """`count` passes over the list, `ahead` of them submitted before the oldest is collected.  The statuses of list k
        are exchanged (ONE all_gather of int8) asynchronously: started when the list's jobs have delivered, finished and
        merged two lists later -- the submitting thread goes on retiring and submitting, it never sleeps on the collective
        (mixed.StatusExchange).  Merged statuses come back in list order."""
        out, handles, exchanges = [], [], []
        lag = 2                                            # exchanges outstanding before the oldest is finished: it has had two lists' time

        def collected(h):
            exchanges.append(exch.start(pipe.collect(h)))
            while len(exchanges) > lag:
                out.append(exch.finish(exchanges.pop(0)))

        for k in range(count):
            handles.append(pipe.submit_list())
            if len(handles) >= ahead:
                collected(handles.pop(0))
        while handles:
            collected(handles.pop(0))
        while exchanges:
            out.append(exch.finish(exchanges.pop(0)))
        return out

    if args.warmup < 0 or args.steps < 1:
        raise SystemExit("bench_mixed: --steps must be >= 1 and --warmup >= 0 (got %d / %d)" % (args.steps, args.warmup))
    warmup, steps = args.warmup, args.steps                # the driver's values, as given: never clamped
    for res in run_lists(warmup):
        assert np.array_equal(res, expect), "warm-up statuses differ from the expected pattern"
    barrier()
    t0 = time.perf_counter()
    results = run_lists(steps)
    barrier()
    dt = time.perf_counter() - t0
    assert len(results) == steps
    for res in results:
        assert np.array_equal(res, expect), "merged statuses differ from the expected pattern"
    tmax = torch.tensor([dt], dtype=torch.float64, device=red_dev)
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    dt = float(tmax.item())
    if rank == 0:
        emit(json.dumps({
            "metric": "BBS+ proof_verify/sec (mixed BN254 + BLS12-381 list of 65536)", "value": total * steps / dt,
            "unit": "proof_verify/s", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": "BASELINE configs[4]: %d proof_verify = %d BN254 + %d BLS12-381 (L=%d, R=%d), sharded by curve "
                                   "then contiguously over %d GPU(s), batches of at most %d items from host buffers, one all_gather of int8 "
                                   "statuses per step" % (total, total // 2, total // 2, L, R, world, batch),
                       "batches_per_rank": len(batches), "batch_sizes_rank0": sorted({b.n for b in batches}, reverse=True),
                       "items_per_rank": sum(b.n for b in batches), "batches_in_flight": depth, "lists_in_flight": ahead,
                       "retire_order": "completion (bbs_jobs_wait_any)",
                       "status_exchange": "asynchronous: one all_gather of int8 per list on a side stream, merged one list later",
                       "backend": args.backend if (world > 1 or dist is not None) else None,
                       "fixed_base_window_bits": {"bls12_381": args.window_bits, "bn254": min(args.window_bits, 16)}},
            "checks": {"merged_statuses_exact_every_step": True, "corrupted": "every 16th global item"},
            "prepare_s_rank0": t_prep}))
    for e in engines.values():
        e.close()


def run_mixed_pool(args, pc, devices, total=65536, lib_path=None, emit=None, L=32, R=8):
    """BASELINE configs[4] with ONE process: the list goes through bbs_pool (csrc/pool.hpp) -- partitioned by curve and then
    contiguously over the pool's member devices INSIDE the library, one submitting thread per member, statuses written into
    the caller's array in list order.  A step = the whole list once; `ahead` lists are kept in flight
    (bbs_pool_proof_verify_submit / bbs_pool_job_wait) so that the members never drain between lists.  Every 16th item of the
    list is corrupted and every list's statuses are compared with that pattern.  Strong scaling over the members."""
    import time as _time

    import numpy as np
    from bbs_sign_amd.pool import Pool
    if emit is None:
        def emit(line):
            print(line, flush=True)
    curve_of_item = ["bls12_381" if (i & 1) else "bn254" for i in range(total)]
    expect = np.array([0 if i % 16 == 0 else 1 for i in range(total)], dtype=np.int8)
    pool = Pool(devices, lib_path)
    pool.set_inflight(max(1, args.inflight))
    t_prep = _time.perf_counter()
    sections = []
    for curve, w in (("bls12_381", args.window_bits), ("bn254", min(args.window_bits, 16))):
        suite, eng, gens, sk = pc.bench_engine(curve, L, lib_path, w, device=devices[0])
        ids = [i for i in range(total) if curve_of_item[i] == curve]
        out_p, out_dm, out_di = [], [], []
        for lo in range(0, len(ids), 2048):
            part = ids[lo:lo + 2048]
            msgs, disclosed, rnds = pc.bench_items(suite, eng, len(part), L, R, ids=part)
            sigs, st = eng.core_sign_batch(msgs)
            assert (st == 1).all()
            proofs, st = eng.core_proof_gen_batch(sigs, msgs, disclosed, rnds)
            assert (st == 1).all()
            for g, p in zip(part, proofs):
                if g % 16 == 0:
                    p.commitments[0] = (p.commitments[0] + 1) % suite.curve.r
            out_p += proofs; out_dm += [m[:R] for m in msgs]; out_di += disclosed
        pk = eng.public_key()
        eng.close()                                    # the generating engine's tables leave before the pool's are built
        pool.set_window_bits(curve, w)
        pool.set_generators(curve, gens, suite.api_id)
        pool.set_public_key(curve, pk)
        sections.append(pool.pack(curve, out_p, out_dm, out_di, global_index=ids))
    t_prep = _time.perf_counter() - t_prep
    jobs_per_list = sum(-(-(-(-s.n // len(devices))) // args.batch) for s in sections)     # per member: ceil(share / batch) per curve
    ahead = getattr(args, "lists_in_flight", None) or max(2, -(-max(args.inflight, 8) // max(1, jobs_per_list)))
    if args.warmup < 0 or args.steps < 1:
        raise SystemExit("bench_mixed: --steps must be >= 1 and --warmup >= 0 (got %d / %d)" % (args.steps, args.warmup))

    def run_lists(count):
        out, flying = [], []
        for _ in range(count):
            flying.append(pool.submit_packed(sections, n_total=total, max_batch=args.batch))
            if len(flying) >= ahead:
                out.append(flying.pop(0).wait())
        while flying:
            out.append(flying.pop(0).wait())
        return out

    for res in run_lists(args.warmup):
        assert np.array_equal(res, expect), "warm-up statuses differ from the expected pattern"
    t0 = _time.perf_counter()
    results = run_lists(args.steps)
    dt = _time.perf_counter() - t0
    for res in results:
        assert np.array_equal(res, expect), "statuses differ from the expected pattern"
    emit(json.dumps({
        "metric": "BBS+ proof_verify/sec (mixed BN254 + BLS12-381 list of 65536)", "value": total * args.steps / dt,
        "unit": "proof_verify/s", "n_gpus": len(set(devices)), "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "u32", "data": "synthetic",
        "config": {"workload": "BASELINE configs[4]: %d proof_verify = %d BN254 + %d BLS12-381 (L=%d, R=%d), ONE process: bbs_pool over "
                               "member devices %s (sharded by curve, then contiguously, inside the library; one submitting thread per "
                               "member; no collective), jobs of at most %d items from host buffers" % (total, total // 2, total // 2, L, R, devices, args.batch),
                   "pool_members": len(devices), "jobs_per_member_per_list": jobs_per_list, "batches_in_flight_per_member": args.inflight,
                   "lists_in_flight": ahead, "single_process": True,
                   "fixed_base_window_bits": {"bls12_381": args.window_bits, "bn254": min(args.window_bits, 16)}},
        "checks": {"statuses_exact_every_step": True, "corrupted": "every 16th global item"},
        "prepare_s": t_prep}))
    pool.close()

