#!/usr/bin/env python3
"""Headline benchmark: BBS+ proof_verify/s, BLS12-381, 32 messages / 8 disclosed, batch 4096 per GPU.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)
    python bench.py --config mixed65536 [--gpus N]     # BASELINE configs[4]: 65 536 mixed BN254 + BLS12-381 proofs

A step = one pass of core_proof_verify (src/proof_verify.rs:64-116) over one batch of 4096 proofs handed over in HOST
buffers (BASELINE.json configs[3]; the proofs are produced untimed by the engine's own sign -> proof_gen).  The timed
region is SURVEY 8(d)'s: one submitting thread per GPU calls bbs_core_proof_verify_submit -- staging in page-locked
memory, ONE asynchronous H2D copy of the algorithmic bytes, validation / unpacking / SoA transposition on the device,
the verification kernels, asynchronous D2H of the statuses -- keeps `--inflight` batches outstanding and reads every
batch's statuses.  `value` is that host-inclusive rate.  The rate from batches already resident in HBM is reported
beside it as `resident` (never the headline).

Every in-flight slot holds a DIFFERENT batch (item ids offset by slot and by rank), one slot has every 16th item
corrupted (one commitment incremented), and the statuses of every timed step are compared with the expected pattern
after the timed region: a kernel that did nothing cannot pass (undecided items make bbs_job_wait fail).

Timing: barrier + synchronize, K steps, barrier + synchronize; wall time = max over ranks.  HIP events around every
stage of every timed step (recorded on the stream the stage is launched on) give the per-kernel durations of the same
timed region.  Prints ONE JSON line (rank 0).
"""
import os

# The engine keeps many independent batches in flight, every batch on streams of its own (three per proof_verify job since
# round 5).  The HIP runtime maps all streams of a process onto 4 hardware queues by default, so a long, narrow kernel (a
# batch's tail) blocks the streams that share its queue (DESIGN.md 5 rule 6).  Must be set before the first HIP call of the
# process (torch initialises HIP before the engine is loaded); the library's own constructor does the same when it is
# loaded first (20: what its scratch budget allows with room to spare, bbs_runtime_queue_budget).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")
import argparse
import json
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# algorithmic bytes per BLS12-381 proof_verify (L=32, R=8): 1040 proof octets
# (3 x 48 + 28 x 32) + 256 (8 disclosed scalars) + 64 (8 indexes) + 1 status
ALG_BYTES_PER_PROOF_VERIFY = 1361
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec


def load_counters():
    """Per-kernel counters of one 4096-item launch, measured by rocprofv3 (--pmc passes, tools/run_profile.sh) and kept
    under profiles/: SQ_INSTS_VALU, FETCH_SIZE (KiB, doubled for gfx950 per MI355X_MICROARCH.md), WRITE_SIZE (KiB),
    and the issue-cost model of tools/valu_model.py (cycles per wave-instruction from the ISA histogram x the
    micro-benchmarked cost of every opcode).  bench.py only reads the newest committed file."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_counters.json")))
    if not files:
        return None, None
    with open(files[-1]) as f:
        return json.load(f), os.path.relpath(files[-1], ROOT)


def host_cores():
    """Threads this process may really use: affinity, cgroup CPU quota, and the GPU box's share (16)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return max(1, min(n, 16))


def cpu_baseline(budget_s=20.0):
    """The oracle's plain-C restatement of the reference path (oracle/c: per-call domain, 38 independent double-and-add
    scalar multiplications, two full pairings) timed on the host cores on a bounded sample of the same workload:
    single thread and one thread per core, all four operations, both curves; config 1 (README.md:64-81) as plumbing."""
    from oracle import c_baseline
    return c_baseline.run(host_cores(), budget_s)


class Slot:
    """One in-flight slot of the headline: a distinct batch, packed once into host buffers."""

    def __init__(self, eng, proofs, dm, disclosed, expect):
        import numpy as np
        self.n, self.keep, self.args = eng._pv_inputs(proofs, dm, disclosed, None, None)
        self.expect = np.asarray(expect, dtype=np.int8)


def make_slots(pc, suite, eng, n, L, R, n_slots, first_item, corrupt_slot=1):
    """n_slots distinct batches (item ids first_item + slot * n ...); slot `corrupt_slot` has every 16th item corrupted
    (one commitment incremented: SURVEY 8d).  Returns (slots, raw data of slot 0 for the other legs)."""
    slots, raw0 = [], None
    for s in range(n_slots):
        msgs, disclosed, rnds = pc.bench_items(suite, eng, n, L, R, first_item + s * n)
        sigs, st = eng.core_sign_batch(msgs)
        assert (st == 1).all()
        proofs, st = eng.core_proof_gen_batch(sigs, msgs, disclosed, rnds)
        assert (st == 1).all()
        dm = [m[:R] for m in msgs]
        if s == 0:
            raw0 = (msgs, disclosed, rnds, sigs, [pc.to_engine_proof(p) for p in proofs], dm)
        expect = [1] * n
        if s == corrupt_slot:
            for i in range(0, n, 16):
                proofs[i].commitments[0] = (proofs[i].commitments[0] + 1) % suite.curve.r
                expect[i] = 0
        slots.append(Slot(eng, proofs, dm, disclosed, expect))
    return slots, raw0


def submit_loop(eng, slots, steps, inflight, first_step=0, collect_times=False, fifo=False):
    """The serving loop of one submitting thread: step k submits slot k % len(slots); at most `inflight` jobs are
    outstanding; every job's statuses are read.  Jobs are retired in COMPLETION order (bbs_jobs_wait_any: the thread sleeps
    until whichever outstanding job finishes first, reads its statuses and submits the next batch in its place) -- waiting
    for the oldest job instead (fifo=True, round 3's loop, kept for the A/B) makes jobs submitted together finish together
    and the loop run in convoys.  Returns (#status mismatches, {stage: summed ms}, summed job ms)."""
    from bbs_sign_amd import Job
    pending, bad = [], 0
    stage_ms, job_ms = {}, 0.0

    def retire():
        nonlocal bad, job_ms
        if fifo:
            k = 0
            pending[0][1].wait()                     # raises if any item was left undecided (BBS_E_STATE)
        else:
            k = Job.wait_any([j for _, j in pending])
        slot, job = pending.pop(k)
        bad += int((job.result != slot.expect).sum())
        if collect_times:
            tot, st = job.stage_times()
            job_ms += tot
            for nm, v in st.items():
                stage_ms[nm] = stage_ms.get(nm, 0.0) + v
        job.free()

    for k in range(first_step, first_step + steps):
        if len(pending) >= inflight:
            retire()
        slot = slots[k % len(slots)]
        pending.append((slot, eng.submit_packed(slot.n, slot.args)))
    while pending:
        retire()
    return bad, stage_ms, job_ms


def resident_rate(eng, slots_data, n, steps, inflight, only_form=None):
    """Batches already resident in HBM (uploaded, validated, unpacked): kernels only.  -> (proof_verify/s, stage ms)"""
    from bbs_sign_amd import Job
    eng.set_latency_mode(False)                  # every resident job in the throughput form (AUTO would give the first two the other)
    jobs = [eng.core_proof_verify_upload(*d) for d in slots_data[:inflight]]
    for j in jobs:
        j.run()
    for j in jobs:
        j.wait()
    Job.run_many_timed(jobs, len(jobs))
    ms, st = Job.run_many_timed(jobs, steps)
    for j in jobs:
        j.free()
    # one batch at a time, in both forms of a job (the library's AUTO picks the latency form for a job that is alone)
    single = {}
    for form, mode in (("throughput_form", False), ("latency_form", True)):
        if only_form is not None and mode != bool(only_form):      # --latency-mode given (profiling): only that form's kernels run
            single[form] = None
            continue
        eng.set_latency_mode(mode)
        j = eng.core_proof_verify_upload(*slots_data[0])
        j.run(); j.wait()
        one_ms, one_st = j.run_timed(3, per_stage=True)
        assert (j.status() == 1).all()
        j.free()
        single[form] = {"ms": one_ms / 3, "proof_verify_per_s": n / (one_ms / 3 * 1e-3), "stage_ms": {k: v / 3 for k, v in one_st.items()}}
    eng.set_latency_mode("auto")
    return n * steps / (ms * 1e-3), {k: v / steps for k, v in st.items()}, single


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--window-bits", type=int, default=20)
    ap.add_argument("--inflight", type=int, default=6, help="batches in flight per GPU (distinct data in every slot).  6 = the measured "
                    "optimum of rounds 3 - 4 (two streams per job on 14 hardware queues: 4 / 6 / 7 / 8 in flight 1.51 / 1.52 / 1.48 / 1.44 M/s, "
                    "profiles/r03_x_inflight_sweep.log); round 5: three streams per job on 20 queues, 6 in flight 1.55 M/s "
                    "(profiles/r05_a_ab_split_msm_layouts.log)")
    ap.add_argument("--config", default="proof_verify_4096", choices=["proof_verify_4096", "mixed65536"])
    ap.add_argument("--total", type=int, default=65536, help="mixed65536 only: length of the list (8192 = one rank's share "
                    "of the 65 536-item list at 8 GPUs, to rehearse the strong-scaling regime on one GPU)")
    ap.add_argument("--lists-in-flight", type=int, default=None, help="mixed65536 only: lists (steps) submitted ahead of the one being "
                    "collected; default = enough for --inflight jobs alive, at least 2")
    ap.add_argument("--min-batch", type=int, default=None, help="mixed65536 only: smallest job a rank's share is cut into")
    ap.add_argument("--latency-mode", type=int, default=None, help="A/B and profiling: bbs_ctx_set_latency_mode 0 = throughput form for every "
                    "job, 1 = latency form, default = the library's AUTO (by live jobs)")
    ap.add_argument("--dedicated-queues", type=int, default=None, help="bbs_runtime_set_dedicated_queues(k): up to k job streams get a "
                    "hardware queue of their own, independent of GPU_MAX_HW_QUEUES (for processes whose first HIP call precedes the library)")
    ap.add_argument("--fifo-retire", action="store_true", help="A/B: wait for the OLDEST outstanding job (round 3's loop) instead of "
                    "bbs_jobs_wait_any's completion order")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed sign/verify/proof_gen/BN254 figures")
    ap.add_argument("--fixed-base-tree", type=int, default=None, help="A/B: bbs_ctx_set_fixed_base_tree on (1) / off (0); default = the library's")
    ap.add_argument("--no-stage-timing", action="store_true", help="no HIP events in the timed region (A/B of their cost)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse "
                    "the N > 1 path on a box with fewer GPUs than ranks)")
    ap.add_argument("--all-ranks-on-device", type=int, default=None, help="rehearsal only: every rank uses this GPU")
    ap.add_argument("--force-dist", action="store_true", help="rehearsal: initialise the process group (RCCL for --backend nccl) even "
                    "with one rank, so that the collectives of the N > 1 path run on the real backend of a one-GPU box")
    ap.add_argument("--single-process", action="store_true", help="mixed65536 only: ONE process drives all --gpus devices through bbs_pool -- "
                    "the multi-GPU fan-out BEHIND the C ABI (include/bbs_sign_amd.h, SURVEY 8(b)), what a Rust / C host gets -- "
                    "instead of one process per GPU and an all_gather")
    ap.add_argument("--pool-devices", default=None, help="--single-process: comma-separated member device ids (default 0 .. --gpus - 1; an "
                    "id may repeat: two members on one GPU)")
    ap.add_argument("--min-region-s", type=float, default=1.0, help="if the K timed steps last less than this, a second, "
                    "longer region of the same loop is timed and reported beside `value` as `long_region` (0 = off)")
    args = ap.parse_args()

    if args.single_process:
        if args.config != "mixed65536":
            raise SystemExit("bench.py: --single-process goes with --config mixed65536")
        from bbs_sign_amd import workload as pc
        from bench_mixed import run_mixed_pool
        devices = [int(x) for x in args.pool_devices.split(",")] if args.pool_devices else list(range(args.gpus))
        run_mixed_pool(args, pc, devices, total=args.total)
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # a bare `python bench.py --gpus N`: start the N ranks ourselves, as a CHILD process, before this process has
        # touched torch.cuda or HIP (a process that initialised the GPU must never exec), and leave with its exit code
        import socket
        import subprocess
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.all_ranks_on_device is not None:
            local_rank = args.all_ranks_on_device
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with --nproc-per-node equal to --gpus, or leave "
                         "WORLD_SIZE unset and bench.py starts the ranks itself)" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    red_dev = "cuda" if (dist is None or args.backend == "nccl") else "cpu"

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    from bbs_sign_amd import workload as pc    # SURVEY 8(d) inputs from the product's own host functions: no oracle, no tests/
    from bbs_sign_amd import Engine  # noqa: F401  (fails loudly if the HIP library is missing)
    if args.dedicated_queues is not None:
        from bbs_sign_amd import _lib
        assert _lib.load_library().bbs_runtime_set_dedicated_queues(args.dedicated_queues) == 0

    if args.config == "mixed65536":
        from bench_mixed import run_mixed
        run_mixed(args, pc, torch, dist, rank, local_rank, world, red_dev, barrier, total=args.total)
        if dist is not None:
            dist.destroy_process_group()
        return

    n, L, R = args.batch, 32, 8
    n_slots = max(1, args.inflight)
    suite, eng, gens, sk = pc.bench_engine("bls12_381", L, None, args.window_bits, device=local_rank)
    if args.fixed_base_tree is not None:
        eng.set_fixed_base_tree(bool(args.fixed_base_tree))
    if args.latency_mode is not None:
        eng.set_latency_mode(args.latency_mode if args.latency_mode in (0, 1) else "auto")
    # every rank and every slot verifies its own batch: item ids offset by rank and slot
    slots, raw0 = make_slots(pc, suite, eng, n, L, R, n_slots, first_item=rank * n_slots * n)
    msgs, disclosed, rnds, sigs, proofs, dm = raw0
    eng.set_stage_timing(not args.no_stage_timing)

    bad, _, _ = submit_loop(eng, slots, max(args.warmup, n_slots), n_slots, fifo=args.fifo_retire)
    assert bad == 0, "warm-up statuses differ from the expected pattern"

    barrier()
    t0 = time.perf_counter()
    bad, stage_ms, job_ms = submit_loop(eng, slots, args.steps, n_slots, first_step=0, collect_times=not args.no_stage_timing, fifo=args.fifo_retire)
    barrier()
    dt = time.perf_counter() - t0
    assert bad == 0, "timed statuses differ from the expected pattern (%d items)" % bad
    eng.set_stage_timing(False)
    passed = sum(int((slots[k % n_slots].expect == 1).sum()) for k in range(args.steps))
    # the K steps the driver asked for may last a few tens of milliseconds, of which filling and draining the
    # `inflight` slots is a visible share: time the same loop once more over a region of at least --min-region-s
    long_region = None
    if args.min_region_s > 0 and dt < args.min_region_s:
        k2 = int(min(4096, max(args.steps, args.min_region_s / (dt / args.steps) * 1.1)))
        barrier()
        t1 = time.perf_counter()
        bad2, _, _ = submit_loop(eng, slots, k2, n_slots, fifo=args.fifo_retire)
        barrier()
        dt2 = time.perf_counter() - t1
        assert bad2 == 0, "long-region statuses differ from the expected pattern"
        long_region = (k2, dt2)

    # ---- untimed legs ------------------------------------------------------------------------------------------
    slots_data = [(proofs, dm, disclosed)]
    res_rate, res_stage, single = resident_rate(eng, slots_data * n_slots, n, max(32, args.steps // 2), n_slots,
                                                args.latency_mode if args.latency_mode in (0, 1) else None) \
        if rank == 0 else (None, None, None)
    extras = None
    if rank == 0 and world == 1 and not args.no_extras:
        from bench_extras import other_ops
        extras = other_ops(args, pc, eng, suite, n, L, R, msgs, disclosed, rnds, sigs, proofs, dm, local_rank, submit_loop, make_slots, slots)

    tmax = torch.tensor([dt], dtype=torch.float64, device=red_dev)
    cnt = torch.tensor([passed], dtype=torch.int64, device=red_dev)
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)       # the only exchange: one pass-count per rank
    dt = float(tmax.item())
    if long_region is not None:
        t2 = torch.tensor([long_region[1]], dtype=torch.float64, device=red_dev)
        if dist is not None:
            dist.all_reduce(t2, op=dist.ReduceOp.MAX)
        long_region = (long_region[0], float(t2.item()))

    if rank == 0:
        value = world * n * args.steps / dt
        counters, counters_file = load_counters()
        lib_hash = eng.lib.bbs_source_hash().decode()
        counters_hash = (counters or {}).get("library_source_hash")
        counters_current = (counters_hash == lib_hash) if counters_hash else None
        out = {
            "metric": "BBS+ proof_verify/sec (BLS12-381, 32-msg)",
            "value": value, "unit": "proof_verify/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": "BLS12-381 core_proof_verify, batch %d per GPU, 32 msgs / 8 disclosed, empty "
                                   "header/ph, one issuer key (BASELINE configs[3], proof_verify leg)" % n,
                       "operating_point": ("ONE issuer key and generator set per GPU with %d-bit fixed-base windows (%.1f GB of tables: a choice for a "
                                           "service with one issuer, sized for the 288 GB of HBM3E); the library's own default, "
                                           "bbs_ctx_set_window_bits(ctx, 0), is 16 bits (2.1 GB) at 32 messages -- that rate is "
                                           "`value_at_library_default_window_bits`" % (
                                               args.window_bits, (L + 2) * ((256 + args.window_bits - 1) // args.window_bits) * (1 << (args.window_bits - 1)) * 128 / 1e9)),
                       "batch_per_gpu": n, "messages": L, "disclosed": R, "fixed_base_window_bits": args.window_bits,
                       "batches_in_flight": n_slots, "retire_order": "fifo (oldest first)" if args.fifo_retire else "completion (bbs_jobs_wait_any)",
                       "hw_queues": int(eng.lib.bbs_runtime_hw_queues()),
                       "dedicated_queues": args.dedicated_queues if args.dedicated_queues is not None else (os.environ.get("BBS_DEDICATED_QUEUES") or 0),
                       "timed_region": "host buffers -> page-locked staging -> one async H2D -> device-side validation/"
                                       "unpack -> kernels -> async D2H of statuses; one submitting thread per GPU "
                                       "(SURVEY 8d)",
                       "distinct_batches": n_slots, "corrupted_slot": "slot 1: every 16th item",
                       "parallelism": "independent batches per GPU, no data-path collective"},
            "checks": {"statuses_exact_every_step": True, "proofs_passed": int(cnt.item()),
                       "proofs_failed_as_expected": world * n * args.steps - int(cnt.item())},
            "resident": {"proof_verify_per_s": res_rate, "host_inclusive_over_resident": value / world / res_rate,
                         "stage_ms_per_step": res_stage,
                         "note": "the same kernels on batches already uploaded / validated / unpacked in HBM, %d in "
                                 "flight; never the headline" % n_slots},
            "single_batch": dict(single["latency_form"] or single["throughput_form"],
                                 form=("latency form = what the library's AUTO mode gives a job that is alone on its context "
                                       "(bbs_ctx_set_latency_mode)") if single["latency_form"] else "throughput form (--latency-mode 0)",
                                 throughput_form=single["throughput_form"]),
        }
        if dt < 1.0:
            out["timed_region_note"] = ("the timed region is %.0f ms (%d steps): filling and draining the %d in-flight slots is "
                                        "inside it; `long_region` times the same loop for >= %.1f s" % (dt * 1e3, args.steps, n_slots, args.min_region_s))
        if long_region is not None:
            out["long_region"] = {"steps": long_region[0], "seconds": long_region[1],
                                  "proof_verify_per_s": world * n * long_region[0] / long_region[1],
                                  "ms_per_step": long_region[1] / long_region[0] * 1e3,
                                  "note": "same submit loop, same slots, statuses checked; never `value`"}
        if stage_ms:
            per_step = {k: v / args.steps for k, v in stage_ms.items()}
            dom = max(per_step, key=per_step.get)
            dom_ms = per_step[dom]
            achieved = ALG_BYTES_PER_PROOF_VERIFY * n / (dom_ms * 1e-3) / 1e9
            kc = (counters or {}).get("kernels", {})
            traffic = None
            if dom in kc:
                traffic = (2 * kc[dom]["FETCH_SIZE_KiB"] + kc[dom]["WRITE_SIZE_KiB"]) * 1024 * (n / 4096.0)
            alg_bytes = ALG_BYTES_PER_PROOF_VERIFY * n
            excl_ms = (single["throughput_form"]["stage_ms"].get(dom) if single and single["throughput_form"] else None)
            out["roofline"] = {
                "bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                "frac_uses": "duration_ms_co_scheduled",
                "duration_ms_co_scheduled": dom_ms,
                "duration_ms_exclusive": excl_ms,
                "frac_exclusive": (alg_bytes / (excl_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if excl_ms else None,
                "frac_per_step": alg_bytes / (dt / args.steps / world) / 1e9 / HBM_PEAK_GBS,
                "traffic_over_algorithmic": (traffic / alg_bytes) if traffic else None,
                "traffic_from_this_library": counters_current,
                "target_40_percent_of_hbm": "NOT MET and not attainable: 1361 algorithmic bytes against ~2.9e5 wavefront "
                                            "instructions per proof (SURVEY 7/8d); the binding resource is integer VALU issue (valu_issue)",
                "note": "algorithmic bytes/unit = %d (SURVEY 8d) x %d units per launch; `frac` divides them by the launch duration "
                        "measured with HIP events around the kernel on its own stream, averaged over the timed region, i.e. the "
                        "wall duration of a kernel CO-SCHEDULED with the kernels of the other %d batches in flight (longer than "
                        "ms_per_step); duration_ms_exclusive = the same kernel with the chip to itself (single_batch, throughput "
                        "form) and frac_exclusive the fraction priced with it; frac_per_step prices the whole step.  traffic = "
                        "2 x FETCH_SIZE + WRITE_SIZE per launch from %s (the x2 is the guide's gfx950 correction for streaming "
                        "reads; the MSM kernel's 112-byte table gathers are probably over-corrected by it)" % (
                            ALG_BYTES_PER_PROOF_VERIFY, n, n_slots - 1, counters_file)}
            out["stage_ms_per_step"] = per_step
            out["gpu_ms_per_job_events"] = job_ms / args.steps
            if counters:
                # the binding resource: cycles the SIMDs need to ISSUE the batch's vector instructions (per kernel:
                # wave-instructions from SQ_INSTS_VALU x modelled cycles per instruction, tools/valu_model.py) against
                # the cycles 1024 SIMDs have in ms_per_step at the clock measured under load (SQ_BUSY_CYCLES)
                tot_cyc = sum(k["valu_insts"] * k["cycles_per_inst"] for k in kc.values()) * (n / 4096.0)
                tot_inst = sum(k["valu_insts"] for k in kc.values()) * (n / 4096.0)
                tot_cyc8 = sum(k["valu_insts"] * k["cycles_per_inst_if_waves_per_simd"]["8"] for k in kc.values()) * (n / 4096.0)
                clk = counters["clock_ghz_under_load"]
                avail = 1024 * clk * 1e9 * (dt / args.steps) / world
                tot_cyc2 = sum(k["valu_insts"] * k["cycles_per_inst_if_waves_per_simd"]["2"] for k in kc.values()) * (n / 4096.0)
                # the same in TIME (the chip clocks down as more wavefronts issue: a cycle count priced at another occupancy's
                # clock over-states what co-residency buys): nanoseconds one SIMD spent per wave-instruction in this run against
                # the micro-benchmarked cost of this opcode mix at 1 / 2 / 4 / 8 wavefronts per SIMD
                ns_mix = None
                if all("ns_per_inst_if_waves_per_simd" in k for k in kc.values()):
                    ns_mix = {w: sum(k["valu_insts"] * k["ns_per_inst_if_waves_per_simd"][w] for k in kc.values()) / sum(k["valu_insts"] for k in kc.values())
                              for w in ("1", "2", "4", "8")}
                ns_run = 1024 * (dt / args.steps / world) * 1e9 / tot_inst
                out["valu_issue"] = {
                    "frac": tot_cyc / avail, "issue_cycles_per_step": tot_cyc, "simd_cycles_available_per_step": avail,
                    "frac_of_multiwave_ceiling": tot_cyc8 / avail,
                    "frac_if_2_waves_per_simd_issue_costs": tot_cyc2 / avail,
                    "cycles_per_inst": {"as_run": tot_cyc / tot_inst, "2_waves_per_simd": tot_cyc2 / tot_inst,
                                        "8_waves_per_simd": tot_cyc8 / tot_inst},
                    "ns_per_wave_inst_per_simd": {"this_run": ns_run, "ubench_mix_by_waves_per_simd": ns_mix,
                                                  "frac_of_2_wave_cost": (ns_mix["2"] / ns_run) if ns_mix else None,
                                                  "frac_of_8_wave_cost": (ns_mix["8"] / ns_run) if ns_mix else None,
                                                  "note": "time, not cycles: 1024 SIMDs x step time / wave-instructions of the step; the "
                                                          "8-wavefront cost of the same opcode mix is the hardware's ceiling, the 2-wavefront "
                                                          "cost what fitting 256 registers could buy at best -- measured: nothing "
                                                          "(profiles/r04_d_ab_split_pairing_two_waves.log)"},
                    "valu_wave_insts_per_step": tot_inst, "achieved_ginstr_s": tot_inst * args.steps * world / dt / 1e9 / world,
                    "clock_ghz_under_load": clk, "source": counters_file,
                    "counters_from_this_library": counters_current,
                    "counters_note": (None if counters_current else
                                      ("the counters in %s were taken from a library built from other sources (%s; this run: %s): "
                                       "instruction counts and traffic are those of THAT build -- re-run tools/run_profile.sh"
                                       % (counters_file, counters_hash or "no hash recorded", lib_hash))),
                    "note": "issue cycles = sum over kernels of SQ_INSTS_VALU x (opcode histogram of the kernel's ISA . "
                            "micro-benchmarked issue cost of every opcode class); reproducible by hand from profiles/ "
                            "(tools/valu_model.py).  gfx950 SIMDs are 32 lanes wide: a wave64 VALU instruction occupies its SIMD "
                            "for 2 cycles (simple VOP2) to ~4 (v_mad_u64_u32, 61 percent of these kernels), but ONE wavefront alone "
                            "issues at most one instruction per 4 - 5 cycles (MI355X_MICROARCH.md:54,473,489; "
                            "profiles/*_ubench_valu_int.csv).  frac = share of the SIMD cycles the step needs to issue its "
                            "instructions at the costs a wavefront ALONE on its SIMD pays -- what the big kernels run at (pairing 420, the "
                            "single multiplication 354 registers: one wavefront per SIMD; T1's chain, capped at 256, and the fixed-base "
                            "chunks, 246, both kernels of their own since round 5: two, priced at the two-wavefront cost).  frac_of_multiwave_ceiling prices the SAME instructions at the costs "
                            "eight co-resident wavefronts see (this instruction mix: %.2f cycles per instruction instead of %.2f): "
                            "the hardware's real ceiling for this mix, which the design cannot reach without fitting 2+ wavefronts "
                            "per SIMD (<= 256 VGPRs, no spills; DESIGN.md 5, profiles/r04_*_occupancy*)" % (
                                tot_cyc8 / tot_inst, tot_cyc / tot_inst)}
                out["kernels"] = [{"kernel": k, "ms_per_launch": per_step.get(k), "valu_wave_insts": v["valu_insts"] * (n / 4096.0),
                                   "cycles_per_inst": v["cycles_per_inst"],
                                   "hbm_bytes": (2 * v["FETCH_SIZE_KiB"] + v["WRITE_SIZE_KiB"]) * 1024 * (n / 4096.0),
                                   "hbm_traffic_over_algorithmic": (2 * v["FETCH_SIZE_KiB"] + v["WRITE_SIZE_KiB"]) * 1024 / (ALG_BYTES_PER_PROOF_VERIFY * 4096.0),
                                   "own_isa_histogram": v.get("own_isa_histogram")}
                                  for k, v in kc.items()]
        if extras is not None:
            out["other_ops"] = extras
            w16 = ((extras.get("bls12_381") or {}).get("host_inclusive_by_window_bits") or {}).get("16")
            if w16 and args.window_bits != 16:
                out["value_at_library_default_window_bits"] = {
                    "window_bits": 16, "proof_verify_per_s": w16["proof_verify_per_s"], "table_bytes": w16["table_bytes"],
                    "note": "the same submit loop on the same batches with the tables the library picks by itself (64 steps, untimed leg)"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
