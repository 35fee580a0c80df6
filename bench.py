#!/usr/bin/env python3
"""Headline benchmark: BBS+ proof_verify/s, BLS12-381, 32 messages / 8 disclosed, batch 4096 per GPU.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A step = one pass of core_proof_verify (src/proof_verify.rs:64-116) over one device-resident batch of
4096 proofs (BASELINE.json configs[3]; the proofs are produced untimed by the engine's own
sign -> proof_gen).  Items are independent, so N GPUs each verify their own 4096-item batch per step
(weak scaling, no data-path collective); rank 0 gathers one pass-count per rank over RCCL.

Each rank keeps `--inflight` device-resident batches (default 8), every batch on its own HIP stream
pair: step k runs on batch k % inflight, so consecutive steps overlap on the GPU (a 4096-item batch
alone is ~10^3 wavefronts on a chip that holds 4096).  `--inflight 1` gives the one-batch-at-a-time
number, also reported in the JSON as `single_batch`.

Timing: barrier + synchronize, K steps enqueued with HIP events around every stage (recorded on the
stream the stage runs on, read after one synchronisation), barrier + synchronize; wall time = max
over ranks.  Prints ONE JSON line (rank 0).
"""
import os

# The engine keeps many independent batches in flight, one HIP stream pair per batch.  The HIP runtime maps all
# streams of a process onto 4 hardware queues by default, so a long, narrow kernel (a batch's tail) blocks the
# streams that share its queue; 16 queues let the batches overlap (measured: 645k -> 780k proof_verify/s).
# Must be set before the first HIP call of the process (torch initialises HIP before the engine is loaded).
os.environ.setdefault("GPU_MAX_HW_QUEUES", "14")
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

# algorithmic bytes per BLS12-381 proof_verify (L=32, R=8): 1040 proof octets
# (3 x 48 + 28 x 32) + 256 (8 disclosed scalars) + 64 (8 indexes) + 1 status
ALG_BYTES_PER_PROOF_VERIFY = 1361
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
# VALU issue ceiling: 1024 SIMDs x one wave-instruction per 4 cycles at 2.4 GHz; every instruction of
# this integer mix (v_mad_u64_u32 included) issues at that rate (tools/ubench/valu_int.hip, 0.5 G/s/SIMD)
VALU_PEAK_GINSTR = 1024 * 2.4 / 4
# VALU wave-instructions per 4096-item launch, from rocprofv3 SQ_INSTS_VALU (profiles/r01_l_pmc.csv; measured at the
# default --window-bits 20: other widths change the number of fixed-base additions and so pv_msm_parts' count)
# FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB (checked on PvScalars: 5984 KiB written
# for 4096 x 36 scalars); FETCH_SIZE doubled per MI355X_MICROARCH.md (128-B requests tallied at 64 B)
PMC = {"pairing_6lane": {"valu_insts": 7.502e8, "fetch_bytes": 2 * 1.9274e4 * 1024, "write_bytes": 9.578e4 * 1024},
       "pv_msm_parts": {"valu_insts": 4.052e8, "fetch_bytes": 2 * 2.5101e5 * 1024, "write_bytes": 7.38e4 * 1024},
       "pv_challenge": {"valu_insts": 1.126e7, "fetch_bytes": 2 * 6057 * 1024, "write_bytes": 6075 * 1024}}


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


def cpu_baseline(items_per_core=48):
    """The oracle's plain-C restatement (oracle/c/bbs_oracle.c: reference operation order, per-call
    domain, 38 independent double-and-add scalar multiplications, two full pairings) timed on the host
    cores on a bounded sample of the same workload (items 0.. of the bench batch)."""
    import concurrent.futures as cf
    from oracle import bbs, c_port
    from oracle.hashing import expand_message, i2osp
    suite = bbs.BLS_SUITE
    L, R = 32, 8
    api_id = suite.api_id
    cores = host_cores()
    n_items = items_per_core * cores
    sk = bbs.key_gen(suite, bytes([1] * 32), b"", b"BBS-SIG-KEYGEN-SALT-")
    pk = c_port.sk_to_pk(sk)
    gens = bbs.create_generators(suite, L + 1, api_id)

    def make(b):
        raw = [expand_message(b"bbs-bench-msg" + i2osp(b, 8) + i2osp(j, 8), b"BBS_BENCH_MSG_DST_", 32) for j in range(L)]
        msgs = bbs.msg_to_scalars(suite, raw, api_id)
        rnd = bbs.seeded_random_scalars(suite, b"bbs-bench-rnd" + i2osp(b, 8), api_id + b"MOCK_RANDOM_SCALARS_DST_", 5 + L - R)
        sig = c_port.core_sign(sk, gens, b"", msgs, api_id)
        proof = c_port.core_proof_gen(pk, sig, b"", gens, b"", msgs, list(range(R)), api_id, rnd)
        return proof, msgs[:R]

    with cf.ThreadPoolExecutor(max_workers=cores) as ex:          # ctypes releases the GIL
        items = list(ex.map(make, range(n_items)))
        t0 = time.perf_counter()
        ok = list(ex.map(lambda it: c_port.core_proof_verify(pk, it[0], gens, b"", b"", it[1], list(range(R)), api_id), items))
        dt = time.perf_counter() - t0
    assert all(ok)
    return {"value": n_items / dt, "unit": "proof_verify/s", "cores": cores, "kind": "port",
            "sample": "%d items of the bench batch (BLS12-381, L=32, R=8), core_proof_verify with caller-supplied "
                      "generators, plain-C port of the reference path (oracle/c, gcc -O3; NOT arkworks), one thread per "
                      "host core, %.1f s wall" % (n_items, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=128)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--window-bits", type=int, default=20)
    ap.add_argument("--inflight", type=int, default=8, help="device-resident batches in flight per GPU")
    ap.add_argument("--batch-verify", action="store_true", help="time the opt-in batch-verification mode (one combined "
                    "pairing check per batch, per-item fallback) instead of the default per-item pairing products")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the untimed sign/verify/proof_gen/BN254 figures")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse "
                    "the N > 1 path on a box with fewer GPUs than ranks)")
    ap.add_argument("--all-ranks-on-device", type=int, default=None, help="rehearsal only: every rank uses this GPU")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.all_ranks_on_device is not None:
            local_rank = args.all_ranks_on_device
        torch.cuda.set_device(local_rank)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend)
    assert world == args.gpus, "launch with --nproc-per-node equal to --gpus"
    torch.cuda.set_device(local_rank)
    red_dev = "cuda" if (dist is None or args.backend == "nccl") else "cpu"

    import parity_cases as pc
    from bbs_sign_amd import Engine  # noqa: F401  (fails loudly if the HIP library is missing)

    n, L, R = args.batch, 32, 8
    # every rank verifies its own batch: item ids offset by rank so the batches differ
    suite, eng, gens, sk, msgs, disclosed, rnds = pc.bench_workload("bls12_381", n, L, R, None, args.window_bits,
                                                                    device=local_rank)
    sigs, st = eng.core_sign_batch(msgs)
    assert (st == 1).all()
    proofs, st = eng.core_proof_gen_batch(sigs, msgs, disclosed, rnds)
    assert (st == 1).all()
    dm = [m[:R] for m in msgs]
    from bbs_sign_amd import Job
    if args.batch_verify:
        eng.set_batch_verification(True)
    jobs = [eng.core_proof_verify_upload(proofs, dm, disclosed) for _ in range(max(1, args.inflight))]   # resident in HBM
    job = jobs[0]

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for k in range(max(args.warmup, len(jobs))):
        jobs[k % len(jobs)].run()
    for j in jobs:
        j.wait()
        assert (j.status() == 1).all(), "warm-up batch did not verify"

    barrier()
    t0 = time.perf_counter()
    total_ms, stage_ms = Job.run_many_timed(jobs, args.steps)
    barrier()
    dt = time.perf_counter() - t0
    passed = n
    for j in jobs:
        passed = min(passed, int((j.status() == 1).sum()))
    assert passed == n, "timed batch did not verify"
    # one batch at a time (latency form), outside the timed region
    single_ms, single_stage = job.run_timed(3, per_stage=True)

    extras = None
    if rank == 0 and world == 1 and not args.no_extras:
        # the other three operations of the path and the second curve, one batch at a time (BASELINE configs[1..3])
        def rate(j, reps=3):
            j.run(); j.wait()
            ms, _ = j.run_timed(reps, per_stage=False)
            return n / (ms / reps * 1e-3)
        def rate8(make, k=8, steps=32):            # k device-resident batches in flight, like the headline
            js = [make() for _ in range(k)]
            for j in js:
                j.run()
            for j in js:
                j.wait()
            ms, _ = Job.run_many_timed(js, steps)
            for j in js:
                j.free()
            return n * steps / (ms * 1e-3)
        extras = {"unit": "items/s, one 4096-item batch at a time; *_8_in_flight: eight resident batches in flight",
                  "bls12_381": {"sign": rate(eng.core_sign_upload(msgs)), "verify": rate(eng.core_verify_upload(sigs, msgs)),
                                "proof_gen": rate(eng.core_proof_gen_upload(sigs, msgs, disclosed, rnds)),
                                "sign_8_in_flight": rate8(lambda: eng.core_sign_upload(msgs)),
                                "verify_8_in_flight": rate8(lambda: eng.core_verify_upload(sigs, msgs)),
                                "proof_gen_8_in_flight": rate8(lambda: eng.core_proof_gen_upload(sigs, msgs, disclosed, rnds))}}
        if not args.batch_verify:
            # opt-in batch verification: same batch, same booleans, one combined pairing check per batch
            eng.set_batch_verification(True)
            bj = [eng.core_proof_verify_upload(proofs, dm, disclosed) for _ in range(32)]
            eng.set_batch_verification(False)
            for j in bj:
                j.run()
            for j in bj:
                j.wait()
                assert (j.status() == 1).all()
            bms, bstage = Job.run_many_timed(bj, 96)
            b1, b1stage = bj[0].run_timed(3, per_stage=True)
            extras["bls12_381"]["proof_verify_batch_verification"] = {
                "proof_verify_per_s": n * 96 / (bms * 1e-3), "batches_in_flight": len(bj),
                "single_batch_ms": b1 / 3, "stage_ms_single_batch": {k: v / 3 for k, v in b1stage.items()},
                "note": "bbs_ctx_set_batch_verification: random-linear-combination check over the batch (bucket-method "
                        "MSM of 2 x 4096 points, 128-bit coefficients) + per-item fallback; not the headline"}
            for j in bj:
                j.free()
            # the combined check's tail (one narrow pairing, ~12 ms) is per job: with 4096-item jobs the rate is bound by
            # the number of hardware queues; four times the items per job amortise it (same proofs repeated: the work does
            # not depend on the data)
            eng.set_batch_verification(True)
            big = [eng.core_proof_verify_upload(proofs * 4, dm * 4, disclosed * 4) for _ in range(12)]
            eng.set_batch_verification(False)
            for j in big:
                j.run()
            for j in big:
                j.wait()
                assert (j.status() == 1).all()
            gms, _ = Job.run_many_timed(big, 48)
            extras["bls12_381"]["proof_verify_batch_verification"]["batch_16384_per_s"] = 4 * n * 48 / (gms * 1e-3)
            for j in big:
                j.free()
            # core_verify in the same mode
            eng.set_batch_verification(True)
            vj = [eng.core_verify_upload(sigs, msgs) for _ in range(32)]
            eng.set_batch_verification(False)
            for j in vj:
                j.run()
            for j in vj:
                j.wait()
                assert (j.status() == 1).all()
            vms, _ = Job.run_many_timed(vj, 96)
            extras["bls12_381"]["verify_batch_verification"] = {"verify_per_s": n * 96 / (vms * 1e-3), "batches_in_flight": len(vj)}
            for j in vj:
                j.free()
            # opt-in subgroup vouching (bbs_ctx_set_points_in_subgroup: GLV split of the variable-base terms), alone and
            # together with batch verification; same proofs, same booleans -- not the headline either
            def checked(make, k):
                js = [make() for _ in range(k)]
                for j in js:
                    j.run()
                for j in js:
                    j.wait()
                    assert (j.status() == 1).all()
                return js
            # (one job set alive at a time: created, checked, timed, freed)
            def measured(make, k, steps, single=False):
                js = checked(make, k)
                one = js[0].run_timed(3, per_stage=True) if single else None
                Job.run_many_timed(js, 8)                  # untimed rounds, as the headline's warm-up
                ms, _ = Job.run_many_timed(js, steps)
                for j in js:
                    j.free()
                return ms, one
            eng.set_points_in_subgroup(True)
            gms8, (g1ms, g1stage) = measured(lambda: eng.core_proof_verify_upload(proofs, dm, disclosed), 8, 32, single=True)
            gvms8, _ = measured(lambda: eng.core_verify_upload(sigs, msgs), 8, 32)
            gpms8, _ = measured(lambda: eng.core_proof_gen_upload(sigs, msgs, disclosed, rnds), 8, 32)
            eng.set_batch_verification(True)
            gbms, _ = measured(lambda: eng.core_proof_verify_upload(proofs * 4, dm * 4, disclosed * 4), 12, 48)
            eng.set_batch_verification(False)
            eng.set_points_in_subgroup(False)
            extras["bls12_381"]["points_in_subgroup"] = {
                "proof_verify_8_in_flight": n * 32 / (gms8 * 1e-3), "verify_8_in_flight": n * 32 / (gvms8 * 1e-3),
                "proof_gen_8_in_flight": n * 32 / (gpms8 * 1e-3),
                "proof_verify_single_batch_ms": g1ms / 3, "pv_msm_parts_ms_single_batch": g1stage.get("pv_msm_parts", 0) / 3,
                "proof_verify_batch_verification_16384_per_s": 4 * n * 48 / (gbms * 1e-3),
                "note": "bbs_ctx_set_points_in_subgroup: caller vouches G1 membership (as the reference's types do); "
                        "variable-base terms use the GLV split; opt-in, not the headline"}
        # host-inclusive form: bbs_core_proof_verify_batch on host buffers = validation + packing (C++),
        # H2D of the proofs, kernels, D2H of the statuses -- never the headline, which starts from HBM-resident batches
        import ctypes
        import threading
        import numpy as np
        from bbs_sign_amd import _lib as _l
        nn, keep, cargs = eng._pv_inputs(proofs, dm, disclosed, None, None)
        def one_call():
            st = np.zeros(nn, dtype=np.int8)
            rc = eng.lib.bbs_core_proof_verify_batch(eng.h, nn, *cargs, st.ctypes.data_as(_l.c_i8p))
            assert rc == 0 and (st == 1).all()
        one_call()
        t1 = time.perf_counter()
        for _ in range(4):
            one_call()
        one_ms = (time.perf_counter() - t1) / 4 * 1e3
        nthreads, per = 8, 6
        def worker():
            for _ in range(per):
                one_call()                     # ctypes releases the GIL; every call builds its own job and streams
        th = [threading.Thread(target=worker) for _ in range(nthreads)]
        t1 = time.perf_counter()
        for t_ in th:
            t_.start()
        for t_ in th:
            t_.join()
        piped = nthreads * per * n / (time.perf_counter() - t1)
        extras["bls12_381"]["proof_verify_host_inclusive"] = {
            "one_call_ms": one_ms, "one_call_per_s": n / (one_ms * 1e-3), "threads": nthreads, "threaded_per_s": piped,
            "note": "one-shot bbs_core_proof_verify_batch from host buffers (1.36 KB/proof over PCIe + host-side validation "
                    "and SoA packing on one core per call); sequential calls, then 8 host threads each issuing calls"}
        # ingest: n proofs as octet strings -> records (3 n point decompressions + subgroup checks on the device)
        from bbs_sign_amd import api as _api
        octs = [_api.proof_to_octets("bls12_381", p_) for p_ in proofs[:n]]
        eng.proofs_from_octets_batch(octs[:64])
        flat_o, off_o = (__import__("bbs_sign_amd.engine", fromlist=["x"])._ragged_bytes(octs))
        rec_o = 6 * eng.fpb + 128
        pf_o = np.zeros(n * rec_o, dtype=np.uint8); cm_o = np.zeros(n * 32 * 32, dtype=np.uint8)
        cmo_o = np.zeros(n + 1, dtype=np.uint64); st_o = np.zeros(n, dtype=np.int8)
        def decode_call():
            rc = eng.lib.bbs_proofs_from_octets_batch(eng.h, n, flat_o.ctypes.data_as(_l.c_u8p), off_o.ctypes.data_as(_l.c_u64p),
                                                      pf_o.ctypes.data_as(_l.c_u8p), cm_o.ctypes.data_as(_l.c_u8p),
                                                      cmo_o.ctypes.data_as(_l.c_u64p), st_o.ctypes.data_as(_l.c_i8p))
            assert rc == 0 and (st_o == 1).all()
        decode_call()
        t1 = time.perf_counter()
        for _ in range(4):
            decode_call()
        extras["bls12_381"]["proofs_from_octets"] = {"proofs_per_s": 4 * n / (time.perf_counter() - t1),
            "note": "bbs_proofs_from_octets_batch: 1040-byte proof octets -> records, 3 x 4096 G1 decompressions and "
                    "subgroup checks on the device, scalars on the host, one call at a time"}
        _, eb, _, _, mb, db, rb = pc.bench_workload("bn254", n, L, R, None, 16, device=local_rank)
        sb, st = eb.core_sign_batch(mb)
        assert (st == 1).all()
        pb, st = eb.core_proof_gen_batch(sb, mb, db, rb)
        assert (st == 1).all()
        jb = eb.core_proof_verify_upload(pb, [m[:R] for m in mb], db)
        dmb = [m[:R] for m in mb]
        extras["bn254"] = {"sign": rate(eb.core_sign_upload(mb)), "verify": rate(eb.core_verify_upload(sb, mb)),
                           "proof_gen": rate(eb.core_proof_gen_upload(sb, mb, db, rb)), "proof_verify": rate(jb),
                           "proof_verify_8_in_flight": rate8(lambda: eb.core_proof_verify_upload(pb, dmb, db))}
        assert (jb.status() == 1).all()
        # BASELINE configs[4] on one GPU: its share of the mixed batch is half BN254, half BLS12-381 -- four resident
        # jobs of each curve in flight together (two contexts, every job on its own streams)
        mj = [eng.core_proof_verify_upload(proofs, dm, disclosed) for _ in range(4)] + \
             [eb.core_proof_verify_upload(pb, dmb, db) for _ in range(4)]
        for j in mj:
            j.run()
        for j in mj:
            j.wait()
            assert (j.status() == 1).all()
        Job.run_many_timed(mj, 8)
        mms, _ = Job.run_many_timed(mj, 64)                # 64 batch runs, round robin over the eight jobs
        extras["mixed_curves"] = {"proof_verify_per_s": n * 64 / (mms * 1e-3),
                                  "note": "BASELINE configs[4] per-GPU share: 4 BLS12-381 + 4 BN254 batches of %d in flight" % n}
        for j in mj:
            j.free()

    tmax = torch.tensor([dt], dtype=torch.float64, device=red_dev)
    cnt = torch.tensor([passed], dtype=torch.int64, device=red_dev)
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(cnt, op=dist.ReduceOp.SUM)       # the only exchange: one pass-count per rank
    dt = float(tmax.item())

    if rank == 0:
        assert int(cnt.item()) == n * world
        value = world * n * args.steps / dt
        dom = max(stage_ms, key=stage_ms.get)
        dom_ms = stage_ms[dom] / args.steps
        achieved = ALG_BYTES_PER_PROOF_VERIFY * n / (dom_ms * 1e-3) / 1e9
        pmc = PMC.get(dom, {})
        valu_total = sum(v["valu_insts"] for v in PMC.values()) * (n / 4096.0)
        out = {
            "metric": "BBS+ proof_verify/sec (BLS12-381, 32-msg)",
            "value": value, "unit": "proof_verify/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": "BLS12-381 core_proof_verify, batch %d per GPU, 32 msgs / 8 disclosed, empty "
                                   "header/ph, one issuer key (BASELINE configs[3], proof_verify leg)%s" % (
                                       n, "; OPT-IN batch-verification mode" if args.batch_verify else ""),
                       "batch_per_gpu": n, "messages": L, "disclosed": R, "fixed_base_window_bits": args.window_bits,
                       "batches_in_flight": len(jobs),
                       "parallelism": "independent batch per GPU, no data-path collective"},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS,
                         "traffic": (pmc["fetch_bytes"] + pmc["write_bytes"]) * (n / 4096.0) if pmc else None,
                         "note": "algorithmic bytes/unit = %d (SURVEY 8d); kernel duration = HIP events with %d batches "
                                 "in flight; traffic = FETCH_SIZE(x2, gfx950) + WRITE_SIZE per launch from "
                                 "profiles/r01_l_pmc.csv; the path is bound by integer VALU issue, not HBM "
                                 "(see valu_issue and DESIGN.md)" % (ALG_BYTES_PER_PROOF_VERIFY, len(jobs))},
            # the binding resource: VALU wave-instructions issued per second vs the chip's issue ceiling
            "valu_issue": {"achieved_ginstr_s": valu_total * world * args.steps / dt / 1e9 / world,
                           "peak_ginstr_s": VALU_PEAK_GINSTR,
                           "frac": valu_total * args.steps / dt / 1e9 / VALU_PEAK_GINSTR,
                           "valu_wave_insts_per_step": valu_total,
                           "source": "SQ_INSTS_VALU of pairing_6lane + pv_msm_parts + pv_challenge (profiles/r01_l_pmc.csv); peak = 1024 "
                                     "SIMDs x 2.4 GHz / 4 cycles per wave-instruction"},
            # per kernel: launch duration (HIP events, this run), VALU wave-instructions and HBM bytes per launch (PMC)
            "kernels": [{"kernel": k, "ms_per_launch": stage_ms[k] / args.steps,
                         "valu_wave_insts": PMC[k]["valu_insts"] * (n / 4096.0),
                         "hbm_bytes": (PMC[k]["fetch_bytes"] + PMC[k]["write_bytes"]) * (n / 4096.0)}
                        for k in PMC if k in stage_ms],
            "stage_ms_per_step": {k: v / args.steps for k, v in stage_ms.items()},
            "gpu_ms_per_step_events": total_ms / args.steps,
            "batches_in_flight": len(jobs),
            "single_batch": {"ms": single_ms / 3, "proof_verify_per_s": n / (single_ms / 3 * 1e-3),
                             "stage_ms": {k: v / 3 for k, v in single_stage.items()}},
        }
        if extras is not None:
            out["other_ops"] = extras
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
