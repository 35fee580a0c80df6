"""Batch verification with 4096-item resident jobs, PACED: a job is (re)submitted at most every `interval` ms, so that jobs alive
are spread over their phases (MSM kernel: chip-filling; tail: a few wavefronts) instead of running in lockstep -- a loop that
resubmits as soon as a job retires keeps the convoys it started with (all MSM kernels together, then all tails together with
the chip idle: profiles/r04_j_bv_trace_12_in_flight.txt).  usage (GPU box): python tools/quick_bv_paced.py N interval_ms [interval_ms ...]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import parity_cases as pc

N = int(sys.argv[1]) if len(sys.argv) > 1 else 12
intervals = [float(x) for x in sys.argv[2:]] or [0.0, 1.0, 1.2]
n = 4096
suite, eng, gens, sk, msgs, disclosed, rnds = pc.bench_workload("bls12_381", n, 32, 8, None, 16)
sigs, st = eng.core_sign_batch(msgs)
proofs, st = eng.core_proof_gen_batch(sigs, msgs, disclosed, rnds)
dm = [m[:8] for m in msgs]
eng.set_latency_mode(False)
eng.set_batch_verification(True)
jobs = [eng.core_proof_verify_upload(proofs, dm, disclosed) for _ in range(N)]
for j in jobs:
    j.run()
for j in jobs:
    j.wait()
    assert (j.status() == 1).all()
poll = eng.lib.bbs_job_poll
for interval in intervals:
    for rep in range(2):
        free, busy, done = list(jobs), [], 0
        t0 = time.perf_counter()
        nxt = t0
        T = 0.6
        while True:
            now = time.perf_counter()
            if busy:
                still = []
                for j in busy:
                    if poll(j.h) == 1:
                        free.append(j); done += 1
                    else:
                        still.append(j)
                busy = still
            if now - t0 > T:
                break
            if free and now >= nxt:
                j = free.pop(0)
                j.run()
                busy.append(j)
                nxt = max(nxt + interval * 1e-3, now) if interval > 0 else now
        t1 = time.perf_counter()
        for j in busy:
            j.wait()
        print("N=%2d interval %.2f ms: %8.0f proof_verify/s (%d jobs in %.3f s)" % (N, interval, done * n / (t1 - t0), done, t1 - t0), flush=True)
for j in jobs:
    assert (j.status() == 1).all()
    j.free()
eng.close()
