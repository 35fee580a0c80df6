"""Queue budget under load (GPU box): K one-stream proof_verify jobs in flight (batch verification's throughput form: a job owns
ONE stream) with whatever GPU_MAX_HW_QUEUES / BBS_DEDICATED_QUEUES the environment holds.  Before round 5, pool + dedicated
queues beyond ~ 20 made the runtime ABORT the process (HSA_STATUS_ERROR_OUT_OF_RESOURCES, profiles/r04_g_*); now the library
bounds the hardware queues it touches by its scratch budget (runtime.hpp queue_budget) and jobs beyond it share streams.
Prints one JSON line: the budget, the statuses' verdict, the rate.  usage: python tools/queue_budget_probe.py [jobs] [items] [rounds]"""
import ctypes
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import parity_cases as pc          # noqa: E402
from bbs_sign_amd import Job       # noqa: E402

K = int(sys.argv[1]) if len(sys.argv) > 1 else 32
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
suite, eng, gens, sk, msgs, disclosed, rnds = pc.bench_workload("bls12_381", n, 32, 8, None, 12)
lib = eng.lib
t, p, d, s = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_size_t()
assert lib.bbs_runtime_queue_budget(0, ctypes.byref(t), ctypes.byref(p), ctypes.byref(d), ctypes.byref(s)) == 0
sigs, st = eng.core_sign_batch(msgs)
proofs, st = eng.core_proof_gen_batch(sigs, msgs, disclosed, rnds)
for i in range(0, n, 16):
    proofs[i].commitments[0] = (proofs[i].commitments[0] + 1) % suite.curve.r
dm = [m[:8] for m in msgs]
want = [0 if i % 16 == 0 else 1 for i in range(n)]
eng.set_latency_mode(False)
eng.set_batch_verification(True)
jobs = [eng.core_proof_verify_upload(proofs, dm, disclosed) for _ in range(K)]
t0 = time.perf_counter()
for _ in range(rounds):
    for j in jobs:
        j.run()
for j in jobs:
    j.wait()
dt = time.perf_counter() - t0
ok = all([int(x) for x in j.status()] == want for j in jobs)
for j in jobs:
    j.free()
eng.close()
print(json.dumps({"jobs_in_flight": K, "items_per_job": n, "rounds": rounds, "statuses_exact": ok, "proof_verify_per_s": K * rounds * n / dt,
                  "queue_budget": {"total": t.value, "pool": p.value, "dedicated_cap": d.value, "scratch_bytes_per_lane": s.value},
                  "env": {k: os.environ.get(k) for k in ("GPU_MAX_HW_QUEUES", "BBS_DEDICATED_QUEUES")}}))
