"""Batch-verification mode: resident 4096-item proof_verify jobs, k in flight (development aid).
usage (GPU box): [GPU_MAX_HW_QUEUES=q] python tools/quick_bv_sweep.py k1 k2 ..."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import parity_cases as pc
from bbs_sign_amd import Job

ks = [int(x) for x in sys.argv[1:]] or [4, 6, 7, 8]
n = 4096
suite, eng, gens, sk, msgs, disclosed, rnds = pc.bench_workload("bls12_381", n, 32, 8, None, 16)
sigs, st = eng.core_sign_batch(msgs)
proofs, st = eng.core_proof_gen_batch(sigs, msgs, disclosed, rnds)
dm = [m[:8] for m in msgs]
eng.set_latency_mode(bool(int(os.environ.get("FORM", "0"))))
print("latency form", os.environ.get("FORM", "0"))
print("hw queues", eng.lib.bbs_runtime_hw_queues(), flush=True)
for bv in ((True,) if os.environ.get("BV_ONLY") else (True, False)):
    eng.set_batch_verification(bv)
    for k in ks:
        js = [eng.core_proof_verify_upload(proofs, dm, disclosed) for _ in range(k)]
        for j in js:
            j.run()
        for j in js:
            j.wait()
            assert (j.status() == 1).all()
        Job.run_many_timed(js, k)
        ms, _ = Job.run_many_timed(js, 8 * k)
        print("batch_verification=%d  %2d in flight: %8.0f /s" % (bv, k, n * 8 * k / (ms * 1e-3)), flush=True)
        for j in js:
            j.free()
eng.close()
