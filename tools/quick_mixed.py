"""Ad-hoc (GPU box): BASELINE config 5 on one GPU -- BN254 and BLS12-381 proof_verify batches in flight together."""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "14")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_cases as pc
from bbs_sign_amd import Job
n = 4096
jobs = []
for curve, wb in (("bls12_381", 16), ("bn254", 16)):
    suite, eng, gens, sk, msgs, disclosed, rnds = pc.bench_workload(curve, n, 32, 8, None, wb)
    sigs, st = eng.core_sign_batch(msgs); assert (st == 1).all()
    proofs, st = eng.core_proof_gen_batch(sigs, msgs, disclosed, rnds); assert (st == 1).all()
    for i in range(0, n, 16):
        proofs[i].commitments[0] = (proofs[i].commitments[0] + 1) % suite.curve.r
    dm = [m[:8] for m in msgs]
    jobs += [eng.core_proof_verify_upload(proofs, dm, disclosed) for _ in range(4)]
order = [jobs[k] for pair in zip(jobs[:4], jobs[4:]) for k in ()] or [j for pair in zip(jobs[:4], jobs[4:]) for j in pair]
for j in order: j.run()
for j in order: j.wait()
ms, _ = Job.run_many_timed(order, 64)
want = [0 if i % 16 == 0 else 1 for i in range(n)]
for j in order:
    assert [int(x) for x in j.status()] == want
print("mixed BN254 + BLS12-381, 4 + 4 batches in flight: %.0f proof_verify/s, statuses exact" % (n * 64 / (ms * 1e-3)))
