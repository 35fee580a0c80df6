"""One batch at a time in both forms of a job, per-stage durations (development aid; bench.py is the contract):
proof_verify and verify, per-item pairing and batch verification, BLS12-381 batch 4096 with 16-bit windows.
usage (GPU box): python tools/quick_forms.py [n] [window_bits]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import parity_cases as pc
from bbs_sign_amd import Job

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
wb = int(sys.argv[2]) if len(sys.argv) > 2 else 16
suite, eng, gens, sk, msgs, disclosed, rnds = pc.bench_workload("bls12_381", n, 32, 8, None, wb)
sigs, st = eng.core_sign_batch(msgs)
proofs, st = eng.core_proof_gen_batch(sigs, msgs, disclosed, rnds)
assert (st == 1).all()
dm = [m[:8] for m in msgs]


def one(name, make, reps=3):
    j = make()
    j.run(); j.wait()
    assert (j.status() == 1).all(), name
    tot, stg = j.run_timed(reps, per_stage=True)
    print("%-44s %6.2f ms  %7.0f /s  %s" % (name, tot / reps, n / (tot / reps) * 1e3, {k: round(v / reps, 2) for k, v in stg.items()}), flush=True)
    j.free()


def many(name, make, k=8, steps=32):
    js = [make() for _ in range(k)]
    for j in js:
        j.run()
    for j in js:
        j.wait()
        assert (j.status() == 1).all(), name
    Job.run_many_timed(js, k)
    ms, _ = Job.run_many_timed(js, steps)
    print("%-44s %d in flight: %7.0f /s" % (name, k, js[0].n * steps / (ms * 1e-3)), flush=True)
    for j in js:
        j.free()


for bv in (False, True):
    eng.set_batch_verification(bv)
    for form in (False, True):
        eng.set_latency_mode(form)
        tag = "%s, %s form" % ("batch verification" if bv else "per-item pairing", "latency" if form else "throughput")
        one("proof_verify " + tag, lambda: eng.core_proof_verify_upload(proofs, dm, disclosed))
        one("verify       " + tag, lambda: eng.core_verify_upload(sigs, msgs))
    eng.set_latency_mode(False)
    tag = "batch verification" if bv else "per-item pairing"
    many("proof_verify " + tag, lambda: eng.core_proof_verify_upload(proofs, dm, disclosed), 8, 32)
    if bv:
        many("proof_verify " + tag, lambda: eng.core_proof_verify_upload(proofs, dm, disclosed), 16, 64)
        many("proof_verify " + tag, lambda: eng.core_proof_verify_upload(proofs, dm, disclosed), 32, 96)
        many("proof_verify 16384-item jobs, " + tag, lambda: eng.core_proof_verify_upload(proofs * 4, dm * 4, disclosed * 4), 8, 24)
    many("verify       " + tag, lambda: eng.core_verify_upload(sigs, msgs), 8, 32)
eng.close()
