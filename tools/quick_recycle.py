"""Does a job set built from recycled streams / buffers run as fast as a fresh one?  (development aid)"""
import os, sys
os.environ.setdefault("GPU_MAX_HW_QUEUES", "14")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import parity_cases as pc
from bbs_sign_amd.engine import Job

n = 4096
suite, eng, gens, sk, msgs, disclosed, rnds = pc.bench_workload("bls12_381", n, 32, 8, None, 16)
sigs, st = eng.core_sign_batch(msgs, [b""] * n)
proofs, st = eng.core_proof_gen_batch(sigs, msgs, disclosed, rnds, [b""] * n, [b""] * n)
dm = [m[:8] for m in msgs]


def measure(tag, k=8, free=True):
    js = [eng.core_proof_verify_upload(proofs, dm, disclosed) for _ in range(k)]
    for j in js:
        j.run()
    for j in js:
        j.wait()
    ms, _ = Job.run_many_timed(js, 32)
    print("%s: %.0f/s" % (tag, n * 32 / (ms * 1e-3)), flush=True)
    if free:
        for j in js:
            j.free()
    return js


measure("fresh")
measure("recycled once")
eng.set_batch_verification(True)
bj = [eng.core_proof_verify_upload(proofs, dm, disclosed) for _ in range(32)]
eng.set_batch_verification(False)
for j in bj:
    j.run()
for j in bj:
    j.wait()
for j in bj:
    j.free()
measure("after 32 batch-verification jobs freed")
keep = measure("kept alive", free=False)
measure("while 8 others are resident")
for j in reversed(keep):
    j.free()
measure("after reversed frees")

eng.set_batch_verification(True)
big = [eng.core_proof_verify_upload(proofs * 4, dm * 4, disclosed * 4) for _ in range(12)]
eng.set_batch_verification(False)
for j in big:
    j.run()
for j in big:
    j.wait()
for j in big:
    j.free()
measure("first set after 12 big batch-verification jobs freed")
measure("second set")
measure("third set")

# the sequence of bench.py's extras: 32 verify jobs in batch-verification mode, timed, freed
eng.set_batch_verification(True)
vj = [eng.core_verify_upload(sigs, msgs) for _ in range(32)]
eng.set_batch_verification(False)
for j in vj:
    j.run()
for j in vj:
    j.wait()
ms, _ = Job.run_many_timed(vj, 96)
print("verify, batch verification, 32 in flight: %.0f/s" % (n * 96 / (ms * 1e-3)), flush=True)
for j in vj:
    j.free()
measure("first set after the verify jobs")
measure("second set")
measure("third set")
