"""Resident batches in flight, one curve, the four operations (development aid for A/B of library builds).
usage: python tools/quick_inflight.py [curve] [window_bits] [ops: comma list of pv,vf,sg,pg] [k in flight]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import parity_cases as pc
from bbs_sign_amd import Job

curve = sys.argv[1] if len(sys.argv) > 1 else "bn254"
wb = int(sys.argv[2]) if len(sys.argv) > 2 else 16
ops = (sys.argv[3] if len(sys.argv) > 3 else "pv,vf,sg,pg").split(",")
k = int(sys.argv[4]) if len(sys.argv) > 4 else 8
n = 4096
suite, eng, gens, sk, msgs, disclosed, rnds = pc.bench_workload(curve, n, 32, 8, None, wb)
sigs, st = eng.core_sign_batch(msgs)
proofs, st = eng.core_proof_gen_batch(sigs, msgs, disclosed, rnds)
dm = [m[:8] for m in msgs]
eng.set_latency_mode(False)
make = {"pv": lambda: eng.core_proof_verify_upload(proofs, dm, disclosed), "vf": lambda: eng.core_verify_upload(sigs, msgs),
        "sg": lambda: eng.core_sign_upload(msgs), "pg": lambda: eng.core_proof_gen_upload(sigs, msgs, disclosed, rnds)}
for op in ops:
    js = [make[op]() for _ in range(k)]
    for j in js:
        j.run()
    for j in js:
        j.wait()
        assert (j.status() == 1).all()
    Job.run_many_timed(js, k)
    rates = []
    for rep in range(3):
        ms, _ = Job.run_many_timed(js, 6 * k)
        rates.append(n * 6 * k / (ms * 1e-3))
    for j in js:
        j.free()
    print("%s %s %d in flight: %s /s" % (curve, op, k, " ".join("%8.0f" % r for r in rates)), flush=True)
eng.close()
