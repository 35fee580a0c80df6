"""One operation of the other three (sign / verify / proof_gen) as a profiling target: a resident 4096-item BLS12-381 job
(L = 32, R = 8, 20-bit windows: BASELINE configs[1] / [2] / the proof_gen leg of [3]) in the throughput form, run a few times one
at a time.  What tools/run_profile_ops.sh wraps in rocprofv3 (--kernel-trace --pmc ...; the program directly after `--`).
usage: python3 tools/prof_ops.py sign|verify|proof_gen [runs] [window_bits]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import parity_cases as pc      # noqa: E402

op = sys.argv[1]
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 6
wb = int(sys.argv[3]) if len(sys.argv) > 3 else 20
n, L, R = 4096, 32, 8
suite, eng, gens, sk, msgs, disclosed, rnds = pc.bench_workload("bls12_381", n, L, R, None, wb)
eng.set_latency_mode(False)
sigs, st = eng.core_sign_batch(msgs)
assert (st == 1).all()
if op == "sign":
    j = eng.core_sign_upload(msgs)
elif op == "verify":
    j = eng.core_verify_upload(sigs, msgs)
elif op == "proof_gen":
    j = eng.core_proof_gen_upload(sigs, msgs, disclosed, rnds)
else:
    raise SystemExit("unknown operation " + op)
for _ in range(runs):
    j.run()
    j.wait()
assert (j.status() == 1).all()
j.free()
eng.close()
print("ok", op, runs)
