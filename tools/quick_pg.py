"""proof_gen, resident 4096-item jobs, k in flight, in both layouts of the variable-base parts (development aid):
latency mode 1 = one lane per multiplication (seven chains per item), 0 = Bbar and T1 as joint chains (five).
usage (GPU box): python tools/quick_pg.py k1 k2 ..."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import parity_cases as pc
from bbs_sign_amd import Job

ks = [int(x) for x in sys.argv[1:]] or [8, 12, 16]
n = 4096
curve = os.environ.get("CURVE", "bls12_381")
suite, eng, gens, sk, msgs, disclosed, rnds = pc.bench_workload(curve, n, 32, 8, None, 16)
if os.environ.get("VOUCH"):
    eng.set_points_in_subgroup(True)          # BLS12-381: the GLV split (BN254 has it always)
print("curve", curve, "vouched", bool(os.environ.get("VOUCH")), "BBS_PG_COMB", os.environ.get("BBS_PG_COMB", "1"), flush=True)
sigs, st = eng.core_sign_batch(msgs)
for form, name in ((False, "throughput form"),):
    eng.set_latency_mode(form)
    for k in ks:
        js = [eng.core_proof_gen_upload(sigs, msgs, disclosed, rnds) for _ in range(k)]
        for j in js:
            j.run()
        for j in js:
            j.wait()
            assert (j.status() == 1).all()
        Job.run_many_timed(js, k)
        ms, stg = Job.run_many_timed(js, 6 * k)
        print("%-18s %2d in flight: %8.0f proof_gen/s" % (name, k, n * 6 * k / (ms * 1e-3)), flush=True)
        for j in js:
            j.free()
eng.close()
