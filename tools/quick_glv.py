"""A/B of bbs_ctx_set_points_in_subgroup on the bench workload: default and vouched jobs timed alternately
(development aid; bench.py is the contract)."""
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


def jobs(glv, k=8, verify=False):
    eng.set_points_in_subgroup(glv)
    js = [(eng.core_verify_upload(sigs, msgs) if verify else eng.core_proof_verify_upload(proofs, dm, disclosed)) for _ in range(k)]
    eng.set_points_in_subgroup(False)
    for j in js:
        j.run()
    for j in js:
        j.wait()
        assert (j.status() == 1).all()
    return js


sets = {"pv default": jobs(False), "pv vouched": jobs(True), "vf default": jobs(False, verify=True), "vf vouched": jobs(True, verify=True)}
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    for name, js in sets.items():
        ms, _ = Job.run_many_timed(js, 32)
        one, stg = js[0].run_timed(3, per_stage=True)
        print("%s: 8 in flight %.0f/s ; single %.2f ms %s" % (name, n * 32 / (ms * 1e-3), one / 3,
              {k: round(v / 3, 2) for k, v in stg.items() if "msm" in k}), flush=True)
