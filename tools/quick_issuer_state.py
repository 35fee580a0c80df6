"""Does the issuer's rate depend on what the process did before?  (development aid behind bench_extras' issuer legs.)
Measures the two-length list with 4 lists in flight (tools/quick_issuer.py's leg) in a fresh process, then again after the
process has had `churn` proof_verify jobs alive at once (each with its three streams) and freed them.
usage: python tools/quick_issuer_state.py [churn_jobs]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bbs_sign_amd import workload as pc
from bbs_sign_amd import Issuer, Job, api as _api

churn = int(sys.argv[1]) if len(sys.argv) > 1 else 32
n = 4096
items = {}
for L, r in ((32, 8), (16, 4)):
    s, e, _, _ = pc.bench_engine("bls12_381", L, None, 16)
    m, d, rn = pc.bench_items(s, e, n // 2, L, r, 0)
    sg, st = e.core_sign_batch(m)
    pf, st = e.core_proof_gen_batch(sg, m, d, rn)
    raw = [[pc.expand_message(b"bbs-bench-msg" + pc.i2osp(b, 8) + pc.i2osp(j, 8), b"BBS_BENCH_MSG_DST_", 32) for j in range(r)] for b in range(n // 2)]
    items[L] = ([_api.proof_to_octets("bls12_381", p_) for p_ in pf], raw, d)
    if L == 32:
        keep = (s, e, pf, [x[:r] for x in m], d)
    else:
        e.close()
suite, eng, proofs, dm, disclosed = keep
iss = Issuer("bls12_381", suite.api_id, window_bits=16)
iss.set_public_key(eng.public_key())
mix = [[x for pair in zip(items[32][k], items[16][k]) for x in pair] for k in range(3)]
n_i, keep_i, args_i = iss.pack_proof_verify(mix[0], mix[1], mix[2])
assert (iss.proof_verify_packed(n_i, args_i) == 1).all()


def leg(tag, k=4):
    pend = []
    for phase in (0, 1):
        t1 = time.perf_counter()
        for _ in range(32):
            if len(pend) >= k:
                j = pend.pop(0); j.wait(); assert (j.result == 1).all(); j.free()
            pend.append(iss.proof_verify_submit_packed(n_i, args_i))
        while pend:
            j = pend.pop(0); j.wait(); assert (j.result == 1).all(); j.free()
    print("%-44s %8.0f proof_verify/s" % (tag, 32 * n / (time.perf_counter() - t1)), flush=True)


leg("fresh process")
leg("fresh process (again)")
js = [eng.core_proof_verify_upload(proofs[:n // 2], dm[:n // 2], disclosed[:n // 2]) for _ in range(churn)]
for j in js:
    j.run()
for j in js:
    j.wait()
    j.free()
leg("after %d jobs alive at once and freed" % churn)
eng.set_batch_verification(True)
js = [eng.core_proof_verify_upload(proofs[:n // 2], dm[:n // 2], disclosed[:n // 2]) for _ in range(churn)]
Job.run_many_timed(js, 2 * churn)
for j in js:
    j.free()
eng.set_batch_verification(False)
leg("after %d batch-verification jobs as well" % churn)
iss.close()
eng.close()
