"""Does the issuer's rate depend on what the process did before?  (development aid behind bench_extras' issuer legs.)
Measures the two-length list (tools/quick_issuer.py's leg) with 2 .. 6 lists in flight while ANOTHER context of the process is
alive (as the bench's main engine is during bench_extras' issuer legs: its stream sits in front of the jobs' streams in the
creation-ordered pool, which shifts which job streams share a hardware queue) and again after that context has been closed.
usage: python tools/quick_issuer_state.py"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bbs_sign_amd import workload as pc
from bbs_sign_amd import Issuer, Job, api as _api

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
        churn_items = (m, d, rn, sg)
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


for k in (2, 3, 4, 5, 6):
    leg("another context alive, %d lists in flight" % k, k)
if os.environ.get("CHURN"):
    # what bench_extras does before its issuer legs: many resident jobs of the other operations alive at once, then freed
    msgs_, disc_, rnds_, sigs_ = churn_items
    for make, k in ((lambda: eng.core_sign_upload(msgs_), 16), (lambda: eng.core_proof_gen_upload(sigs_, msgs_, disc_, rnds_), 16),
                    (lambda: eng.core_proof_verify_upload(proofs, dm, disclosed), 32)):
        js = [make() for _ in range(k)]
        for j in js:
            j.run()
        for j in js:
            j.wait()
        Job.run_many_timed(js, 2 * k)
        for j in js:
            j.free()
        leg("after %d more resident jobs alive at once, 6 lists in flight" % k, 6)
eng.close()
for k in (2, 3, 4, 5, 6):
    leg("no other context, %d lists in flight" % k, k)
iss.close()
