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


big = None
if os.environ.get("BIG"):                        # a third context with 20-bit tables (29.7 GB), as the bench's main engine
    big = pc.bench_engine("bls12_381", 32, None, 20)[1]
    leg("a 20-bit context alive as well, 6 lists in flight", 6)
for k in ((6,) if os.environ.get("BIG") else (2, 3, 4, 5, 6)):
    leg("another context alive, %d lists in flight" % k, k)
if os.environ.get("WIRE"):                       # bench_extras' wire legs run right before its issuer legs: 8 wire jobs in flight
    oct32, raw32, d32 = items[32]
    pend = []
    for _ in range(32):
        if len(pend) >= 8:
            j = pend.pop(0); j.wait(); j.free()
        pend.append(eng.proof_verify_wire_submit(oct32, raw32, d32))
    while pend:
        j = pend.pop(0); j.wait(); j.free()
    leg("after 32 wire jobs (8 in flight), 6 lists in flight", 6)
if os.environ.get("CHURN"):
    # what bench_extras does before its issuer legs: many resident jobs of the other operations alive at once, then freed.
    # CHURN = comma list of kind:jobs, kind in sg, pg, pv, vf, pvbv, vfbv (bv = batch verification on)
    msgs_, disc_, rnds_, sigs_ = churn_items
    eng.set_latency_mode(False)
    makers = {"sg": lambda: eng.core_sign_upload(msgs_), "pg": lambda: eng.core_proof_gen_upload(sigs_, msgs_, disc_, rnds_),
              "pv": lambda: eng.core_proof_verify_upload(proofs, dm, disclosed), "vf": lambda: eng.core_verify_upload(sigs_, msgs_)}
    for spec in os.environ["CHURN"].split(","):
        kind, k = spec.split(":")
        k = int(k)
        bv = kind.endswith("bv")
        eng.set_batch_verification(bv)
        js = [makers[kind[:2]]() for _ in range(k)]
        for j in js:
            j.run()
        for j in js:
            j.wait()
        Job.run_many_timed(js, 2 * k)
        for j in js:
            j.free()
        eng.set_batch_verification(False)
        leg("after %d %s jobs alive at once, 6 lists in flight" % (k, kind), 6)
eng.close()
for k in (2, 3, 4, 5, 6):
    leg("no other context, %d lists in flight" % k, k)
iss.close()
