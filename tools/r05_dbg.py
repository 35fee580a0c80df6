"""debug aid (round 5): the failing step of check_batch_verification with the statuses printed, per job form"""
import os, random, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import parity_cases as pc
from parity_cases import *
curve, n, L, seed, window_bits = "bls12_381", 9, 4, 5, 8
rng = random.Random(seed)
suite = bbs.SUITES[curve]; c = suite.curve; api_id = suite.api_id
gens = gens_for(suite, L + 1)
sk = rng.randrange(1, c.r)
exact = make_engine(curve, gens, api_id, None, sk=sk, window_bits=window_bits)
batch = make_engine(curve, gens, api_id, None, sk=sk, window_bits=window_bits)
batch.set_batch_verification(True, bytes(rng.randrange(256) for _ in range(32)))
msgs = [[rng.randrange(c.r) for _ in range(L)] for _ in range(n)]
headers = [bytes(rng.randrange(256) for _ in range(rng.choice([0, 3, 40]))) for _ in range(n)]
phs = [bytes(rng.randrange(256) for _ in range(rng.choice([0, 9]))) for _ in range(n)]
disclosed = [sorted(rng.sample(range(L), rng.randrange(0, L + 1))) for _ in range(n)]
rnds = [[rng.randrange(1, c.r) for _ in range(5 + L - len(d))] for d in disclosed]
sigs, st = exact.core_sign_batch(msgs, headers)
proofs, st = exact.core_proof_gen_batch(sigs, msgs, disclosed, rnds, headers, phs)
dm = [[msgs[i][j] for j in disclosed[i]] for i in range(n)]
forged = [Signature(s.a, s.e) for s in sigs]
forged[2] = Signature(c.g1_add(sigs[2].a, c.g1), sigs[2].e)
forged[7] = Signature(c.g1_mul(sigs[7].a, 2), sigs[7].e)
fp, st = exact.core_proof_gen_batch(forged, msgs, disclosed, rnds, headers, phs)
for ident in (False, True):
    if ident:
        fp[3].a_bar = None
    for mode in (2, 0, 1):
        exact.set_latency_mode(mode); batch.set_latency_mode(mode)
        want = list(exact.core_proof_verify_batch(fp, dm, disclosed, headers, phs))
        got = list(batch.core_proof_verify_batch(fp, dm, disclosed, headers, phs))
        j = batch.core_proof_verify_upload(fp, dm, disclosed, headers, phs)
        print("identity", ident, "latency_mode", mode, "want", want, "got", got, "OK" if want == got else "DIFF", [(batch.lib.bbs_job_stage_name(j.h, k) or b"").decode() for k in range(14) if batch.lib.bbs_job_stage_name(j.h, k)], flush=True)
        j.free()
