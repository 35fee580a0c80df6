"""bbs_issuer: lists of 4096 proofs of two lengths (half 32 messages / 8 disclosed, half 16 / 4), k lists in flight (development
aid; the same leg as bench_extras' issuer_proof_verify_two_lengths_*).  usage: python tools/quick_issuer.py [lists_in_flight...]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bbs_sign_amd import workload as pc
from bbs_sign_amd import Issuer, api as _api

ks = [int(x) for x in sys.argv[1:]] or [1, 2, 4]
n, R = 4096, 8
items = {}
for L, r in ((32, 8), (16, 4)):
    s, e, _, _ = pc.bench_engine("bls12_381", L, None, 16)
    m, d, rn = pc.bench_items(s, e, n // 2, L, r, 0)
    sg, st = e.core_sign_batch(m)
    pf, st = e.core_proof_gen_batch(sg, m, d, rn)
    assert (st == 1).all()
    raw = [[pc.expand_message(b"bbs-bench-msg" + pc.i2osp(b, 8) + pc.i2osp(j, 8), b"BBS_BENCH_MSG_DST_", 32) for j in range(r)] for b in range(n // 2)]
    items[L] = ([_api.proof_to_octets("bls12_381", p_) for p_ in pf], raw, d)
    pk = e.public_key()
    suite = s
    e.close()
iss = Issuer("bls12_381", suite.api_id, window_bits=16)
iss.set_public_key(pk)
mix = [[x for pair in zip(items[32][k], items[16][k]) for x in pair] for k in range(3)]
n_i, keep_i, args_i = iss.pack_proof_verify(mix[0], mix[1], mix[2])
assert (iss.proof_verify_packed(n_i, args_i) == 1).all()
for k in ks:
    pend = []
    for phase in (0, 1):
        t1 = time.perf_counter()
        for _ in range(32):
            if len(pend) >= k:
                j = pend.pop(0); j.wait(); assert (j.result == 1).all(); j.free()
            pend.append(iss.proof_verify_submit_packed(n_i, args_i))
        while pend:
            j = pend.pop(0); j.wait(); assert (j.result == 1).all(); j.free()
    print("BBS_PV_MSM_LAYOUT=%s issuer two lengths, %d lists in flight: %8.0f proof_verify/s" % (os.environ.get("BBS_PV_MSM_LAYOUT"), k, 32 * n / (time.perf_counter() - t1)), flush=True)
iss.close()
