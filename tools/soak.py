"""Ad-hoc (GPU box): the serving loop for a while -- submit / wait / free with 8 batches in flight over distinct 4096-item
batches, all four operations and the wire path mixed in -- device memory, host RSS and the statuses stay what they were.
usage: python tools/soak.py [seconds]"""
import os, sys, time, resource
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import parity_cases as pc
from bbs_sign_amd import api
import bench

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 45.0
torch.cuda.mem_get_info()          # torch's HIP context first (initialising it after the engine has been using the device has failed on this pool)
n, L, R = 4096, 32, 8
suite, eng, gens, sk = pc.bench_engine("bls12_381", L, None, 16)
slots, raw0 = bench.make_slots(pc, suite, eng, n, L, R, 4, first_item=0)
msgs, disclosed, rnds, sigs, proofs, dm = raw0
octs = [api.proof_to_octets("bls12_381", p) for p in proofs]
no, keep_o, args_o = eng._oct_inputs(octs, dm, disclosed, None, None)


raw512 = [[bytes([b & 255, j, 7]) * (1 + (b + j) % 40) for j in range(L)] for b in range(512)]
from bbs_sign_amd import Issuer
iss = Issuer("bls12_381", suite.api_id, window_bits=8)
iss.set_secret_key(sk)
iraw = [[bytes([b & 255, j, 9]) * (1 + j) for j in range(b % 5)] for b in range(600)]          # 0 .. 4 messages per item


def snap():
    free, total = torch.cuda.mem_get_info()
    return (total - free) / 2**20, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024


t0 = time.time()
rounds = 0
base = None
while time.time() - t0 < secs:
    bad, _, _ = bench.submit_loop(eng, slots, 64, 8)
    assert bad == 0
    pend = [eng.proof_verify_octets_submit_packed(no, args_o) for _ in range(4)]
    for j in pend:
        j.wait(); assert (j.result == 1).all(); j.free()
    vj = [eng.core_verify_submit(sigs, msgs) for _ in range(2)]
    for j in vj:
        j.wait(); assert (j.result == 1).all(); j.free()
    s2, st = eng.core_sign_batch(msgs[:512]); assert (st == 1).all()
    p2, st = eng.core_proof_gen_batch(sigs[:512], msgs[:512], disclosed[:512], rnds[:512]); assert (st == 1).all()
    # the public one-call forms (raw messages hashed on the device, octet strings on both sides)
    so, st = eng.sign_wire_batch(raw512); assert (st == 1).all()
    assert (eng.verify_wire_batch(so, raw512) == 1).all()
    po, st = eng.proof_gen_wire_batch(so, raw512, disclosed[:512], rnds[:512]); assert (st == 1).all()
    assert (eng.proof_verify_wire_batch(po, [m[:R] for m in raw512], disclosed[:512]) == 1).all()
    # batch-verification mode and both forms of a job on the same context, and the multi-length issuer
    eng.set_batch_verification(True)
    for form in (False, True):
        eng.set_latency_mode(form)
        bj = [eng.core_proof_verify_submit(proofs[:1024], dm[:1024], disclosed[:1024]) for _ in range(3)]
        for j in bj:
            j.wait(); assert (j.result == 1).all(); j.free()
    eng.set_batch_verification(False)
    eng.set_latency_mode("auto")
    iso, ist = iss.sign(iraw)
    assert (ist == 1).all()
    assert (iss.verify(iso, iraw) == 1).all()
    rounds += 1
    if rounds in (2, 4) or rounds % 10 == 0:
        d, h = snap()
        if rounds == 4:
            base = (d, h)
        print("round %3d  %5.0f s: device used %.0f MiB, host max RSS %.0f MiB" % (rounds, time.time() - t0, d, h), flush=True)
d, h = snap()
print("done: %d rounds (%d batches of %d proofs verified); device %.0f MiB, host %.0f MiB" % (rounds, rounds * 70, n, d, h))
if base:
    assert d - base[0] < 64 and h - base[1] < 256, "memory grew: device %+.0f MiB, host %+.0f MiB since round 4" % (d - base[0], h - base[1])
    print("flat since round 4: device %+.0f MiB, host %+.0f MiB" % (d - base[0], h - base[1]))
