"""One batch at a time, per-stage durations, for one curve (development aid).  usage: python tools/quick_single_bn.py [curve] [window_bits]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import parity_cases as pc

curve = sys.argv[1] if len(sys.argv) > 1 else "bn254"
wb = int(sys.argv[2]) if len(sys.argv) > 2 else 16
n = 4096
suite, eng, gens, sk, msgs, disclosed, rnds = pc.bench_workload(curve, n, 32, 8, None, wb)
sigs, st = eng.core_sign_batch(msgs)
proofs, st = eng.core_proof_gen_batch(sigs, msgs, disclosed, rnds)
dm = [m[:8] for m in msgs]
for form in (False, True, "auto"):
    eng.set_latency_mode(form)
    j = eng.core_proof_verify_upload(proofs, dm, disclosed)
    j.run(); j.wait()
    assert (j.status() == 1).all()
    tot, stg = j.run_timed(3, per_stage=True)
    print("%s proof_verify form=%-5s %6.2f ms  %7.0f /s  %s" % (curve, form, tot / 3, n / (tot / 3) * 1e3, {k: round(v / 3, 2) for k, v in stg.items()}), flush=True)
    j.free()
eng.close()
