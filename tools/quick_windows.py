"""Ad-hoc: effect of the fixed-base window width on setup time and MSM stage time."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import parity_cases as pc
n = 4096
for wb in [int(x) for x in sys.argv[1:]] or [8, 12, 16]:
    t = time.time()
    suite, eng, gens, sk, msgs, disclosed, rnds = pc.bench_workload("bls12_381", n, 32, 8, None, wb)
    setup = time.time() - t
    sigs, s = eng.core_sign_batch(msgs); assert (s == 1).all()
    proofs, s = eng.core_proof_gen_batch(sigs, msgs, disclosed, rnds); assert (s == 1).all()
    dm = [m[:8] for m in msgs]
    j = eng.core_proof_verify_upload(proofs, dm, disclosed)
    j.run(); j.wait()
    tot, st = j.run_timed(3)
    assert (j.status() == 1).all()
    js = eng.core_sign_upload(msgs); js.run(); js.wait(); tots, sts = js.run_timed(3)
    print("window %2d: setup %.2fs  pv %.2f ms/batch (msm %.2f, pairing %.2f)  sign %.2f ms (msm %.2f)" % (
        wb, setup, tot / 3, st["pv_msm_parts"] / 3, st["pairing_6lane"] / 3, tots / 3, sts["sg_msm_parts"] / 3), flush=True)
    j.free(); js.free(); eng.close()
