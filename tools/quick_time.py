"""Ad-hoc timing of the four ops on the bench workload (development aid; bench.py is the contract)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import parity_cases as pc

curve = sys.argv[1] if len(sys.argv) > 1 else "bls12_381"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
wb = int(sys.argv[3]) if len(sys.argv) > 3 else 8
t = time.time()
suite, eng, gens, sk, msgs, disclosed, rnds = pc.bench_workload(curve, n, 32, 8, None, wb)
print("setup %.2fs (window %d)" % (time.time() - t, wb), flush=True)
j = eng.core_sign_upload(msgs)
tot, st = j.run_timed(1)
print("sign      n=%d %.2f ms -> %.0f/s" % (n, tot, n / tot * 1e3), {k: round(v, 2) for k, v in st.items()}, flush=True)
sigs, s = j.signatures(); assert (s == 1).all()
j = eng.core_verify_upload(sigs, msgs)
tot, st = j.run_timed(1)
print("verify    n=%d %.2f ms -> %.0f/s" % (n, tot, n / tot * 1e3), {k: round(v, 2) for k, v in st.items()}, flush=True)
assert (j.status() == 1).all()
j = eng.core_proof_gen_upload(sigs, msgs, disclosed, rnds)
tot, st = j.run_timed(1)
print("proof_gen n=%d %.2f ms -> %.0f/s" % (n, tot, n / tot * 1e3), {k: round(v, 2) for k, v in st.items()}, flush=True)
proofs, s = j.proofs(); assert (s == 1).all()
dm = [m[:8] for m in msgs]
j = eng.core_proof_verify_upload(proofs, dm, disclosed)
for rep in range(2):
    tot, st = j.run_timed(1)
    print("proof_ver n=%d %.2f ms -> %.0f/s" % (n, tot, n / tot * 1e3), {k: round(v, 2) for k, v in st.items()}, flush=True)
assert (j.status() == 1).all()
