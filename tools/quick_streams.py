"""Ad-hoc: throughput of proof_verify with S device-resident batches in flight (one stream pair each)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import parity_cases as pc

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
wb = int(sys.argv[2]) if len(sys.argv) > 2 else 8
suite, eng, gens, sk, msgs, disclosed, rnds = pc.bench_workload("bls12_381", n, 32, 8, None, wb)
sigs, s = eng.core_sign_batch(msgs); assert (s == 1).all()
proofs, s = eng.core_proof_gen_batch(sigs, msgs, disclosed, rnds); assert (s == 1).all()
dm = [m[:8] for m in msgs]
for S in (1, 2, 4, 8, 16):
    jobs = [eng.core_proof_verify_upload(proofs, dm, disclosed) for _ in range(S)]
    for j in jobs: j.run()
    for j in jobs: j.wait()
    K = 4 * S
    t0 = time.perf_counter()
    for k in range(K): jobs[k % S].run()
    for j in jobs: j.wait()
    dt = time.perf_counter() - t0
    ok = all((j.status() == 1).all() for j in jobs)
    print("streams=%2d  %d batches of %d in %.1f ms -> %.0f proof_verify/s  ok=%s" % (S, K, n, dt * 1e3, K * n / dt, ok), flush=True)
    for j in jobs: j.free()
