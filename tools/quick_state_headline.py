"""Is the serving loop's rate stable over a process's history?  (development aid.)  The headline's submit loop (bench.submit_loop,
6 distinct batches in flight, host buffers -> statuses) is timed in a fresh process, then again after every round of "history":
many resident jobs of other operations alive at once and freed (what a long-lived service, or the later legs of bench_extras, do).
usage: python tools/quick_state_headline.py [window_bits]"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")
import bench                      # noqa: E402
import parity_cases as pc         # noqa: E402
from bbs_sign_amd import Job      # noqa: E402

wb = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n, L, R = 4096, 32, 8
suite, eng, gens, sk = pc.bench_engine("bls12_381", L, None, wb)
slots, raw0 = bench.make_slots(pc, suite, eng, n, L, R, 6, 0)
msgs, disclosed, rnds, sigs, proofs, dm = raw0


def rate(tag):
    bad, _, _ = bench.submit_loop(eng, slots, 12, 6)
    assert bad == 0
    out = []
    for rep in range(3):
        t0 = time.perf_counter()
        bad, _, _ = bench.submit_loop(eng, slots, 96, 6)
        out.append(n * 96 / (time.perf_counter() - t0))
        assert bad == 0
    print("%-64s %s proof_verify/s" % (tag, " ".join("%8.0f" % x for x in out)), flush=True)


def churn(make, k, steps):
    eng.set_latency_mode(False)                    # one form for every job of the set (bbs_jobs_run_timed wants equal stage lists)
    js = [make() for _ in range(k)]
    for j in js:
        j.run()
    for j in js:
        j.wait()
    Job.run_many_timed(js, steps)
    for j in js:
        j.free()
    eng.set_latency_mode("auto")


rate("fresh process")
churn(lambda: eng.core_sign_upload(msgs), 16, 32)
rate("after 16 resident sign jobs alive at once")
churn(lambda: eng.core_proof_gen_upload(sigs, msgs, disclosed, rnds), 16, 32)
rate("after 16 resident proof_gen jobs as well")
churn(lambda: eng.core_proof_verify_upload(proofs, dm, disclosed), 12, 24)
rate("after 12 resident proof_verify jobs as well (36 streams)")
eng.set_batch_verification(True)
churn(lambda: eng.core_proof_verify_upload(proofs, dm, disclosed), 32, 64)
eng.set_batch_verification(False)
rate("after 32 batch-verification jobs as well")
eng.set_batch_verification(True)
churn(lambda: eng.core_proof_verify_upload(proofs * 4, dm * 4, disclosed * 4), 12, 24)
eng.set_batch_verification(False)
rate("after 12 batch-verification jobs of 16384 items as well")
eng.set_batch_verification(True)
churn(lambda: eng.core_verify_upload(sigs, msgs), 32, 64)
eng.set_batch_verification(False)
rate("after 32 batch-verification VERIFY jobs as well")
eng.close()
