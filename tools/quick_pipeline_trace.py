"""Host timestamps of every submit and every retire of the headline loop over K steps (development aid): where the time of a
short timed region goes (filling and draining the in-flight slots).  usage (GPU box): python tools/quick_pipeline_trace.py [K] [inflight]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from bbs_sign_amd import workload as pc

K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
inflight = int(sys.argv[2]) if len(sys.argv) > 2 else 6
suite, eng, gens, sk = pc.bench_engine("bls12_381", 32, None, 20)
slots, _ = bench.make_slots(pc, suite, eng, 4096, 32, 8, inflight, 0)
bench.submit_loop(eng, slots, 16, inflight)
from bbs_sign_amd import Job
for order in ("fifo (wait for the oldest job)", "completion (bbs_jobs_wait_any)"):
    print("== retire order:", order)
    for rep in range(3):
        t0 = time.perf_counter()
        pending, sub, ret, which = [], [], [], []

        def retire():
            if order.startswith("fifo"):
                k = 0
                pending[0][1].wait()
            else:
                k = Job.wait_any([j for _, j in pending])
            step, j = pending.pop(k)
            ret.append(time.perf_counter() - t0); which.append(step); j.free()

        for k in range(K):
            if len(pending) >= inflight:
                retire()
            pending.append((k, eng.submit_packed(slots[k % inflight].n, slots[k % inflight].args))); sub.append(time.perf_counter() - t0)
        while pending:
            retire()
        print("rep %d total %.2f ms -> %.0f /s" % (rep, ret[-1] * 1e3, K * 4096 / ret[-1]))
        print("  submit at ms:", " ".join("%.1f" % (x * 1e3) for x in sub))
        print("  retire at ms:", " ".join("%.1f" % (x * 1e3) for x in ret))
        print("  retired step:", " ".join("%d" % x for x in which))
        print("  gaps between retires:", " ".join("%.1f" % ((b - a) * 1e3) for a, b in zip(ret, ret[1:])))
eng.close()
