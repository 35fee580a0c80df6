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
for rep in range(3):
    t0 = time.perf_counter()
    pending, sub, ret = [], [], []
    for k in range(K):
        if len(pending) >= inflight:
            j = pending.pop(0); j.wait(); ret.append(time.perf_counter() - t0); j.free()
        pending.append(eng.submit_packed(slots[k % inflight].n, slots[k % inflight].args)); sub.append(time.perf_counter() - t0)
    while pending:
        j = pending.pop(0); j.wait(); ret.append(time.perf_counter() - t0); j.free()
    print("rep %d total %.2f ms -> %.0f /s" % (rep, ret[-1] * 1e3, K * 4096 / ret[-1]))
    print("  submit at ms:", " ".join("%.1f" % (x * 1e3) for x in sub))
    print("  retire at ms:", " ".join("%.1f" % (x * 1e3) for x in ret))
    print("  gaps between retires:", " ".join("%.1f" % ((b - a) * 1e3) for a, b in zip(ret, ret[1:])))
eng.close()
