"""Fixed-base sums alone (core_sign: 34 bases, 8 chunk lanes per item) by window width: kernel time of one resident
4096-item batch and the rate with 8 in flight (development aid).  usage (GPU box): python tools/quick_fixed_base.py [widths...]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import parity_cases as pc
from bbs_sign_amd import Job

widths = [int(x) for x in sys.argv[1:]] or [8, 12, 14, 16, 18, 20]
n = 4096
for w in widths:
    suite, eng, gens, sk = pc.bench_engine("bls12_381", 32, None, w)
    eng.set_latency_mode(False)
    batches = [pc.bench_items(suite, eng, n, 32, 8, first_item=k * n)[0] for k in range(4)]
    j = eng.core_sign_upload(batches[0])
    j.run(); j.wait()
    tot, st = j.run_timed(3, per_stage=True)
    j.free()
    js = [eng.core_sign_upload(batches[k % 4]) for k in range(8)]
    for x in js:
        x.run()
    for x in js:
        x.wait()
    Job.run_many_timed(js, 8)
    ms, _ = Job.run_many_timed(js, 64)
    for x in js:
        x.free()
    windows = (256 + w - 1) // w
    print("w=%2d windows=%2d table=%8.1f MB  sg_msm_parts %.3f ms (%.1f us per addition)  one batch %.2f ms  8 in flight %.2f M sign/s"
          % (w, windows, 34 * windows * (1 << (w - 1)) * 128 / 1e6, st.get("sg_msm_parts", 0) / 3, st.get("sg_msm_parts", 0) / 3 * 1e3 / (34 * windows / 8.0),
             tot / 3, n * 64 / (ms * 1e-3) / 1e6), flush=True)
    eng.close()
