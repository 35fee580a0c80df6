"""Ad-hoc (GPU box): where the one-shot proof_verify call spends its host time."""
import os, sys, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import parity_cases as pc
from bbs_sign_amd import _lib
n = 4096
suite, eng, gens, sk, msgs, disclosed, rnds = pc.bench_workload("bls12_381", n, 32, 8, None, 16)
sigs, st = eng.core_sign_batch(msgs); proofs, st = eng.core_proof_gen_batch(sigs, msgs, disclosed, rnds)
dm = [m[:8] for m in msgs]
nn, keep, cargs = eng._pv_inputs(proofs, dm, disclosed, None, None)
lib = eng.lib
for rep in range(3):
    t0 = time.perf_counter()
    j = ctypes.c_void_p()
    assert lib.bbs_core_proof_verify_upload(eng.h, nn, *cargs, ctypes.byref(j)) == 0
    t1 = time.perf_counter()
    assert lib.bbs_job_run(j) == 0 and lib.bbs_job_wait(j) == 0
    t2 = time.perf_counter()
    st = np.zeros(nn, dtype=np.int8)
    assert lib.bbs_job_fetch_status(j, st.ctypes.data_as(_lib.c_i8p)) == 0
    t3 = time.perf_counter()
    lib.bbs_job_free(j)
    t4 = time.perf_counter()
    print("upload %.2f ms  run+wait %.2f ms  fetch %.2f ms  free %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3), flush=True)
