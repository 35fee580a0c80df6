"""Ad-hoc (GPU box): device and host memory stay flat over many one-shot calls and job create/free cycles."""
import os, sys, resource
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import parity_cases as pc
from bbs_sign_amd import _lib
n = 1024
suite, eng, gens, sk, msgs, disclosed, rnds = pc.bench_workload("bls12_381", n, 32, 8, None, 8)
sigs, st = eng.core_sign_batch(msgs); proofs, st = eng.core_proof_gen_batch(sigs, msgs, disclosed, rnds)
dm = [m[:8] for m in msgs]
nn, keep, cargs = eng._pv_inputs(proofs, dm, disclosed, None, None)
def snap():
    free, total = torch.cuda.mem_get_info()
    return (total - free) / 2**20, resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024
for mode in (False, True):
    eng.set_batch_verification(mode)
    for it in range(401):
        stt = np.zeros(nn, dtype=np.int8)
        assert eng.lib.bbs_core_proof_verify_batch(eng.h, nn, *cargs, stt.ctypes.data_as(_lib.c_i8p)) == 0 and (stt == 1).all()
        if it in (0, 100, 400):
            print("batch_verify=%s call %3d: device used %.0f MiB, host max RSS %.0f MiB" % (mode, it, *snap()), flush=True)
