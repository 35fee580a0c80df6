"""Ad-hoc (GPU box): device memory of a context (window tables) and of one resident job per operation -- the numbers of
INTEGRATION.md "Sizing".  usage: python tools/mem_sizing.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import parity_cases as pc
import bench


def used():
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    return (total - free) / 2**20


L, R = 32, 8
u0 = used()
for w in (16, 20):
    a = used()
    suite, eng, gens, sk = pc.bench_engine("bls12_381", L, None, w)
    b = used()
    print("context BLS12-381, L = %d, %d-bit windows: %.0f MiB" % (L, w, b - a), flush=True)
    if w == 20:
        break
    eng.close()
for n in (4096, 16384):
    slots, raw = bench.make_slots(pc, suite, eng, n, L, R, 1, first_item=0)
    msgs, disclosed, rnds, sigs, proofs, dm = raw
    makers = {"proof_verify": lambda: eng.core_proof_verify_upload(proofs, dm, disclosed),
              "verify": lambda: eng.core_verify_upload(sigs, msgs),
              "sign": lambda: eng.core_sign_upload(msgs),
              "proof_gen": lambda: eng.core_proof_gen_upload(sigs, msgs, disclosed, rnds)}
    for name, mk in makers.items():
        j = mk()
        print("%-12s n = %5d: %.1f MiB per job (bbs_job_device_bytes)" % (name, n, j.device_bytes() / 2**20), flush=True)
        j.free()
    eng.set_fixed_base_tree(True)
    j = eng.core_proof_verify_upload(proofs, dm, disclosed)
    print("%-12s n = %5d: %.1f MiB per job with the fixed-base tree's work arrays" % ("proof_verify", n, j.device_bytes() / 2**20), flush=True)
    j.free()
    eng.set_fixed_base_tree(False)
