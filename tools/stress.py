"""Ad-hoc (GPU box): seeded random batches with ragged inputs and corruptions against the oracle, many seeds."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_cases as pc
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
t0 = time.time()
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 100
done = 0
while time.time() - t0 < budget:
    pc.LATENCY_MODE = [None, False, True][seed % 3]          # the library's AUTO / throughput / latency form of every job
    for curve in ("bls12_381", "bn254"):
        L = [0, 1, 2, 5, 7, 12][seed % 6]
        pc.check_random_batch(curve, None, n=8 + seed % 9, L=L, seed=seed)
        pc.check_batch_verification(curve, None, n=9, L=4 + seed % 3, seed=seed)
        pc.check_points_in_subgroup(curve, None, n=10, L=3 + seed % 4, seed=seed)
        pc.check_submit(curve, None, n=7 + seed % 5, L=3 + seed % 4, seed=seed)
        pc.check_latency_mode(curve, None, n=12, L=4 + seed % 3, seed=seed)
        pc.check_proof_verify_octets(curve, None, n=14, L=5 + seed % 3, seed=seed)
        pc.check_verify_octets(curve, None, n=16 + seed % 7, L=2 + seed % 5, seed=seed)
        pc.check_octets_out(curve, None, n=12 + seed % 6, L=3 + seed % 5, seed=seed)
        pc.check_proof_verify_wire(curve, None, n=12 + seed % 5, L=3 + seed % 5, seed=seed)
        pc.check_sign_verify_wire(curve, None, n=10 + seed % 5, L=1 + seed % 6, seed=seed)
        import random as _r
        rr = _r.Random(seed)
        lens = [rr.randrange(0, 7) for _ in range(9)] + [7 + seed % 3]
        rr.shuffle(lens)
        if max(lens[:7]) > 6:                              # the case tampers with items 0 .. 6: the over-long item goes behind them
            lens.sort()
        pc.check_issuer_mixed_lengths(curve, None, seed=seed, lengths=tuple(lens), oracle_items=(0,))
        if seed % 4 == 0:
            pc.check_issuer_budget(curve, None, seed=seed)          # bounded contexts: eviction, rebuild, -43, refused configuration
        if seed % 5 == 0:
            pc.check_fail_closed_submit(curve, None)                # round 5: every submit entry point refuses undecided items
        done += 1
    if seed % 3 == 0:
        pc.check_pool(None, devices=(0, 0), per_curve=24 + seed % 17, L=3 + seed % 4, R=1 + seed % 2, window_bits=8, max_batch=5 + seed % 9)      # round 5: mixed list through bbs_pool
    seed += 1
    print("seed", seed, "cases", done, "elapsed %.0f s" % (time.time() - t0), flush=True)
print("stress ok:", done, "cases")
