"""GPU parity tests proper: the product library (hand-written HIP for gfx950) through the C ABI,
against the oracle on the same seeded inputs, the committed golden fixtures, the reference's
known-answer vectors, and size-independent properties at BASELINE.json's full batch size."""
import pytest

import parity_cases as pc

pytestmark = pytest.mark.gpu


def test_kat_vectors():
    pc.check_kat_vectors(None)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_golden(curve):
    pc.check_golden(curve, None)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_random_batch(curve):
    pc.check_random_batch(curve, None, n=24, L=5, seed=1)
    pc.check_random_batch(curve, None, n=70, L=3, seed=2)     # crosses a wavefront boundary


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_error_semantics(curve):
    pc.check_error_semantics(curve, None)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_primitives(curve):
    pc.check_primitives(curve, None)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_pippenger(curve):
    pc.check_pippenger(curve, None, n=200)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
@pytest.mark.parametrize("n", [4097, 9000, 16384])
def test_pippenger_tiles(curve, n):
    """n > 4096: several tiles per (set, window) in k_pip_window -- one extra item, a ragged third tile, four full tiles."""
    pc.check_pippenger_tiles(curve, None, n=n)


@pytest.mark.parametrize("curve,window_bits", [("bls12_381", 20), ("bn254", 16)])
def test_batch_verification_16384_item_job(curve, window_bits):
    """one 16384-item job (n_tiles = 4), the job size DESIGN section 5 rule 7 quotes batch-verification rates for"""
    pc.check_bv_tiles(curve, None, n=16384, window_bits=window_bits)


def test_batch_verification_ragged_tiles():
    pc.check_bv_tiles("bls12_381", None, n=4096 + 4096 + 37, L=4, R=1, window_bits=16)


# Fixed-base window widths: 8 = the library default, 16 / 20 = what bench.py runs (BN254 / BLS12-381 headline).  Width 20
# is the one whose digits straddle 32-bit words, whose top window holds the carry of the signed recoding and whose tables take 26 GB at L = 32.
WIDTHS = [("bls12_381", 8), ("bls12_381", 16), ("bls12_381", 20), ("bn254", 8), ("bn254", 16), ("bn254", 20)]


@pytest.mark.parametrize("curve,window_bits", WIDTHS)
def test_batch_verification(curve, window_bits):
    pc.check_batch_verification(curve, None, window_bits=window_bits)


@pytest.mark.parametrize("curve,window_bits", WIDTHS)
def test_points_in_subgroup(curve, window_bits):
    pc.check_points_in_subgroup(curve, None, window_bits=window_bits)


@pytest.mark.parametrize("curve,window_bits", WIDTHS)
def test_full_batch_4096(curve, window_bits):
    pc.check_big_batch(curve, None, n=4096, L=32, R=8, window_bits=window_bits)


@pytest.mark.parametrize("curve,window_bits", [("bls12_381", 8), ("bls12_381", 16), ("bn254", 16)])
def test_every_item_against_c_oracle(curve, window_bits):
    pc.check_batch_vs_c_oracle(None, n=1024, window_bits=window_bits, curve=curve)


def test_every_item_of_the_baseline_batch_against_c_oracle():
    """BASELINE's exact batch -- 4096 items, BLS12-381, L = 32, R = 8 -- at the window width bench.py runs (20 bits): every
    signature, proof and boolean against the plain-C oracle (about 50 ms of CPU per item, spread over the host cores)."""
    pc.check_batch_vs_c_oracle(None, n=4096, window_bits=20, curve="bls12_381")


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_window_widths(curve):
    pc.check_window_widths(curve, None, widths=(5, 7, 11, 13, 17, 19, 20, 22), L=2)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_fail_closed(curve):
    pc.check_fail_closed(curve, None)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_submit(curve):
    pc.check_submit(curve, None)
    pc.check_submit(curve, None, n=130, L=6, seed=24)         # crosses wavefront boundaries


def test_mixed_curves_in_flight():
    pc.check_mixed_curves_in_flight(None)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_empty_batches(curve):
    pc.check_empty_batches(curve, None)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_latency_mode(curve):
    pc.check_latency_mode(curve, None)
    pc.check_latency_mode(curve, None, n=80, L=6, seed=42)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_large_shapes(curve):
    pc.check_large_shapes(curve, None, L=100, n=4)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_proof_verify_octets(curve):
    pc.check_proof_verify_octets(curve, None)
    pc.check_proof_verify_octets(curve, None, n=150, L=7, seed=62)
    pc.check_proof_verify_octets(curve, None, seed=63, disclose_all_3=True)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_verify_octets(curve):
    pc.check_verify_octets(curve, None)
    pc.check_verify_octets(curve, None, n=200, L=6, seed=72)


def test_threads():
    pc.check_threads(None, threads=6, rounds=4, n=64, L=5)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
@pytest.mark.parametrize("window_bits", [8, 13, 20])
def test_fixed_base_tree(curve, window_bits):
    if curve == "bn254" and window_bits == 20:
        window_bits = 16
    pc.check_fixed_base_tree(curve, None, window_bits=window_bits, n_pv=70)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_octets_out(curve):
    pc.check_octets_out(curve, None)
    pc.check_octets_out(curve, None, n=300, L=9, seed=96)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_proof_verify_wire(curve):
    pc.check_proof_verify_wire(curve, None)
    pc.check_proof_verify_wire(curve, None, n=200, L=8, seed=98)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_sign_verify_wire(curve):
    pc.check_sign_verify_wire(curve, None)
    pc.check_sign_verify_wire(curve, None, n=150, L=9, seed=100)



@pytest.mark.parametrize("total", [65536, 8192])
def test_mixed_list_configs4_one_gpu(total):
    """BASELINE configs[4] through the real path on one GPU: bench.py --config mixed65536's own driver
    (bench_mixed.run_mixed -> sharding.shard_plan -> mixed.prepare_rank -> packed batches in host buffers ->
    bbs_core_proof_verify_submit with both curves in flight -> gather -> merge_status); every 16th global item is
    corrupted and run_mixed compares the merged statuses of every step with that pattern.  8192 = one rank's share of the
    list at 8 GPUs (the strong-scaling regime: one job per curve, which the library runs in its latency form)."""
    import argparse
    import json
    import torch
    import bench_mixed
    args = argparse.Namespace(batch=4096, inflight=8, window_bits=20, warmup=1, steps=3, backend="nccl")
    lines = []
    # (single process: every job is waited for by the library itself; torch only holds the timing scalar, on the CPU --
    # torch.cuda is not initialised here, after the engine has been using the device for the whole session)
    bench_mixed.run_mixed(args, pc, torch, None, 0, 0, 1, "cpu", lambda: None, total=total, emit=lines.append)
    line = json.loads(lines[0])
    assert line["checks"]["merged_statuses_exact_every_step"] is True and line["n_gpus"] == 1
    assert line["config"]["items_per_rank"] == total
    assert line["config"]["batches_per_rank"] == total // 4096
    assert line["config"]["batch_sizes_rank0"] == [4096]


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` from a bare command (no torchrun, WORLD_SIZE unset): bench.py starts the two ranks itself
    as a child process before touching the GPU.  Rehearsal of the N > 1 plumbing only: both ranks share this box's one
    GPU and talk over gloo."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--backend", "gloo", "--all-ranks-on-device", "0",
                        "--steps", "8", "--warmup", "2", "--batch", "1024", "--inflight", "2", "--window-bits", "16",
                        "--no-extras", "--no-cpu-baseline", "--min-region-s", "0"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["steps"] == 8 and line["checks"]["statuses_exact_every_step"] is True


# ---- the latency form of a job (T1 on three lanes, the two Miller loops on separate wavefronts): the same cases ----------
LATENCY = pytest.mark.job_form(True)


@LATENCY
@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_latency_form_random_batch_and_errors(curve):
    pc.check_random_batch(curve, None, n=70, L=3, seed=2)
    pc.check_error_semantics(curve, None)
    pc.check_verify_octets(curve, None)
    pc.check_proof_verify_octets(curve, None)
    pc.check_empty_batches(curve, None)
    pc.check_batch_verification(curve, None, window_bits=8)


@LATENCY
@pytest.mark.parametrize("curve,window_bits", [("bls12_381", 20), ("bn254", 16)])
def test_latency_form_full_batch_4096(curve, window_bits):
    pc.check_big_batch(curve, None, n=4096, L=32, R=8, window_bits=window_bits)


@LATENCY
def test_latency_form_every_item_against_c_oracle():
    pc.check_batch_vs_c_oracle(None, n=512, window_bits=16, curve="bls12_381")


@pytest.mark.job_form(None)
@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_auto_form(curve):
    """The library's default: a job with at most one other live job on its context is laid out in the latency form (verify:
    the pairing as two stages), the third and later live jobs in the throughput form (one fused pairing stage); the
    statuses are the same.  (proof_verify keeps the fused pairing kernel in both forms; its forms differ in the MSM parts.)"""
    import ctypes
    suite, eng, gens, sk, msgs, disclosed, rnds = pc.bench_workload(curve, 130, 6, 2, None, 8)
    sigs, st = eng.core_sign_batch(msgs)
    proofs, st = eng.core_proof_gen_batch(sigs, msgs, disclosed, rnds)
    assert (st == 1).all()
    proofs[5].r1_cap = (proofs[5].r1_cap + 1) % suite.curve.r
    dm = [m[:2] for m in msgs]
    vsigs = list(sigs)
    vsigs[5] = pc.Signature(sigs[5].a, (sigs[5].e + 1) % suite.curve.r)
    jobs = [eng.core_verify_upload(vsigs, msgs) for _ in range(4)]
    eng.lib.bbs_job_stage_name.restype = ctypes.c_char_p

    def names(j):
        out, k = [], 0
        while True:
            nm = eng.lib.bbs_job_stage_name(j.h, k)
            if not nm:
                return out
            out.append(nm.decode()); k += 1
    # latency form: the two Miller loops of an item on separate wavefronts; throughput form: both on one six-lane group
    # (either as its own kernel in front of the final exponentiation, or fused with it: BBS_PAIR_SPLIT2)
    assert "pair_miller" in names(jobs[0]) and "pair_miller" in names(jobs[1])
    assert ("pair_miller_both" in names(jobs[2]) or "pairing_6lane" in names(jobs[2])) and "pair_miller" not in names(jobs[2])
    assert ("pair_miller_both" in names(jobs[3]) or "pairing_6lane" in names(jobs[3])) and "pair_miller" not in names(jobs[3])
    pjobs = [eng.core_proof_verify_upload(proofs, dm, disclosed) for _ in range(2)]      # live jobs 5 and 6: throughput form
    for j in jobs + pjobs:
        j.run()
    want = [0 if i == 5 else 1 for i in range(130)]
    for j in jobs + pjobs:
        j.wait()
        assert [int(x) for x in j.status()] == want
        j.free()
    pjobs = [eng.core_proof_verify_upload(proofs, dm, disclosed) for _ in range(2)]      # alone again: latency form
    for j in pjobs:
        j.run()
    for j in pjobs:
        j.wait()
        assert [int(x) for x in j.status()] == want
        j.free()
    eng.close()


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_issuer_mixed_lengths(curve):
    pc.check_issuer_mixed_lengths(curve, None)
    pc.check_issuer_mixed_lengths(curve, None, seed=137, lengths=(6, 6, 2, 4, 4, 9, 0, 1, 6, 2, 3, 3, 5), oracle_items=(0, 2))
    # table width chosen by the library from the free device memory (20 bits on an empty MI355X)
    pc.check_issuer_mixed_lengths(curve, None, seed=139, lengths=(2, 2, 1, 2, 1, 2, 1, 8, 2), oracle_items=(0,), window_bits=0)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_issuer_threads(curve):
    pc.check_issuer_threads(curve, None, threads=4, rounds=3)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_issuer_budget(curve):
    pc.check_issuer_budget(curve, None)


def test_dedicated_queues_under_the_runtimes_default_pool():
    """A process whose HIP runtime was initialised with the default pool of 4 hardware queues (GPU_MAX_HW_QUEUES=4 here stands
    for "torch touched the GPU before the library was loaded"): with bbs_runtime_set_dedicated_queues the job streams get
    hardware queues of their own.  Functional check through bench.py's own loop (distinct batches in flight, one of them
    corrupted, statuses compared on every step); the rates are in profiles/r04_f_dedicated_queues.log."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["GPU_MAX_HW_QUEUES"] = "4"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--dedicated-queues", "12", "--steps", "12", "--warmup", "3", "--batch", "1024",
                        "--inflight", "8", "--window-bits", "16", "--no-extras", "--no-cpu-baseline", "--min-region-s", "0"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["checks"]["statuses_exact_every_step"] is True and line["config"]["dedicated_queues"] == 12 and line["config"]["hw_queues"] == 4


@pytest.mark.parametrize("pool,dedicated", [("14", "16"), ("32", "0"), ("4", "16")])
def test_hardware_queue_budget_never_aborts(pool, dedicated):
    """Pooled + dedicated hardware queues x the largest kernel frame is a CHECKED budget (runtime.hpp queue_budget): with
    GPU_MAX_HW_QUEUES=14 and bbs_runtime_set_dedicated_queues(16) -- a legal configuration that made the runtime abort the
    process in round 4 once more than ~ 20 queues were in use -- 32 one-stream jobs in flight run to completion with exact
    statuses; likewise with a pool that is itself over budget (32: the library stops creating streams at the budget and the
    jobs beyond it share their context's stream).  The process must END NORMALLY: a runtime abort is a non-zero exit."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["GPU_MAX_HW_QUEUES"] = pool
    env["BBS_DEDICATED_QUEUES"] = dedicated
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "queue_budget_probe.py"), "32", "1024", "3"], env=env,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert r.returncode == 0, (r.returncode, r.stderr[-3000:])
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["statuses_exact"] is True
    b = line["queue_budget"]
    assert b["pool"] == int(pool) and b["total"] >= 8 and b["dedicated_cap"] == max(0, b["total"] - b["pool"]), b
    assert b["scratch_bytes_per_lane"] <= 1900, b      # the largest kernel frame of the library (round 4: 2848)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_proof_gen_unusual_points(curve):
    pc.check_proof_gen_unusual_points(curve, None)


@pytest.mark.job_form(True)
@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_proof_gen_unusual_points_latency_form(curve):
    pc.check_proof_gen_unusual_points(curve, None)


def test_pool_two_members_on_one_gpu():
    """bbs_pool with device ids (0, 0): two context sets, two submitting threads inside the library, one GPU.  A mixed BN254 +
    BLS12-381 list with corrupted, forged and malformed items: statuses in list order = one context per curve."""
    pc.check_pool(None, devices=(0, 0), per_curve=300, L=8, R=3, window_bits=8, max_batch=64)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_fail_closed_every_submit_entry_point(curve):
    pc.check_fail_closed_submit(curve, None)

