"""Untimed legs of bench.py (rank 0, N = 1): the other three operations of the path, the second curve, the opt-in
modes, the window-width comparison on distinct data and the reference's own bench sweeps.  Nothing here feeds `value`."""
import ctypes
import time


def op_roofline(counters, counters_file, lib_hash, op, items_per_s, in_flight):
    """SURVEY 8(d) asks for sign / verify / proof_gen "plus fraction of roofline" as well.  From the per-kernel counters of
    one 4096-item job of the operation (rocprofv3 --pmc, tools/run_profile_ops.sh -> profiles/*_counters.json, key "ops") and
    the rate measured HERE with `in_flight` resident jobs:
      * hbm (the roofline the contract names): algorithmic bytes per item x items/s against 8 TB/s, and the measured traffic of
        one launch (2 x FETCH_SIZE + WRITE_SIZE, the guide's gfx950 correction) over the algorithmic bytes;
      * valu_issue (the resource that binds): the SIMD time this run spent per wavefront instruction (1024 SIMDs / rate /
        instructions) against the micro-benchmarked issue cost of each kernel's opcode mix at the occupancy its registers
        allow -- frac = modelled / spent = how much of the SIMDs' time went into issuing the operation's instructions."""
    spec = ((counters or {}).get("ops") or {}).get(op)
    if not spec or not items_per_s:
        return None
    ks = spec["kernels"]
    n0 = float(spec.get("items_per_launch", 4096))
    insts = sum(k["valu_insts"] for k in ks.values())
    model_ns = sum(k["valu_insts"] * k["ns_per_inst_at_that_occupancy"] for k in ks.values()) / insts
    spent_ns = 1024.0 * (n0 / items_per_s) * 1e9 / insts
    traffic = sum(2 * k["FETCH_SIZE_KiB"] + k["WRITE_SIZE_KiB"] for k in ks.values()) * 1024.0
    alg = spec["alg_bytes_per_item"] * n0
    current = (counters.get("library_source_hash") == lib_hash) if counters.get("library_source_hash") else None
    return {"items_per_s": items_per_s, "jobs_in_flight": in_flight,
            "valu_issue": {"wave_insts_per_item": insts / n0, "valu_wave_insts_per_job": insts,
                           "ns_per_wave_inst_per_simd": {"this_run": spent_ns, "ubench_mix_at_kernel_occupancy": model_ns},
                           "frac": model_ns / spent_ns,
                           "frac_note": "modelled issue cost (micro-benchmark of the kernels' opcode mixes, +-3 %: clock and mix) over SIMD time "
                                        "spent; at or slightly above 1 = the operation runs at the issue ceiling within the model's error",
                           "kernels": {st: {"valu_wave_insts": k["valu_insts"], "waves_per_simd": k["waves_per_simd"],
                                            "ns_per_inst": k["ns_per_inst_at_that_occupancy"]} for st, k in ks.items()}},
            "roofline": {"bound": "hbm", "algorithmic_bytes_per_item": spec["alg_bytes_per_item"], "achieved": spec["alg_bytes_per_item"] * items_per_s / 1e9,
                         "peak": 8000.0, "unit": "GB/s", "frac": spec["alg_bytes_per_item"] * items_per_s / 1e9 / 8000.0,
                         "traffic": traffic, "hbm_traffic_over_algorithmic": traffic / alg},
            "source": counters_file, "counters_from_this_library": current}


def other_ops(args, pc, eng, suite, n, L, R, msgs, disclosed, rnds, sigs, proofs, dm, device, submit_loop, make_slots, slots=None):
    from bbs_sign_amd import Job

    def rate(j, reps=3):
        j.run(); j.wait()
        ms, _ = j.run_timed(reps, per_stage=False)
        assert (j.status() == 1).all()
        j.free()
        return n / (ms / reps * 1e-3)

    def rate_k(make, k=8, steps=32, expect_all_true=True):        # k device-resident batches in flight
        js = [make() for _ in range(k)]
        for j in js:
            j.run()
        for j in js:
            j.wait()
            if expect_all_true:
                assert (j.status() == 1).all()
        Job.run_many_timed(js, k)
        ms, _ = Job.run_many_timed(js, steps)
        size = js[0].n
        for j in js:
            j.free()
        return size * steps / (ms * 1e-3)

    def rate_k_any(make, k=16, steps=128):          # k resident batches in flight, re-run in COMPLETION order (bbs_jobs_wait_any)
        js = [make() for _ in range(k)]
        for j in js:
            j.run()
        for _ in range(k):                          # warm: every job once more, whichever finishes first goes again
            js[Job.wait_any(js)].run()
        t0 = time.perf_counter()
        for _ in range(steps):
            js[Job.wait_any(js)].run()
        dt = time.perf_counter() - t0               # `steps` jobs retired (and re-submitted) in dt
        for j in js:
            j.wait()
            assert (j.status() == 1).all()
        size = js[0].n
        for j in js:
            j.free()
        return size * steps / dt

    # development aid (BBS_BENCH_ISSUER_PROBE=1): the issuer's two-length list, six lists in flight, measured at several points
    # of this function -- which leg leaves the state in which the issuer legs further down run at half their stand-alone rate?
    probe = {"iss": None}

    def issuer_probe(tag):
        import os as _os
        if not _os.environ.get("BBS_BENCH_ISSUER_PROBE"):
            return
        from bbs_sign_amd import Issuer, api as _api2
        if probe["iss"] is None:
            s16, e16, _, _ = pc.bench_engine("bls12_381", 16, None, 16, device=device)
            m16, d16, r16 = pc.bench_items(s16, e16, n // 2, 16, 4, 0)
            sg16, _st = e16.core_sign_batch(m16)
            pf16, _st = e16.core_proof_gen_batch(sg16, m16, d16, r16)
            e16.close()
            raw = lambda cnt, r: [[pc.expand_message(b"bbs-bench-msg" + pc.i2osp(b, 8) + pc.i2osp(j, 8), b"BBS_BENCH_MSG_DST_", 32) for j in range(r)] for b in range(cnt)]
            o32 = [_api2.proof_to_octets("bls12_381", p_) for p_ in proofs[:n // 2]]
            o16 = [_api2.proof_to_octets("bls12_381", p_) for p_ in pf16]
            iss = Issuer("bls12_381", suite.api_id, device=device, window_bits=16)
            iss.set_public_key(eng.public_key())
            mo = [x for pair in zip(o32, o16) for x in pair]
            mr = [x for pair in zip(raw(n // 2, R), raw(n // 2, 4)) for x in pair]
            mi = [x for pair in zip(disclosed[:n // 2], d16) for x in pair]
            probe["iss"] = (iss,) + tuple(iss.pack_proof_verify(mo, mr, mi))
            assert (iss.proof_verify_packed(probe["iss"][1], probe["iss"][3]) == 1).all()
        iss, n_p, keep_p, args_p = probe["iss"]
        pend = []
        for phase in (0, 1):
            t9 = time.perf_counter()
            for _ in range(32):
                if len(pend) >= 6:
                    j = pend.pop(0); j.wait(); j.free()
                pend.append(iss.proof_verify_submit_packed(n_p, args_p))
            while pend:
                j = pend.pop(0); j.wait(); j.free()
        out.setdefault("issuer_probe", {})[tag] = 32 * n / (time.perf_counter() - t9)

    eng.set_latency_mode(False)        # every leg below states its form: throughput unless it says otherwise
    out = {"unit": "items/s; single = one 4096-item resident batch at a time, *_8_in_flight = eight resident batches; jobs in "
                   "the throughput form unless a leg says otherwise"}
    bls = {"sign": rate(eng.core_sign_upload(msgs)), "verify": rate(eng.core_verify_upload(sigs, msgs)),
           "proof_gen": rate(eng.core_proof_gen_upload(sigs, msgs, disclosed, rnds)),
           "sign_8_in_flight": rate_k(lambda: eng.core_sign_upload(msgs)),
           "verify_8_in_flight": rate_k(lambda: eng.core_verify_upload(sigs, msgs)),
           "proof_gen_8_in_flight": rate_k(lambda: eng.core_proof_gen_upload(sigs, msgs, disclosed, rnds)),
           # (the comb's table stage makes a proof_gen job longer: it takes 12 in flight to fill the chip where 8 did)
           "proof_gen_12_in_flight": rate_k(lambda: eng.core_proof_gen_upload(sigs, msgs, disclosed, rnds), 12, 48),
           # (round 5, profiles/r05_m_other_ops_by_inflight.log: the short jobs of sign and the five-stage jobs of proof_gen fill
           # the chip at 16 in flight -- sign 8.4 - 9.0 M at 8, 10.0 - 10.2 M at 12, 10.2 - 10.7 M at 16, 9.4 M at 20; proof_gen
           # 2.45 / 2.73 / 2.80 / 2.45 M at 8 / 12 / 16 / 20; verify is flat from 8: 1.83 / 1.86 / 1.84 M)
           "sign_16_in_flight": rate_k(lambda: eng.core_sign_upload(msgs), 16, 96),
           "verify_12_in_flight": rate_k(lambda: eng.core_verify_upload(sigs, msgs), 12, 48),
           "proof_gen_16_in_flight": rate_k(lambda: eng.core_proof_gen_upload(sigs, msgs, disclosed, rnds), 16, 64)}
    out["bls12_381"] = bls
    issuer_probe("1 after the basic legs (sign / verify / proof_gen, up to 16 in flight)")
    # fraction of roofline of the other three operations (counters: profiles/*_counters.json "ops")
    try:
        import bench as _bench
        counters, counters_file = _bench.load_counters()
        lib_hash = eng.lib.bbs_source_hash().decode()
        for op, key, k_in in (("sign", "sign_16_in_flight", 16), ("verify", "verify_12_in_flight", 12), ("proof_gen", "proof_gen_16_in_flight", 16)):
            r = op_roofline(counters, counters_file, lib_hash, op, bls.get(key), k_in)
            if r:
                out[op] = r
    except Exception as e:       # a missing / older counters file must not take the bench down
        out["ops_roofline_error"] = repr(e)

    # ---- the reference's only usable proof_verify bench sweep (benches/proof_verify.rs:145-175): L = 32, R in {1..32}
    sweep = {}
    for r in (1, 2, 4, 8, 16, 32):
        d_r = [list(range(r))] * n
        rn_r = [rn[:5 + L - r] for rn in rnds] if r >= R else None
        if rn_r is None:                               # fewer disclosed -> more undisclosed scalars than the workload drew
            rn_r = [pc.seeded_random_scalars(suite, b"bbs-bench-rnd-r%d-" % r + pc.i2osp(b, 8),
                                                suite.api_id + b"MOCK_RANDOM_SCALARS_DST_", 5 + L - r) for b in range(n)]
        pr, st = eng.core_proof_gen_batch(sigs, msgs, d_r, rn_r)
        assert (st == 1).all()
        dm_r = [m[:r] for m in msgs]
        sweep["R=%d" % r] = {"proof_verify_8_in_flight": rate_k(lambda: eng.core_proof_verify_upload(pr, dm_r, d_r)),
                             "proof_gen_8_in_flight": rate_k(lambda: eng.core_proof_gen_upload(sigs, msgs, d_r, rn_r))}
    bls["disclosed_sweep_L32"] = sweep
    issuer_probe("2 after the disclosed sweep")

    # ---- window widths of the fixed-base tables (signed digits: 2^(w-1) entries per base and window), the headline's loop on
    # the headline's own packed batches (distinct data in every slot: a 26 GB table is touched at different entries by every
    # batch).  The packed host buffers do not depend on the context, only the tables do.
    cmp_w = {}
    for w in sorted({8, 12, 16, 20} - {args.window_bits}):
        s2, e2, _, _ = pc.bench_engine("bls12_381", L, None, w, device=device)
        use = slots if slots is not None else make_slots(pc, s2, e2, n, L, R, max(1, args.inflight), first_item=0)[0]
        bad, _, _ = submit_loop(e2, use, 2 * len(use), len(use))
        assert bad == 0
        t0 = time.perf_counter()
        bad, _, _ = submit_loop(e2, use, 64, len(use))
        cmp_w[str(w)] = {"proof_verify_per_s": n * 64 / (time.perf_counter() - t0),
                         "table_bytes": (L + 2) * ((256 + w - 1) // w) * (1 << (w - 1)) * 128}
        assert bad == 0
        e2.close()
    wb = args.window_bits
    bls["host_inclusive_by_window_bits"] = dict(cmp_w, note="same loop as the headline (distinct batches, %d in flight); the headline's own width "
                                                "(%d bits, %d table bytes) is `value`; the library default, bbs_ctx_set_window_bits(ctx, 0), picks "
                                                "the widest of 20 / 16 / 12 / 8 that fits an eighth of the free device memory"
                                                % (args.inflight, wb, (L + 2) * ((256 + wb - 1) // wb) * (1 << (wb - 1)) * 128))

    # ---- one batch at a time in latency mode (bbs_ctx_set_latency_mode: T1's three terms on three lanes), and the
    # price of that mode with eight batches in flight
    eng.set_latency_mode(True)
    lj = eng.core_proof_verify_upload(proofs, dm, disclosed)
    lj.run(); lj.wait()
    assert (lj.status() == 1).all()
    lms, lst = lj.run_timed(3, per_stage=True)
    lj.free()
    bls["latency_mode"] = {"single_batch_ms": lms / 3, "single_batch_proof_verify_per_s": n / (lms / 3 * 1e-3),
                           "stage_ms": {k: v / 3 for k, v in lst.items()},
                           "proof_verify_8_in_flight": rate_k(lambda: eng.core_proof_verify_upload(proofs, dm, disclosed))}
    eng.set_latency_mode(False)
    issuer_probe("3 after the window-width engines (created and closed)")
    # ---- larger batches need fewer batches in flight to fill the chip (one 16384-item batch is 2560 + 1640 wavefronts
    # of the two long kernels on 1024 SIMDs): resident, two and four in flight
    big = (proofs * 4, dm * 4, disclosed * 4)
    bls["batch_16384"] = {"proof_verify_2_in_flight": rate_k(lambda: eng.core_proof_verify_upload(*big), 2, 8),
                          "proof_verify_4_in_flight": rate_k(lambda: eng.core_proof_verify_upload(*big), 4, 12)}

    issuer_probe("4 after the 16384-item jobs")
    # ---- opt-in modes (not the headline): batch verification, subgroup vouching
    eng.set_batch_verification(True)
    bls["proof_verify_batch_verification_32_in_flight"] = rate_k(lambda: eng.core_proof_verify_upload(proofs, dm, disclosed), 32, 96)
    issuer_probe("5a after proof_verify batch verification, 32 jobs in flight")
    bls["proof_verify_batch_verification_16384_items"] = rate_k(
        lambda: eng.core_proof_verify_upload(proofs * 4, dm * 4, disclosed * 4), 12, 48)
    issuer_probe("5b after 16384-item batch-verification jobs, 12 in flight")
    bls["verify_batch_verification_32_in_flight"] = rate_k(lambda: eng.core_verify_upload(sigs, msgs), 32, 96)
    issuer_probe("5c after verify batch verification, 32 jobs in flight")
    # the same 4096-item jobs retired in completion order, as a serving loop does (round 4): jobs whose tails end early go
    # again at once instead of waiting for an older job (profiles/r04_k_bv_paced.log)
    bls["proof_verify_batch_verification_16_in_flight_completion_order"] = rate_k_any(lambda: eng.core_proof_verify_upload(proofs, dm, disclosed), 16, 160)
    eng.set_batch_verification(False)
    issuer_probe("5 after the batch-verification legs (32 jobs in flight)")
    eng.set_points_in_subgroup(True)
    bls["points_in_subgroup"] = {"proof_verify_8_in_flight": rate_k(lambda: eng.core_proof_verify_upload(proofs, dm, disclosed)),
                                 "verify_8_in_flight": rate_k(lambda: eng.core_verify_upload(sigs, msgs)),
                                 "proof_gen_8_in_flight": rate_k(lambda: eng.core_proof_gen_upload(sigs, msgs, disclosed, rnds))}
    # ... and the headline's own submit loop on its own host buffers with the proof points vouched for: what a host gets whose
    # point type guarantees membership of the prime-order subgroup -- arkworks' G1 after deserialisation, i.e. every caller
    # of the reference's core_proof_verify, and the Rust shim (bindings/rust) sets this always.  `value` does NOT assume it.
    if slots is not None:
        bad, _, _ = submit_loop(eng, slots, 2 * len(slots), len(slots))
        assert bad == 0
        t0 = time.perf_counter()
        bad, _, _ = submit_loop(eng, slots, 96, len(slots))
        bls["points_in_subgroup"]["proof_verify_host_inclusive"] = n * 96 / (time.perf_counter() - t0)
        assert bad == 0
    eng.set_points_in_subgroup(False)
    issuer_probe("6 after the vouched legs")

    # ---- ingest of octet strings: 3 n point decompressions + subgroup checks on the device
    import numpy as np
    from bbs_sign_amd import _lib as _l
    from bbs_sign_amd import api as _api
    from bbs_sign_amd.engine import _ragged_bytes
    octs = [_api.proof_to_octets("bls12_381", p_) for p_ in proofs[:n]]
    flat_o, off_o = _ragged_bytes(octs)
    rec_o = 6 * eng.fpb + 128
    pf_o = np.zeros(n * rec_o, dtype=np.uint8); cm_o = np.zeros(n * 32 * 32, dtype=np.uint8)
    cmo_o = np.zeros(n + 1, dtype=np.uint64); st_o = np.zeros(n, dtype=np.int8)

    def decode_call():
        rc = eng.lib.bbs_proofs_from_octets_batch(eng.h, n, flat_o.ctypes.data_as(_l.c_u8p), off_o.ctypes.data_as(_l.c_u64p),
                                                  pf_o.ctypes.data_as(_l.c_u8p), cm_o.ctypes.data_as(_l.c_u8p),
                                                  cmo_o.ctypes.data_as(_l.c_u64p), st_o.ctypes.data_as(_l.c_i8p))
        assert rc == 0 and (st_o == 1).all()
    decode_call()
    t1 = time.perf_counter()
    for _ in range(4):
        decode_call()
    bls["proofs_from_octets_per_s"] = 4 * n / (time.perf_counter() - t1)
    # ... and the fused wire path: proof octets in host buffers -> statuses (decompression, subgroup checks, verification
    # in one pipeline; GLV for the variable-base terms because the decoder has checked membership), 8 batches in flight
    nn_o, keep_o, args_o = eng._oct_inputs(octs, dm, disclosed, None, None)
    pend = []
    def retire_o():
        j = pend.pop(0)
        j.wait()
        assert (j.result == 1).all()
        j.free()
    for k in range(8):
        pend.append(eng.proof_verify_octets_submit_packed(nn_o, args_o))
    while pend:
        retire_o()
    t1 = time.perf_counter()
    for k in range(64):
        if len(pend) >= 8:
            retire_o()
        pend.append(eng.proof_verify_octets_submit_packed(nn_o, args_o))
    while pend:
        retire_o()
    bls["proof_verify_from_octets_host_inclusive"] = 64 * n / (time.perf_counter() - t1)
    # ... and with the disclosed messages as raw bytes too (msg_to_scalars on the device: the reference's public
    # proof_verify in one call), the 32-byte messages of the workload
    raw_msgs = [[pc.expand_message(b"bbs-bench-msg" + pc.i2osp(b, 8) + pc.i2osp(j, 8), b"BBS_BENCH_MSG_DST_", 32) for j in range(R)]
                for b in range(n)]
    nn_w, keep_w, args_w = eng._wire_inputs(octs, raw_msgs, disclosed, None, None)

    def wire_submit():
        st_ = np.full(n, -128, dtype=np.int8)
        jh = ctypes.c_void_p()
        eng._chk(eng.lib.bbs_proof_verify_wire_submit(eng.h, nn_w, *args_w, st_.ctypes.data_as(_l.c_i8p), ctypes.byref(jh)), "bbs_proof_verify_wire_submit")
        j = Job(eng, jh, n)
        j.result = st_
        return j
    for k in range(16):
        if len(pend) >= 8:
            retire_o()
        pend.append(wire_submit())
    while pend:
        retire_o()
    t1 = time.perf_counter()
    for k in range(64):
        if len(pend) >= 8:
            retire_o()
        pend.append(wire_submit())
    while pend:
        retire_o()
    bls["proof_verify_wire_raw_messages_host_inclusive"] = 64 * n / (time.perf_counter() - t1)
    issuer_probe("7 after the wire legs")

    # ---- bbs_issuer: ONE call over proofs of two different lengths (half 32 messages / 8 disclosed, half 16 / 4), raw
    # disclosed messages, the library routing the items to the context of their own message count (two groups in flight);
    # the call is synchronous, so this is one list at a time
    from bbs_sign_amd import Issuer
    s16, e16, _, _ = pc.bench_engine("bls12_381", 16, None, 16, device=device)
    m16, d16, r16 = pc.bench_items(s16, e16, n // 2, 16, 4, 0)
    sg16, st16 = e16.core_sign_batch(m16)
    pf16, st16 = e16.core_proof_gen_batch(sg16, m16, d16, r16)
    assert (st16 == 1).all()
    raw16 = [[pc.expand_message(b"bbs-bench-msg" + pc.i2osp(b, 8) + pc.i2osp(j, 8), b"BBS_BENCH_MSG_DST_", 32) for j in range(4)] for b in range(n // 2)]
    oct16 = [_api.proof_to_octets("bls12_381", p_) for p_ in pf16]
    e16.close()
    iss = Issuer("bls12_381", suite.api_id, device=device, window_bits=16)
    iss.set_public_key(eng.public_key())
    mix_oct = [x for pair in zip(octs[:n // 2], oct16) for x in pair]
    mix_raw = [x for pair in zip(raw_msgs[:n // 2], raw16) for x in pair]
    mix_idx = [x for pair in zip(disclosed[:n // 2], d16) for x in pair]
    n_i, keep_i, args_i = iss.pack_proof_verify(mix_oct, mix_raw, mix_idx)
    assert (iss.proof_verify_packed(n_i, args_i) == 1).all()
    t1 = time.perf_counter()
    for _ in range(16):
        st_i = iss.proof_verify_packed(n_i, args_i)
    bls["issuer_proof_verify_two_lengths_one_list_at_a_time"] = 16 * n / (time.perf_counter() - t1)
    assert (st_i == 1).all() and iss.context_count() == 2
    pend_i = []

    def retire_i():
        j = pend_i.pop(0)
        j.wait()
        assert (j.result == 1).all()
        j.free()
    # lists in flight: 24 job streams (four lists) on 20 hardware queues is an unlucky point -- which job streams share a queue
    # depends on what else the process has alive (profiles/r05_r_issuer_lists_in_flight.log: 0.73 - 0.92 M at four, 1.05 - 1.13 M
    # at six); both are reported
    for lists in (4, 6):
        for phase in (0, 1):                          # warm, then timed
            t1 = time.perf_counter()
            for _ in range(32):
                if len(pend_i) >= lists:
                    retire_i()
                pend_i.append(iss.proof_verify_submit_packed(n_i, args_i))
            while pend_i:
                retire_i()
        bls["issuer_proof_verify_two_lengths_%d_lists_in_flight" % lists] = 32 * n / (time.perf_counter() - t1)
    bls["issuer_note"] = ("lists in flight: four lists are 24 job streams on 20 hardware queues, a pothole (0.73 - 0.92 M stand-alone); six: 1.05 - "
                          "1.13 M (tools/quick_issuer_state.py, profiles/r05_r_issuer_lists_in_flight.log).  Until the buffer pools evicted by age "
                          "(round 5) these legs read 0.4 - 0.75 M here: the 16384-item jobs of earlier legs had filled the pools "
                          "(profiles/r05_x_issuer_probe_inside_bench.log)")
    iss.close()

    # ---- the other three operations from HOST buffers through their submit forms, 8 batches in flight, results checked:
    # verify from signature records and from signature octets (decompression + subgroup check on the device), sign and
    # proof_gen with the records delivered at wait
    def host_loop(submit, check, steps=48, depth=8):
        pend = []
        any_order = depth > 8                          # the deeper loops retire in completion order (bbs_jobs_wait_any), as bench.submit_loop
        def retire():
            if any_order:
                j = pend.pop(Job.wait_any(pend))
            else:
                j = pend.pop(0)
                j.wait()
            check(j)
            j.free()
        for _ in range(4 * depth):                     # warm: pools (page-locked and device buffers, streams) reach their size
            if len(pend) >= depth:
                retire()
            pend.append(submit())
        while pend:
            retire()
        t = time.perf_counter()
        for _ in range(steps):
            if len(pend) >= depth:
                retire()
            pend.append(submit())
        while pend:
            retire()
        return steps * n / (time.perf_counter() - t)

    def all_true(j):
        assert (j.result == 1).all()
    sig_octs = [_api.signature_to_octets("bls12_381", s_) for s_ in sigs[:n]]
    ob_s, _ = eng._sig_octets(sig_octs)
    ms_v, mo_v = eng._scalars(msgs)
    hb_v, ho_v = _ragged_bytes([b""] * n)
    sg_v = eng._sigs(sigs)
    import ctypes as _ct

    def packed_submit(fn, name, *a):
        st_ = np.full(n, -128, dtype=np.int8)
        jh = _ct.c_void_p()
        eng._chk(fn(eng.h, n, *a, st_.ctypes.data_as(_l.c_i8p), _ct.byref(jh)), name)
        j = Job(eng, jh, n)
        j.result = st_
        return j
    u8 = lambda x: x.ctypes.data_as(_l.c_u8p)
    u64 = lambda x: x.ctypes.data_as(_l.c_u64p)
    bls["verify_from_octets_host_inclusive"] = host_loop(
        lambda: packed_submit(eng.lib.bbs_verify_octets_submit, "bbs_verify_octets_submit", u8(ob_s), u8(ms_v), u64(mo_v), u8(hb_v), u64(ho_v)), all_true)
    bls["verify_host_inclusive"] = host_loop(
        lambda: packed_submit(eng.lib.bbs_core_verify_submit, "bbs_core_verify_submit", u8(sg_v), u8(ms_v), u64(mo_v), u8(hb_v), u64(ho_v)), all_true)
    sig_out = [np.zeros(n * (2 * eng.fpb + 32), dtype=np.uint8) for _ in range(9)]
    turn = [0]

    def sign_submit():
        o = sig_out[turn[0] % 9]; turn[0] += 1
        return packed_submit(eng.lib.bbs_core_sign_submit, "bbs_core_sign_submit", u8(ms_v), u64(mo_v), u8(hb_v), u64(ho_v), u8(o))
    bls["sign_host_inclusive"] = host_loop(sign_submit, all_true)
    assert any(o.any() for o in sig_out)
    pgn, pgkeep, pgargs = eng._pg_inputs(sigs, msgs, disclosed, rnds, None, None)
    pf_out = [np.zeros(n * (6 * eng.fpb + 128), dtype=np.uint8) for _ in range(9)]
    cm_out = [np.zeros(n * L * 32, dtype=np.uint8) for _ in range(9)]
    cmo_out = [np.zeros(n + 1, dtype=np.uint64) for _ in range(9)]

    def pg_submit():
        k = turn[0] % 9; turn[0] += 1
        return packed_submit(eng.lib.bbs_core_proof_gen_submit, "bbs_core_proof_gen_submit", *pgargs, u8(pf_out[k]), u8(cm_out[k]), u64(cmo_out[k]))
    bls["proof_gen_host_inclusive"] = host_loop(pg_submit, all_true, steps=72, depth=12)       # (the comb's table stage: a proof_gen job is longer, 12 in flight fill the chip)
    assert all(int(c_[n]) == n * (L - R) for c_ in cmo_out[:8])
    # ... and to the wire: signature / proof octet strings compressed on the device
    so_out = [np.zeros(n * (eng.fpb + 32), dtype=np.uint8) for _ in range(9)]

    def sign_oct_submit():
        o = so_out[turn[0] % 9]; turn[0] += 1
        return packed_submit(eng.lib.bbs_sign_octets_submit, "bbs_sign_octets_submit", u8(ms_v), u64(mo_v), u8(hb_v), u64(ho_v), u8(o))
    bls["sign_to_octets_host_inclusive"] = host_loop(sign_oct_submit, all_true)
    po_len = 3 * eng.fpb + 32 * (4 + L - R)
    po_out = [np.zeros(n * (3 * eng.fpb + 32 * (4 + L)), dtype=np.uint8) for _ in range(9)]
    poo_out = [np.zeros(n + 1, dtype=np.uint64) for _ in range(9)]

    def pg_oct_submit():
        k = turn[0] % 9; turn[0] += 1
        return packed_submit(eng.lib.bbs_proof_gen_octets_submit, "bbs_proof_gen_octets_submit", *pgargs, u8(po_out[k]), u64(poo_out[k]))
    bls["proof_gen_to_octets_host_inclusive"] = host_loop(pg_oct_submit, all_true, steps=72, depth=12)       # (the comb's table stage: a proof_gen job is longer, 12 in flight fill the chip)
    assert all(int(o_[n]) == n * po_len for o_ in poo_out[:8])

    # ---- the PUBLIC functions in one call each: raw 32-byte messages in (32 per item, hashed on the device), octet strings
    # on both sides
    raw_all = [[pc.expand_message(b"bbs-bench-msg" + pc.i2osp(b, 8) + pc.i2osp(j, 8), b"BBS_BENCH_MSG_DST_", 32) for j in range(L)]
               for b in range(n)]
    mb_w, mbo_w, mio_w = eng._raw_msgs(raw_all)
    bls["verify_wire_raw_messages_host_inclusive"] = host_loop(
        lambda: packed_submit(eng.lib.bbs_verify_wire_submit, "bbs_verify_wire_submit", u8(ob_s), u8(mb_w), u64(mbo_w), u64(mio_w), u8(hb_v), u64(ho_v)), all_true)

    def sign_wire_submit():
        o = so_out[turn[0] % 9]; turn[0] += 1
        return packed_submit(eng.lib.bbs_sign_wire_submit, "bbs_sign_wire_submit", u8(mb_w), u64(mbo_w), u64(mio_w), u8(hb_v), u64(ho_v), u8(o))
    bls["sign_wire_raw_messages_host_inclusive"] = host_loop(sign_wire_submit, all_true)
    assert bytes(so_out[0][:eng.fpb + 32]) == sig_octs[0]
    di_w, dio_w = eng._indexes(disclosed)
    rs_w, ro_w = eng._scalars(rnds)
    pb_w, po_w = _ragged_bytes([b""] * n)

    def pg_wire_submit():
        k = turn[0] % 9; turn[0] += 1
        return packed_submit(eng.lib.bbs_proof_gen_wire_submit, "bbs_proof_gen_wire_submit", u8(ob_s), u8(mb_w), u64(mbo_w), u64(mio_w),
                             u64(di_w), u64(dio_w), u8(rs_w), u64(ro_w), u8(hb_v), u64(ho_v), u8(pb_w), u64(po_w), u8(po_out[k]), u64(poo_out[k]))
    bls["proof_gen_wire_raw_messages_host_inclusive"] = host_loop(pg_wire_submit, all_true, steps=72, depth=12)       # (the comb's table stage: a proof_gen job is longer, 12 in flight fill the chip)
    assert bytes(po_out[0][:po_len]) == octs[0]

    # ---- BN254 (16-bit windows) and the per-GPU share of BASELINE configs[4]
    sb_, eb, _, _ = pc.bench_engine("bn254", L, None, 16, device=device)
    eb.set_latency_mode(False)
    mb, db, rb = pc.bench_items(sb_, eb, n, L, R, 0)
    sb, st = eb.core_sign_batch(mb)
    assert (st == 1).all()
    pb, st = eb.core_proof_gen_batch(sb, mb, db, rb)
    assert (st == 1).all()
    dmb = [m[:R] for m in mb]
    out["bn254"] = {"sign": rate(eb.core_sign_upload(mb)), "verify": rate(eb.core_verify_upload(sb, mb)),
                    "proof_gen": rate(eb.core_proof_gen_upload(sb, mb, db, rb)), "proof_verify": rate(eb.core_proof_verify_upload(pb, dmb, db)),
                    "sign_8_in_flight": rate_k(lambda: eb.core_sign_upload(mb)),
                    "verify_8_in_flight": rate_k(lambda: eb.core_verify_upload(sb, mb)),
                    "proof_gen_8_in_flight": rate_k(lambda: eb.core_proof_gen_upload(sb, mb, db, rb)),
                    "proof_verify_8_in_flight": rate_k(lambda: eb.core_proof_verify_upload(pb, dmb, db)),
                    # (profiles/r05_m_other_ops_by_inflight_bn254.log: BN254's shorter jobs fill the chip at 16 in flight)
                    "sign_16_in_flight": rate_k(lambda: eb.core_sign_upload(mb), 16, 96),
                    "verify_12_in_flight": rate_k(lambda: eb.core_verify_upload(sb, mb), 12, 48),
                    "proof_gen_16_in_flight": rate_k(lambda: eb.core_proof_gen_upload(sb, mb, db, rb), 16, 64),
                    "proof_verify_16_in_flight": rate_k(lambda: eb.core_proof_verify_upload(pb, dmb, db), 16, 64)}
    mj = [eng.core_proof_verify_upload(proofs, dm, disclosed) for _ in range(4)] + \
         [eb.core_proof_verify_upload(pb, dmb, db) for _ in range(4)]
    for j in mj:
        j.run()
    for j in mj:
        j.wait()
        assert (j.status() == 1).all()
    Job.run_many_timed(mj, 8)
    mms, _ = Job.run_many_timed(mj, 64)
    out["mixed_curves_resident"] = {"proof_verify_per_s": n * 64 / (mms * 1e-3),
                                    "note": "4 BLS12-381 + 4 BN254 resident batches of %d in flight; the whole configs[4] "
                                            "path (host buffers, sharding, gather) is bench.py --config mixed65536" % n}
    for j in mj:
        j.free()
    eb.close()

    # ---- the reference's message-count sweep for sign / verify (benches/sign.rs:40, benches/verify.rs:49), BN254 like
    # the reference's benches, 16-bit windows, one resident batch at a time
    msweep = {}
    for Lm in (1, 2, 4, 8, 16, 64, 128):
        sm_, em, _, _ = pc.bench_engine("bn254", Lm, None, 16, device=device)
        em.set_latency_mode(False)
        mm, _, _ = pc.bench_items(sm_, em, n, Lm, min(R, Lm), 0)
        sg, st = em.core_sign_batch(mm)
        assert (st == 1).all()
        msweep["msgs=%d" % Lm] = {"sign": rate(em.core_sign_upload(mm)), "verify": rate(em.core_verify_upload(sg, mm))}
        em.close()
    out["bn254"]["message_count_sweep"] = msweep

    # ---- the reference's message-SIZE sweeps (benches/sign.rs:18, verify.rs:21, proof_gen.rs:22: ONE message of 32 ... 4096
    # bytes, BN254) through the public functions in one call each: raw message bytes in host buffers, hashed on the device
    # (msg_to_scalars), octet strings on both sides, 8 batches of 4096 in flight, one submitting thread
    bsweep = {}
    sb1, e1, _, _ = pc.bench_engine("bn254", 1, None, 16, device=device)
    e1.set_latency_mode(False)
    hb1, ho1 = _ragged_bytes([b""] * n)
    so1 = [np.zeros(n * (e1.fpb + 32), dtype=np.uint8) for _ in range(9)]
    di1, dio1 = e1._indexes([[0]] * n)
    rs1, ro1 = e1._scalars([[7 + b, 11 + b, 13 + b, 17 + b, 19 + b] for b in range(n)])
    po1 = [np.zeros(n * (3 * e1.fpb + 32 * 5), dtype=np.uint8) for _ in range(9)]
    poo1 = [np.zeros(n + 1, dtype=np.uint64) for _ in range(9)]

    def packed1(fn, name, *a):
        st_ = np.full(n, -128, dtype=np.int8)
        jh = _ct.c_void_p()
        e1._chk(fn(e1.h, n, *a, st_.ctypes.data_as(_l.c_i8p), _ct.byref(jh)), name)
        j = Job(e1, jh, n)
        j.result = st_
        return j
    for size in (32, 128, 512, 2048, 4096):
        raw1 = [[pc.expand_message(b"bbs-bench-bytes" + pc.i2osp(b, 8), b"BBS_BENCH_MSG_DST_", 32) * (size // 32)] for b in range(n)]
        mb1, mbo1, mio1 = e1._raw_msgs(raw1)

        def sign1():
            o = so1[turn[0] % 9]; turn[0] += 1
            return packed1(e1.lib.bbs_sign_wire_submit, "bbs_sign_wire_submit", u8(mb1), u64(mbo1), u64(mio1), u8(hb1), u64(ho1), u8(o))
        r_sign = host_loop(sign1, all_true, steps=32)
        sig1 = so1[(turn[0] - 1) % 9].copy()

        def pg1():
            k = turn[0] % 9; turn[0] += 1
            return packed1(e1.lib.bbs_proof_gen_wire_submit, "bbs_proof_gen_wire_submit", u8(sig1), u8(mb1), u64(mbo1), u64(mio1), u64(di1), u64(dio1),
                           u8(rs1), u64(ro1), u8(hb1), u64(ho1), u8(hb1), u64(ho1), u8(po1[k]), u64(poo1[k]))
        bsweep["bytes=%d" % size] = {
            "sign": r_sign,
            "verify": host_loop(lambda: packed1(e1.lib.bbs_verify_wire_submit, "bbs_verify_wire_submit", u8(sig1), u8(mb1), u64(mbo1), u64(mio1),
                                                u8(hb1), u64(ho1)), all_true, steps=32),
            "proof_gen": host_loop(pg1, all_true, steps=32),
            "message_megabytes_per_s_at_sign_rate": r_sign * size / 1e6}
    e1.close()
    out["bn254"]["single_message_bytes_sweep_wire_host_inclusive"] = bsweep
    eng.set_latency_mode("auto")
    return out
