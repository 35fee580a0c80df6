"""Misuse of the C ABI must be refused with an error code, never crash or read out of bounds: NULL where data is needed,
offsets that decrease or are absurdly large, operations on a context that is not set up, wrong job kinds.  CPU: through
the test build of the stage code (tests/hosttwin)."""
import ctypes
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import parity_cases as pc      # noqa: E402
from bbs_sign_amd import _lib  # noqa: E402
from oracle import bbs         # noqa: E402

E_ARG, E_STATE = -100, -102


@pytest.fixture(scope="module")
def twin():
    from bbs_sign_amd import build as b
    return b.build(twin=True, verbose=False)


def _u8(a):
    return a.ctypes.data_as(_lib.c_u8p)


def _u64(a):
    return a.ctypes.data_as(_lib.c_u64p)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_misuse_is_refused(twin, curve):
    suite = bbs.SUITES[curve]
    L, n = 3, 4
    eng = pc.make_engine(curve, pc.gens_for(suite, L + 1), suite.api_id, twin, sk=7)
    lib, h = eng.lib, eng.h
    fpb = eng.fpb
    st = np.zeros(n, dtype=np.int8)
    i8 = st.ctypes.data_as(_lib.c_i8p)
    msgs = np.zeros(n * L * 32, dtype=np.uint8)
    good = np.arange(0, (n + 1) * L, L, dtype=np.uint64)
    sigs = np.zeros(n * (2 * fpb + 32), dtype=np.uint8)
    job = ctypes.c_void_p()
    none8, none64 = ctypes.cast(None, _lib.c_u8p), ctypes.cast(None, _lib.c_u64p)
    # NULL where the batch needs data
    assert lib.bbs_core_verify_batch(h, n, none8, _u8(msgs), _u64(good), none8, none64, i8) == E_ARG
    assert lib.bbs_core_verify_batch(h, n, _u8(sigs), _u8(msgs), none64, none8, none64, i8) == E_ARG
    assert lib.bbs_core_verify_batch(h, n, _u8(sigs), none8, _u64(good), none8, none64, i8) == E_ARG      # offsets say 12 scalars, no data
    assert lib.bbs_core_verify_submit(h, n, _u8(sigs), _u8(msgs), _u64(good), none8, none64, ctypes.cast(None, _lib.c_i8p), ctypes.byref(job)) == E_ARG
    assert lib.bbs_verify_octets_batch(h, n, none8, _u8(msgs), _u64(good), none8, none64, i8) == E_ARG
    # offsets that decrease, or are absurd (a wrapped total would read far outside the caller's buffer)
    dec = good.copy(); dec[2] = 1
    assert lib.bbs_core_verify_batch(h, n, _u8(sigs), _u8(msgs), _u64(dec), none8, none64, i8) == E_ARG
    huge = good.copy(); huge[n] = np.uint64(1) << np.uint64(59)
    assert lib.bbs_core_verify_batch(h, n, _u8(sigs), _u8(msgs), _u64(huge), none8, none64, i8) == E_ARG
    assert lib.bbs_core_sign_batch(h, n, _u8(msgs), _u64(huge), none8, none64, _u8(sigs), i8) == E_ARG
    hdr = np.zeros(8, dtype=np.uint8)
    assert lib.bbs_core_sign_batch(h, n, _u8(msgs), _u64(good), _u8(hdr), _u64(huge), _u8(sigs), i8) == E_ARG
    # proof_verify: every ragged section is checked
    pf = np.zeros(n * (6 * fpb + 128), dtype=np.uint8)
    zero = np.zeros(n + 1, dtype=np.uint64)
    for bad_at in range(3):
        offs = [zero.copy(), zero.copy(), zero.copy()]
        offs[bad_at][n] = np.uint64(1) << np.uint64(50)
        rc = lib.bbs_core_proof_verify_batch(h, n, _u8(pf), _u8(msgs), _u64(offs[0]), _u8(msgs), _u64(offs[1]),
                                             _u64(good), _u64(offs[2]), none8, none64, none8, none64, i8)
        assert rc == E_ARG, (bad_at, rc)
    assert lib.bbs_core_proof_verify_batch(h, n, none8, none8, _u64(zero), none8, _u64(zero), none64, _u64(zero), none8, none64,
                                           none8, none64, i8) == E_ARG
    # NULL handles
    assert lib.bbs_job_wait(None) == E_ARG and lib.bbs_job_run(None) == E_ARG
    assert lib.bbs_job_fetch_status(None, i8) == E_ARG and lib.bbs_job_device_bytes(None) == 0
    assert lib.bbs_core_verify_batch(None, n, _u8(sigs), _u8(msgs), _u64(good), none8, none64, i8) == E_ARG
    # the wrong kind of job: a verify job has no signatures / proofs to fetch
    assert lib.bbs_core_verify_upload(h, n, _u8(sigs), _u8(msgs), _u64(good), none8, none64, ctypes.byref(job)) == 0
    assert lib.bbs_job_fetch_signatures(job, _u8(sigs)) == E_ARG
    assert lib.bbs_job_fetch_proofs(job, _u8(pf), none8, none64) == E_ARG
    lib.bbs_job_free(job)
    eng.close()
    # a context that is not set up: no generators / no key
    bare = ctypes.c_void_p()
    assert lib.bbs_ctx_create(0 if curve == "bls12_381" else 1, 0, ctypes.byref(bare)) == 0
    assert lib.bbs_core_verify_batch(bare, n, _u8(sigs), _u8(msgs), _u64(good), none8, none64, i8) == E_STATE
    assert lib.bbs_core_sign_batch(bare, n, _u8(msgs), _u64(good), none8, none64, _u8(sigs), i8) == E_STATE
    lib.bbs_ctx_destroy(bare)


@pytest.mark.parametrize("curve", ["bls12_381", "bn254"])
def test_wire_forms_without_messages_and_rebased_item_offsets(twin, curve):
    """A wire-form batch in which no item has a message may pass msg_byte_off = NULL, and item offsets need not start at
    zero: the library must not index the (absent) byte offsets with the caller's first item offset."""
    suite = bbs.SUITES[curve]
    n = 3
    eng = pc.make_engine(curve, pc.gens_for(suite, 1), suite.api_id, twin, sk=11)        # L = 0
    lib, h = eng.lib, eng.h
    none8, none64 = ctypes.cast(None, _lib.c_u8p), ctypes.cast(None, _lib.c_u64p)
    item_off = np.full(n + 1, 1 << 40, dtype=np.uint64)            # empty items, offsets far from zero
    st = np.full(n, -128, dtype=np.int8)
    so = np.zeros(n * (eng.fpb + 32), dtype=np.uint8)
    rc = lib.bbs_sign_wire_batch(h, n, none8, none64, _u64(item_off), none8, none64, _u8(so), st.ctypes.data_as(_lib.c_i8p))
    assert rc == 0 and (st == 1).all()
    want = bbs.core_sign(suite, 11, pc.gens_for(suite, 1), b"", [], suite.api_id)
    from bbs_sign_amd import api
    assert bytes(so[:eng.fpb + 32]) == api.signature_to_octets(curve, pc.Signature(want.a, want.e))
    st[:] = -128
    rc = lib.bbs_verify_wire_batch(h, n, _u8(so), none8, none64, _u64(item_off), none8, none64, st.ctypes.data_as(_lib.c_i8p))
    assert rc == 0 and (st == 1).all()
    eng.close()


@pytest.mark.parametrize("curve", ["bls12_381"])
def test_rerun_of_a_submitted_job_delivers_the_new_results(twin, curve):
    """bbs_job_run on a submit-form job followed by bbs_job_wait hands over THAT run's statuses (here: the context's
    public key changed in between, so every pairing product fails), not the first run's."""
    suite = bbs.SUITES[curve]
    L, R, n = 3, 1, 4
    gens = pc.gens_for(suite, L + 1)
    eng = pc.make_engine(curve, gens, suite.api_id, twin, sk=7)
    rng = __import__("random").Random(5)
    msgs = [[rng.randrange(suite.curve.r) for _ in range(L)] for _ in range(n)]
    disclosed = [[1]] * n
    rnds = [[rng.randrange(1, suite.curve.r) for _ in range(5 + L - R)] for _ in range(n)]
    sigs, st = eng.core_sign_batch(msgs)
    proofs, st = eng.core_proof_gen_batch(sigs, msgs, disclosed, rnds)
    assert (st == 1).all()
    nn, keep, args = eng._pv_inputs(proofs, [[m[1]] for m in msgs], disclosed, None, None)
    job = eng.submit_packed(nn, args)
    job.wait()
    assert (job.result == 1).all()
    eng.set_public_key(bbs.sk_to_pk(suite, 8))
    # (context constants reach the device with the next upload)
    assert list(eng.core_proof_verify_batch(proofs[:1], [[msgs[0][1]]], disclosed[:1])) == [0]
    job.result[:] = -99
    job.run()
    job.wait()
    assert (job.result == 0).all(), list(job.result)
    job.free()
    eng.close()


def test_runtime_and_wait_any_arguments(twin):
    lib = _lib.load_library(twin)
    assert lib.bbs_runtime_set_dedicated_queues(-1) == E_ARG and lib.bbs_runtime_set_dedicated_queues(17) == E_ARG
    assert lib.bbs_runtime_set_dedicated_queues(0) == 0
    idx = ctypes.c_size_t(5)
    assert lib.bbs_jobs_wait_any(None, 0, ctypes.byref(idx)) == E_ARG
    arr = (ctypes.c_void_p * 2)(None, None)
    assert lib.bbs_jobs_wait_any(arr, 2, None) == E_ARG
    assert lib.bbs_jobs_wait_any(arr, 2, ctypes.byref(idx)) == E_STATE and idx.value == 5      # nothing to wait for; index untouched
    assert lib.bbs_job_poll(None) == E_ARG
    assert lib.bbs_ctx_table_bytes(None) == 0 and lib.bbs_issuer_table_bytes(None) == 0
    assert lib.bbs_issuer_set_budget(None, 4, 0) == E_ARG


def test_pool_misuse_is_refused(twin):
    """bbs_pool_*: NULL handles, devices that do not exist, sections with NULL parts or garbage offsets are refused BY THE
    SUBMITTING CALL (nothing is queued to the members' threads), a curve without generators / key is BBS_E_STATE, and the pool
    still verifies a good list afterwards."""
    from bbs_sign_amd.pool import Pool
    lib = _lib.load_library(twin)
    E_NO_DEVICE = -104
    h = ctypes.c_void_p()
    ids = (ctypes.c_int * 2)(0, 0)
    assert lib.bbs_pool_create(None, 2, ctypes.byref(h)) == E_ARG
    assert lib.bbs_pool_create(ids, 0, ctypes.byref(h)) == E_ARG
    assert lib.bbs_pool_create(ids, 2, None) == E_ARG
    bad = (ctypes.c_int * 2)(0, 4096)
    assert lib.bbs_pool_create(bad, 2, ctypes.byref(h)) == E_NO_DEVICE
    neg = (ctypes.c_int * 1)(-1)
    assert lib.bbs_pool_create(neg, 1, ctypes.byref(h)) == E_NO_DEVICE
    assert lib.bbs_pool_device_count(None) == 0
    assert lib.bbs_pool_set_window_bits(None, 0, 4) == E_ARG and lib.bbs_pool_set_inflight(None, 4) == E_ARG
    assert lib.bbs_pool_set_generators(None, 0, None, 0, None, 0) == E_ARG and lib.bbs_pool_set_public_key(None, 0, None, 1) == E_ARG
    assert lib.bbs_pool_proof_verify(None, None, 0, 0) == E_ARG and lib.bbs_pool_job_wait(None) == E_ARG
    lib.bbs_pool_job_free(None)
    lib.bbs_pool_destroy(None)

    suite = bbs.SUITES["bls12_381"]
    L, R, n = 3, 1, 6
    gens = pc.gens_for(suite, L + 1)
    eng = pc.make_engine("bls12_381", gens, suite.api_id, twin, sk=11)
    msgs, disclosed, rnds = pc.bench_items(suite, eng, n, L, R)
    sigs, st = eng.core_sign_batch(msgs)
    proofs, st = eng.core_proof_gen_batch(sigs, msgs, disclosed, rnds)
    assert (st == 1).all()
    dm, didx = [m[:R] for m in msgs], [list(d) for d in disclosed]
    pool = Pool([0, 0], twin)
    ph = pool.h
    assert lib.bbs_pool_set_inflight(ph, 0) == E_ARG and lib.bbs_pool_set_inflight(ph, 65) == E_ARG
    assert lib.bbs_pool_set_window_bits(ph, 7, 4) == E_ARG                           # no such curve
    ctx = ctypes.c_void_p()
    assert lib.bbs_pool_context(ph, 0, 2, ctypes.byref(ctx)) == E_ARG               # two members: 0 and 1
    assert lib.bbs_pool_context(ph, 0, 0, None) == E_ARG
    sec = pool.pack("bls12_381", proofs, dm, didx)
    job = ctypes.c_void_p()

    status = np.zeros(n, dtype=np.int8)

    def submit(mutate=None, n_lists=1):
        arr = (_lib.PvList * 1)()
        sec.fill(arr[0], status)
        if mutate:
            mutate(arr[0])
        return lib.bbs_pool_proof_verify_submit(ph, arr, n_lists, 4, ctypes.byref(job))

    assert submit() == E_STATE                                                        # no generators / key for the curve yet
    pool.set_window_bits("bls12_381", 4)
    pool.set_generators("bls12_381", gens, suite.api_id)
    pool.set_public_key("bls12_381", eng.public_key())
    assert lib.bbs_pool_proof_verify_submit(ph, None, 1, 4, ctypes.byref(job)) == E_ARG
    assert lib.bbs_pool_proof_verify_submit(ph, (_lib.PvList * 1)(), 1, 4, None) == E_ARG
    none8, none64 = ctypes.cast(None, _lib.c_u8p), ctypes.cast(None, _lib.c_u64p)
    for field, null in (("status", ctypes.cast(None, _lib.c_i8p)), ("proofs_fixed", none8), ("commit_off", none64), ("dmsg_off", none64),
                        ("didx_off", none64), ("commitments", none8), ("disclosed_msgs", none8), ("disclosed_idx", none64)):
        assert submit(lambda s, f=field, v=null: setattr(s, f, v)) == E_ARG, field
    assert submit(lambda s: setattr(s, "curve", 9)) == E_ARG
    # offsets that decrease / are absurd, in every ragged part: refused by the call, not by a member's thread later
    for field in ("commit_off", "dmsg_off", "didx_off"):
        ref = (_lib.PvList * 1)()
        sec.fill(ref[0], status)
        good = np.ctypeslib.as_array(getattr(ref[0], field), shape=(n + 1,)).copy()
        dec = good.copy(); dec[2] = dec[1] - 1 if dec[1] else np.uint64(1 << 40); dec[3] = 0
        huge = good.copy(); huge[n] = np.uint64(1) << np.uint64(50)
        for arr in (dec, huge):
            assert submit(lambda s, f=field, a=arr: setattr(s, f, a.ctypes.data_as(_lib.c_u64p))) == E_ARG, field
    hdr_off = np.zeros(n + 1, dtype=np.uint64); hdr_off[n] = 5

    def headers_without_data(s):
        s.hdr_off, s.headers = hdr_off.ctypes.data_as(_lib.c_u64p), none8
    assert submit(headers_without_data) == E_ARG                                      # 5 header bytes, no header data
    # nothing of the refused calls is in flight: reconfiguration is allowed, and a good list verifies
    pool.set_inflight(2)
    got = pool.proof_verify_packed([sec], max_batch=4)
    assert [int(x) for x in got] == [1] * n
    pool.close()
    eng.close()
