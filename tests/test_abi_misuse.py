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
