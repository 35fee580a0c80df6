"""ctypes binding of the C ABI declared in include/bbs_sign_amd.h.

The product library (bbs_sign_amd/libbbs_sign_amd.so, built by __graft_entry__.build() with
hipcc --offload-arch=gfx950) is the ONLY thing loaded by default; there is no CPU fallback: a
missing library or a missing GPU raises.  Tests of the host logic may pass an explicit path to
the host-twin test library (tests/hosttwin) instead.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
PRODUCT_LIB = os.path.join(_HERE, "libbbs_sign_amd.so")

c_u8p = ctypes.POINTER(ctypes.c_uint8)
c_i8p = ctypes.POINTER(ctypes.c_int8)
c_u64p = ctypes.POINTER(ctypes.c_uint64)
c_f32p = ctypes.POINTER(ctypes.c_float)
vp = ctypes.c_void_p
sz = ctypes.c_size_t
ci = ctypes.c_int

class PvList(ctypes.Structure):
    """struct bbs_pv_list of include/bbs_sign_amd.h: the items of one curve of a list (bbs_pool_proof_verify)"""
    _fields_ = [("curve", ctypes.c_int), ("n", ctypes.c_size_t),
                ("proofs_fixed", c_u8p), ("commitments", c_u8p), ("commit_off", c_u64p),
                ("disclosed_msgs", c_u8p), ("dmsg_off", c_u64p), ("disclosed_idx", c_u64p), ("didx_off", c_u64p),
                ("headers", c_u8p), ("hdr_off", c_u64p), ("ph", c_u8p), ("ph_off", c_u64p),
                ("global_index", c_u64p), ("status", c_i8p)]


# name -> (restype, argtypes); every symbol include/bbs_sign_amd.h declares
SIGNATURES = {
    "bbs_fp_bytes": (sz, [ci]),
    "bbs_issuer_create": (ci, [ci, ci, c_u8p, sz, ctypes.POINTER(vp)]),
    "bbs_issuer_destroy": (None, [vp]),
    "bbs_issuer_set_public_key": (ci, [vp, c_u8p, ci]),
    "bbs_issuer_set_secret_key": (ci, [vp, c_u8p]),
    "bbs_issuer_set_limits": (ci, [vp, sz, ci]),
    "bbs_issuer_set_modes": (ci, [vp, ci, ci, ci]),
    "bbs_issuer_context": (ci, [vp, sz, ctypes.POINTER(vp)]),
    "bbs_issuer_context_count": (sz, [vp]),
    "bbs_issuer_proof_verify_submit": (ci, [vp, sz, c_u8p, c_u64p, c_u8p, c_u64p, c_u64p, c_u64p, c_u64p, c_u8p, c_u64p, c_u8p, c_u64p, c_i8p,
                                            ctypes.POINTER(vp)]),
    "bbs_issuer_verify_submit": (ci, [vp, sz, c_u8p, c_u8p, c_u64p, c_u64p, c_u8p, c_u64p, c_i8p, ctypes.POINTER(vp)]),
    "bbs_issuer_sign_submit": (ci, [vp, sz, c_u8p, c_u64p, c_u64p, c_u8p, c_u64p, c_u8p, c_i8p, ctypes.POINTER(vp)]),
    "bbs_issuer_proof_gen_submit": (ci, [vp, sz, c_u8p, c_u8p, c_u64p, c_u64p, c_u64p, c_u64p, c_u8p, c_u64p, c_u8p, c_u64p, c_u8p, c_u64p,
                                         c_u8p, c_u64p, c_i8p, ctypes.POINTER(vp)]),
    "bbs_issuer_job_wait": (ci, [vp]),
    "bbs_issuer_job_free": (None, [vp]),
    "bbs_issuer_proof_verify": (ci, [vp, sz, c_u8p, c_u64p, c_u8p, c_u64p, c_u64p, c_u64p, c_u64p, c_u8p, c_u64p, c_u8p, c_u64p, c_i8p]),
    "bbs_issuer_verify": (ci, [vp, sz, c_u8p, c_u8p, c_u64p, c_u64p, c_u8p, c_u64p, c_i8p]),
    "bbs_issuer_sign": (ci, [vp, sz, c_u8p, c_u64p, c_u64p, c_u8p, c_u64p, c_u8p, c_i8p]),
    "bbs_issuer_proof_gen": (ci, [vp, sz, c_u8p, c_u8p, c_u64p, c_u64p, c_u64p, c_u64p, c_u8p, c_u64p, c_u8p, c_u64p, c_u8p, c_u64p,
                                  c_u8p, c_u64p, c_i8p]),
    "bbs_version": (ctypes.c_char_p, []),
    "bbs_source_hash": (ctypes.c_char_p, []),
    "bbs_runtime_hw_queues": (ci, []),
    "bbs_device_count": (ci, []),
    "bbs_ctx_create": (ci, [ci, ci, ctypes.POINTER(vp)]),
    "bbs_ctx_destroy": (None, [vp]),
    "bbs_ctx_set_window_bits": (ci, [vp, ci]),
    "bbs_ctx_set_batch_verification": (ci, [vp, ci, c_u8p]),
    "bbs_ctx_set_points_in_subgroup": (ci, [vp, ci]),
    "bbs_ctx_set_latency_mode": (ci, [vp, ci]),
    "bbs_ctx_set_fixed_base_tree": (ci, [vp, ci]),
    "bbs_selftest_glv_split": (ci, [ci, c_u8p, c_u8p, c_u8p, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    "bbs_selftest_lin_pm": (ci, [ci, ci, c_u8p, c_u8p, c_u8p]),
    "bbs_selftest_mul3": (ci, [ci, ci, c_u8p, c_u8p, c_u8p]),
    "bbs_ctx_set_generators": (ci, [vp, c_u8p, sz, c_u8p, sz]),
    "bbs_ctx_set_public_key": (ci, [vp, c_u8p, ci]),
    "bbs_ctx_set_secret_key": (ci, [vp, c_u8p]),
    "bbs_ctx_get_public_key": (ci, [vp, c_u8p, ctypes.POINTER(ci)]),
    "bbs_ctx_get_public_key_compressed": (ci, [vp, c_u8p, sz, ctypes.POINTER(sz)]),
    "bbs_core_proof_verify_upload": (ci, [vp, sz, c_u8p, c_u8p, c_u64p, c_u8p, c_u64p, c_u64p, c_u64p,
                                          c_u8p, c_u64p, c_u8p, c_u64p, ctypes.POINTER(vp)]),
    "bbs_core_proof_verify_batch": (ci, [vp, sz, c_u8p, c_u8p, c_u64p, c_u8p, c_u64p, c_u64p, c_u64p,
                                         c_u8p, c_u64p, c_u8p, c_u64p, c_i8p]),
    "bbs_core_proof_verify_submit": (ci, [vp, sz, c_u8p, c_u8p, c_u64p, c_u8p, c_u64p, c_u64p, c_u64p,
                                          c_u8p, c_u64p, c_u8p, c_u64p, c_i8p, ctypes.POINTER(vp)]),
    "bbs_proof_verify_octets_submit": (ci, [vp, sz, c_u8p, c_u64p, c_u8p, c_u64p, c_u64p, c_u64p, c_u8p, c_u64p, c_u8p, c_u64p,
                                            c_i8p, ctypes.POINTER(vp)]),
    "bbs_proof_verify_octets_batch": (ci, [vp, sz, c_u8p, c_u64p, c_u8p, c_u64p, c_u64p, c_u64p, c_u8p, c_u64p, c_u8p, c_u64p, c_i8p]),
    "bbs_proof_verify_wire_submit": (ci, [vp, sz, c_u8p, c_u64p, c_u8p, c_u64p, c_u64p, c_u64p, c_u64p, c_u8p, c_u64p, c_u8p, c_u64p,
                                          c_i8p, ctypes.POINTER(vp)]),
    "bbs_proof_verify_wire_batch": (ci, [vp, sz, c_u8p, c_u64p, c_u8p, c_u64p, c_u64p, c_u64p, c_u64p, c_u8p, c_u64p, c_u8p, c_u64p,
                                         c_i8p]),
    "bbs_core_verify_upload": (ci, [vp, sz, c_u8p, c_u8p, c_u64p, c_u8p, c_u64p, ctypes.POINTER(vp)]),
    "bbs_core_verify_batch": (ci, [vp, sz, c_u8p, c_u8p, c_u64p, c_u8p, c_u64p, c_i8p]),
    "bbs_core_verify_submit": (ci, [vp, sz, c_u8p, c_u8p, c_u64p, c_u8p, c_u64p, c_i8p, ctypes.POINTER(vp)]),
    "bbs_verify_octets_submit": (ci, [vp, sz, c_u8p, c_u8p, c_u64p, c_u8p, c_u64p, c_i8p, ctypes.POINTER(vp)]),
    "bbs_verify_octets_batch": (ci, [vp, sz, c_u8p, c_u8p, c_u64p, c_u8p, c_u64p, c_i8p]),
    "bbs_core_sign_upload": (ci, [vp, sz, c_u8p, c_u64p, c_u8p, c_u64p, ctypes.POINTER(vp)]),
    "bbs_core_sign_batch": (ci, [vp, sz, c_u8p, c_u64p, c_u8p, c_u64p, c_u8p, c_i8p]),
    "bbs_proof_gen_wire_submit": (ci, [vp, sz, c_u8p, c_u8p, c_u64p, c_u64p, c_u64p, c_u64p, c_u8p, c_u64p, c_u8p, c_u64p, c_u8p, c_u64p,
                                       c_u8p, c_u64p, c_i8p, ctypes.POINTER(vp)]),
    "bbs_proof_gen_wire_batch": (ci, [vp, sz, c_u8p, c_u8p, c_u64p, c_u64p, c_u64p, c_u64p, c_u8p, c_u64p, c_u8p, c_u64p, c_u8p, c_u64p,
                                      c_u8p, c_u64p, c_i8p]),
    "bbs_verify_wire_submit": (ci, [vp, sz, c_u8p, c_u8p, c_u64p, c_u64p, c_u8p, c_u64p, c_i8p, ctypes.POINTER(vp)]),
    "bbs_verify_wire_batch": (ci, [vp, sz, c_u8p, c_u8p, c_u64p, c_u64p, c_u8p, c_u64p, c_i8p]),
    "bbs_sign_wire_submit": (ci, [vp, sz, c_u8p, c_u64p, c_u64p, c_u8p, c_u64p, c_u8p, c_i8p, ctypes.POINTER(vp)]),
    "bbs_sign_wire_batch": (ci, [vp, sz, c_u8p, c_u64p, c_u64p, c_u8p, c_u64p, c_u8p, c_i8p]),
    "bbs_sign_octets_submit": (ci, [vp, sz, c_u8p, c_u64p, c_u8p, c_u64p, c_u8p, c_i8p, ctypes.POINTER(vp)]),
    "bbs_sign_octets_batch": (ci, [vp, sz, c_u8p, c_u64p, c_u8p, c_u64p, c_u8p, c_i8p]),
    "bbs_proof_gen_octets_submit": (ci, [vp, sz, c_u8p, c_u8p, c_u64p, c_u64p, c_u64p, c_u8p, c_u64p,
                                         c_u8p, c_u64p, c_u8p, c_u64p, c_u8p, c_u64p, c_i8p, ctypes.POINTER(vp)]),
    "bbs_proof_gen_octets_batch": (ci, [vp, sz, c_u8p, c_u8p, c_u64p, c_u64p, c_u64p, c_u8p, c_u64p,
                                        c_u8p, c_u64p, c_u8p, c_u64p, c_u8p, c_u64p, c_i8p]),
    "bbs_core_sign_submit": (ci, [vp, sz, c_u8p, c_u64p, c_u8p, c_u64p, c_u8p, c_i8p, ctypes.POINTER(vp)]),
    "bbs_core_proof_gen_submit": (ci, [vp, sz, c_u8p, c_u8p, c_u64p, c_u64p, c_u64p, c_u8p, c_u64p,
                                       c_u8p, c_u64p, c_u8p, c_u64p, c_u8p, c_u8p, c_u64p, c_i8p, ctypes.POINTER(vp)]),
    "bbs_core_proof_gen_upload": (ci, [vp, sz, c_u8p, c_u8p, c_u64p, c_u64p, c_u64p, c_u8p, c_u64p,
                                       c_u8p, c_u64p, c_u8p, c_u64p, ctypes.POINTER(vp)]),
    "bbs_core_proof_gen_batch": (ci, [vp, sz, c_u8p, c_u8p, c_u64p, c_u64p, c_u64p, c_u8p, c_u64p,
                                      c_u8p, c_u64p, c_u8p, c_u64p, c_u8p, c_u8p, c_u64p, c_i8p]),
    "bbs_runtime_set_dedicated_queues": (ci, [ci]),
    "bbs_pool_create": (ci, [ctypes.POINTER(ci), sz, ctypes.POINTER(vp)]),
    "bbs_pool_destroy": (None, [vp]),
    "bbs_pool_device_count": (sz, [vp]),
    "bbs_pool_set_window_bits": (ci, [vp, ci, ci]),
    "bbs_pool_set_generators": (ci, [vp, ci, c_u8p, sz, c_u8p, sz]),
    "bbs_pool_set_public_key": (ci, [vp, ci, c_u8p, ci]),
    "bbs_pool_set_inflight": (ci, [vp, ci]),
    "bbs_pool_context": (ci, [vp, ci, sz, ctypes.POINTER(vp)]),
    "bbs_pool_proof_verify": (ci, [vp, ctypes.POINTER(PvList), sz, sz]),
    "bbs_pool_proof_verify_submit": (ci, [vp, ctypes.POINTER(PvList), sz, sz, ctypes.POINTER(vp)]),
    "bbs_pool_job_wait": (ci, [vp]),
    "bbs_pool_job_free": (None, [vp]),
    "bbs_runtime_queue_budget": (ci, [ci, ctypes.POINTER(ci), ctypes.POINTER(ci), ctypes.POINTER(ci), ctypes.POINTER(sz)]),
    "bbs_device_free_bytes": (sz, [ci]),
    "bbs_ctx_table_bytes": (sz, [vp]),
    "bbs_issuer_set_budget": (ci, [vp, sz, sz]),
    "bbs_issuer_table_bytes": (sz, [vp]),
    "bbs_job_run": (ci, [vp]),
    "bbs_job_wait": (ci, [vp]),
    "bbs_jobs_wait_any": (ci, [ctypes.POINTER(vp), sz, ctypes.POINTER(sz)]),
    "bbs_job_poll": (ci, [vp]),
    "bbs_job_size": (sz, [vp]),
    "bbs_job_device_bytes": (sz, [vp]),
    "bbs_job_fetch_status": (ci, [vp, c_i8p]),
    "bbs_job_fetch_signatures": (ci, [vp, c_u8p]),
    "bbs_job_fetch_proofs": (ci, [vp, c_u8p, c_u8p, c_u64p]),
    "bbs_job_free": (None, [vp]),
    "bbs_job_run_timed": (ci, [vp, ci, c_f32p, c_f32p, ci, ctypes.POINTER(ci)]),
    "bbs_job_stage_name": (ctypes.c_char_p, [vp, ci]),
    "bbs_jobs_run_timed": (ci, [ctypes.POINTER(vp), ci, ci, c_f32p, c_f32p, ci, ctypes.POINTER(ci)]),
    "bbs_ctx_set_stage_timing": (ci, [vp, ci]),
    "bbs_job_stage_times": (ci, [vp, c_f32p, c_f32p, ci, ctypes.POINTER(ci)]),
    "bbs_hash_to_scalar_batch": (ci, [vp, sz, c_u8p, c_u64p, c_u8p, sz, c_u8p]),
    "bbs_g1_msm_batch": (ci, [vp, sz, c_u8p, sz, c_u8p, c_u8p, sz, c_u8p, c_i8p]),
    "bbs_g1_msm_pippenger": (ci, [vp, sz, c_u8p, c_u8p, c_u8p, ctypes.POINTER(ci), c_i8p]),
    "bbs_pairing_product2_is_one_batch": (ci, [vp, sz, c_u8p, c_u8p, c_i8p]),
    "bbs_selftest_f12": (ci, [vp, ci, c_u8p, c_u8p, c_u8p, c_u8p]),
    "bbs_selftest_inv": (ci, [ci, ci, c_u8p, c_u8p, c_u8p]),
    "bbs_selftest_fp4sqr": (ci, [ci, ci, c_u8p, c_u8p, c_u8p]),
    "bbs_selftest_f2dot": (ci, [ci, sz, c_u8p, c_u8p, c_u8p, c_u8p]),
    "bbs_selftest_f2dot2": (ci, [ci, sz, c_u8p, c_u8p, c_u8p, c_u8p]),
    "bbs_create_generators": (ci, [ci, sz, c_u8p, sz, c_u8p]),
    "bbs_hash_to_g1": (ci, [ci, c_u8p, sz, c_u8p, sz, c_u8p]),
    "bbs_scalar_from_okm": (ci, [ci, c_u8p, c_u8p]),
    "bbs_key_gen": (ci, [ci, c_u8p, sz, c_u8p, sz, c_u8p, sz, c_u8p]),
    "bbs_signature_to_octets": (ci, [ci, c_u8p, c_u8p]),
    "bbs_signature_from_octets": (ci, [ci, c_u8p, c_u8p]),
    "bbs_proof_to_octets": (ci, [ci, c_u8p, c_u8p, sz, c_u8p]),
    "bbs_proof_from_octets": (ci, [ci, c_u8p, sz, c_u8p, c_u8p, sz, ctypes.POINTER(sz)]),
    "bbs_g1_decompress_batch": (ci, [vp, sz, c_u8p, c_u8p, c_i8p]),
    "bbs_signatures_from_octets_batch": (ci, [vp, sz, c_u8p, c_u8p, c_i8p]),
    "bbs_proofs_to_octets_batch": (ci, [ci, sz, c_u8p, c_u8p, c_u64p, c_u8p, c_u64p, c_i8p]),
    "bbs_proofs_from_octets_batch": (ci, [vp, sz, c_u8p, c_u64p, c_u8p, c_u8p, c_u64p, c_i8p]),
    "bbs_public_key_to_octets": (ci, [ci, c_u8p, ci, c_u8p]),
    "bbs_public_key_from_octets": (ci, [ci, c_u8p, c_u8p, ctypes.POINTER(ci)]),
}

_cache = {}


class LibraryMissing(RuntimeError):
    pass


def load_library(path=None):
    """Load the C-ABI library.  ``path=None`` means the product library; it is an error if it has
    not been built (run ``python -c 'import __graft_entry__ as g; g.build()'``)."""
    # BBS_SIGN_AMD_LIB: development override (A/B builds of the same product library)
    path = os.path.abspath(path or os.environ.get("BBS_SIGN_AMD_LIB") or PRODUCT_LIB)
    if path in _cache:
        return _cache[path]
    if not os.path.exists(path):
        raise LibraryMissing(
            "%s not found: the HIP extension is not built; there is no CPU fallback "
            "(build it with __graft_entry__.build())" % path)
    lib = ctypes.CDLL(path)
    # (A/B against an OLDER build of the library, development override only: symbols that build does not have yet)
    optional = set(filter(None, os.environ.get("BBS_SIGN_AMD_LIB_OPTIONAL", "").split(","))) if os.environ.get("BBS_SIGN_AMD_LIB") else set()
    for name, (res, args) in SIGNATURES.items():
        if name in optional and not hasattr(lib, name):
            continue
        fn = getattr(lib, name)          # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _cache[path] = lib
    return lib
