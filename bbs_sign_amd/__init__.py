"""bbs_sign_amd -- MI355X-native batched BBS+ sign / verify / proof_gen / proof_verify engine.

The compute path is hand-written HIP for gfx950 behind the C ABI of include/bbs_sign_amd.h
(bbs_sign_amd/csrc).  This package is only the host-side mirror of the reference's core_*
interface; importing it does not load the library, using it does, and there is no CPU fallback.
"""
from .engine import (BLS12_381, BN254, BbsError, BbsRuntimeError, Engine, Issuer, Job, Proof, Signature,  # noqa: F401
                     STATUS_NAMES)
from ._lib import PRODUCT_LIB, LibraryMissing, load_library  # noqa: F401
from . import api  # noqa: F401,E402
