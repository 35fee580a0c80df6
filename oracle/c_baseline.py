"""CPU baseline of bench.py: the plain-C restatement of the reference path (oracle/c) timed on the host cores.

ORACLE = test infrastructure (only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this).

What is timed is the reference's operation order -- per-call domain hash over the compressed generators, every
`point * scalar` an independent double-and-add, two full pairings each with its own final exponentiation
(src/proof_verify.rs:163-182, :112-115) -- written in plain C with 64-bit limbs (gcc -O3, no -march flags).  It is a
restatement, NOT arkworks: the reference is Rust and cannot be built in this image.

BASELINE.md's plan: single thread and one thread per host core; sign / verify / proof_gen / proof_verify; BLS12-381
(L = 32, R = 8) and BN254; config 1 (README.md:64-81: BN254, 4 messages, one sign + verify) as the plumbing check.
"""
import concurrent.futures as cf
import time

from . import bbs, c_port
from .hashing import expand_message, i2osp


def _items(suite, port, sk, pk, gens, n, L, R):
    api_id = suite.api_id
    out = []
    for b in range(n):
        raw = [expand_message(b"bbs-bench-msg" + i2osp(b, 8) + i2osp(j, 8), b"BBS_BENCH_MSG_DST_", 32) for j in range(L)]
        msgs = bbs.msg_to_scalars(suite, raw, api_id)
        rnd = bbs.seeded_random_scalars(suite, b"bbs-bench-rnd" + i2osp(b, 8), api_id + b"MOCK_RANDOM_SCALARS_DST_", 5 + L - R)
        out.append((msgs, rnd))
    return out


def _time_ops(suite, port, cores, per_core, L=32, R=8, single=32):
    """-> {op: {"single_thread": items/s, "all_cores": items/s}} for one curve"""
    api_id = suite.api_id
    gens = bbs.create_generators(suite, L + 1, api_id) if suite.curve.name == "bls12_381" else bbs.synthetic_generators(suite, L + 1)
    sk = bbs.key_gen(suite, bytes([1] * 32), b"", b"BBS-SIG-KEYGEN-SALT-")
    pk = port.sk_to_pk(sk)
    disclosed = list(range(R))
    n = per_core * cores
    data = _items(suite, port, sk, pk, gens, n, L, R)
    res = {}

    def run(name, fn, inputs, check):
        t0 = time.perf_counter()
        one = [fn(x) for x in inputs[:single]]
        t1 = time.perf_counter() - t0
        with cf.ThreadPoolExecutor(max_workers=cores) as ex:          # ctypes releases the GIL
            t0 = time.perf_counter()
            allr = list(ex.map(fn, inputs))
            tn = time.perf_counter() - t0
        assert all(check(r) for r in one) and all(check(r) for r in allr), name
        res[name] = {"single_thread": single / t1, "all_cores": len(inputs) / tn, "items_single": single, "items_all": len(inputs)}
        return allr

    sigs = run("sign", lambda d: port.core_sign(sk, gens, b"", d[0], api_id), data, lambda s: s is not None)
    both = list(zip(data, sigs))
    run("verify", lambda x: port.core_verify(pk, x[1], gens, b"", x[0][0], api_id), both, lambda ok: ok is True)
    proofs = run("proof_gen", lambda x: port.core_proof_gen(pk, x[1], b"", gens, b"", x[0][0], disclosed, api_id, x[0][1]), both,
                 lambda p: p is not None)
    trip = list(zip(data, proofs))
    run("proof_verify", lambda x: port.core_proof_verify(pk, x[1], gens, b"", b"", x[0][0][:R], disclosed, api_id), trip,
        lambda ok: ok is True)
    return res


def config1(port_bn):
    """BASELINE configs[0] = README.md:64-81: BN254, messages b"message1", b"message2", b"msg3", b"msg4", IKM [5u8;32],
    dst "BBS-SIG-KEYGEN-SALT-", empty header: one sign + verify.  Plumbing check of the restatement."""
    suite = bbs.BN_SUITE
    api_id = suite.api_id
    raw = [b"message1", b"message2", b"msg3", b"msg4"]
    sk = bbs.key_gen(suite, bytes([5] * 32), b"", b"BBS-SIG-KEYGEN-SALT-")
    gens = bbs.create_generators(suite, len(raw) + 1, api_id)
    msgs = bbs.msg_to_scalars(suite, raw, api_id)
    t0 = time.perf_counter()
    pk = port_bn.sk_to_pk(sk)
    sig = port_bn.core_sign(sk, gens, b"", msgs, api_id)
    ok = port_bn.core_verify(pk, sig, gens, b"", msgs, api_id)
    ms = (time.perf_counter() - t0) * 1e3
    want = bbs.core_sign(suite, sk, gens, b"", msgs, api_id)
    assert ok is True and (sig.a, sig.e) == (want.a, want.e)
    return {"sign_plus_verify_ms": ms, "verified": True, "matches_python_oracle": True}


def run(cores, budget_s=20.0):
    """The contract's cpu_baseline object (value = BLS12-381 proof_verify/s on all cores) plus the rest as extras."""
    t_start = time.perf_counter()
    per_core = 64        # ~10-20 s of CPU work per curve over the four operations
    bls = _time_ops(bbs.BLS_SUITE, c_port.port("bls12_381"), cores, per_core)
    out = {
        "value": bls["proof_verify"]["all_cores"], "unit": "proof_verify/s", "cores": cores, "kind": "port",
        "single_thread_value": bls["proof_verify"]["single_thread"],
        "sample": "%d items of the bench workload per operation (BLS12-381, L=32, R=8; %d of them also single-thread), core_* with "
                  "caller-supplied generators, plain-C restatement of the reference's operation order (oracle/c, gcc -O3, "
                  "64-bit limbs; NOT arkworks), one thread per host core" % (per_core * cores, 32),
        "bls12_381": bls,
    }
    try:
        port_bn = c_port.port("bn254")
    except NotImplementedError as e:
        out["bn254"] = None
        out["bn254_note"] = str(e)
    else:
        out["bn254"] = _time_ops(bbs.BN_SUITE, port_bn, cores, per_core)
        out["config1_bn254_4msgs"] = config1(port_bn)
    out["wall_s"] = time.perf_counter() - t_start
    return out
