#!/bin/bash
# round 5, GPU session 12: proof_gen's comb form as a kernel of its own (PgVarComb, 256 registers, 4 spilled: two wavefronts per
# SIMD) against the previous library (one kernel for all forms, 350 registers): the proof_gen parity cases, then resident jobs
# in flight on both curves, with and without subgroup vouching (the GLV recoding of the comb), alternating
set -e
O=gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
NEW=$GRAFT_REPO_ROOT/bbs_sign_amd/libbbs_sign_amd.so
PREV=$GRAFT_REPO_ROOT/gpurun_ab/prev/libbbs_sign_amd.so
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "gen or unusual or kat or golden or vector or issuer or roundtrip or wire" > $O/r05_q_pytest_pg.log 2>&1 || { tail -30 $O/r05_q_pytest_pg.log; exit 1; }
tail -1 $O/r05_q_pytest_pg.log
for rep in 1 2 3; do
  for v in new prev; do
    lib=$NEW; [ $v = prev ] && lib=$PREV
    echo "== $v rep $rep"
    for k in 12 16; do
      BBS_SIGN_AMD_LIB=$lib timeout -k 10 200 python tools/quick_inflight.py bls12_381 20 pg $k 2>&1 | grep -v amdgpu.ids
    done
    BBS_SIGN_AMD_LIB=$lib timeout -k 10 200 python tools/quick_inflight.py bn254 20 pg 16 2>&1 | grep -v amdgpu.ids
  done
done | tee $O/r05_q_proof_gen_comb_kernel.log
