#!/bin/bash
# round 5, GPU session 9: chain kernels capped at 256 registers (the library's new default: BLS12-381 T1 chain; BN254 T1 chain,
# single multiplications, challenge) against the build with every cap at one wavefront per SIMD (gpurun_ab/w1):
# parity of the new default on the whole -m gpu suite first, then BN254 and BLS12-381 resident batches in flight, the headline
# loop and the mixed 65 536-item list, alternating
set -e
O=gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
NEW=$GRAFT_REPO_ROOT/bbs_sign_amd/libbbs_sign_amd.so
W1=$GRAFT_REPO_ROOT/gpurun_ab/w1/libbbs_sign_amd.so
python -m pytest tests -x -q -m gpu > $O/r05_j_pytest_gpu.log 2>&1 || { tail -40 $O/r05_j_pytest_gpu.log; exit 1; }
tail -2 $O/r05_j_pytest_gpu.log
for rep in 1 2; do
  for v in new w1; do
    lib=$NEW; [ $v = w1 ] && lib=$W1
    echo "== $v rep $rep" | tee -a $O/r05_j_inflight.log
    BBS_SIGN_AMD_LIB=$lib timeout -k 10 200 python tools/quick_inflight.py bn254 16 pv,vf 8 2>&1 | grep -v amdgpu.ids | tee -a $O/r05_j_inflight.log
    BBS_SIGN_AMD_LIB=$lib timeout -k 10 200 python tools/quick_inflight.py bls12_381 16 pv,vf 8 2>&1 | grep -v amdgpu.ids | tee -a $O/r05_j_inflight.log
  done
done
run() {
  name=$1; lib=$2; shift 2
  BBS_SIGN_AMD_LIB=$lib timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras "$@" > $O/r05_j_$name.json 2> $O/r05_j_$name.err || { echo "$name failed"; tail -5 $O/r05_j_$name.err; return 1; }
  python - <<PY
import json
a=json.load(open("$O/r05_j_$name.json"))
print("%-14s value %8.0f  long_region %s  ms_per_step %.3f" % ("$name", a["value"], a.get("long_region",{}).get("proof_verify_per_s"), a["ms_per_step"]))
PY
}
for rep in 1 2; do
  run head_new_$rep $NEW --steps 96
  run head_w1_$rep $W1 --steps 96
  run mixed_new_$rep $NEW --config mixed65536
  run mixed_w1_$rep $W1 --config mixed65536
done
