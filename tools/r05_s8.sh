#!/bin/bash
# round 5, GPU session 8: the doubling-chain kernels of proof_verify at TWO wavefronts per SIMD (256 registers, spills) against
# one (300 / 354 registers): parity of the variant libraries on the proof_verify cases, then the headline loop, alternating
set -e
O=gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
for v in t1w2 cw22; do
  BBS_SIGN_AMD_LIB=$GRAFT_REPO_ROOT/gpurun_ab/$v/libbbs_sign_amd.so BBS_SIGN_AMD_LIB_NOHASH=1 timeout -k 10 400 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "proof_verify or every_item or big_batch" > $O/r05_i_pytest_$v.log 2>&1 || { tail -30 $O/r05_i_pytest_$v.log; exit 1; }
  tail -1 $O/r05_i_pytest_$v.log
done
run() {
  name=$1; lib=$2
  BBS_SIGN_AMD_LIB=$lib timeout -k 10 240 python bench.py --no-cpu-baseline --no-extras --steps 96 > $O/r05_i_$name.json 2> $O/r05_i_$name.err || { echo "$name failed"; tail -5 $O/r05_i_$name.err; return 1; }
  python - <<PY
import json
a=json.load(open("$O/r05_i_$name.json"))
print("%-10s value %8.0f/s  long_region %8.0f  resident %8.0f/s (single %.2f ms: %s)" % ("$name", a["value"], a.get("long_region",{}).get("proof_verify_per_s",0), a["resident"]["proof_verify_per_s"], a["single_batch"]["ms"], {k: round(x,2) for k,x in a["single_batch"]["stage_ms"].items() if x > 0.1}))
PY
}
for rep in 1 2 3; do
  run base_$rep $GRAFT_REPO_ROOT/bbs_sign_amd/libbbs_sign_amd.so
  run t1w2_$rep $GRAFT_REPO_ROOT/gpurun_ab/t1w2/libbbs_sign_amd.so
  run cw22_$rep $GRAFT_REPO_ROOT/gpurun_ab/cw22/libbbs_sign_amd.so
done
