#!/bin/bash
# round 5, GPU session 10: every chain kernel a kernel per multiplication routine (plain / GLV) and capped at 256 registers
# (new default) against the same source with only T1's chain capped (gpurun_ab/t2): the proof_verify parity cases on the new
# default, then the headline loop, batch verification with 4096-item jobs (20 / 24 / 32 in flight), one batch at a time in
# every form, subgroup vouching (GLV kernels) 8 in flight -- alternating
# NOTE: the source this session measured (chain kernels per multiplication routine) was not kept -- profiles/r05_k_*.log, DESIGN.md 7.
set -e
O=gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
NEW=$GRAFT_REPO_ROOT/bbs_sign_amd/libbbs_sign_amd.so
T2=$GRAFT_REPO_ROOT/gpurun_ab/t2/libbbs_sign_amd.so
timeout -k 10 500 python -m pytest tests/test_parity_gpu.py -x -q -m gpu -k "proof_verify or every_item or big_batch or subgroup or batch_verif or bv or pool or identity or small_order" > $O/r05_k_pytest_pv.log 2>&1 || { tail -30 $O/r05_k_pytest_pv.log; exit 1; }
tail -1 $O/r05_k_pytest_pv.log
run() {
  name=$1; lib=$2; shift 2
  BBS_SIGN_AMD_LIB=$lib timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 96 > $O/r05_k_$name.json 2> $O/r05_k_$name.err || { echo "$name failed"; tail -5 $O/r05_k_$name.err; return 1; }
  python - <<PY
import json
a=json.load(open("$O/r05_k_$name.json"))
print("%-10s value %8.0f  long_region %8.0f  resident %8.0f  single %.2f ms  co-scheduled %s" % ("$name", a["value"], a["long_region"]["proof_verify_per_s"], a["resident"]["proof_verify_per_s"], a["single_batch"]["ms"], {k: round(x,2) for k,x in a["stage_ms_per_step"].items() if x > 0.5}))
PY
}
for rep in 1 2 3; do
  run new_$rep $NEW
  run t2_$rep $T2
done | tee $O/r05_k_headline.log
for rep in 1 2; do
  for v in new t2; do
    lib=$NEW; [ $v = t2 ] && lib=$T2
    echo "== $v rep $rep" | tee -a $O/r05_k_bv.log
    BBS_SIGN_AMD_LIB=$lib BV_ONLY=1 timeout -k 10 300 python tools/quick_bv_sweep.py 20 24 32 2>&1 | grep -v amdgpu.ids | tee -a $O/r05_k_bv.log
  done
done
for v in new t2; do
  lib=$NEW; [ $v = t2 ] && lib=$T2
  echo "== $v" | tee -a $O/r05_k_forms.log
  BBS_SIGN_AMD_LIB=$lib timeout -k 10 300 python tools/quick_forms.py 2>&1 | grep -v amdgpu.ids | cut -c1-260 | tee -a $O/r05_k_forms.log
done
