#!/bin/bash
# round 5, first GPU session: the queue budget as the library computes it, the whole -m gpu suite on the split MSM kernels,
# then A/B of the headline loop: round 4's library against the new one with the doubling chains on a side stream of their
# own (layout 3), as one launch on that stream (2), on the main stream (1); hardware-queue pool 14 / 18 / 20 for layout 3;
# batch verification with 4096-item jobs, 16 / 20 / 24 / 32 in flight, dedicated queues as the budget grants them.
set -e
O=gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
python - > $O/r05_a_budget.txt 2>&1 <<'PY'
import ctypes, sys
sys.path.insert(0, '.')
from bbs_sign_amd import _lib
lib = _lib.load_library()
t, p, d, s = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_size_t()
rc = lib.bbs_runtime_queue_budget(0, ctypes.byref(t), ctypes.byref(p), ctypes.byref(d), ctypes.byref(s))
print("queue budget: rc", rc, "total", t.value, "pool", p.value, "dedicated_cap", d.value, "scratch bytes/lane", s.value)
PY
cat $O/r05_a_budget.txt
python -m pytest tests -x -q -m gpu > $O/r05_a_pytest_gpu.log 2>&1 || { tail -40 $O/r05_a_pytest_gpu.log; exit 1; }
tail -3 $O/r05_a_pytest_gpu.log
run() {  # name lib layout [env...]
  name=$1; lib=$2; layout=$3; shift 3
  env "$@" BBS_SIGN_AMD_LIB=$lib BBS_PV_MSM_LAYOUT=$layout timeout -k 10 240 python bench.py --no-cpu-baseline --no-extras --steps 96 > $O/r05_a_$name.json 2> $O/r05_a_$name.err || { echo "$name failed"; tail -5 $O/r05_a_$name.err; return 1; }
  python - <<PY
import json
a=json.load(open("$O/r05_a_$name.json"))
print("%-14s host-inclusive %8.0f/s  long_region %s  resident %8.0f/s (single %.2f ms: %s)" % ("$name", a["value"], a.get("long_region",{}).get("proof_verify_per_s"), a["resident"]["proof_verify_per_s"], a["single_batch"]["ms"], {k: round(x,2) for k,x in a["single_batch"]["stage_ms"].items() if x > 0.1}))
PY
}
NEW=$GRAFT_REPO_ROOT/bbs_sign_amd/libbbs_sign_amd.so
OLD=$GRAFT_REPO_ROOT/gpurun_ab/r04/libbbs_sign_amd.so
for rep in 1 2; do
  run r04_$rep $OLD 0 BBS_SIGN_AMD_LIB_OPTIONAL=bbs_runtime_queue_budget
  run new3_$rep $NEW 3
  run new2_$rep $NEW 2
  run new1_$rep $NEW 1
done
run new3_q18 $NEW 3 GPU_MAX_HW_QUEUES=18
run new3_q20 $NEW 3 GPU_MAX_HW_QUEUES=20
run new3_q24 $NEW 3 GPU_MAX_HW_QUEUES=24
# batch verification, 4096-item jobs: old library at 4 + 16 dedicated (round 4's best), new library likewise and with what the budget grants
for v in r04 new; do
  lib=$NEW; [ $v = r04 ] && lib=$OLD
  echo "== bv $v: GPU_MAX_HW_QUEUES=4 BBS_DEDICATED_QUEUES=16" | tee -a $O/r05_a_bv.log
  BBS_SIGN_AMD_LIB_OPTIONAL=bbs_runtime_queue_budget GPU_MAX_HW_QUEUES=4 BBS_DEDICATED_QUEUES=16 BV_ONLY=1 BBS_SIGN_AMD_LIB=$lib timeout -k 10 300 python tools/quick_bv_sweep.py 12 16 20 2>&1 | tee -a $O/r05_a_bv.log
done
echo "== bv new: pool 14" | tee -a $O/r05_a_bv.log
BV_ONLY=1 timeout -k 10 300 python tools/quick_bv_sweep.py 12 16 20 24 32 2>&1 | tee -a $O/r05_a_bv.log
echo "== bv new: pool 24" | tee -a $O/r05_a_bv.log
GPU_MAX_HW_QUEUES=24 BV_ONLY=1 timeout -k 10 300 python tools/quick_bv_sweep.py 16 20 24 32 2>&1 | tee -a $O/r05_a_bv.log
