#!/bin/bash
# round 5, GPU session 4: xi applied by the publishing lane (pairing_dist.hpp BBS_DIST_XI_AT_SOURCE): six-lane self-tests and the
# pairing / parity tests, then A/B of the headline loop against the same sources built with the knob off
set -e
O=gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu -k "selftest or kat or golden or random_batch or pairing or primitives or big_batch or every_item or verify or proof_gen" > $O/r05_f_pytest_xi.log 2>&1 || { tail -40 $O/r05_f_pytest_xi.log; exit 1; }
tail -2 $O/r05_f_pytest_xi.log
run() {
  name=$1; lib=$2
  BBS_SIGN_AMD_LIB=$lib timeout -k 10 240 python bench.py --no-cpu-baseline --no-extras --steps 96 > $O/r05_f_$name.json 2> $O/r05_f_$name.err || { echo "$name failed"; tail -5 $O/r05_f_$name.err; return 1; }
  python - <<PY
import json
a=json.load(open("$O/r05_f_$name.json"))
print("%-10s value %8.0f/s  long_region %8.0f  resident %8.0f/s (single %.2f ms: %s)" % ("$name", a["value"], a.get("long_region",{}).get("proof_verify_per_s",0), a["resident"]["proof_verify_per_s"], a["single_batch"]["ms"], {k: round(x,2) for k,x in a["single_batch"]["stage_ms"].items() if x > 0.1}))
PY
}
for rep in 1 2 3; do
  run xi1_$rep $GRAFT_REPO_ROOT/bbs_sign_amd/libbbs_sign_amd.so
  run xi0_$rep $GRAFT_REPO_ROOT/gpurun_ab/xi0/libbbs_sign_amd.so
done
