#!/bin/bash
# round 5, GPU session 7: fixed-base table entries padded to 128 bytes (one line per gather) against packed 112-byte entries:
# parity tests that exercise every window width, then the headline loop and sign (fixed-base work only), alternating
set -e
O=gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu -k "window or widths or kat or golden or big_batch or every_item or fixed_base" > $O/r05_h_pytest_pad.log 2>&1 || { tail -40 $O/r05_h_pytest_pad.log; exit 1; }
tail -2 $O/r05_h_pytest_pad.log
run() {
  name=$1; lib=$2
  BBS_SIGN_AMD_LIB=$lib timeout -k 10 240 python bench.py --no-cpu-baseline --no-extras --steps 96 > $O/r05_h_$name.json 2> $O/r05_h_$name.err || { echo "$name failed"; tail -5 $O/r05_h_$name.err; return 1; }
  python - <<PY
import json
a=json.load(open("$O/r05_h_$name.json"))
print("%-10s value %8.0f/s  long_region %8.0f  resident %8.0f/s (single %.2f ms: %s)" % ("$name", a["value"], a.get("long_region",{}).get("proof_verify_per_s",0), a["resident"]["proof_verify_per_s"], a["single_batch"]["ms"], {k: round(x,2) for k,x in a["single_batch"]["stage_ms"].items() if x > 0.1}))
PY
}
for rep in 1 2; do
  run pad128_$rep $GRAFT_REPO_ROOT/bbs_sign_amd/libbbs_sign_amd.so
  run pad0_$rep $GRAFT_REPO_ROOT/gpurun_ab/pad0/libbbs_sign_amd.so
done
for v in pad128 pad0; do
  lib=$GRAFT_REPO_ROOT/bbs_sign_amd/libbbs_sign_amd.so; [ $v = pad0 ] && lib=$GRAFT_REPO_ROOT/gpurun_ab/pad0/libbbs_sign_amd.so
  echo "== $v" | tee -a $O/r05_h_fixed_base.log
  BBS_SIGN_AMD_LIB=$lib timeout -k 10 300 python tools/quick_fixed_base.py 16 20 2>&1 | tee -a $O/r05_h_fixed_base.log
done
