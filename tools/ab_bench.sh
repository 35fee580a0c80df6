#!/bin/bash
# on the GPU box: headline (exact) and batch-verification throughput of each A/B library: tools/ab_bench.sh name1 name2 ...
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ab
for v in "$@"; do
  lib=$GRAFT_REPO_ROOT/gpurun_ab/$v/libbbs_sign_amd.so
  BBS_SIGN_AMD_LIB=$lib timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras --steps 64 > gpurun_out/ab/$v.json 2> gpurun_out/ab/$v.err || { echo "$v exact failed"; tail -3 gpurun_out/ab/$v.err; exit 1; }
  BBS_SIGN_AMD_LIB=$lib timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras --batch-verify --inflight 32 --steps 128 > gpurun_out/ab/${v}_bv.json 2>> gpurun_out/ab/$v.err || { echo "$v bv failed"; tail -3 gpurun_out/ab/$v.err; exit 1; }
  python - <<PY
import json
a=json.load(open("gpurun_out/ab/$v.json")); b=json.load(open("gpurun_out/ab/${v}_bv.json"))
print("%-12s exact %8.0f/s (single %.2f ms: %s)   batch-verify %8.0f/s" % ("$v", a["value"], a["single_batch"]["ms"], {k: round(x,2) for k,x in a["single_batch"]["stage_ms"].items() if x > 0.1}, b["value"]))
PY
done
