#!/bin/bash
# on the GPU box: headline (host-inclusive) and resident throughput of each A/B library: tools/ab_bench.sh name1 name2 ...
# ("base" = the in-tree product library)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ab
for v in "$@"; do
  lib=$GRAFT_REPO_ROOT/gpurun_ab/$v/libbbs_sign_amd.so
  [ "$v" = base ] && lib=$GRAFT_REPO_ROOT/bbs_sign_amd/libbbs_sign_amd.so
  BBS_SIGN_AMD_LIB=$lib timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras --steps 96 > gpurun_out/ab/$v.json 2> gpurun_out/ab/$v.err || { echo "$v failed"; tail -3 gpurun_out/ab/$v.err; exit 1; }
  python - <<PY
import json
a=json.load(open("gpurun_out/ab/$v.json"))
print("%-12s host-inclusive %8.0f/s  resident %8.0f/s (single %.2f ms: %s)" % ("$v", a["value"], a["resident"]["proof_verify_per_s"], a["single_batch"]["ms"], {k: round(x,2) for k,x in a["single_batch"]["stage_ms"].items() if x > 0.1}))
PY
done
