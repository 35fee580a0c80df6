#!/bin/bash
# on the GPU box: time each A/B library: tools/ab_run.sh name1 name2 ...
for v in "$@"; do
  echo "=== $v"
  BBS_SIGN_AMD_LIB=$GRAFT_REPO_ROOT/gpurun_ab/$v/libbbs_sign_amd.so timeout -k 10 200 python tools/quick_time.py bls12_381 4096 8 2>&1 | tail -1
  BBS_SIGN_AMD_LIB=$GRAFT_REPO_ROOT/gpurun_ab/$v/libbbs_sign_amd.so timeout -k 10 200 python tools/quick_streams.py 4096 8 2>&1 | grep -E "streams= (1|8)"
done
