#!/bin/bash
# on the GPU box: other-operation rates (bench extras) of each A/B library
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ab
for v in "$@"; do
  BBS_SIGN_AMD_LIB=$GRAFT_REPO_ROOT/gpurun_ab/$v/libbbs_sign_amd.so timeout -k 10 400 python bench.py --no-cpu-baseline > gpurun_out/ab/$v.ops.json 2> gpurun_out/ab/$v.err || { echo "$v failed"; tail -3 gpurun_out/ab/$v.err; exit 1; }
  python -c "import json;d=json.load(open('gpurun_out/ab/$v.ops.json'));x=d['other_ops']['bls12_381'];print('$v', round(d['value']), {k: round(x[k]) for k in x if isinstance(x[k], float)})"
done
