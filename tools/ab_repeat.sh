#!/bin/bash
# on the GPU box: alternate two A/B libraries several times (headline mode only) to see through run-to-run noise
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/ab
a=$1; b=$2; reps=${3:-3}
for r in $(seq 1 $reps); do for v in $a $b; do
  BBS_SIGN_AMD_LIB=$GRAFT_REPO_ROOT/gpurun_ab/$v/libbbs_sign_amd.so timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras --steps 96 > gpurun_out/ab/$v.$r.json 2> gpurun_out/ab/$v.err || { echo "$v failed"; exit 1; }
  python -c "import json;d=json.load(open('gpurun_out/ab/$v.$r.json'));print('$v', $r, round(d['value']), round(d['single_batch']['stage_ms']['pairing_6lane'],2))"
done; done
