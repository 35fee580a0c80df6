#!/bin/bash
# hardware queues vs the runtime's scratch pool (GPU box)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/q
run() { # name, env..., then args
  name=$1; shift
  env "$@" timeout -k 10 150 python bench.py --no-cpu-baseline --no-extras --batch-verify --inflight 32 --steps 128 > gpurun_out/q/$name.json 2> gpurun_out/q/$name.err
  rc=$?
  if [ $rc -ne 0 ]; then echo "$name FAILED rc=$rc: $(grep -o 'HSA_STATUS[A-Z_]*' gpurun_out/q/$name.err | head -1)"; return 1; fi
  python -c "import json;d=json.load(open('gpurun_out/q/$name.json'));print('$name', round(d['value']))"
}
run q12 GPU_MAX_HW_QUEUES=12 || exit 1
run q16_mem64g GPU_MAX_HW_QUEUES=16 HSA_SCRATCH_MEM=68719476736 || exit 1
run q24_mem128g GPU_MAX_HW_QUEUES=24 HSA_SCRATCH_MEM=137438953472 || exit 1
