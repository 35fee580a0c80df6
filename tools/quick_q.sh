#!/bin/bash
# exact mode: hardware queues x batches in flight (GPU box)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/q
run() { # name, inflight, env...
  name=$1; k=$2; shift; shift
  env "$@" timeout -k 10 150 python bench.py --no-cpu-baseline --no-extras --inflight $k --steps 96 > gpurun_out/q/$name.json 2> gpurun_out/q/$name.err
  rc=$?
  if [ $rc -ne 0 ]; then echo "$name FAILED rc=$rc: $(grep -o 'HSA_STATUS[A-Z_]*' gpurun_out/q/$name.err | head -1)"; return 1; fi
  python -c "import json;d=json.load(open('gpurun_out/q/$name.json'));print('$name', round(d['value']))"
}
run q12_k8 8 GPU_MAX_HW_QUEUES=12 || exit 1
run q12_k12 12 GPU_MAX_HW_QUEUES=12 || exit 1
run q12_k16 16 GPU_MAX_HW_QUEUES=12 || exit 1
run q16_k8 8 GPU_MAX_HW_QUEUES=16 HSA_SCRATCH_MEM=68719476736 || exit 1
run q16_k16 16 GPU_MAX_HW_QUEUES=16 HSA_SCRATCH_MEM=68719476736 || exit 1
run q8_k8 8 GPU_MAX_HW_QUEUES=8 || exit 1
run q4_k8 8 GPU_MAX_HW_QUEUES=4 || exit 1
