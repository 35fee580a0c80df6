#!/bin/bash
# batches in flight x hardware queues, both modes (run on the GPU box)
OUT=$GRAFT_REPO_ROOT/gpurun_out/bv; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for q in 16 32; do for k in 8 16 32; do
  GPU_MAX_HW_QUEUES=$q timeout -k 10 120 python bench.py --no-cpu-baseline --no-extras --inflight $k --steps 64 > $OUT/ex_q${q}_k${k}.json 2>> $OUT/err.log || exit 1
  python -c "import json;d=json.load(open('$OUT/ex_q${q}_k${k}.json'));print('exact queues $q inflight $k', round(d['value']), round(d['ms_per_step'],2))"
done; done
for q in 24 32; do for k in 32 64; do
  GPU_MAX_HW_QUEUES=$q timeout -k 10 120 python bench.py --no-cpu-baseline --no-extras --batch-verify --inflight $k --steps 128 > $OUT/bv_q${q}_k${k}.json 2>> $OUT/err.log || exit 1
  python -c "import json;d=json.load(open('$OUT/bv_q${q}_k${k}.json'));print('bv queues $q inflight $k', round(d['value']), round(d['ms_per_step'],2))"
done; done
