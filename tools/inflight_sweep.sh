#!/bin/bash
# GPU box: headline loop (host-inclusive, 4096-item batches) over batches in flight x hardware queues, two repeats each
# usage: tools/inflight_sweep.sh outdir
out=${1:-gpurun_out/sweep}; mkdir -p $out
for rep in 1 2; do
  for q in 14 16; do
    for k in 8 12 16 24; do
      GPU_MAX_HW_QUEUES=$q timeout -k 10 150 python bench.py --no-extras --no-cpu-baseline --inflight $k > $out/q${q}_if${k}_r${rep}.json 2> $out/err || { echo "failed q=$q k=$k"; tail -3 $out/err; exit 1; }
      python - <<PY
import json
d = json.loads(open("$out/q${q}_if${k}_r${rep}.json").read().strip().splitlines()[-1])
print("queues $q inflight $k rep $rep: %.0f host-inclusive, %.0f resident" % (d["value"], d["resident"]["proof_verify_per_s"]), flush=True)
PY
    done
  done
done
