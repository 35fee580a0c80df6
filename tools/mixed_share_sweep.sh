#!/bin/bash
# Strong scaling of BASELINE configs[4] rehearsed on ONE GPU: the list a rank would own at N = 8 / 4 / 2 GPUs
# (8192 / 16384 / 32768 items), cut into jobs of different sizes, with and without latency mode.
# usage (GPU box): tools/mixed_share_sweep.sh > gpurun_out/mixed_share_sweep.log
for total in ${TOTALS:-8192 16384 32768}; do
  for mb in 512 1024 2048 4096; do
    for lat in 0 1; do
      python bench.py --config mixed65536 --total $total --min-batch $mb --latency-mode $lat --steps 8 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('total=%d min_batch=%d latency=%d -> %.3f M/s, %.2f ms/step, jobs=%d sizes=%s' % ($total,$mb,$lat,d['value']/1e6,d['ms_per_step'],d['config']['batches_per_rank'],d['config']['batch_sizes_rank0']))"
    done
  done
done
