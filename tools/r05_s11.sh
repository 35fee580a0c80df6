#!/bin/bash
# round 5, GPU session 11: the driver's form of the bench (--steps 20 --warmup 5) with 4 / 5 / 6 / 7 batches in flight, three
# alternating repeats: is the plateau of the long run (6) also the best choice for a 57 ms region with one fill and one drain?
set -e
O=gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for k in 4 5 6 7; do
    timeout -k 10 200 python bench.py --no-extras --no-cpu-baseline --steps 20 --warmup 5 --inflight $k > $O/r05_l_s20_k${k}_$rep.json 2> $O/r05_l_s20_k${k}_$rep.err || { tail -5 $O/r05_l_s20_k${k}_$rep.err; exit 1; }
    python - <<PY
import json
a=json.load(open("$O/r05_l_s20_k${k}_$rep.json"))
print("rep $rep  in flight $k  value %8.0f  ms/step %.3f  long_region %8.0f" % (a["value"], a["ms_per_step"], a["long_region"]["proof_verify_per_s"]))
PY
  done
done | tee $O/r05_l_steps20_by_inflight.log
