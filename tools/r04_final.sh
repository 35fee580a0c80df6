#!/bin/bash
# round 4 measurement session on the GPU box: profile set (kernel stats, counters, micro-benchmark), the bench in the driver's
# form and in its default form, the pipeline trace of the serving loop.  usage: tools/r04_final.sh <tag>
TAG=${1:-r04_p}
O=$GRAFT_REPO_ROOT/gpurun_out
bash $GRAFT_REPO_ROOT/tools/run_profile.sh $TAG || exit 1
cd $GRAFT_REPO_ROOT
hipcc -O3 --offload-arch=gfx950 tools/ubench/valu_int.hip -o /tmp/valu_int && timeout -k 10 300 /tmp/valu_int > $O/$TAG/ubench_valu_int.csv || { echo ubench failed; exit 1; }
echo "ubench done"
timeout -k 10 500 python bench.py --steps 20 --warmup 5 > $O/${TAG}_bench_steps20.json 2> $O/${TAG}_bench_steps20.err || { echo bench20 failed; tail -5 $O/${TAG}_bench_steps20.err; exit 1; }
echo "bench steps 20 done"
timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline > $O/${TAG}_bench_default.json 2> $O/${TAG}_bench_default.err || { echo bench default failed; exit 1; }
timeout -k 10 200 python tools/quick_pipeline_trace.py 20 6 > $O/${TAG}_pipeline_trace_k20.log 2>&1 || echo "pipeline trace failed"
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out", "*_bench_*.json"))):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(os.path.basename(f), round(d["value"]), round(d["ms_per_step"], 3), (d.get("long_region") or {}).get("proof_verify_per_s"), round(d["single_batch"]["ms"], 2))
    except Exception as e:
        print(f, "ERR", e)
PY
