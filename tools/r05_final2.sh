#!/bin/bash
# round 5, final measurement session part 2 (GPU box): the whole -m gpu suite on the final library, the bench in the driver's form
# (other operations + CPU baseline) and in its default form, configs[4] through the launcher path (with and without an RCCL
# process group of one rank) and through ONE process (bbs_pool), the pipeline trace
set -e
O=gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu > $O/r05_z_pytest_gpu.log 2>&1 || { tail -40 $O/r05_z_pytest_gpu.log; exit 1; }
tail -3 $O/r05_z_pytest_gpu.log
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > $O/r05_z_bench_steps20.json 2> $O/r05_z_bench_steps20.err || { tail -20 $O/r05_z_bench_steps20.err; exit 1; }
timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline > $O/r05_z_bench_default.json 2> $O/r05_z_bench_default.err || { tail -5 $O/r05_z_bench_default.err; exit 1; }
timeout -k 10 400 python bench.py --config mixed65536 --total 8192 --steps 60 --warmup 6 > $O/r05_z_mixed8192.json 2> $O/r05_z_mixed8192.err || { tail -5 $O/r05_z_mixed8192.err; exit 1; }
timeout -k 10 400 python bench.py --config mixed65536 --total 8192 --steps 60 --warmup 6 --force-dist > $O/r05_z_mixed8192_rccl_world1.json 2> $O/r05_z_mixed8192_rccl_world1.err || { tail -5 $O/r05_z_mixed8192_rccl_world1.err; exit 1; }
timeout -k 10 400 python bench.py --config mixed65536 --total 65536 --steps 8 --warmup 2 > $O/r05_z_mixed65536.json 2> $O/r05_z_mixed65536.err || { tail -5 $O/r05_z_mixed65536.err; exit 1; }
timeout -k 10 400 python bench.py --config mixed65536 --single-process --total 65536 --steps 8 --warmup 2 > $O/r05_z_pool65536.json 2> $O/r05_z_pool65536.err || { tail -5 $O/r05_z_pool65536.err; exit 1; }
timeout -k 10 400 python bench.py --config mixed65536 --single-process --total 8192 --steps 60 --warmup 6 > $O/r05_z_pool8192.json 2> $O/r05_z_pool8192.err || { tail -5 $O/r05_z_pool8192.err; exit 1; }
for k in 5 7 8; do
  timeout -k 10 200 python bench.py --no-extras --no-cpu-baseline --steps 96 --inflight $k > $O/r05_z_bench_inflight$k.json 2> $O/r05_z_bench_inflight$k.err || { tail -5 $O/r05_z_bench_inflight$k.err; exit 1; }
done
timeout -k 10 200 python tools/quick_pipeline_trace.py 20 6 > $O/r05_pt_pipeline_trace_k20.log 2>&1 || echo "pipeline trace failed"
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob("gpurun_out/r05_z_*.json")):
    try:
        d = json.loads([l for l in open(f) if l.startswith("{")][-1])
        print(os.path.basename(f), "value", round(d["value"]), "ms/step", round(d["ms_per_step"], 3), "long", round((d.get("long_region") or {}).get("proof_verify_per_s", 0)),
              "single", round(d["single_batch"]["ms"], 2) if d.get("single_batch") else "", "valu frac", round(d.get("valu_issue", {}).get("frac", 0), 3),
              "counters current", d.get("valu_issue", {}).get("counters_from_this_library"))
    except Exception as e:
        print(f, "ERR", e)
PY
