#!/bin/bash
# round 5, GPU session 2b: the bench in the driver's form (with the other operations and the CPU baseline), the default form,
# batches in flight 4 .. 8, batch verification with 4096-item jobs, configs[4] with and without an RCCL process group
set -e
O=gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
show() { python - "$@" <<'PY'
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f.split('/')[-1], 'value %.0f' % d['value'], 'ms/step %.3f' % d['ms_per_step'], 'long', round(d.get('long_region', {}).get('proof_verify_per_s', 0)), 'resident', round(d.get('resident', {}).get('proof_verify_per_s', 0)) if d.get('resident') else '', 'single', round(d['single_batch']['ms'], 2) if d.get('single_batch') else '', d['config'].get('lists_in_flight', ''))
    except Exception as e:
        print(f, 'ERR', e)
PY
}
timeout -k 10 900 python bench.py --steps 20 --warmup 5 > $O/r05_d_bench_steps20.json 2> $O/r05_d_bench_steps20.err || { tail -20 $O/r05_d_bench_steps20.err; exit 1; }
show $O/r05_d_bench_steps20.json
for k in 4 5 6 7 8; do
  timeout -k 10 300 python bench.py --no-extras --no-cpu-baseline --steps 96 --inflight $k > $O/r05_d_inflight$k.json 2> $O/r05_d_inflight$k.err || { tail -5 $O/r05_d_inflight$k.err; exit 1; }
  show $O/r05_d_inflight$k.json
done
echo "== bv: pool 20 (library default)" | tee -a $O/r05_d_bv.log
BV_ONLY=1 timeout -k 10 300 python tools/quick_bv_sweep.py 12 16 20 24 2>&1 | tee -a $O/r05_d_bv.log
echo "== bv: GPU_MAX_HW_QUEUES=27 (the whole budget as pool)" | tee -a $O/r05_d_bv.log
GPU_MAX_HW_QUEUES=27 BV_ONLY=1 timeout -k 10 300 python tools/quick_bv_sweep.py 16 20 24 27 32 2>&1 | tee -a $O/r05_d_bv.log
echo "== bv: GPU_MAX_HW_QUEUES=4 BBS_DEDICATED_QUEUES=16 (granted: min(16, 27 - 4))" | tee -a $O/r05_d_bv.log
GPU_MAX_HW_QUEUES=4 BBS_DEDICATED_QUEUES=16 BV_ONLY=1 timeout -k 10 300 python tools/quick_bv_sweep.py 16 20 24 2>&1 | tee -a $O/r05_d_bv.log
timeout -k 10 400 python bench.py --config mixed65536 --total 8192 --steps 60 --warmup 6 > $O/r05_d_mixed8192.json 2> $O/r05_d_mixed8192.err || { tail -5 $O/r05_d_mixed8192.err; exit 1; }
timeout -k 10 400 python bench.py --config mixed65536 --total 8192 --steps 60 --warmup 6 --force-dist > $O/r05_d_mixed8192_rccl_world1.json 2> $O/r05_d_mixed8192_rccl_world1.err || { tail -5 $O/r05_d_mixed8192_rccl_world1.err; exit 1; }
timeout -k 10 400 python bench.py --config mixed65536 --total 65536 --steps 8 --warmup 2 > $O/r05_d_mixed65536.json 2> $O/r05_d_mixed65536.err || { tail -5 $O/r05_d_mixed65536.err; exit 1; }
show $O/r05_d_mixed8192.json $O/r05_d_mixed8192_rccl_world1.json $O/r05_d_mixed65536.json
