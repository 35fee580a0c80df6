#!/bin/bash
# batch-verification mode: kernel stats (32 in flight) + VALU counters (1 in flight)
TAG=${1:-bv}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras --batch-verify --inflight 32 --steps 128 > $OUT/bench_under_rocprof.json 2> $OUT/stats.err || { echo stats failed; tail -5 $OUT/stats.err; exit 1; }
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_FLAT SQ_INSTS_LDS SQ_WAIT_INST_ANY --output-format csv -d $OUT/p1 -o p1 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --inflight 1 --no-cpu-baseline --no-extras --batch-verify > $OUT/p1.log 2>&1 || { echo pmc failed; tail -3 $OUT/p1.log; exit 1; }
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS --output-format csv -d $OUT/p2 -o p2 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --inflight 1 --no-cpu-baseline --no-extras --batch-verify > $OUT/p2.log 2>&1 || { echo pmc2 failed; tail -3 $OUT/p2.log; exit 1; }
cat $OUT/bench_under_rocprof.json | python3 -c "import json,sys; d=json.load(sys.stdin); print(d['value'])"
