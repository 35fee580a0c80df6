#!/bin/bash
# On the GPU box: full GPU test suite, bench, rocprofv3 kernel stats and PMC passes of the same bench command.
# usage: tools/run_profile.sh <tag> [profile-only]
TAG=${1:-r01}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
if [ "$2" != "profile-only" ]; then
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc" | tee -a $OUT/pytest_gpu.log; tail -3 $OUT/pytest_gpu.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 400 python bench.py > $OUT/bench.json 2> $OUT/bench.err || { echo bench failed; tail -5 $OUT/bench.err; exit 1; }
timeout -k 10 300 python bench.py --inflight 1 --no-cpu-baseline --no-extras > $OUT/bench_inflight1.json 2>> $OUT/bench.err || exit 1
fi
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras > $OUT/bench_under_rocprof.json 2> $OUT/stats.err || { echo stats failed; tail -5 $OUT/stats.err; exit 1; }
i=0
for set in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_FLAT SQ_INSTS_LDS SQ_WAIT_INST_ANY" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" \
  "FETCH_SIZE" "WRITE_SIZE" ; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -o p$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --inflight 1 --no-cpu-baseline --no-extras > $OUT/p$i.log 2>&1 || { echo "pmc pass $i failed"; tail -3 $OUT/p$i.log; exit 1; }
done
ls $OUT $OUT/stats | head -30
