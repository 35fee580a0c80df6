#!/bin/bash
# On the GPU box: rocprofv3 kernel stats of the default bench command and PMC passes (one 4096-item batch at a time AND
# eight in flight, the headline condition).  usage: tools/run_profile.sh <tag>
TAG=${1:-r02}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
B="$GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-extras"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o stats -- python3 $B --steps 64 > $OUT/bench_under_rocprof.json 2> $OUT/stats.err || { echo stats failed; tail -5 $OUT/stats.err; exit 1; }
echo "stats done"
i=0
for set in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_FLAT SQ_INSTS_LDS SQ_WAIT_INST_ANY" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" \
  "GRBM_GUI_ACTIVE GRBM_COUNT" \
  "FETCH_SIZE" "WRITE_SIZE" ; do
  i=$((i+1))
  # (--latency-mode 0: the throughput form of a job, what the headline's steady state runs; alone, a job would get the latency form)
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -o p$i -- python3 $B --steps 4 --warmup 1 --inflight 1 --latency-mode 0 > $OUT/p$i.log 2>&1 || { echo "pmc pass $i failed"; tail -3 $OUT/p$i.log; exit 1; }
  echo "pmc pass $i (inflight 1) done"
done
# the headline condition: eight batches in flight (counters per dispatch while other dispatches run beside it)
i=0
for set in \
  "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_FLAT SQ_INSTS_LDS SQ_WAIT_INST_ANY" \
  "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" \
  "GRBM_GUI_ACTIVE GRBM_COUNT" ; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/q$i -o q$i -- python3 $B --steps 24 --warmup 8 --inflight 8 --latency-mode 0 > $OUT/q$i.log 2>&1 || { echo "pmc pass q$i failed"; tail -3 $OUT/q$i.log; exit 1; }
  echo "pmc pass $i (inflight 8) done"
done
ls $OUT | head -40
