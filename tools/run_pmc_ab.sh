#!/bin/bash
# SQ counter passes for A/B libraries: tools/run_pmc_ab.sh name1 name2 ...
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  export BBS_SIGN_AMD_LIB=$GRAFT_REPO_ROOT/gpurun_ab/$v/libbbs_sign_amd.so
  OUT=$GRAFT_REPO_ROOT/gpurun_out/pmcab_$v; mkdir -p $OUT
  i=0
  for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_FLAT SQ_INSTS_LDS SQ_WAIT_INST_ANY" \
             "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -o p$i -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --inflight 1 --no-cpu-baseline > $OUT/p$i.log 2>&1 || { echo "pass $i failed"; tail -3 $OUT/p$i.log; }
  done
  echo "$v done"
done
