#!/bin/bash
# On the GPU box: counters of the kernels of sign / verify / proof_gen (SURVEY 8(d) asks for their fraction of roofline too):
# four rocprofv3 --pmc passes per operation over tools/prof_ops.py (one resident 4096-item job at a time, throughput form).
# usage: tools/run_profile_ops.sh <tag>      -> gpurun_out/<tag>/<op>_p<k>/ ; then tools/rocprof_summary_ops.py
TAG=${1:-r05_o}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG; mkdir -p $OUT
P=$GRAFT_REPO_ROOT/tools/prof_ops.py
cd /tmp && export TMPDIR=/tmp
for op in sign verify proof_gen; do
  i=0
  for set in \
    "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_FLAT SQ_INSTS_LDS SQ_WAIT_INST_ANY" \
    "GRBM_GUI_ACTIVE GRBM_COUNT" \
    "FETCH_SIZE" "WRITE_SIZE" ; do
    i=$((i+1))
    timeout -k 10 240 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/${op}_p$i -o p$i -- python3 $P $op 6 > $OUT/${op}_p$i.log 2>&1 || { echo "$op pmc pass $i failed"; tail -3 $OUT/${op}_p$i.log; exit 1; }
    echo "$op pmc pass $i done"
  done
done
