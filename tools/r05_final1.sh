#!/bin/bash
# round 5, final measurement session part 1 (GPU box): rocprofv3 kernel stats + counters of the headline's kernels on the final
# library (tools/run_profile.sh), the opcode micro-benchmark, counters of sign / verify / proof_gen (tools/run_profile_ops.sh)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out
bash tools/run_profile.sh r05_p || exit 1
hipcc -O3 --offload-arch=gfx950 tools/ubench/valu_int.hip -o /tmp/valu_int && timeout -k 10 300 /tmp/valu_int > $O/r05_p/ubench_valu_int.csv || { echo ubench failed; exit 1; }
echo "ubench done"
bash tools/run_profile_ops.sh r05_o || exit 1
# keep what travels back small: the per-dispatch counter CSVs and the stats, not the traces
find $O/r05_p $O/r05_o -name "*kernel_trace.csv" -size +4M -delete 2>/dev/null
du -sh $O/r05_p $O/r05_o
