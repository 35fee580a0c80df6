#!/bin/bash
# round 5, on the CPU after tools/r05_final1.sh: condense gpurun_out/r05_p + r05_o into profiles/r05_p_* and the counters file bench.py loads
set -e
cd "$(dirname "$0")/.."
python tools/rocprof_summary.py gpurun_out/r05_p profiles/r05_p | grep "Pv\|Pair" || true
python tools/rocprof_summary_ops.py gpurun_out/r05_o profiles/r05_p
bash tools/kernel_meta.sh bbs_sign_amd/libbbs_sign_amd.so > profiles/r05_p_kernel_meta.txt 2>/dev/null
python tools/isa_histogram.py bbs_sign_amd/libbbs_sign_amd.so profiles/r05_p_isa_histogram.csv 'k_stage<bbs::PairDist<bbs::BlsCurve>' 'k_stage<bbs::PvT1Chain<bbs::BlsCurve>' \
  'k_stage<bbs::PvVarMul<bbs::BlsCurve>' 'k_stage<bbs::PvFixedChunk<bbs::BlsCurve>' 'k_stage<bbs::PvChallenge<bbs::BlsCurve>' 'k_stage<bbs::PvScalars<bbs::BlsCurve>' 'k_stage<bbs::PvFinish,' \
  'k_stage<bbs::PairMillerHalf<bbs::BlsCurve>' 'k_stage<bbs::PairFinalDist<bbs::BlsCurve>' 'k_stage<bbs::PvIngest<bbs::BlsCurve>' 'k_stage<bbs::SgMsmPart<bbs::BlsCurve>' \
  'k_stage<bbs::VfVarMul<bbs::BlsCurve>' 'k_stage<bbs::PgVarPart<bbs::BlsCurve>' 'k_stage<bbs::PgBPart<bbs::BlsCurve>' 'k_stage<bbs::PgTables<bbs::BlsCurve>' | tail -1
python tools/valu_model.py profiles/r05_p | tail -14
