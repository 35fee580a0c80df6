#!/bin/bash
# round 4, A/B 2: per-item pairing check as PairMillerBoth + PairFinalDist (two wavefronts per SIMD) vs the fused PairDist kernel
set -e
O=gpurun_out
python -m pytest tests -x -q -m gpu -k "kat or golden or random_batch or error_semantics or primitives or auto_form or latency_form_random or selftest or (full_batch_4096 and bls12_381-20) or (full_batch_4096 and bn254-16) or batch_verification or baseline_batch or empty" > $O/r04_d_tests.log 2>&1 || { tail -30 $O/r04_d_tests.log; exit 1; }
tail -3 $O/r04_d_tests.log
for rep in 1 2; do
  bash tools/ab_bench.sh base fused split_w1 2>&1 | tee -a $O/r04_d_ab.log
done
