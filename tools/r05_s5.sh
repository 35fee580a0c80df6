#!/bin/bash
# round 5, GPU session 5: single batch at a time (latency form): the pairing as one kernel or split in two (BBS_PV_LAT_SPLIT)
set -e
O=gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in 0 1; do
  echo "== BBS_PV_LAT_SPLIT=$v" | tee -a $O/r05_g_single_batch.log
  BBS_PV_LAT_SPLIT=$v timeout -k 10 200 python tools/quick_forms.py 4096 20 2>&1 | grep -i "proof_verify" | head -6 | tee -a $O/r05_g_single_batch.log
done; done
