#!/bin/bash
# round 5, GPU session 6: the whole -m gpu suite on the final library; one batch at a time in every form
set -e
O=gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu > $O/r05_z_pytest_gpu.log 2>&1 || { tail -40 $O/r05_z_pytest_gpu.log; exit 1; }
tail -3 $O/r05_z_pytest_gpu.log
timeout -k 10 300 python tools/quick_forms.py 4096 20 2>&1 | tee $O/r05_z_quick_forms.log | cut -c1-260
