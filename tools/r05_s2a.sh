#!/bin/bash
# round 5, GPU session 2a: queue budget, the whole -m gpu suite on the final kernels
set -e
O=gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
python - > $O/r05_c_budget.txt 2>&1 <<'PY'
import ctypes, sys
sys.path.insert(0, '.')
from bbs_sign_amd import _lib
lib = _lib.load_library()
t, p, d, s = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(), ctypes.c_size_t()
rc = lib.bbs_runtime_queue_budget(0, ctypes.byref(t), ctypes.byref(p), ctypes.byref(d), ctypes.byref(s))
print("queue budget: rc", rc, "total", t.value, "pool", p.value, "dedicated_cap", d.value, "scratch bytes/lane", s.value, "source hash", lib.bbs_source_hash().decode())
PY
cat $O/r05_c_budget.txt
python -m pytest tests -x -q -m gpu > $O/r05_c_pytest_gpu.log 2>&1 || { tail -40 $O/r05_c_pytest_gpu.log; exit 1; }
tail -3 $O/r05_c_pytest_gpu.log
