#!/bin/bash
# round 5, GPU session 3: the pool on the GPU (tests, configs[4] through ONE process), verify's variable-base multiplication on a side
# stream or not, batch verification with 4096-item jobs over (hardware queues x jobs in flight)
set -e
O=gpurun_out; mkdir -p $O
cd $GRAFT_REPO_ROOT
python -m pytest tests -x -q -m gpu -k "pool or queue_budget or mixed" > $O/r05_e_pytest_pool.log 2>&1 || { tail -40 $O/r05_e_pytest_pool.log; exit 1; }
tail -2 $O/r05_e_pytest_pool.log
show() { python - "$@" <<'PY'
import json, sys
for f in sys.argv[1:]:
    try:
        d = json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f.split('/')[-1], 'value %.0f' % d['value'], 'ms/step %.3f' % d['ms_per_step'], d['config'].get('lists_in_flight', ''), d['config'].get('pool_members', ''))
    except Exception as e:
        print(f, 'ERR', e)
PY
}
timeout -k 10 400 python bench.py --config mixed65536 --single-process --total 65536 --steps 8 --warmup 2 > $O/r05_e_pool65536.json 2> $O/r05_e_pool65536.err || { tail -5 $O/r05_e_pool65536.err; exit 1; }
timeout -k 10 400 python bench.py --config mixed65536 --single-process --total 8192 --steps 60 --warmup 6 > $O/r05_e_pool8192.json 2> $O/r05_e_pool8192.err || { tail -5 $O/r05_e_pool8192.err; exit 1; }
timeout -k 10 400 python bench.py --config mixed65536 --single-process --pool-devices 0,0 --total 16384 --steps 30 --warmup 4 > $O/r05_e_pool16384_two_members_one_gpu.json 2> $O/r05_e_pool16384_two_members_one_gpu.err || { tail -5 $O/r05_e_pool16384_two_members_one_gpu.err; exit 1; }
show $O/r05_e_pool65536.json $O/r05_e_pool8192.json $O/r05_e_pool16384_two_members_one_gpu.json
cat > /tmp/vf_ab.py <<'PY'
import os, sys
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"]); sys.path.insert(0, os.path.join(os.environ["GRAFT_REPO_ROOT"], "tests"))
import parity_cases as pc
from bbs_sign_amd import Job
n = 4096
suite, eng, gens, sk, msgs, disclosed, rnds = pc.bench_workload("bls12_381", n, 32, 8, None, 20)
eng.set_latency_mode(False)
sigs, st = eng.core_sign_batch(msgs)
for k in (6, 8, 12):
    js = [eng.core_verify_upload(sigs, msgs) for _ in range(k)]
    for j in js: j.run()
    for j in js: j.wait(); assert (j.status() == 1).all()
    Job.run_many_timed(js, k)
    ms, _ = Job.run_many_timed(js, 6 * k)
    print("BBS_VF_SIDE=%s verify %2d in flight: %8.0f /s" % (os.environ.get("BBS_VF_SIDE"), k, n * 6 * k / (ms * 1e-3)), flush=True)
    for j in js: j.free()
eng.close()
PY
for rep in 1 2; do for v in 0 1; do BBS_VF_SIDE=$v timeout -k 10 200 python /tmp/vf_ab.py 2>&1 | tee -a $O/r05_e_vf_side_stream.log; done; done
for q in 22 24 27; do
  echo "== bv: GPU_MAX_HW_QUEUES=$q" | tee -a $O/r05_e_bv_fine.log
  GPU_MAX_HW_QUEUES=$q BV_ONLY=1 timeout -k 10 300 python tools/quick_bv_sweep.py 18 20 22 24 18 20 22 24 2>&1 | tee -a $O/r05_e_bv_fine.log
done
echo "== bv: GPU_MAX_HW_QUEUES=27, chains on a side stream (BBS_PV_MSM_LAYOUT=3: two queues per job)" | tee -a $O/r05_e_bv_fine.log
BBS_PV_MSM_LAYOUT=3 GPU_MAX_HW_QUEUES=27 BV_ONLY=1 timeout -k 10 300 python tools/quick_bv_sweep.py 8 10 12 13 16 2>&1 | tee -a $O/r05_e_bv_fine.log
