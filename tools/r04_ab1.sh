#!/bin/bash
# round 4, A/B 1: completion-order retire (bbs_jobs_wait_any) vs FIFO in the headline loop; configs[4] pipelined vs one list at a time
set -e
O=gpurun_out
python -m pytest tests -x -q -m gpu -k "submit or ffi_sequence or mixed_list or auto_form or issuer or threads or fail_closed or random_batch" > $O/r04_b_tests.log 2>&1 || { tail -30 $O/r04_b_tests.log; exit 1; }
tail -3 $O/r04_b_tests.log
for rep in 1 2; do
  python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/r04_b_bench_any_$rep.json 2> $O/r04_b_bench_any_$rep.err
  python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --fifo-retire > $O/r04_b_bench_fifo_$rep.json 2> $O/r04_b_bench_fifo_$rep.err
done
python bench.py --config mixed65536 --total 8192 --steps 60 --warmup 6 > $O/r04_b_mixed8192_pipe.json 2> $O/r04_b_mixed8192_pipe.err
python bench.py --config mixed65536 --total 8192 --steps 60 --warmup 6 --lists-in-flight 1 > $O/r04_b_mixed8192_one.json 2> $O/r04_b_mixed8192_one.err
python bench.py --config mixed65536 --total 8192 --steps 60 --warmup 6 --lists-in-flight 2 > $O/r04_b_mixed8192_two.json 2> $O/r04_b_mixed8192_two.err
python bench.py --config mixed65536 --total 8192 --steps 60 --warmup 6 --lists-in-flight 4 > $O/r04_b_mixed8192_four.json 2> $O/r04_b_mixed8192_four.err
python bench.py --config mixed65536 --total 65536 --steps 8 --warmup 2 > $O/r04_b_mixed65536.json 2> $O/r04_b_mixed65536.err
python - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04_b_*.json')):
    try:
        d=json.loads([l for l in open(f) if l.startswith('{')][-1])
        print(f, round(d['value']), round(d['ms_per_step'],3), d.get('long_region',{}).get('proof_verify_per_s'), d['config'].get('lists_in_flight'))
    except Exception as e:
        print(f, 'ERR', e)
PY
