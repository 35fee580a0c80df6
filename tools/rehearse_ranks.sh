#!/bin/bash
# On a one-GPU box: rehearse the N > 1 paths of bench.py with 2 ranks over gloo, both ranks on GPU 0 (the real N > 1
# run -- one rank per GPU over RCCL -- is the driver's; this checks the rank plumbing: distinct data per rank, barrier,
# max-over-ranks timing, the pass-count all-reduce, and for mixed65536 the shard plan, all_gather and merge).
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-rehearse}; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --backend gloo --all-ranks-on-device 0 --steps 32 --warmup 8 --no-extras --no-cpu-baseline > $OUT/bench_2rank_gloo.json 2> $OUT/bench_2rank_gloo.err || { echo "2-rank headline failed"; tail -5 $OUT/bench_2rank_gloo.err; exit 1; }
tail -c 600 $OUT/bench_2rank_gloo.json; echo
timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --backend gloo --all-ranks-on-device 0 --config mixed65536 --steps 2 --warmup 1 > $OUT/bench_mixed_2rank_gloo.json 2> $OUT/bench_mixed_2rank_gloo.err || { echo "2-rank mixed failed"; tail -5 $OUT/bench_mixed_2rank_gloo.err; exit 1; }
cat $OUT/bench_mixed_2rank_gloo.json
