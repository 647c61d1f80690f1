#!/bin/bash
# Round 3, GPU job 31: rehearsal of bench.py --gpus 2 on one GPU over gloo (the N > 1 path after this round's changes), and the sharded GPU tests.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_job31
mkdir -p $OUT
cd $ROOT
RDST_BENCH_SINGLE_DEVICE=1 RDST_BENCH_BACKEND=gloo timeout -k 10 600 python3 bench.py --gpus 2 --keys 100000000 --steps 3 --warmup 1 > $OUT/bench_gloo2.json 2> $OUT/bench_gloo2.err; echo "rc=$?"; tail -c 400 $OUT/bench_gloo2.err; head -c 900 $OUT/bench_gloo2.json
echo
RDST_BENCH_SINGLE_DEVICE=1 RDST_BENCH_BACKEND=gloo timeout -k 10 600 python3 bench.py --gpus 2 --keys 300000000 --dtype u32 --steps 2 --warmup 1 > $OUT/bench_gloo2_u32.json 2> $OUT/bench_gloo2_u32.err; echo "rc=$?"; tail -c 300 $OUT/bench_gloo2_u32.err; head -c 600 $OUT/bench_gloo2_u32.json
echo done
