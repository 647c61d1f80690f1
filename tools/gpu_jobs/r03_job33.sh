#!/bin/bash
# Round 3, GPU job 33: the bench line with the size points among its configs (the driver's command).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_job33
mkdir -p $OUT
cd $ROOT
( time timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | tail -4; echo "rc=$?"; tail -c 600 $OUT/bench.err; head -c 300 $OUT/bench.json
echo done
