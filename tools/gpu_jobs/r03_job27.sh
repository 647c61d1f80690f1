#!/bin/bash
# Round 3, GPU job 27: the suites and the bench line on the round's build.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_job27
mkdir -p $OUT
cd $ROOT
echo "== pytest gpu (all)"; timeout -k 10 1200 python3 -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.log 2>&1; echo "rc=$?"; tail -6 $OUT/pytest_gpu.log
echo "== smoke"; timeout -k 10 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
echo "== bench (driver's command)"; ( time timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err ) 2>&1 | tail -4; echo "rc=$?"; tail -c 600 $OUT/bench.err; head -c 500 $OUT/bench.json
echo done
