#!/bin/bash
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_job14
mkdir -p $OUT
cd $ROOT
echo "== pytest sharded"; timeout -k 10 600 python3 -m pytest tests/test_gpu_sharded.py -m gpu -x -q > $OUT/pytest.log 2>&1; echo "rc=$?"; tail -15 $OUT/pytest.log
echo "== bench --gpus 1 through nccl path? (N=1 is not distributed)"; echo skip
echo done
