#!/bin/bash
# Round 3, GPU job 24: the sort rate against the length, 2^24 .. 2^32.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_job24
mkdir -p $OUT
cd $ROOT
timeout -k 10 800 python3 tools/size_sweep.py $OUT/size_sweep.json 2>&1 | grep -v amdgpu.ids | tee $OUT/size_sweep.log
echo done
