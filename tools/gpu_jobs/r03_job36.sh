#!/bin/bash
# Round 3, GPU job 36: K4 on small buckets (4 096 keys each at 2^28 keys): the counting kernels against the generic ranked kernel.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_job36
mkdir -p $OUT
cd $ROOT
for t in uint32 uint64; do echo "n=2^28 $t"; RDST_N=268435456 timeout -k 10 200 python3 tools/stage_times.py $t 1 7 2 2>&1 | grep mode; done | tee $OUT/stages.log
echo done
