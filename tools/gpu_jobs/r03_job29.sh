#!/bin/bash
# Round 3, GPU job 29: where a short sort's time goes (2^24 / 2^26 keys, LSD route): stage times with profiling events, and wall time without.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_job29
mkdir -p $OUT
cd $ROOT
for n in 16777216 67108863; do for t in uint32 uint64; do echo "n=$n $t"; RDST_N=$n timeout -k 10 200 python3 tools/stage_times.py $t 1 2>&1 | grep mode; done; done | tee $OUT/small_stages.log
echo done
