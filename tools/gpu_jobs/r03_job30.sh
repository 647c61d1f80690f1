#!/bin/bash
# Round 3, GPU job 30: rocprof records of the u32 / u64 / f32 commands on the round's final build (profiles/r03c_*).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_job30
mkdir -p $OUT
cd $ROOT
echo "== profiles"
timeout -k 10 500 bash tools/profile.sh r03c > $OUT/profile_u32.log 2>&1 || echo "profile u32 failed"
echo "u32 done"
timeout -k 10 500 bash tools/profile.sh r03c_u64 --dtype u64 > $OUT/profile_u64.log 2>&1 || echo "profile u64 failed"
echo "u64 done"
timeout -k 10 500 bash tools/profile.sh r03c_f32 --dtype f32 > $OUT/profile_f32.log 2>&1 || echo "profile f32 failed"
echo "== stage times"; for t in uint32 uint64 float32 float64; do timeout -k 10 200 python3 tools/stage_times.py $t 1 0 2>&1 | grep mode; done | tee $OUT/stages.log
echo done
