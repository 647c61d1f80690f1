#!/bin/bash
# Round 3, GPU job 10: rocprof records of the u32 / u64 / f32 commands on the round's build; stage times of the skewed inputs.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_job10
mkdir -p $OUT
cd $ROOT
echo "== skew stages (default)"; RDST_STAGES=1 timeout -k 10 300 python3 tools/skew_bench.py 1000000000 "gaussian,reverse sorted,bimodal,f32 normal" > $OUT/skew_default.log 2>&1; tail -12 $OUT/skew_default.log
echo "== skew stages (LSD only)"; RDST_MODE=0 RDST_STAGES=1 timeout -k 10 300 python3 tools/skew_bench.py 1000000000 "gaussian,reverse sorted,bimodal,f32 normal" > $OUT/skew_lsd.log 2>&1; tail -12 $OUT/skew_lsd.log
echo "== profiles"
timeout -k 10 500 bash tools/profile.sh r03b > $OUT/profile_u32.log 2>&1 || echo "profile u32 failed"
timeout -k 10 500 bash tools/profile.sh r03b_u64 --dtype u64 > $OUT/profile_u64.log 2>&1 || echo "profile u64 failed"
timeout -k 10 500 bash tools/profile.sh r03b_f32 --dtype f32 > $OUT/profile_f32.log 2>&1 || echo "profile f32 failed"
echo done
