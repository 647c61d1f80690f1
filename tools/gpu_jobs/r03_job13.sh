#!/bin/bash
# Round 3, GPU job 13: A/B of the linear counter tables against the thread-major ones, same box.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_job13
mkdir -p $OUT
cd $ROOT
echo "== A/B u32"; timeout -k 10 400 python3 tools/ab_stages.py uint32 rdst_amd/librdst_hip.so tools/_build/librdst_prevk4.so 2>&1 | tee $OUT/ab_u32.log
echo "== A/B u64"; timeout -k 10 400 python3 tools/ab_stages.py uint64 rdst_amd/librdst_hip.so tools/_build/librdst_prevk4.so 2>&1 | tee $OUT/ab_u64.log
echo "== A/B f32"; timeout -k 10 400 python3 tools/ab_stages.py float32 rdst_amd/librdst_hip.so tools/_build/librdst_prevk4.so 2>&1 | tee $OUT/ab_f32.log
echo done
