#!/bin/bash
# Round 3, GPU job 37: where a workgroup's time goes in the 8-byte K4 on small buckets (2^27 keys: 2 048 per bucket), and in the 4-byte one's neighbour (pass B) for scale.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_job37
mkdir -p $OUT
cd $ROOT
RDST_HIP_LIB=$ROOT/tools/_build/librdst_hip_exp.so timeout -k 10 200 python3 tools/timeline2.py uint64 1 134217728 2>&1 | grep -v amdgpu.ids | tee $OUT/timeline_wide3_small.log
RDST_HIP_LIB=$ROOT/tools/_build/librdst_hip_exp.so timeout -k 10 200 python3 tools/timeline2.py uint64 1 536870912 2>&1 | grep -v amdgpu.ids | tee $OUT/timeline_wide3_mid.log
echo done
