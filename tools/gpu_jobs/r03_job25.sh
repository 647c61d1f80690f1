#!/bin/bash
# Round 3, GPU job 25: outside the window: short slices with the threshold lowered; u32 slices of 1.6 .. 3.2 x 10^9 keys, hybrid against LSD-only.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_job25
mkdir -p $OUT
cd $ROOT
timeout -k 10 400 python3 tools/window_probe.py small 2>&1 | grep -v amdgpu.ids | tee $OUT/small.log
timeout -k 10 400 python3 tools/window_probe.py big 2>&1 | grep -v amdgpu.ids | tee $OUT/big.log
echo done
