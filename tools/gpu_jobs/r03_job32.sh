#!/bin/bash
# Round 3, GPU job 32: randomized runs around the lowered thresholds (4 x 10^7 .. 3 x 10^8 keys, 64-bit slices too) and at all sizes.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_job32
mkdir -p $OUT
cd $ROOT
echo "== stress (4e7 .. 3e8 keys)"; RDST_STRESS_BIG64=1 timeout -k 10 450 python3 tools/stress.py 61 420 7.6 0.9 2>&1 | tee $OUT/stress_a.log | tail -3
echo "== stress (10^3 .. 4e7 keys)"; timeout -k 10 450 python3 tools/stress.py 67 420 2>&1 | tee $OUT/stress_b.log | tail -3
echo done
