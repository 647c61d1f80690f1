#!/bin/bash
# Round 3, GPU job 28: randomized runs on the round's build (the split forced now and then; 64-bit slices up to 4.5 x 10^8 keys in the second run).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_job28
mkdir -p $OUT
cd $ROOT
echo "== stress (all sizes)"; timeout -k 10 330 python3 tools/stress.py 53 300 2>&1 | tee $OUT/stress_a.log | tail -4
echo "== stress (2e7 .. 6e8 keys, 64-bit slices too)"; RDST_STRESS_BIG64=1 timeout -k 10 330 python3 tools/stress.py 59 300 7.3 1.5 2>&1 | tee $OUT/stress_b.log | tail -4
echo done
