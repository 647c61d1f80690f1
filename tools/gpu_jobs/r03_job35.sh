#!/bin/bash
# Round 3, GPU job 35: randomized runs on the round's last build (after the giants' counting kernel and the expanding K4's count changed).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_job35
mkdir -p $OUT
cd $ROOT
echo "== stress (2e7 .. 6e8 keys, 64-bit slices too)"; RDST_STRESS_BIG64=1 timeout -k 10 500 python3 tools/stress.py 71 470 7.3 1.5 2>&1 | tee $OUT/stress_a.log | tail -3
echo "== stress (10^3 .. 4e7 keys)"; timeout -k 10 500 python3 tools/stress.py 73 470 2>&1 | tee $OUT/stress_b.log | tail -3
echo done
