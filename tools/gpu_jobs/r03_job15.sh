#!/bin/bash
# Round 3, GPU job 15: randomized runs against torch.sort on the round's build (not part of the suites).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_job15
mkdir -p $OUT
cd $ROOT
echo "== stress small/medium (10^3 .. 4e7)"; timeout -k 10 420 python3 tools/stress.py 31 330 3.0 4.6 2>&1 | tee $OUT/stress_small.log | tail -15
echo "== stress large (2.7e8 .. 8.5e8)"; RDST_STRESS_BIG64=1 timeout -k 10 560 python3 tools/stress.py 32 450 8.43 0.5 2>&1 | tee $OUT/stress_large.log | tail -15
echo done
