#!/bin/bash
# Round 3, GPU job 19: the expanding K4's mark + max-scan output; what it does to the inputs whose buckets are over one tile.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_job19
mkdir -p $OUT
cd $ROOT
echo "== pytest (hybrid, lengths, skew)"; timeout -k 10 900 python3 -m pytest tests/test_gpu_hybrid.py tests/test_gpu_lengths.py -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?; echo "rc=$rc"; tail -5 $OUT/pytest.log
[ $rc -eq 0 ] || exit 1
CASES="gaussian,ids below,one rank,bimodal,256 distinct,f32 normal,uniform random"
echo "== default"; RDST_STAGES=1 timeout -k 10 300 python3 tools/skew_bench.py 1000000000 "$CASES" > $OUT/skew_default.log 2>&1 && cat $OUT/skew_default.log | grep -v amdgpu.ids
echo "== mode 16 (no mid rule)"; RDST_MODE=16 RDST_STAGES=1 timeout -k 10 300 python3 tools/skew_bench.py 1000000000 "$CASES" > $OUT/skew_mode16.log 2>&1 && cat $OUT/skew_mode16.log | grep -v amdgpu.ids
echo "== LSD only"; RDST_MODE=0 timeout -k 10 300 python3 tools/skew_bench.py 1000000000 "$CASES" > $OUT/skew_lsd.log 2>&1; grep -v amdgpu.ids $OUT/skew_lsd.log
echo done
