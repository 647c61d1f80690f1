#!/bin/bash
# Round 3, GPU job 21: the expanding K4 with its marks in the cleared table: parity, timeline, the inputs with buckets over one tile.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r03_job21
mkdir -p $OUT
cd $ROOT
echo "== pytest (hybrid, lengths)"; timeout -k 10 900 python3 -m pytest tests/test_gpu_hybrid.py tests/test_gpu_lengths.py -m gpu -x -q > $OUT/pytest.log 2>&1; rc=$?; echo "rc=$rc"; tail -5 $OUT/pytest.log
[ $rc -eq 0 ] || exit 1
echo "== timeline expand"; RDST_HIP_LIB=$ROOT/tools/_build/librdst_hip_exp.so timeout -k 10 200 python3 tools/timeline2.py uint32g 5 > $OUT/timeline_expand.log 2>&1 && tail -10 $OUT/timeline_expand.log
CASES="gaussian,bimodal,256 distinct,f32 normal,f32 uniform,uniform random"
echo "== default"; RDST_STAGES=1 timeout -k 10 300 python3 tools/skew_bench.py 1000000000 "$CASES" > $OUT/skew_default.log 2>&1 && cat $OUT/skew_default.log | grep -v amdgpu.ids
echo "== mode 16 (no mid rule)"; RDST_MODE=16 RDST_STAGES=1 timeout -k 10 300 python3 tools/skew_bench.py 1000000000 "$CASES" > $OUT/skew_mode16.log 2>&1 && cat $OUT/skew_mode16.log | grep -v amdgpu.ids
echo done
